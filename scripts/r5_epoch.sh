#!/bin/bash
# GPU box: device-side parity of the five-launch mode: its tests, the md examples (graphs), then host-parity library / new A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_five_launches.py tests/test_md_examples.py -m gpu -q -x > gpurun_out/r5s_five.log 2>&1
echo "five + md rc=$?"; tail -5 gpurun_out/r5s_five.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r5s_parity.log 2>&1
echo "parity rc=$?"; tail -5 gpurun_out/r5s_parity.log
bash scripts/abx.sh 3 "hostpar|build/diag/lib_prev.so|" "epoch|-|" 2>&1 | tee gpurun_out/r5s_abx.log
