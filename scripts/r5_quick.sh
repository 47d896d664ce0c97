#!/bin/bash
# GPU box: a subset of the parity tests, then previous library / new on one box
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_five_launches.py -m gpu -q -x -k "config or golden or five or second_larger or random_walk or one_round" > gpurun_out/r5q_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5q_tests.log
bash scripts/abx.sh 3 "prev|build/diag/lib_prev.so|" "new|-|" 2>&1 | tee gpurun_out/r5q_abx.log
