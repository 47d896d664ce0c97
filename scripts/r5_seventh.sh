#!/bin/bash
# GPU box, round 5, seventh call: the rounds rule again (LDS region sized for the rule, not for its fallback), the replica tests.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replicas.py -m gpu -q -k "one_round or second_larger or heals_on_the_device or forest_packing or big_subtrees or shared_subtrees or rank or ranks" > gpurun_out/r5g_pack.log 2>&1
echo "pack tests rc=$?"; tail -5 gpurun_out/r5g_pack.log
BENCH_ARGS="--system 2clr --steps 200 --warmup 20 --cpu-evals 2 --secondary 0" bash scripts/abx.sh 2 "2clr_classes|build/diag/lib_norounds.so|" "2clr_rounds|-|" 2>&1 | tee gpurun_out/r5g_abx_2clr.log
timeout -k 10 200 python scripts/forest_probe.py 2clr 1dwc 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5g_probe.log
