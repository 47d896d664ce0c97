#!/bin/bash
# GPU box, round 5, sixth call: the rounds rule of the forest packing (2clr in one round of forests).
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_round or second_larger or heals_on_the_device or forest_packing or big_subtrees or shared_subtrees" > gpurun_out/r5f_pack.log 2>&1
echo "pack tests rc=$?"; tail -5 gpurun_out/r5f_pack.log
BENCH_ARGS="--system 2clr --steps 200 --warmup 20 --cpu-evals 0 --secondary 0" bash scripts/abx.sh 2 "2clr_classes|build/diag/lib_norounds.so|" "2clr_rounds|-|" 2>&1 | tee gpurun_out/r5f_abx_2clr.log
bash scripts/abx.sh 2 "1dwc_classes|build/diag/lib_norounds.so|" "1dwc_new|-|" 2>&1 | tee gpurun_out/r5f_abx_1dwc.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5f_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5f_pytest.log
