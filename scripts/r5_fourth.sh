#!/bin/bash
# GPU box, round 5, fourth call: far GB strips: parity, then the lattice and 2clr with the test off / on.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "far_strips or config4" > gpurun_out/r5d_far.log 2>&1
echo "far rc=$?"; tail -5 gpurun_out/r5d_far.log
BENCH_ARGS="--system 1dwc_x4 --steps 60 --warmup 6 --cpu-evals 0 --secondary 0" bash scripts/abx.sh 2 "lattice_far0|-|AGBNP_HIP_GB_FAR=0" "lattice_far1|-|AGBNP_HIP_GB_FAR=1" 2>&1 | tee gpurun_out/r5d_abx_lattice.log
BENCH_ARGS="--system 2clr --steps 200 --warmup 10 --cpu-evals 0 --secondary 0" bash scripts/abx.sh 2 "2clr_far0|-|AGBNP_HIP_GB_FAR=0" "2clr_far1|-|AGBNP_HIP_GB_FAR=1" 2>&1 | tee gpurun_out/r5d_abx_2clr.log
