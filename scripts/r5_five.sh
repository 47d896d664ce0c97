#!/bin/bash
# GPU box, round 5: the experimental five-launch mode (AGBNP_HIP_FIVE_LAUNCHES=1): its own tests, the parity suite under it,
# then previous library / default / five A/B.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_five_launches.py -m gpu -q -x > gpurun_out/r5m_five_own.log 2>&1
echo "five own rc=$?"; tail -6 gpurun_out/r5m_five_own.log
AGBNP_HIP_FIVE_LAUNCHES=1 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r5m_five.log 2>&1
echo "parity suite under five rc=$?"; tail -12 gpurun_out/r5m_five.log
bash scripts/abx.sh 2 "prev|build/diag/lib_prev.so|" "default|-|" "five|-|AGBNP_HIP_FIVE_LAUNCHES=1" "five08|-|AGBNP_HIP_FIVE_LAUNCHES=1 AGBNP_HIP_MASK_SKIN=0.08" "five04|-|AGBNP_HIP_FIVE_LAUNCHES=1 AGBNP_HIP_MASK_SKIN=0.04" 2>&1 | tee gpurun_out/r5m_abx.log
