#!/bin/bash
mkdir -p gpurun_out
echo "== 2clr"; bash scripts/ab_env.sh "AGBNP_HIP_TIGHT_ROUNDS=0" 2 --system 2clr --steps 200 --warmup 20
echo "== 1dwc"; bash scripts/ab_env.sh "AGBNP_HIP_TIGHT_ROUNDS=0" 1
echo "== lattice"; bash scripts/ab_env.sh "AGBNP_HIP_TIGHT_ROUNDS=0" 1 --system 1dwc_x4 --steps 60 --warmup 6
python3 scripts/forest_probe.py 2clr 1dwc_x4 2>&1 | grep -v amdgpu
