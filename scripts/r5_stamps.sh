#!/bin/bash
# GPU box: per-phase cycles of a forest (tree-stamped build) with five and with six launches
mkdir -p gpurun_out
for five in 1 0; do
  echo "==== AGBNP_HIP_FIVE_LAUNCHES=$five"
  AGBNP_HIP_FIVE_LAUNCHES=$five AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_stamps.so timeout -k 10 200 python scripts/stamps.py 1dwc 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5p_stamps.txt
