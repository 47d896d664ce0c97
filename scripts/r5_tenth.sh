#!/bin/bash
mkdir -p gpurun_out
AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_pstamps.so timeout -k 10 200 python scripts/rows_timeline.py 1dwc 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j_rows_timeline.txt
