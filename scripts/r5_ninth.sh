#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python scripts/pack_probe.py 2clr 6 220 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5i_probe_2clr.log
bash scripts/abx.sh 2 "1dwc_prep_lean|-|" 2>&1 | tee gpurun_out/r5i_abx_1dwc.log
