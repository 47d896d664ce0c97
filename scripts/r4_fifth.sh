#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r4j_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r4j_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
bash scripts/abn.sh "build/diag/lib_base.so openmm_agbnp_plugin_amd/libagbnp_hip.so" 2 2>&1 | tee gpurun_out/r4j_abn.log
bash scripts/abn.sh "build/diag/lib_base.so openmm_agbnp_plugin_amd/libagbnp_hip.so" 2 --system 1dwc_x4 --steps 60 --warmup 6 2>&1 | tee gpurun_out/r4j_abn_x4.log
echo "== GB light items first"
bash scripts/ab_env.sh "AGBNP_HIP_GB_LIGHT_FIRST=1" 3 2>&1 | tee gpurun_out/r4j_gbfirst.log
