#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r4k_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r4k_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
bash scripts/abn.sh "build/diag/lib_base.so openmm_agbnp_plugin_amd/libagbnp_hip.so" 2 2>&1 | tee gpurun_out/r4k_abn.log
bash scripts/abn.sh "build/diag/lib_base.so openmm_agbnp_plugin_amd/libagbnp_hip.so" 2 --system 1dwc_x4 --steps 60 --warmup 6 2>&1 | tee gpurun_out/r4k_abn_x4.log
echo "== lattice, forces through k_outputs (AGBNP_HIP_FUSE_QUEUED=0) against the default"
bash scripts/ab_env.sh "AGBNP_HIP_FUSE_QUEUED=0" 2 --system 1dwc_x4 --steps 60 --warmup 6 2>&1 | tee gpurun_out/r4k_fuseq.log
