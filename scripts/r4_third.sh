#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4c_pytest.log 2>&1
rc=$?; tail -40 gpurun_out/r4c_pytest.log
timeout -k 10 300 python3 scripts/drift_probe.py 2000 0.001 > gpurun_out/r4c_probe.log 2>&1
echo "probe rc=$?"; tail -22 gpurun_out/r4c_probe.log
