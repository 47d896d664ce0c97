#!/bin/bash
# GPU box, round 5, third call: the three tests repaired after the second call, then the whole suite.
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_replicas.py -m gpu -q > gpurun_out/r5c_replicas.log 2>&1
echo "replicas rc=$?"; tail -3 gpurun_out/r5c_replicas.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5c_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5c_pytest.log
