#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replicas.py -m gpu -q -k "one_round or ranks or rank" > gpurun_out/r5l_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5l_tests.log
