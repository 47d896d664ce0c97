#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r5r_durations.log 2>&1
echo "rc=$?"; tail -40 gpurun_out/r5r_durations.log
