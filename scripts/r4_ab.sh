#!/bin/bash
# GPU box: parity tests on an experimental library, then the A/B of scripts/abn.sh.  Usage: scripts/r4_ab.sh <tag> "<libs>" [rounds] [test lib]
tag=$1; libs=$2; rounds=${3:-2}; testlib=$4
mkdir -p gpurun_out
if [ -n "$testlib" ]; then
  AGBNP_HIP_LIBRARY=$testlib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1
  rc=$?; tail -8 gpurun_out/${tag}_pytest.log
  if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
fi
bash scripts/abn.sh "$libs" $rounds 2>&1 | tee gpurun_out/${tag}_abn.log
