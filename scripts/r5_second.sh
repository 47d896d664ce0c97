#!/bin/bash
# GPU box, round 5, second call: the N > 1 tests on hardware, the device-side reaction to a lone oversized subtree, the whole
# suite on the new status-word layout, then the bookkeeping role restructured (statistics without the histogram on non-plan
# evaluations, straight-line parts_of, replan every 16 + drift trigger) against the round-4 library.
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_replicas.py -m gpu -q -x > gpurun_out/r5b_replicas.log 2>&1
echo "replicas rc=$?"; tail -5 gpurun_out/r5b_replicas.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "heals_on_the_device or big_subtrees or second_larger" > gpurun_out/r5b_heal.log 2>&1
echo "heal rc=$?"; tail -5 gpurun_out/r5b_heal.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5b_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5b_pytest.log
for every in 1 16; do
  echo "---- pair timeline, AGBNP_HIP_REPLAN_EVERY=$every"
  AGBNP_HIP_REPLAN_EVERY=$every AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_pstamps.so timeout -k 10 200 python scripts/pair_timeline.py 1dwc > gpurun_out/r5b_timeline_$every.txt 2>&1
  grep -A1 -E "bookkeeping" gpurun_out/r5b_timeline_$every.txt | head -8
  grep -E "last end" gpurun_out/r5b_timeline_$every.txt
done
bash scripts/abx.sh 2 "r4|build/diag/lib_r4.so|" "new|-|" "new4|-|AGBNP_HIP_REPLAN_EVERY=4" "plan1|-|AGBNP_HIP_REPLAN_EVERY=1" 2>&1 | tee gpurun_out/r5b_abx.log
