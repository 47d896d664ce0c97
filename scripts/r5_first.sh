#!/bin/bash
# GPU box, round 5, first call: state of the suite on this round's box, the GB launch's bookkeeping role on plan / non-plan
# evaluations (stamped build), and the A/B that brackets the k_gb_tiles regression of VERDICT r04 item 2.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5a_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r5a_pytest.log
for every in 1 1000; do
  echo "---- pair timeline, AGBNP_HIP_REPLAN_EVERY=$every"
  AGBNP_HIP_REPLAN_EVERY=$every AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_pstamps.so timeout -k 10 200 python scripts/pair_timeline.py 1dwc > gpurun_out/r5a_timeline_$every.txt 2>&1
  grep -A1 -E "bookkeeping|k_gb_tiles|last end" gpurun_out/r5a_timeline_$every.txt | head -30
done
bash scripts/abx.sh 2 "head|-|" "old|build/diag/lib_ff74847.so|" "nofit|-|AGBNP_HIP_SPLIT_FIT=0" "noplan|-|AGBNP_HIP_REPLAN_EVERY=1000" "plan1|-|AGBNP_HIP_REPLAN_EVERY=1" 2>&1 | tee gpurun_out/r5a_abx.log
