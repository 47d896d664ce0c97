#!/bin/bash
# GPU box: rocprofv3 evidence for one round.  Usage: scripts/profile_round.sh r02
# Writes raw output under gpurun_out/prof_<tag>/ (scratch); the summaries go to profiles/<tag>/ by scripts/summarize_*.py.
# Counter passes are separate runs with --kernel-trace only (the pool refuses --pmc beside the trace domains).
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out profiles/$tag
# what these summaries were measured on: the library's own build id (SHA-256 of its sources, csrc/Makefile) and the git head
# that scripts/profile_here.sh wrote beside the snapshot (the GPU box has no .git)
python3 - <<PY > profiles/$tag/profile_head.json
import json, os, time
from openmm_agbnp_plugin_amd import _lib
head = open("profiles/.git_head").read().strip() if os.path.exists("profiles/.git_head") else None
print(json.dumps({"library_build_id": _lib.build_id(), "git_head": head, "taken": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                  "command": "python3 bench.py --steps 100 --warmup 10 --cpu-evals 0 --secondary 0 under rocprofv3 (scripts/profile_round.sh)"}, indent=1))
PY
cat profiles/$tag/profile_head.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 100 --warmup 10 --cpu-evals 0 --secondary 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $BENCH > $out/stats_bench.log 2>&1
echo "stats pass done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/pmc_$c -o bench -- $BENCH > $out/pmc_$c.log 2>&1
  echo "pmc $c done"
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  -d $out/pmc_sq -o bench -- $BENCH > $out/pmc_sq.log 2>&1
echo "pmc sq done"
find $out -name "*.csv" | head -40
stats=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp "$stats" profiles/$tag/bench_1dwc_kernel_stats.csv
# which launches are the maxima of that summary: from the kernel trace of the same run
trace=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_outliers.py "$trace" 3 > profiles/$tag/bench_1dwc_kernel_outliers.txt 2>&1 || true
grep '^{"metric"' $out/stats_bench.log | tail -1 > profiles/$tag/bench_1dwc_line_under_rocprof.json
fetch=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
write=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 scripts/summarize_pmc.py "$fetch" "$write" $tag 1dwc k_tree_cavity
sq=$(find $out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 scripts/summarize_sq.py "$sq" profiles/$tag/pmc_utilization.csv
cp profiles/$tag/*.csv profiles/$tag/*.json profiles/$tag/*.txt $out/ 2>/dev/null || true
cp profiles/traffic_pmc.json $out/
