#!/bin/bash
# THIS container: writes the git head (+ "-dirty" when the engine's sources differ from it) to profiles/.git_head, where it
# travels with the snapshot, then runs scripts/profile_round.sh <tag> on a GPU box; the summaries come back under
# gpurun_out/prof_<tag>/ and are copied into profiles/<tag>/.   Usage: scripts/profile_here.sh r05
set -e
tag=${1:-r05}
cd "$(dirname "$0")/.."
head=$(git rev-parse --short=12 HEAD)
[ -n "$(git status --porcelain -- openmm_agbnp_plugin_amd/csrc include bench.py)" ] && head="$head-dirty"
echo "$head" > profiles/.git_head
make -C openmm_agbnp_plugin_amd/csrc > /dev/null
/usr/local/graft/bin/gpurun --timeout 1100 -- "bash scripts/profile_round.sh $tag" > gpurun_out/profile_${tag}_call.log 2>&1 || true
tail -5 gpurun_out/profile_${tag}_call.log
mkdir -p profiles/$tag
for f in bench_1dwc_kernel_stats.csv bench_1dwc_kernel_outliers.txt bench_1dwc_line_under_rocprof.json pmc_summary.csv pmc_utilization.csv profile_head.json; do
  [ -f gpurun_out/prof_$tag/$f ] && cp gpurun_out/prof_$tag/$f profiles/$tag/$f
done
[ -f gpurun_out/prof_$tag/traffic_pmc.json ] && cp gpurun_out/prof_$tag/traffic_pmc.json profiles/traffic_pmc.json
python3 scripts/resource_table.py --out profiles/$tag/resource_table.csv > /dev/null 2>&1 || true
ls -la profiles/$tag
