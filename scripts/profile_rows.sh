#!/bin/bash
# GPU box: rocprofv3 kernel statistics and SQ counters of the other configurations: fast mode (row form incl. GB rows) and
# the reference mode on the tile kernels (AGBNP_HIP_ROWS=0).  Usage: scripts/profile_rows.sh r03
set -e
tag=${1:-r03}
out=gpurun_out/prof_rows_$tag
mkdir -p $out profiles/$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for cfg in fast tiles; do
  if [ $cfg = fast ]; then args="--mode fast"; unset AGBNP_HIP_ROWS; else args=""; export AGBNP_HIP_ROWS=0; fi
  BENCH="python3 bench.py --steps 100 --warmup 10 --cpu-evals 0 --secondary 0 $args"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -o bench -- $BENCH > $out/stats_$cfg.log 2>&1
  cp "$(find $out/stats_$cfg -name '*kernel_stats.csv' | head -1)" profiles/$tag/${cfg}_1dwc_kernel_stats.csv
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    -d $out/pmc_$cfg -o bench -- $BENCH > $out/pmc_$cfg.log 2>&1
  python3 scripts/summarize_sq.py "$(find $out/pmc_$cfg -name '*counter_collection.csv' | head -1)" profiles/$tag/${cfg}_pmc_utilization.csv > /dev/null
  echo "$cfg done"
done
cp profiles/$tag/*.csv $out/
