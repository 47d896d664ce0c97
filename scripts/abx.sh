#!/bin/bash
# GPU box: any number of (tag, library, environment) configurations alternated on ONE box (box-to-box differences are larger
# than the effects looked for).  Usage: scripts/abx.sh <rounds> "<tag>|<lib or ->|VAR=v VAR=v" ...   [BENCH_ARGS="..."]
rounds=$1; shift
mkdir -p gpurun_out
args=${BENCH_ARGS:---steps 300 --cpu-evals 0 --secondary 0}
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    IFS='|' read -r tag lib envs <<< "$spec"
    pre=""
    [ "$lib" != "-" ] && [ -n "$lib" ] && pre="AGBNP_HIP_LIBRARY=$lib"
    env $pre $envs timeout -k 10 240 python bench.py $args 2> gpurun_out/abx_${tag}_$i.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
if not lines:
    print('$tag', 'NO JSON LINE -- see gpurun_out/abx_${tag}_$i.err')
else:
    d = json.loads(lines[-1])
    print('$tag', round(d['ms_per_step'] * 1e3, 2), d['kernel_avg_us'], d.get('parity_on_sample', {}).get('max_abs_dF_kJmolnm'), 'forests', d['config'].get('forests'), 'level', d['config'].get('pack_level'), 'tries', [r.get('timed_tries') for r in d.get('ranks', [])])"
  done
done
