#!/bin/bash
# GPU box: the SAME build under two environments, alternated on one box.  Usage: scripts/ab_env.sh "VAR=value" [rounds] [bench args]
setting=$1; rounds=${2:-3}; shift; shift
mkdir -p gpurun_out
for i in $(seq 1 $rounds); do
  for v in default "$setting"; do
    if [ "$v" = default ]; then pre=""; else pre="$v"; fi
    env $pre timeout -k 10 180 python bench.py --steps 300 --cpu-evals 0 --secondary 0 "$@" 2> gpurun_out/abenv_$i.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
if not lines:
    print('$v', 'NO JSON LINE')
else:
    d = json.loads(lines[-1])
    print('$v', round(d['ms_per_step'] * 1e3, 2), d['kernel_avg_us'])"
  done
done
