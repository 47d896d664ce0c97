#!/bin/bash
# GPU box: several builds of the engine alternated on ONE box (see scripts/ab.sh).  Usage: scripts/abn.sh "<lib> <lib> ..." [rounds] [bench args]
libs=$1; rounds=${2:-2}; shift; shift
mkdir -p gpurun_out
for i in $(seq 1 $rounds); do
  for lib in $libs; do
    tag=$(basename $lib .so)
    AGBNP_HIP_LIBRARY=$lib timeout -k 10 180 python bench.py --steps 300 --cpu-evals 0 --secondary 0 "$@" 2> gpurun_out/abn_${tag}_$i.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
if not lines:
    print('$tag', 'NO JSON LINE -- see gpurun_out/abn_${tag}_$i.err')
else:
    d = json.loads(lines[-1])
    print('$tag', round(d['ms_per_step'] * 1e3, 2), d['kernel_avg_us'], d.get('parity_on_sample', {}).get('max_abs_dF_kJmolnm'))"
    tail -2 gpurun_out/abn_${tag}_$i.err | grep -v amdgpu.ids
  done
done
