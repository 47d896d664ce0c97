#!/bin/bash
# GPU box: A/B of two builds of the engine on ONE box (box-to-box differences are 1-1.5 us per kernel, within a box a
# kernel's mean repeats to 0.1 us): the libraries alternate, three runs each.  stderr is kept beside the numbers.
#   scripts/ab.sh build/diag/libagbnp_hip_A.so build/diag/libagbnp_hip_B.so [bench args]
A=$1; B=$2; shift; shift
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    AGBNP_HIP_LIBRARY=$lib timeout -k 10 180 python bench.py --steps 300 --cpu-evals 0 --secondary 0 "$@" 2> gpurun_out/ab_${v}_$i.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
if not lines:
    print('$v', 'NO JSON LINE -- see gpurun_out/ab_${v}_$i.err')
else:
    d = json.loads(lines[-1])
    print('$v', round(d['ms_per_step'] * 1e3, 2), d['kernel_avg_us'])"
    tail -2 gpurun_out/ab_${v}_$i.err | grep -v amdgpu.ids
  done
done
