#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "single or fast" > gpurun_out/r4r_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r4r_pytest.log
for m in fast fast+single; do
  for i in 1 2; do
  timeout -k 10 180 python bench.py --steps 300 --cpu-evals 3 --secondary 0 --mode $m 2> gpurun_out/r4r_$m.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')][-1])
print('$m', round(d['ms_per_step'] * 1e3, 2), d['kernel_avg_us'], d.get('parity_on_sample'))"
  done
done
