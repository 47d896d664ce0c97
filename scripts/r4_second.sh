#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/drift_probe.py 2000 0.001 > gpurun_out/r4b_probe.log 2>&1
echo "probe rc=$?"; tail -25 gpurun_out/r4b_probe.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4b_bench.err
python3 - <<PY
import json
for f in ("r4b_bench",):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r['n_gpus'], r.get('kernel_avg_us'), r.get('parity_on_sample'), r.get('cpu_baseline',{}).get('ms_per_eval'))
        print(' rows', r.get('neighbour_rows')); print(' drift', r.get('drift'))
        for s in r.get('secondary',[]): print(' ', s['config'], round(s['ms_per_eval'],4))
        for s in r.get('other_modes',[]): print(' ', s['mode'], round(s['ms_per_eval'],4))
    except Exception as e:
        print(f, 'no line', e)
PY
