#!/bin/bash
# GPU box, round 4, first call: all GPU tests, the driver's bench command line (full line incl. rebuild + drift records), the
# two-rank rehearsal of the self-launcher (gloo: both ranks share the one GPU).
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4a_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r4a_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
echo "bench rc=$?"
AGBNP_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4a_g2.json 2> gpurun_out/r4a_g2.err
echo "g2 rc=$?"
python3 - <<PY
import json
for f in ("r4a_bench","r4a_g2"):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r['n_gpus'], r.get('kernel_avg_us'), r.get('parity_on_sample'), r.get('cpu_baseline',{}).get('ms_per_eval'))
        print(' rows', r.get('neighbour_rows')); print(' drift', r.get('drift'))
        for s in r.get('secondary',[]): print(' ', s['config'], round(s['ms_per_eval'],4))
    except Exception as e:
        print(f, 'no line', e)
PY
