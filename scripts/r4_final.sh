#!/bin/bash
# GPU box: what the driver runs at round end -- all GPU tests, the smoke test, the bench line -- plus the two-rank rehearsal.
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r4z_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r4z_pytest.log
timeout -k 10 120 python3 __graft_entry__.py smoke > gpurun_out/r4z_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r4z_smoke.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err
echo "bench rc=$?"
AGBNP_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4z_g2.json 2> gpurun_out/r4z_g2.err
echo "g2 rc=$?"
python3 - <<PY
import json
for f in ("r4z_bench","r4z_g2"):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r['n_gpus'], r.get('launcher'), r.get('collectives'), r.get('kernel_avg_us'), round(r['roofline']['frac'],3), r.get('cpu_baseline',{}).get('ms_per_eval'))
        if 'drift' in r: print(' drift', r['drift']['ms_per_eval'], ' rebuild', r['neighbour_rows'].get('rebuild_eval_ms'))
        for s in r.get('secondary',[]): print(' ', s['config'], round(s['ms_per_eval'],4))
        for s in r.get('other_modes',[]): print(' ', s['mode'], round(s['ms_per_eval'],4))
    except Exception as e:
        print(f, 'no line', e)
PY
