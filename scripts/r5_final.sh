#!/bin/bash
# GPU box: what the driver runs at round end -- all GPU tests, the smoke test, the bench line (driver's command line) -- plus a
# 300-step line and the two-rank rehearsal.
tag=${1:-r5z}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log
timeout -k 10 120 python3 __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 300 --cpu-evals 0 --secondary 0 > gpurun_out/${tag}_bench300.json 2> gpurun_out/${tag}_bench300.err
AGBNP_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${tag}_g2.json 2> gpurun_out/${tag}_g2.err
echo "g2 rc=$?"
python3 - <<PY
import json
for f in ("${tag}_bench","${tag}_bench300","${tag}_g2"):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r['n_gpus'], r.get('launcher'), r.get('collectives'), r.get('kernel_avg_us'), round(r['roofline']['frac'],3), r.get('cpu_baseline',{}).get('ms_per_eval'), r['roofline'].get('profile_head'))
        if 'drift' in r: print(' drift', r['drift'])
        if 'neighbour_rows' in r: print(' rebuild', r['neighbour_rows'].get('rebuild_eval_ms'))
        for s in r.get('secondary',[]): print(' ', s['config'][:60], round(s['ms_per_eval'],4), s.get('parity_on_sample'))
        for s in r.get('other_modes',[]): print(' ', s['mode'], round(s['ms_per_eval'],4))
        for k in ('concurrent_replicas_on_one_gpu','openmm_entry','md_loop'):
            if k in r: print(' ', k, r[k])
    except Exception as e:
        print(f, 'no line', e)
PY
