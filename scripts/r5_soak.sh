#!/bin/bash
# GPU box: the whole GPU suite in the default (five-launch) mode and again with AGBNP_HIP_FIVE_LAUNCHES=0, then the driver's bench
# line and a 300-step line.
tag=${1:-r5n}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_five.log 2>&1
echo "suite (default: five launches) rc=$?"; tail -6 gpurun_out/${tag}_pytest_five.log
AGBNP_HIP_FIVE_LAUNCHES=0 timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_five_launches.py > gpurun_out/${tag}_pytest_six.log 2>&1
echo "suite (six launches) rc=$?"; tail -6 gpurun_out/${tag}_pytest_six.log
timeout -k 10 120 python3 __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$?"; tail -1 gpurun_out/${tag}_smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench.err
timeout -k 10 300 python3 bench.py --steps 300 --cpu-evals 3 --secondary 0 > gpurun_out/${tag}_bench300.json 2> gpurun_out/${tag}_bench300.err
python3 - <<PY
import json
for f in ("${tag}_bench","${tag}_bench300"):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r.get('kernel_avg_us'), r.get('launches_per_evaluation'), r.get('parity_on_sample'))
        if 'drift' in r: print(' drift', {k:r['drift'][k] for k in ('ms_per_eval','builds_in_timed_region','forest_plans_in_timed_region','withheld_evaluations')})
        if 'neighbour_rows' in r: print(' rebuild', r['neighbour_rows'].get('rebuild_eval_ms'))
        for s in r.get('secondary',[]): print(' ', s['config'][:60], round(s['ms_per_eval'],4), s.get('parity_on_sample'))
        for s in r.get('other_modes',[]): print(' ', s['mode'], round(s['ms_per_eval'],4))
        for k in ('concurrent_replicas_on_one_gpu','openmm_entry','md_loop'):
            if k in r: print(' ', k, r[k])
    except Exception as e:
        print(f, 'no line', e)
PY
