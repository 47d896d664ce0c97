#!/bin/bash
# GPU box, round 4: rocprofv3 evidence (kernel stats, HBM bytes, SQ counters) of the default configuration, the one-rank
# rehearsal of the RCCL plumbing, and the driver's bench command line.
mkdir -p gpurun_out
bash scripts/profile_round.sh r04 > gpurun_out/r4p_profile.log 2>&1
echo "profile rc=$?"; tail -5 gpurun_out/r4p_profile.log
AGBNP_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --secondary 0 --cpu-evals 2 > gpurun_out/r4p_rccl1.json 2> gpurun_out/r4p_rccl1.err
echo "rccl one-rank rc=$?"; tail -2 gpurun_out/r4p_rccl1.err
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4p_bench.json 2> gpurun_out/r4p_bench.err
echo "bench rc=$?"
python3 - <<PY
import json
for f in ("r4p_rccl1","r4p_bench"):
    try:
        r=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r['ms_per_step'],5), round(r['value'],1), r['n_gpus'], r.get('collectives'), r.get('kernel_avg_us'), r.get('roofline',{}).get('frac'))
        if 'drift' in r: print(' drift', {k:v for k,v in r['drift'].items() if 'note' not in k})
        for s in r.get('secondary',[]): print(' ', s['config'], round(s['ms_per_eval'],4))
    except Exception as e:
        print(f, 'no line', e)
PY
