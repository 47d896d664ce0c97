#!/bin/bash
# GPU box: the tests named by $2 (pytest -k expression, "" = all), then a short bench line.  Usage: scripts/gpu_quick.sh <tag> [k-expr] [bench args]
tag=$1; kexpr=$2; shift; shift
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1
else
  timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
fi
rc=$?
tail -12 gpurun_out/${tag}_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: no bench"; exit $rc; fi
timeout -k 10 300 python bench.py --secondary 0 --cpu-evals 3 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
    print(round(r['ms_per_step'],5), round(r['value'],1), r.get('kernel_avg_us'), r.get('parity_on_sample'))
except Exception as e:
    print('no bench line', e)
PY
