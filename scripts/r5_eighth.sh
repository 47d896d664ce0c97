#!/bin/bash
# GPU box, round 5, eighth call: whole suite; 2clr classes / rounds rule (forces fused into the replay when the forests fit one
# round); what k_prep is made of (timing builds: empty launch / no mask tiles / mask tiles alone; raw event intervals).
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5h_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5h_pytest.log
BENCH_ARGS="--system 2clr --steps 200 --warmup 20 --cpu-evals 2 --secondary 0" bash scripts/abx.sh 2 "2clr_classes|build/diag/lib_norounds.so|" "2clr_rounds|-|" 2>&1 | tee gpurun_out/r5h_abx_2clr.log
for v in full prep1 prep2 prep3; do
  lib=build/diag/lib_$v.so; [ $v = full ] && lib=openmm_agbnp_plugin_amd/libagbnp_hip.so
  AGBNP_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 300 --cpu-evals 0 --secondary 0 2> gpurun_out/r5h_$v.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
if lines:
    d = json.loads(lines[-1]); print('$v', round(d['ms_per_step'] * 1e3, 2), 'raw event intervals', d['kernel_event_us'])
else:
    print('$v no line')"
done 2>&1 | tee gpurun_out/r5h_prep_anatomy.log
