for i in 1 2 3; do
  for v in fused launch; do
    if [ $v = launch ]; then export AGBNP_HIP_OUTPUT_LAUNCH=1; else unset AGBNP_HIP_OUTPUT_LAUNCH; fi
    timeout -k 10 180 python bench.py --steps 300 --cpu-evals 0 --secondary 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v', round(d['ms_per_step']*1e3,2), d['kernel_avg_us'])"
  done
done
