#!/bin/bash
# the driver's bench command N times on one box, one line each: how often does a run contain a stalled step?   tools/bench_repeat.sh [N] [steps] [warmup]
N=${1:-6}; STEPS=${2:-20}; W=${3:-5}
for i in $(seq $N); do
  python bench.py --gpus 1 --steps $STEPS --warmup $W --no-secondary --no-eager-baseline --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i: %.1f samples/s  mean %.3f ms  median %s  p10/p90/max %s  slow %s  first (gpu, host) %s' % (d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d.get('ms_per_step_p10_p90_max'), d.get('slow_steps'), d.get('first_steps_gpu_host_ms')))
"
done
