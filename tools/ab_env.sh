#!/bin/bash
# alternate bench.py runs with an environment switch off / on (same box):  tools/ab_env.sh SAM_GEMM12W [steps] [extra bench args]
VAR=$1; STEPS=${2:-40}; shift; shift
for v in 0 1 0 1; do
  env $VAR=$v python bench.py --steps $STEPS --warmup 8 --no-secondary --no-eager-baseline --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('$VAR=$v  %.1f samples/s  %.3f ms mean  median %s  dominant %s: %s us, frac %s' % (d['value'], d['ms_per_step'], d.get('ms_per_step_median'), r.get('kernel', '?')[:40], r.get('avg_us'), r.get('frac')))
"
done
