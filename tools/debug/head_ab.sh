#!/bin/bash
# A/B on one box: the classifier's / pointer net's weight gradients as their own GEMMs (0) against riding on the first MMT pair launch (1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in 0 1; do
    SAM_DEFER_HEAD_WGRAD=$m python bench.py --no-cpu-baseline --no-eager-baseline --no-secondary --no-roofline --steps 120 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head_defer=$m median %.3f mean %.3f loss %.3f' % (d['ms_per_step_median'], d['ms_per_step'], d['final_loss']))"
  done
done
