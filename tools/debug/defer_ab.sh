#!/bin/bash
# A/B of the deferred TextBert weight gradients (SAM_DEFER_TB_WGRAD): ms per replayed step, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do SAM_DEFER_TB_WGRAD=$v python bench.py --steps 60 --warmup 15 --no-secondary --no-cpu-baseline --no-eager-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFER','$v', d['ms_per_step'], d['value'])"; done; done
