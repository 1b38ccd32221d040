#!/bin/bash
# A/B of HIP runtime knobs on the captured training step (bench.py, c3 B = 64): median ms per step of 60 replays each, two passes
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; env $1 python bench.py --steps 60 --warmup 8 --no-eager-baseline --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_median'], d['ms_per_step'])"; }
for pass in 1 2; do
  run "SAM_NOOP=1"
  run "HIP_FORCE_DEV_KERNARG=1"
  run "HIP_FORCE_DEV_KERNARG=0"
  run "GPU_MAX_HW_QUEUES=2"
  run "GPU_MAX_HW_QUEUES=6"
  run "GPU_MAX_HW_QUEUES=8"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
  run "HSA_ENABLE_SDMA=0"
done
