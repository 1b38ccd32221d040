#!/bin/bash
# A/B of the late weight-gradient stream (SAM_WGRAD_LATE) + a kernel trace with the HIP API calls beside it
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-eager-baseline --no-roofline --no-secondary"
for i in 1 2; do
  for v in 0 1; do
    SAM_WGRAD_LATE=$v $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('late=$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/late_trace -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-eager-baseline --no-roofline --no-secondary > $R/gpurun_out/late_trace.log 2>&1
ls $R/gpurun_out/late_trace
