#!/bin/bash
# kernel trace of the graph-replayed training step: gpurun_out/<tag>_trace/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-roofline --no-secondary "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
tail -1 $R/gpurun_out/${tag}_trace.log | cut -c1-200
