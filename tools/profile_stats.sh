#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of the bench command only -> gpurun_out/<tag>_stats (see profile_bench.sh for the PMC passes)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-eager-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_stats.log 2>&1
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-160
