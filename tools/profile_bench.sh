#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + the two PMC passes of the bench command, into gpurun_out/<tag>_{stats,FETCH,WRITE}
# usage: tools/profile_bench.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_$c -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_$c.log 2>&1
done
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-160
