#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + the two PMC passes of the bench command, into gpurun_out/<tag>_{stats,FETCH,WRITE}
# usage: tools/profile_bench.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_$c -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_$c.log 2>&1
done
# MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy, summed over SIMDs) against GRBM_GUI_ACTIVE (kernel cycles)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_MFMA -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_MFMA.log 2>&1
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-160
