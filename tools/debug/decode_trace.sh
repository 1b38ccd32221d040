cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_decode_trace -o dec -- python $R/tools/debug/decode_prof.py > $R/gpurun_out/r3z_decode_trace.log 2>&1
tail -2 $R/gpurun_out/r3z_decode_trace.log | cut -c1-200
