cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/beam_trace -o dec -- python $R/tools/debug/beam_prof.py > $R/gpurun_out/beam_trace.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/beam_trace/dec_kernel_stats.csv")))
for r in rows[:22]: print(r["Name"][:90].ljust(90), r["Calls"], round(float(r["TotalDurationNs"])/1e3/8), round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
