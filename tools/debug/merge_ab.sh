#!/bin/bash
# A/B on one box: the MMT's last weight-gradient pair launched beside the tail + TextBert's own launch (0) against one launch of 20 problems (1)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for m in 0 1; do
    SAM_WGRAD_MERGE_TB=$m python bench.py --no-cpu-baseline --no-eager-baseline --no-secondary --steps 120 > gpurun_out/merge_$m.json 2> gpurun_out/merge_$m.err || tail -5 gpurun_out/merge_$m.err
    python - <<PY
import json
d=json.load(open("gpurun_out/merge_$m.json"))
r=d["roofline"]
print("merge=$m median %.3f ms mean %.3f | wgrad %d launches avg %.1f us %.1f TFLOP/s frac %.3f" % (d["ms_per_step_median"], d["ms_per_step"], r["launches_per_step"], r["avg_launch_us"], r["achieved"], r["frac"]))
PY
  done
done
