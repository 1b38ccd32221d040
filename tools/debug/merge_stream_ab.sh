#!/bin/bash
# A/B on one box: the launch of 20 weight-gradient problems on the issuing stream (0) against a stream of its own beside TextBert's embedding backward (1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in 0 1; do
    SAM_WGRAD_MERGE_STREAM=$m python bench.py --no-cpu-baseline --no-eager-baseline --no-secondary --no-roofline --steps 120 > gpurun_out/ms_$m.json 2> gpurun_out/ms_$m.err || tail -5 gpurun_out/ms_$m.err
    python - <<PY
import json
d=json.load(open("gpurun_out/ms_$m.json"))
print("merge_stream=$m median %.3f ms mean %.3f" % (d["ms_per_step_median"], d["ms_per_step"]))
PY
  done
done
