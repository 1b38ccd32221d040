#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 1 8; do
for k in fwd wgrad; do
  SAM_GEMM_GROUP_M=$g rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc_l2_${k}_$g -o g -- python $R/tools/one_gemm.py $k "$@" > /dev/null 2>&1
done; done
python - <<PY
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for g in (1,8):
  for k in ("fwd","wgrad"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(R+"/gpurun_out/pmc_l2_%s_%d/g_counter_collection.csv"%(k,g))):
        if "gemm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d={n: sum(v)/len(v) for n,v in agg.items()}
    print("GROUP_M=%d %s"%(g,k), {n: "%.3g"%v for n,v in d.items()}, "hit rate %.3f"%(d["TCC_HIT_sum"]/max(d["TCC_HIT_sum"]+d["TCC_MISS_sum"],1)))
PY
