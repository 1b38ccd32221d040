#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_attn1 -o a -- python $R/tools/bench_attn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_attn2 -o a -- python $R/tools/bench_attn.py > /dev/null 2>&1
python - <<PY
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_attn1","pmc_attn2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(R+"/gpurun_out/%s/a_counter_collection.csv"%d)):
        if "attn_" in r["Kernel_Name"]:
            kn = r["Kernel_Name"]
            kn = kn[kn.index("attn_"):].split("(")[0][:44]
            agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn,c in agg.items():
        print(kn, {n: "%.3g"%(sum(v)/len(v)) for n,v in c.items()})
PY
