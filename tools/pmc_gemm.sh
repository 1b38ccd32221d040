#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm.sh  -> SQ counters for the three GEMM layouts at the FFN1 shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in fwd dgrad wgrad; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/pmc_gemm_$k -o g -- python $R/tools/one_gemm.py $k "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for k in ("fwd","dgrad","wgrad"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(R+"/gpurun_out/pmc_gemm_%s/g_counter_collection.csv"%k)):
        if "gemm_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn,c in agg.items():
        d={n: sum(v)/len(v) for n,v in c.items()}
        wc=d.get("SQ_WAVE_CYCLES",1)
        print(k, "mfma_busy/busy_cycles=%.3f"%(d.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(d.get("SQ_BUSY_CYCLES",1),1)), {n: "%.3g"%v for n,v in d.items()},
              "wait_any/wave=%.2f wait_inst/wave=%.2f active/wave=%.2f lds_conf/lds_active=%.3f"%(d["SQ_WAIT_ANY"]/wc, d["SQ_WAIT_INST_ANY"]/wc, d["SQ_ACTIVE_INST_ANY"]/wc, d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1)))
PY
