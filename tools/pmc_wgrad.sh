#!/bin/bash
# usage (on the GPU box): tools/pmc_wgrad.sh  -> SQ counters of the grouped weight-gradient kernel (k loop only), per loop variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARS:-0}; do
  SAM_GEMM8W_VAR=$v SAM_GEMM8W_DBG=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/pmc_wgrad_a$v -o g -- python $R/tools/bench_wgrad.py > /dev/null 2>&1
  SAM_GEMM8W_VAR=$v SAM_GEMM8W_DBG=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
    --output-format csv -d $R/gpurun_out/pmc_wgrad_b$v -o g -- python $R/tools/bench_wgrad.py > /dev/null 2>&1
done
python - <<PY
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for v in os.environ.get("VARS", "0").split():
    d = {}
    for p in "ab":
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(R+"/gpurun_out/pmc_wgrad_%s%s/g_counter_collection.csv"%(p, v))):
            if "gemm8w_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        d.update({n: sum(x)/len(x) for n,x in agg.items()})
    wc=d.get("SQ_WAVE_CYCLES",1)
    print("VAR", v, {n: "%.4g"%x for n,x in sorted(d.items())})
    print("   wait_any/wave=%.2f wait_inst/wave=%.2f active/wave=%.2f lds_conflict/lds_active=%.3f mfma_busy/(busy*4)=%.3f" % (
        d["SQ_WAIT_ANY"]/wc, d["SQ_WAIT_INST_ANY"]/wc, d["SQ_ACTIVE_INST_ANY"]/wc, d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1),
        d["SQ_VALU_MFMA_BUSY_CYCLES"]/max(d["SQ_BUSY_CYCLES"],1)/4))
PY
