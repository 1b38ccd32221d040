#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm12.sh  -> SQ counters of the 12-wave and 8-wave kernels at [11648, 768, 3072] (192 x 192 tile, EPI_NONE) and [11648, 3072, 768] (192 x 256)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "12 768 3072" "8 768 3072" "12 3072 768" "8 3072 768"; do
  set -- $cfg
  SAM_GEMM12=$([ $1 = 12 ] && echo 1 || echo 0) rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/pmc_gemm12_$1_$2 -o g -- python $R/tools/one_gemm.py fwd $2 $3 > /dev/null 2>&1
  SAM_GEMM12=$([ $1 = 12 ] && echo 1 || echo 0) rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
    --output-format csv -d $R/gpurun_out/pmc_gemm12b_$1_$2 -o g -- python $R/tools/one_gemm.py fwd $2 $3 > /dev/null 2>&1
done
python - <<PY
import csv, collections, os, glob
R=os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R+"/gpurun_out/pmc_gemm12*_*")):
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::","")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn,c in agg.items():
        dd={n: sum(v)/len(v) for n,v in c.items()}
        print(os.path.basename(d), kn, {n: "%.4g"%v for n,v in sorted(dd.items())})
PY
