"""List the kernels of ONE training step (between the last two adam_kernel launches) from a rocprofv3 kernel trace:
    python tools/step_kernels.py gpurun_out/<run>_stats [--all]
prints per-name totals and, in launch order, every kernel that is not one of the hot-path kernels (what is left in torch glue)."""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
def short(n):
    for a in ("void ", "(anonymous namespace)::", "at::native::"):
        n = n.replace(a, "")
    return n[:100]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(dur(r) for r in step)
ours = ("gemm_", "gemm8", "gemm12", "copy_blocks", "enc_", "step_advance", "pack_masks", "ge_u8", "add_dropout", "embedding_bwd", "attn_", "ln_", "partial_finalize", "splitk", "adam", "sumsq", "bce_", "ptr_", "spatial_", "prefix_lm", "relation_", "embedding_bwd_kernel",
        "l2norm_pack", "embed_", "gather2", "cast_bf16", "colsum", "from_additive", "from_int8")
torch_us = sum(dur(r) for r in step if not any(h in r["Kernel_Name"] for h in ours))
print("%d kernels, %.1f us busy, span %.1f us; torch-native: %d kernels, %.1f us (%.1f%%)" % (
    len(step), tot, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3,
    sum(1 for r in step if not any(h in r["Kernel_Name"] for h in ours)), torch_us, 100 * torch_us / tot))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += dur(r)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:(None if "--all" in sys.argv else 25)]:
    print("%8.1f us %5.1f%% %3d  %s" % (v[1], 100 * v[1] / tot, v[0], k))
print("---- launch order, non-hot kernels")
run = 0
for i, r in enumerate(step):
    if any(h in r["Kernel_Name"] for h in ours):
        run += 1; continue
    if run: print("      ... %d hot" % run); run = 0
    print("%4d %6.1f  %s" % (i, dur(r), short(r["Kernel_Name"])))
if "--census" in sys.argv:
    # every node of ONE replayed step in launch order: index, start (us after the step's first kernel), duration, queue, kernel
    t0 = int(step[0]["Start_Timestamp"])
    print("---- census of one replayed step: %d kernels (of which %d copyBuffer / fillBuffer)" % (len(step), sum(1 for r in step if "rocclr" in r["Kernel_Name"])))
    for i, r in enumerate(step):
        print("%4d %9.1f %8.1f  q%-3s %s" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, dur(r), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
