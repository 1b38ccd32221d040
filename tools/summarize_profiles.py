#!/usr/bin/env python3
"""Turn rocprofv3 output directories under gpurun_out/ into the small summaries committed under profiles/.

  python tools/summarize_profiles.py <tag> <kernel_stats_dir> [<pmc_FETCH_SIZE_dir> <pmc_WRITE_SIZE_dir>] [--steps N]

* profiles/<tag>_kernel_stats.csv : rocprofv3 --kernel-trace --stats summary, verbatim
* profiles/<tag>_pmc_traffic.json : per-kernel HBM bytes per launch from the PMC passes, corrected as
  /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are in KiB;
  FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so reads = 2 * FETCH_SIZE * 1024
  (calibrated in this run on cast_bf16_kernel / adam_kernel / sumsq_partial_kernel, whose byte counts are known
  exactly: see the `calibration` block); writes = WRITE_SIZE * 1024 (matches the known counts as is).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(dirname, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(dirname, "bench_counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(os.path.join(stats_dir, "bench_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    if len(sys.argv) >= 5 and not sys.argv[3].startswith("--"):
        f, w = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        out = {"units": "bytes per launch (mean over launches); read = 2*FETCH_SIZE*1024, write = WRITE_SIZE*1024", "kernels": {}}
        for k in f:
            rd = 2.0 * 1024 * f[k][1] / f[k][0]
            wr = 1024.0 * w[k][1] / max(w[k][0], 1)
            out["kernels"][k] = {"launches_profiled": f[k][0], "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
        known = {"cast_bf16_kernel": (4, 2), "sumsq_partial_kernel": (4, 0), "adam_kernel": (16, 14)}
        cal = {}
        for k, (rb, wb) in known.items():
            if k in out["kernels"]:
                n = out["kernels"]["cast_bf16_kernel"]["hbm_write_bytes"] / 2.0 if "cast_bf16_kernel" in out["kernels"] else None
                if n:
                    cal[k] = {"expected_read": round(rb * n), "measured_read": out["kernels"][k]["hbm_read_bytes"],
                              "expected_write": round(wb * n), "measured_write": out["kernels"][k]["hbm_write_bytes"]}
        out["calibration"] = cal
        json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    mfma_dir = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--mfma=")]
    if mfma_dir:
        # MFMA utilisation per kernel = busy cycles of the matrix pipes (SQ_VALU_MFMA_BUSY_CYCLES, summed over all 1024 SIMDs; one 16x16x32
        # bf16 MFMA = 16 busy cycles, 1024 flop/cycle/SIMD = the 2.5 PFLOP/s dense peak at 2.4 GHz) over the SIMD-cycles the kernel had.
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked: 8 x duration x 2.4 GHz), so kernel cycles = GRBM_GUI_ACTIVE / 8.
        cs = {c: pmc(mfma_dir[0], c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
        out = {"units": "per launch means; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); wait/issue fractions are of SQ_WAVE_CYCLES", "kernels": {}}
        for k, (n, busy) in cs["SQ_VALU_MFMA_BUSY_CYCLES"].items():
            gui = cs["GRBM_GUI_ACTIVE"][k][1] / max(cs["GRBM_GUI_ACTIVE"][k][0], 1)
            wc = cs["SQ_WAVE_CYCLES"][k][1] / max(cs["SQ_WAVE_CYCLES"][k][0], 1)
            e = {"launches_profiled": n, "mfma_busy_cycles": round(busy / n), "gui_active_cycles": round(gui), "mfma_util": round(busy / n / max(gui / 8 * 1024, 1), 4)}
            if wc:
                for c, key in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                    e[key] = round(cs[c][k][1] / max(cs[c][k][0], 1) / wc, 3)
            if busy or "gemm" in k or "attn" in k:
                out["kernels"][k] = e
        json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_mfma.json"), "w"), indent=1, sort_keys=True)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
