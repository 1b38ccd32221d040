#!/usr/bin/env python3
"""CU head-room under RCCL, emulated on ONE GPU (VERDICT r5 missing #3).

At N > 1 RCCL's channel kernels hold CUs beside the backward pass.  Every hot kernel of the step is persistent with one block per CU, so blocks that find their CU
taken form a second launch round.  Emulation: bench.py in a 1-rank RCCL group (SAM_FORCE_DIST=1: reducer, buckets released at their finality marks on the
reducer's stream, all of it inside the captured step) with SAM_EMULATE_COMM="<channels>:<GB/s>": behind every (identity) collective the reducer launches, on the
same stream, <channels> workgroups that hold one CU each for bytes / <GB/s> (sam_debug_cu_hog) -- what an N-GPU all-reduce of that bucket would occupy.  Against it:
SAM_DP_CU_RESERVE = CUs the backward's persistent grids leave free (Trainer; the forward keeps the chip).

    python tools/bench_cu_reserve.py [--comm none,16:150,32:150,32:75] [--reserves 0,16,32] [--steps 40]

(An earlier form of this tool ran a hog on a side stream beside replays of the plain captured step.  It measured the runtime's queues, not the CUs: on a
normal-priority stream the replays simply waited for the hog to end (shared hardware queue); on a high-priority stream EVERY kernel of the step started ~20 us
late while the hog was resident (ln_fwd 8 -> 25 us, a 4 us element-wise add -> 28 us); with GPU_MAX_HW_QUEUES=16 the plain step itself took 11 ms.  The eager
single-kernel probe tools/debug/hog_probe.py is free of all three and shows the effect itself: an MMT-size GEMM 56 -> 82 us beside 32 held CUs, 67 us with 32 withheld.)"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(comm, reserve, steps):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SAM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               SAM_DP_CU_RESERVE=str(reserve))
    env.pop("SAM_EMULATE_COMM", None)
    if comm != "none":
        env["SAM_EMULATE_COMM"] = comm
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "8", "--no-eager-baseline", "--no-cpu-baseline", "--no-roofline",
           "--no-secondary"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        return None, (r.stderr or r.stdout)[-400:]
    d = json.loads(line[-1])
    assert d.get("cu_reserved") == reserve and d.get("step_mode") == "hipGraph replay", (d.get("cu_reserved"), d.get("step_mode"))
    return d, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--comm", default="none,16:150,32:150,32:75")
    ap.add_argument("--reserves", default="0,16,32")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    comms, reserves = args.comm.split(","), [int(x) for x in args.reserves.split(",")]
    print("# bench.py, c3 B = 64, 1-rank RCCL group, captured data-parallel step; ms per step = median GPU time between per-step events over %d replays" % args.steps)
    print("# columns: emulated collectives (<channels>:<GB/s>; 292 MB leave per step in 64 MB buckets); rows: CUs the backward's persistent grids leave free")
    print("%-10s" % "reserve" + "".join("%-12s" % c for c in comms), flush=True)
    table = {}
    for r in reserves:
        row = []
        for c in comms:
            d, err = run(c, r, args.steps)
            if d is None:
                print("# reserve %d comm %s FAILED: %s" % (r, c, err))
                row.append(float("nan"))
                continue
            table[(r, c)] = d["ms_per_step_median"]
            row.append(d["ms_per_step_median"])
        print("%-10d" % r + "".join("%-12.3f" % v for v in row), flush=True)
    base = table.get((0, "none"))
    if base:
        print("# relative to reserve 0 / no emulated collectives (%.3f ms):" % base)
        for r in reserves:
            print("%-10d" % r + "".join("%-12.3f" % (table.get((r, c), float("nan")) / base) for c in comms))


if __name__ == "__main__":
    main()
