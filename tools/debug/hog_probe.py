#!/usr/bin/env python3
"""Does sam_debug_cu_hog really take CUs away from a persistent kernel?  (a) how long the hog runs; (b) one MMT-size GEMM (grid = CUs) timed alone and beside
hogs of 32 / 64 / 128 blocks on another stream, with the persistent grids at full size and withheld by the same number."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_textvqa_amd import _capi as capi, ops

dev = torch.device("cuda", 0)
x = torch.randn(11648, 768, device=dev).to(torch.bfloat16)
w = (torch.randn(3072, 768, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(3072, device=dev)
hs = torch.cuda.Stream()


def timed_gemm(reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.gemm(x, w, epilogue=capi.EPI_BIAS, bias=b)
    e0.record()
    for _ in range(reps):
        ops.gemm(x, w, epilogue=capi.EPI_BIAS, bias=b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(hs):
    a0.record()
    ops.debug_cu_hog(8, 20000.0, hs)
    a1.record()
torch.cuda.synchronize()
print("hog of 8 blocks asked for 20000 us ran %.1f us" % (a0.elapsed_time(a1) * 1e3))
for hog in (0, 32, 64, 128):
    for reserve in sorted({0, hog}):
        ops.set_cu_reserve(reserve)
        torch.cuda.synchronize()
        if hog:
            ops.debug_cu_hog(hog, 60000.0, hs)
            time.sleep(0.005)
        t = timed_gemm()
        torch.cuda.synchronize()
        print("hog %3d  reserve %3d   gemm 11648x3072x768 bias: %.1f us" % (hog, reserve, t), flush=True)
ops.set_cu_reserve(0)
