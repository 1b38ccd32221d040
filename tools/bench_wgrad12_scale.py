"""Experiment 13: per-k-tile time of the two weight-gradient loops as a function of how many CUs run them -- one problem dW = dy^T x with dy [R, 3072] and
x [R, N]: N = 768 .. 4096 gives the 12-wave kernel (192 x 256 tiles, one per CU) 48 .. 256 tiles and the 8-wave pair kernel (256 x 256 x two K halves) 72 .. 384 blocks.
Environment: SAM_GEMM12_DBG / SAM_GEMM8W_DBG (1 = no epilogue, 4 = no DMA in the loop, 8 = L2-resident DMA) apply as usual."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops

R, M = 11648, 3072
g = torch.Generator(device="cuda").manual_seed(0)


def t(fn, sets, n=10):
    for s in sets: fn(*s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for s in sets: fn(*s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * len(sets))


for N in [int(x) for x in sys.argv[1:]] or [768, 1536, 2304, 3072, 4096]:
    sets = []
    for _ in range(3):
        dy = (torch.randn(R, M, device="cuda", generator=g) * 0.5).bfloat16()
        x = (torch.randn(R, N, device="cuda", generator=g) * 0.5).bfloat16()
        sets.append((dy, x, torch.zeros(M, N, device="cuda")))
    a = t(lambda dy, x, dw: ops.wgrad_grouped([(dy, x, dw, None)], force_tile=1256), sets)
    b = t(lambda dy, x, dw: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=dw, accumulate=True, force_tile=12448), sets)
    t8, t12 = (M // 256) * (N // 256), (M // 192) * (N // 256)
    r8 = -(-t8 * 2 // 256)          # rounds of the 8-wave kernel's blocks (one per CU)
    r12 = -(-t12 // 256)
    print("N=%4d  8-wave: %3d tiles x 2 halves = %3d blocks, %7.1f us = %.3f us per 256x256 k-tile and round | 12-wave: %3d tiles, %7.1f us = %.3f us per 192x256 k-tile and round (%.3f per 64K MACs)"
          % (N, t8, 2 * t8, a, a / 91 / r8, t12, b, b / 182 / r12, b / 182 / r12 / 0.75), flush=True)
    del sets
    torch.cuda.empty_cache()
