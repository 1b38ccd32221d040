"""Experiment 13b: where the loader-wave grouped kernel (gemm12w.hip, force_tile 12448) loses against its own core: (a) one problem of 256 whole tiles (no slices),
with and without a bias gradient; (b) an MMT layer pair (288 tiles: 256 whole + 32 x 8 slices) with and without bias gradients; the 8-wave pair kernel beside it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops

R = 11648
g = torch.Generator(device="cuda").manual_seed(0)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def mk(shapes, bias, copies=3):
    sets = []
    for _ in range(copies):
        jobs = []
        for m, n in shapes:
            dy = (torch.randn(R, m, device="cuda", generator=g) * 0.5).bfloat16()
            x = (torch.randn(R, n, device="cuda", generator=g) * 0.5).bfloat16()
            jobs.append((dy, x, torch.zeros(m, n, device="cuda"), torch.zeros(m, device="cuda") if bias else None))
        sets.append(jobs)
    return sets


def run(tag, shapes, bias, ft):
    sets = mk(shapes, bias)
    k = [0]
    def fn():
        ops.wgrad_grouped(sets[k[0] % len(sets)], force_tile=ft); k[0] += 1
    us = t(fn, 12)
    tiles = sum(-(-m // 192) * -(-n // 256) for m, n in shapes)
    print("%-46s bias=%d ft=%5d: %7.1f us  (%d tiles of 192x256: %.3f us per k-tile at %.3f tiles per CU)" % (tag, bias, ft, us, tiles, us / 182 / (tiles / 256), tiles / 256), flush=True)
    del sets
    torch.cuda.empty_cache()


one = [(3072, 4096)]
layer = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
for bias in (0, 1):
    run("one problem [3072, 4096], 256 whole tiles", one, bias, 12448)
for bias in (0, 1):
    run("MMT layer pair, 288 tiles (32 x 8 slices)", layer * 2, bias, 12448)
    run("MMT layer pair, 8-wave pair kernel", layer * 2, bias, 1256)
