"""Experiment: per-k-tile time of the weight-gradient loop (both operands k-strided, fp32 accumulate) on the 8-wave grouped kernel (256 x 256 tiles, K split over a pair)
and on the 12-wave loader-wave core (192 x 256 tiles, one block per tile, force_tile 12448), one problem dW1 = dh^T a: [3072, 768] over R = 11648 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops

R, M, N = 11648, 3072, 768
g = torch.Generator(device="cuda").manual_seed(0)
sets = []
for _ in range(4):
    dy = (torch.randn(R, M, device="cuda", generator=g) * 0.5).bfloat16()
    x = (torch.randn(R, N, device="cuda", generator=g) * 0.5).bfloat16()
    sets.append((dy, x, torch.zeros(M, N, device="cuda")))


def t(fn, n=20):
    for s in sets: fn(*s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for s in sets: fn(*s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * len(sets))

a = t(lambda dy, x, dw: ops.wgrad_grouped([(dy, x, dw, None)], force_tile=1256))
b = t(lambda dy, x, dw: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=dw, accumulate=True, force_tile=12448))
ref = sets[0][0].float().t() @ sets[0][1].float()
d1, d2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
ops.wgrad_grouped([(sets[0][0], sets[0][1], d1, None)], force_tile=1256)
ops.gemm(sets[0][0], sets[0][1], a_kcontig=False, b_kcontig=False, out=d2, accumulate=True, force_tile=12448)
print("max err 8w %.3g 12w %.3g (scale %.3g)" % ((d1 - ref).abs().max().item(), (d2 - ref).abs().max().item(), ref.abs().max().item()))
print("8-wave grouped, 36 tiles of 256x256 x 2 k-halves (91 k-tiles each): %.1f us -> %.3f us per 256x256 k-tile (%.3f per 64K MACs)" % (a, a / 91, a / 91))
print("12-wave, 48 tiles of 192x256 (182 k-tiles each): %.1f us -> %.3f us per 192x256 k-tile (%.3f per 64K MACs)" % (b, b / 182, b / 182 / 0.75))
