"""micro-benchmark of the fused attention kernels at the c3 shape (B=64, N=182, H=12)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
from sam_textvqa_amd.synthetic import make_batch
B, H, N = 64, 12, 182
bd = make_batch(B, device="cuda")
kv = torch.cat([bd["question_mask"], bd["pad_obj_mask"], bd["pad_ocr_mask"]], 1).to(torch.uint8).contiguous()
base = ops.mask_bits_prefix_lm(kv, 12)
sp = ops.mask_bits_spatial(base, bd["spatial_adj_matrices"]["3"], 20, H, (1, 2))
qkv = torch.randn(B * N, 2304, device="cuda").to(torch.bfloat16)
dout = torch.randn(B * N, 768, device="cuda").to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, allow in (("spatial", sp), ("plain", base)):
    for p in (0.0, 0.1):
        out, lse2, keep = ops.attn_fwd(qkv, allow, B, H, 0.125, p, 1, 1)
        nw = allow.shape[-1]
        fb = B * (4 * N * 768 * 2 + H * N * nw * 4 * (2 if p else 1) + H * N * 4)
        bb = B * (8 * N * 768 * 2 + H * N * (nw * 4 * (2 if p else 1) + 8))
        uf = t(lambda: ops.attn_fwd(qkv, allow, B, H, 0.125, p, 1, 1))
        ub = t(lambda: ops.attn_bwd(dout, qkv, lse2, allow, keep, B, H, 0.125, p))
        print("%-8s p=%.1f  fwd %6.1f us %6.0f GB/s (%.1f%% of 8 TB/s)   bwd %6.1f us %6.0f GB/s" % (name, p, uf, fb / uf / 1e3, fb / uf / 1e3 / 80, ub, bb / ub / 1e3))
