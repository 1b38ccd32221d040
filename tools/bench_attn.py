"""micro-benchmark of the fused attention kernels: c3 shape (B=64, N=182) and the stress shape (B=32, N=350), H=12"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
from sam_textvqa_amd.synthetic import make_batch, SHAPES
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
H = 12
for tag, B, shape in (("c3", 64, SHAPES["c3"]), ("stress", 32, SHAPES["stress"])):
    T, n_obj, n_ocr, n_dec = shape
    N = sum(shape)
    bd = make_batch(B, *shape, device="cuda")
    kv = torch.cat([bd["question_mask"], bd["pad_obj_mask"], bd["pad_ocr_mask"]], 1).to(torch.uint8).contiguous()
    base = ops.mask_bits_prefix_lm(kv, n_dec)
    sp = ops.mask_bits_spatial(base, bd["spatial_adj_matrices"]["3"], T, H, (1, 2))
    qkv = torch.randn(B * N, 2304, device="cuda").to(torch.bfloat16)
    dout = torch.randn(B * N, 768, device="cuda").to(torch.bfloat16)
    for name, allow in (("spatial", sp), ("plain", base)):
        for p in (0.0, 0.1):
            out, lse2, keep, out_lo = ops.attn_fwd(qkv, allow, B, H, 0.125, p, 1, 1, want_residual=True)
            nw = allow.shape[-1]
            fb = B * (4 * N * 768 * 2 + H * N * nw * 4 * (2 if p else 1) + H * N * 4)
            bb = B * (8 * N * 768 * 2 + H * N * (nw * 4 * (2 if p else 1) + 8))
            uf = t(lambda: ops.attn_fwd(qkv, allow, B, H, 0.125, p, 1, 1))
            ut = t(lambda: ops.attn_fwd(qkv, allow, B, H, 0.125, p, 1, 1, want_residual=True))
            ub = t(lambda: ops.attn_bwd(dout, qkv, lse2, allow, keep, B, H, 0.125, p))
            line = "%-6s %-8s p=%.1f  fwd %6.1f us %6.0f GB/s (%.1f%% of 8 TB/s)   fwd+residual %6.1f us   bwd(2 kernels) %6.1f us %6.0f GB/s (%.1f%%)" % (
                tag, name, p, uf, fb / uf / 1e3, fb / uf / 1e3 / 80, ut, ub, bb / ub / 1e3, bb / ub / 1e3 / 80)
            if N <= ops.attn_bwd_fused_max_n():
                b1 = B * (10 * N * 768 * 2 + H * N * (nw * 4 * (2 if p else 1) + 8))
                u1 = t(lambda: ops.attn_bwd(dout, qkv, lse2, allow, keep, B, H, 0.125, p, out=out, out_lo=out_lo))
                line += "   bwd(fused) %6.1f us %6.0f GB/s (%.1f%%)" % (u1, b1 / u1 / 1e3, b1 / u1 / 1e3 / 80)
            print(line, flush=True)
