"""time sam_attn_dec_row alone at the beam-5 bench shape (64 samples x 5 beams, 182 tokens, position 6)"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from sam_textvqa_amd import ops
b0, k, n, n_dec, h, t = 64, 5, 182, 12, 12, 6
b = b0 * k
enc = torch.randn(b0 * n, 3 * h * 64, device="cuda").to(torch.bfloat16)
dec = torch.randn(b * n_dec, 3 * h * 64, device="cuda").to(torch.bfloat16)
allow = ops.mask_bits_prefix_lm(torch.ones(b0, n - n_dec, dtype=torch.uint8, device="cuda"), n_dec)
for _ in range(3):
    ops.attn_dec_row(enc, dec, allow, b, n, n_dec, t, h, 0.125, kv_group=k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attn_dec_row(enc, dec, allow, b, n, n_dec, t, h, 0.125, kv_group=k)
e1.record(); torch.cuda.synchronize()
print("kernel %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
