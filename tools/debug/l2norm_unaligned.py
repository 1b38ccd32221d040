"""l2norm_pack on the OCR feature slices (rows of 3002 floats: 8-byte aligned only -> the scalar kernel) against F.normalize"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_textvqa_amd import ops
torch.manual_seed(0)
for M in (300, 1500):
    x = torch.randn(M, 3002, device="cuda")
    for c0, D in ((0, 300), (300, 604), (904, 2048), (2952, 50)):
        sl = x[:, c0:c0 + D]
        out = torch.zeros(M, 3008, dtype=torch.bfloat16, device="cuda")
        ops.l2norm_pack(sl, out, col0=c0)
        want = torch.nn.functional.normalize(sl, dim=-1)
        err = (out[:, c0:c0 + D].float() - want).abs().max(dim=1).values
        bad = (err > 2e-3 * want.abs().max()).nonzero().flatten()
        print("M=%d cols [%d, %d): max err %.3e, bad rows %d %s" % (M, c0, c0 + D, err.max().item(), bad.numel(), bad[:12].tolist()))
