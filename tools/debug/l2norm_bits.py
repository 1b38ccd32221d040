"""l2norm_pack (embed.hip) against an fp32 emulation of its exact addition order (lane partials, xor-butterfly 32..1), bit for bit"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_textvqa_amd import ops
for M, D in ((100, 2048), (50, 1024), (7, 512), (33, 256)):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g)
    out = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    ops.l2norm_pack(x.cuda(), out)
    nch = D // 256
    v = x.numpy().reshape(M, nch, 64, 4)
    q = np.zeros((M, 64), np.float32)
    for j in range(nch):
        sq = v[:, j] * v[:, j]
        q = q + ((sq[..., 0] + sq[..., 1]) + (sq[..., 2] + sq[..., 3]))
    lanes = np.arange(64)
    for o in (32, 16, 8, 4, 2, 1):
        q = q + q[:, lanes ^ o]
    scale = np.float32(1.0) / np.maximum(np.sqrt(q[:, :1]), np.float32(1e-12))
    want = torch.from_numpy(x.numpy() * scale).to(torch.bfloat16)
    got = out.cpu()
    print(M, D, "mismatching elements:", int((want.view(torch.int16) != got.view(torch.int16)).sum()), "of", M * D)
