#!/usr/bin/env python3
"""time of the O-projection / FFN2 product at MMT size with the LayerNorm inside the launch, against GEMM + sam_layernorm_fwd; SAM_GEMM8_DBG bits 8 (no wait),
16 (no counters at all), 32 (pass switched off in the kernel) take the pass apart (results are then wrong)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_textvqa_amd import ops, _capi as capi
def t(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, n, k) in ((11648, 768, 768), (11648, 768, 3072)):
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16); w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.zeros(n, device="cuda"); res = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda")
    kw = dict(epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.1, seed=1, offset=2)
    t_gemm = t(lambda: ops.gemm(a, w, **kw))
    z = ops.gemm(a, w, **kw)
    t_ln = t(lambda: ops.layernorm_fwd(z, g, b, 1e-12))
    t_both = t(lambda: ops.gemm_ln(a, w, g, b, 1e-12, **kw))
    print("%dx%dx%d  gemm %.1f us  layernorm %.1f us  gemm_ln (SAM_GEMM_LN_FUSE=%s, dbg=%s) %.1f us" % (m, n, k, t_gemm, t_ln, os.environ.get("SAM_GEMM_LN_FUSE", "1"), os.environ.get("SAM_GEMM8_DBG", "0"), t_both))
