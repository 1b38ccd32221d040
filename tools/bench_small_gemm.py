"""skinny GEMMs of the step (TextBert: 1280 rows; classifier dgrad: 768 rows; OCR / object projections): automatic epilogue split-K vs un-split"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops, _capi as capi
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
for name, M, N, K, bk in [("TB qkv fwd", 1280, 2304, 768, True), ("TB ffn1 fwd", 1280, 3072, 768, True), ("TB ffn2 dgrad", 1280, 3072, 768, False), ("TB ffn2 fwd", 1280, 768, 3072, True), ("TB ffn1 dgrad", 1280, 768, 3072, False), ("TB qkv dgrad", 1280, 768, 2304, False),
                          ("TB o-proj fwd", 1280, 768, 768, True), ("cls dgrad", 768, 768, 5000, False), ("ocr proj", 3200, 768, 3008, True),
                          ("obj proj", 6400, 768, 2048, True)]:
    a = rnd(M, K); b = rnd(N, K) if bk else rnd(K, N)
    bias = torch.zeros(N, device="cuda"); res = rnd(M, N)
    kw = dict(b_kcontig=bk, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.1, seed=1, offset=2)
    row = [name, "%dx%dx%d" % (M, N, K)]
    for label, extra in (("auto", {}), ("tile64", dict(force_tile=64)), ("tile128", dict(force_tile=128)), ("split2", dict(split_k=2)), ("split3", dict(split_k=3)),
                         ("split4", dict(split_k=4)), ("split6", dict(split_k=6)), ("split8", dict(split_k=8))):
        try:
            row.append("%s %.1f" % (label, t(lambda: ops.gemm(a, b, **kw, **extra))))
        except Exception as e:
            row.append("%s err" % label)
    print("  ".join(row))
