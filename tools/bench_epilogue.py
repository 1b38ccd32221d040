"""cost of the GEMM epilogues at the encoder shapes (M = 11648): plain vs bias+residual vs bias+dropout+residual, GELU, dGELU"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops, _capi as capi
R = 11648
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
for (N, K, bk, name) in [(768, 768, True, "O-proj fwd"), (768, 3072, True, "FFN2 fwd"), (768, 3072, False, "dFFN1"), (768, 2304, False, "dQKV")]:
    x = rnd(R, K); w = rnd(N, K) if bk else rnd(K, N); bias = torch.zeros(N, device="cuda"); res = rnd(R, N)
    a = t(lambda: ops.gemm(x, w, b_kcontig=bk))
    b = t(lambda: ops.gemm(x, w, b_kcontig=bk, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.0))
    c = t(lambda: ops.gemm(x, w, b_kcontig=bk, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.1, seed=1, offset=2))
    print("%-11s N=%d K=%d  plain %.1f us   +bias+res %.1f   +dropout %.1f" % (name, N, K, a, b, c))
x = rnd(R, 768); w = rnd(3072, 768); bias = torch.zeros(3072, device="cuda"); pre = torch.empty(R, 3072, device="cuda", dtype=torch.bfloat16)
print("FFN1 fwd    plain %.1f   +bias+gelu(+pre) %.1f" % (t(lambda: ops.gemm(x, w)), t(lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS_GELU, bias=bias, aux_out=pre))))
dy = rnd(R, 768); w2 = rnd(768, 3072)
print("dFFN2       plain %.1f   *gelu' %.1f" % (t(lambda: ops.gemm(dy, w2, b_kcontig=False)), t(lambda: ops.gemm(dy, w2, b_kcontig=False, epilogue=capi.EPI_DGELU, aux_in=pre))))
