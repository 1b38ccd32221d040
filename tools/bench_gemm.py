"""micro-benchmark of sam_gemm_bf16 over the shapes of one SA-M4C step (B=64 -> 11648 rows)"""
import sys, torch
sys.path.insert(0, ".")
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
rows = []
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    x, w = rnd(R, K), rnd(N, K)
    us = t(lambda: ops.gemm(x, w)); rows.append(("fwd  M=%d N=%d K=%d" % (R, N, K), us, 2.0 * R * N * K))
    dy = rnd(R, N)
    us = t(lambda: ops.gemm(dy, w, b_kcontig=False)); rows.append(("dgrad M=%d N=%d K=%d" % (R, K, N), us, 2.0 * R * N * K))
    out = torch.zeros(N, K, device="cuda")
    for sk in (0, 2, 4, 8, -1):
        us = t(lambda: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=out, accumulate=True, split_k=sk))
        rows.append(("wgrad out=%dx%d rows=%d split=%d" % (N, K, R, sk), us, 2.0 * R * N * K))
for name, us, fl in rows:
    print("%-44s %8.1f us  %7.1f TFLOP/s" % (name, us, fl / us / 1e6))
