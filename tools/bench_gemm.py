"""micro-benchmark of sam_gemm_bf16 over the shapes of one SA-M4C step (B=64 -> 11648 rows)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    dy = rnd(R, N)
    out = torch.zeros(N, K, device="cuda")
    for ft in (128, 160, 192, 0):
        us = t(lambda: ops.gemm(x, w, force_tile=ft)); rows.append(("fwd  M=%d N=%d K=%d tile=%d" % (R, N, K, ft), us, 2.0 * R * N * K))
        us = t(lambda: ops.gemm(dy, w, b_kcontig=False, force_tile=ft)); rows.append(("dgrad M=%d N=%d K=%d tile=%d" % (R, K, N, ft), us, 2.0 * R * N * K))
    # library calibration (hipBLASLt / rocBLAS through torch.matmul) on the same three products -- not part of the product path
    us = t(lambda: torch.matmul(x, w.t())); rows.append(("LIB fwd  M=%d N=%d K=%d" % (R, N, K), us, 2.0 * R * N * K))
    us = t(lambda: torch.matmul(dy, w)); rows.append(("LIB dgrad M=%d N=%d K=%d" % (R, K, N), us, 2.0 * R * N * K))
    us = t(lambda: torch.matmul(dy.t(), x)); rows.append(("LIB wgrad out=%dx%d rows=%d" % (N, K, R), us, 2.0 * R * N * K))
    for sk in (-1, 3, 4, 5, 6, 8, 12, 14):
        try:
            us = t(lambda: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=out, accumulate=True, split_k=sk))
            rows.append(("wgrad out=%dx%d rows=%d split=%d" % (N, K, R, sk), us, 2.0 * R * N * K))
        except Exception as e:
            pass
for name, us, fl in rows:
    print("%-44s %8.1f us  %7.1f TFLOP/s" % (name, us, fl / us / 1e6))
