"""8-wave persistent kernels (gemm8.hip) against the 4-wave kernels and the library, on the GEMM shapes of one SA-M4C step (B=64 -> 11648 rows)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops, _capi as capi
R = int(sys.argv[1]) if len(sys.argv) > 1 else 11648
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
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
rows = []
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)]:
    x, w, wT = rnd(R, K), rnd(N, K), rnd(K, N)
    bias = torch.randn(N, device="cuda")
    res, pre = rnd(R, N), rnd(R, N)
    aux = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * R * N * K
    for ft in (0, 1192, 1256):
        try:
            rows.append(("fwd bias       N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS, bias=bias, force_tile=ft)), fl))
            if True:
                rows.append(("fwd gelu       N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=bias, aux_out=aux, force_tile=ft)), fl))
                rows.append(("fwd drop+res   N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.1, seed=1, offset=2, force_tile=ft)), fl))
            rows.append(("dgrad none     N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, wT, b_kcontig=False, force_tile=ft)), fl))
            if True:
                rows.append(("dgrad dgelu    N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, wT, b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=pre, force_tile=ft)), fl))
                rows.append(("dgrad +res     N=%d K=%d tile=%d" % (N, K, ft), t(lambda: ops.gemm(x, wT, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=res, force_tile=ft)), fl))
        except capi.SamHipError as e:
            rows.append(("ERR %s tile=%d" % (str(e)[:60], ft), 1.0, 0.0))
    rows.append(("LIB fwd        N=%d K=%d" % (N, K), t(lambda: torch.matmul(x, w.t())), fl))
    rows.append(("LIB dgrad      N=%d K=%d" % (N, K), t(lambda: torch.matmul(x, wT)), fl))
for name, us, fl in rows:
    print("%-44s %8.1f us  %7.1f TFLOP/s" % (name, us, fl / us / 1e6))
