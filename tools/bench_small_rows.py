"""TextBert-size GEMMs (1280 token rows at B = 64): which tile of the 4-wave / 8-wave kernels is fastest?  usage: python tools/bench_small_rows.py [rows]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops, _capi as capi  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1280


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def rnd(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16)


for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x, w, wT = rnd(R, K), rnd(N, K), rnd(K, N)
    bias, res, aux = torch.randn(N, device="cuda"), rnd(R, N), torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    for ft in (0, 64, 128, 160, 192, 1192, 1256):
        row = "N=%4d K=%4d tile=%4d " % (N, K, ft)
        for name, fn in (("fwd bias", lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS, bias=bias, force_tile=ft)),
                         ("fwd drop+res", lambda: ops.gemm(x, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.1, seed=1, offset=2, force_tile=ft)),
                         ("dgrad", lambda: ops.gemm(x, wT, b_kcontig=False, force_tile=ft))):
            try:
                row += " %s %6.1f us" % (name, t(fn))
            except capi.SamHipError:
                row += " %s    n/a  " % name
        print(row, flush=True)
