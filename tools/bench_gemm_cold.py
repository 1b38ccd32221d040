"""The encoder layer's big GEMMs with COLD operands: every call works on another buffer set (activations of a training step are written once
and read once; the 256 MB Infinity Cache holds none of them by the time they are needed), weights shared.  tools/bench_gemm8.py reuses one
set and measures the cache-resident case, which the step does not see (FFN1 forward: 86 us there, 111 us in the step).
usage: python tools/bench_gemm_cold.py [rows] [nsets]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops, _capi as capi  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 11648
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def rnd(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16)


def timed(fns, reps=4):
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for f in fns:
                f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * len(fns)))
    return best


def main():
    D, I = 768, 3072
    w1, w2 = rnd(I, D), rnd(D, I)                 # FFN1 / FFN2 weights ([out, in])
    wqkv, wo = rnd(3 * D, D), rnd(D, D)
    b1, b2, bqkv = torch.randn(I, device="cuda"), torch.randn(D, device="cuda"), torch.randn(3 * D, device="cuda")
    xs = [rnd(R, D) for _ in range(NS)]
    hs = [rnd(R, I) for _ in range(NS)]
    auxs = [rnd(R, I) for _ in range(NS)]
    outs_i = [torch.empty(R, I, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
    outs_d = [torch.empty(R, D, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
    outs_q = [torch.empty(R, 3 * D, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
    ft = int(os.environ.get("FT", "0"))
    cases = [
        ("QKV  fwd bias        [R,2304,768]", 2.0 * R * 3 * D * D, [lambda i=i: ops.gemm(xs[i], wqkv, epilogue=capi.EPI_BIAS, bias=bqkv, out=outs_q[i], force_tile=ft) for i in range(NS)]),
        ("O    fwd drop+res    [R,768,768]", 2.0 * R * D * D, [lambda i=i: ops.gemm(xs[i], wo, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2, residual=xs[(i + 1) % NS], p_drop=0.1, seed=1, offset=2, out=outs_d[i], force_tile=ft) for i in range(NS)]),
        ("FFN1 fwd gelu+grad   [R,3072,768]", 2.0 * R * I * D, [lambda i=i: ops.gemm(xs[i], w1, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=b1, aux_out=auxs[i], out=outs_i[i], force_tile=ft) for i in range(NS)]),
        ("FFN2 fwd drop+res    [R,768,3072]", 2.0 * R * I * D, [lambda i=i: ops.gemm(hs[i], w2, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2, residual=xs[i], p_drop=0.1, seed=1, offset=2, out=outs_d[i], force_tile=ft) for i in range(NS)]),
        ("FFN2 dgrad x gelu'   [R,3072,768]", 2.0 * R * I * D, [lambda i=i: ops.gemm(xs[i], w2, b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=auxs[i], out=outs_i[i], force_tile=ft) for i in range(NS)]),
        ("FFN1 dgrad +res      [R,768,3072]", 2.0 * R * I * D, [lambda i=i: ops.gemm(hs[i], w1, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=xs[i], out=outs_d[i], force_tile=ft) for i in range(NS)]),
        ("QKV  dgrad +res      [R,768,2304]", 2.0 * R * 3 * D * D, [lambda i=i: ops.gemm(outs_q[i], wqkv, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=xs[i], out=outs_d[i], force_tile=ft) for i in range(NS)]),
    ]
    for name, fl, fns in cases:
        try:
            us = timed(fns)
            print("%-36s %8.1f us  %7.1f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
        except capi.SamHipError as e:
            print(name, "ERR", str(e)[:80])


if __name__ == "__main__":
    main()
