"""sam_gemm_bf16 through its dispatcher (force_tile = 0: whichever kernel the heuristics pick) on random shapes, layouts, epilogues and leading dimensions
against an fp32 matmul of the same bf16 operands:  python tools/fuzz_gemm.py [count] [seed]"""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sam_textvqa_amd import _capi as capi  # noqa: E402
from sam_textvqa_amd import ops  # noqa: E402
from tests.util import assert_close_bf16  # noqa: E402

BF = torch.bfloat16
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(rnd.randint(0, 1 << 30))
torch.backends.cuda.matmul.allow_tf32 = False


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def dim(big):
    k = rnd.random()
    if k < 0.25:
        return 8 * rnd.randint(1, 16)
    if k < 0.6:
        return rnd.choice([64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2304, 3072]) + rnd.choice([0, 0, 0, 8, -8, 64])
    return 8 * rnd.randint(1, big // 8)


def mat(rows, cols, scale=1.0):
    """[rows, cols] bf16 on the GPU, sometimes a view with a padded leading dimension"""
    pad = rnd.choice([0, 0, 8, 64])
    t = (torch.randn(rows, cols + pad, device="cuda") * scale).to(BF)
    return t[:, :cols]


bad = 0
for it in range(count):
    lay = rnd.choice(["fwd", "fwd", "dgrad", "wgrad"])
    M = dim(12000) if rnd.random() < 0.5 else rnd.choice([11648, 1280, 6400, 3200, 23296 // 2, 64, 192, 768])
    N, K = dim(3200), dim(3200)
    if rnd.random() < 0.1:
        M = max(1, M + rnd.choice([-7, -3, 1, 5])) if lay != "wgrad" else M
    tag = "%s M=%d N=%d K=%d" % (lay, M, N, K)
    try:
        if lay == "wgrad":
            R = K if rnd.random() < 0.5 else rnd.choice([11648, 1280, 728, 5000, 182 * 8])
            dy, x = mat(R, M), mat(R, N)
            ref = dy.float().t() @ x.float()
            acc = rnd.random() < 0.5
            c0 = torch.randn(M, N, device="cuda") if acc else torch.empty(M, N, device="cuda")
            want = ref + c0 if acc else ref
            bg0 = torch.randn(M, device="cuda")
            bg = bg0.clone()
            use_bg = rnd.random() < 0.5
            split = rnd.choice([0, 0, -1, 3]) if M * N <= (2 << 20) else 0
            tag += " R=%d acc=%d bias_grad=%d split=%d" % (R, acc, use_bg, split)
            out = c0.clone()
            ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=out, accumulate=acc, bias_grad=bg if use_bg else None, split_k=split if acc else 0)
            assert_close_bf16(out, want, frac=2e-3, ulps=0, name="wgrad")
            if use_bg:
                assert_close_bf16(bg, dy.float().sum(0) + (bg0 if acc else 0), frac=2e-3, ulps=0, name="bias grad")
        else:
            a = mat(M, K)
            b = mat(N, K, 0.05) if lay == "fwd" else mat(K, N, 0.05)
            acc32 = a.float() @ (b.float().t() if lay == "fwd" else b.float())
            # (the epilogues each layout is instantiated with: forward = the nn.Linear sites, dgrad = their backward; anything else is refused loudly)
            epi = rnd.choice(["none", "none32", "bias", "gelu", "gelu_grad", "res"] if lay == "fwd" else ["none", "none32", "res", "dgelu", "mul"])
            tag += " epi=%s" % epi
            bias = torch.randn(N, device="cuda") * 0.1
            kw = dict(b_kcontig=lay == "fwd")
            if epi == "none":
                assert_close_bf16(ops.gemm(a, b, **kw), acc32, name=tag)
            elif epi == "none32":
                assert_close_bf16(ops.gemm(a, b, out_dtype=torch.float32, **kw), acc32, ulps=0, name=tag)
            elif epi == "bias":
                assert_close_bf16(ops.gemm(a, b, epilogue=capi.EPI_BIAS, bias=bias, **kw), acc32 + bias, name=tag)
            elif epi == "gelu":
                pre = torch.empty(M, N, dtype=BF, device="cuda")
                h = ops.gemm(a, b, epilogue=capi.EPI_BIAS_GELU, bias=bias, aux_out=pre, **kw)
                assert_close_bf16(pre, acc32 + bias, name=tag + " pre")
                assert_close_bf16(h, gelu(acc32 + bias), ulps=2, name=tag + " gelu")
            elif epi == "gelu_grad":
                d = torch.empty(M, N, dtype=BF, device="cuda")
                h = ops.gemm(a, b, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=bias, aux_out=d, **kw)
                assert_close_bf16(h, gelu(acc32 + bias), ulps=2, name=tag + " gelu")
                assert_close_bf16(d, dgelu(acc32 + bias), frac=4e-3, ulps=2, name=tag + " gelu'")
            elif epi == "res":
                res = mat(M, N)
                y = ops.gemm(a, b, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=0.0, **kw)
                assert_close_bf16(y, acc32 + bias + res.float(), name=tag)
            elif epi == "dgelu":
                aux = mat(M, N)
                y = ops.gemm(a, b, epilogue=capi.EPI_DGELU, aux_in=aux, **kw)
                assert_close_bf16(y, acc32 * dgelu(aux.float()), frac=2e-3, ulps=2, name=tag)
            else:
                aux = mat(M, N)
                y = ops.gemm(a, b, epilogue=capi.EPI_MUL_AUX, aux_in=aux, **kw)
                assert_close_bf16(y, acc32 * aux.float(), frac=2e-3, ulps=2, name=tag)
        torch.cuda.synchronize()
        print("ok  ", tag, flush=True)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("FAIL", tag, "::", str(e).splitlines()[0][:220], flush=True)
print("fuzz_gemm: %d failures of %d" % (bad, count))
