"""BASELINE.json configs[1] sizes (B=64, N=182, H=12, D=768): too big for the fp32 oracle in seconds, so the hot path is checked through
size-independent properties, and tied back to the oracle through per-sample equality (a sample computed inside the 64-batch is bit-identical
to the same sample computed alone, and single samples ARE compared with the oracle)."""
import math

import pytest
import torch

from oracle import sa_m4c_oracle as O
from tests.test_attention_gpu import make_problem, oracle_attention
from tests.util import assert_close_bf16

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
B, T, N_OBJ, N_OCR, N_DEC, H, HD = 64, 20, 100, 50, 12, 12, 64


def _ops():
    from sam_textvqa_amd import ops
    return ops


def _setup(seed=11):
    ops = _ops()
    pr = make_problem(B, T, N_OBJ, N_OCR, N_DEC, seed=seed)
    base = ops.mask_bits_prefix_lm(pr["key_valid"].to(torch.uint8).cuda(), pr["n_dec"])
    bits = ops.mask_bits_spatial(base, pr["adj"].cuda(), pr["T"], H, (1, 2))
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B * pr["N"], 3 * H * HD, generator=g) * 1.5).to(BF16).cuda()
    return ops, pr, base, bits, qkv


def test_attention_rows_sum_to_one_and_dead_rows_are_zero():
    ops, pr, base, bits, qkv = _setup()
    n = pr["N"]
    q3 = qkv.view(B * n, 3, H * HD).clone()
    q3[:, 2] = 1.0                                                        # V = 1  =>  out = sum_k P = 1 on live rows, exactly 0 on fully masked rows
    out, lse2, _ = ops.attn_fwd(q3.view(B * n, -1), bits, B, H, 1 / math.sqrt(HD), 0.0)
    alive = ~torch.isinf(lse2)                                            # [B, H, N]
    o = out.float().view(B, n, H, HD).permute(0, 2, 1, 3)
    assert (o[~alive] == 0).all()
    assert (o[alive] - 1).abs().max().item() <= 2 ** -8                   # one bf16 ulp of 1.0
    text_rows = alive[:, :, :T]
    assert not text_rows.any()                                            # spatial layer, quadrants (1,2): every text query is fully masked (sa_m4c.py:574-584)
    assert alive[:, :, T:].float().mean().item() > 0.9


def test_attention_is_linear_in_v_and_batch_separable():
    ops, pr, base, bits, qkv = _setup()
    n, scale = pr["N"], 1 / math.sqrt(HD)
    q3 = qkv.view(B * n, 3, H * HD)
    q3[:, 2] = (torch.round(q3[:, 2].float() * 8) / 8).to(BF16)             # V on a 1/8 grid: V1 + V2 is then exact in bf16
    out, lse2, _ = ops.attn_fwd(qkv, bits, B, H, scale, 0.0)
    v2 = (torch.round(torch.randn(B * n, H * HD, generator=torch.Generator().manual_seed(3)) * 0.7 * 8) / 8).to(BF16).cuda()
    a = q3.clone(); a[:, 2] = v2
    s = q3.clone(); s[:, 2] = (q3[:, 2].float() + v2.float()).to(BF16)
    assert torch.equal(s[:, 2].float(), q3[:, 2].float() + v2.float())
    out_b, _, _ = ops.attn_fwd(a.view(B * n, -1), bits, B, H, scale, 0.0)
    out_s, _, _ = ops.attn_fwd(s.view(B * n, -1), bits, B, H, scale, 0.0)
    # three stored results, each rounded to bf16 at ITS OWN magnitude (half an ulp = 2^-9 relative; the sum may cancel)
    bound = 2.0 ** -8 * (out.float().abs() + out_b.float().abs() + out_s.float().abs()) + 1e-3
    assert ((out_s.float() - (out.float() + out_b.float())).abs() <= bound).all()
    # every sample is an independent problem: sample b inside the batch == sample b alone, bit for bit; and alone it matches the oracle
    for b in (0, 17, 63):
        rows = slice(b * n, (b + 1) * n)
        o1, l1, _ = ops.attn_fwd(qkv[rows].contiguous(), bits[b:b + 1].contiguous(), 1, H, scale, 0.0)
        assert torch.equal(o1, out[rows]) and torch.equal(l1, lse2[b:b + 1])
    b = 17
    allow = O.allow_mask(pr["key_valid"][b:b + 1], pr["T"], pr["n_oo"], pr["n_dec"], pr["adj"][b:b + 1], (1, 2), H)
    ref, _ = oracle_attention(qkv[b * n:(b + 1) * n].float().cpu(), allow, 1, H, scale)
    assert_close_bf16(out[b * n:(b + 1) * n], ref, name="sample 17 of the 64-batch vs oracle")


def test_attention_backward_batch_separable_and_directional_derivative():
    ops, pr, base, bits, qkv = _setup()
    n, scale = pr["N"], 1 / math.sqrt(HD)
    dout = torch.randn(B * n, H * HD, generator=torch.Generator().manual_seed(9)).to(BF16).cuda()
    out, lse2, keep = ops.attn_fwd(qkv, bits, B, H, scale, 0.1, seed=5, offset=3)
    dqkv = ops.attn_bwd(dout, qkv, lse2, bits, keep, B, H, scale, 0.1)
    for b in (5, 40):
        rows = slice(b * n, (b + 1) * n)
        d1 = ops.attn_bwd(dout[rows].contiguous(), qkv[rows].contiguous(), lse2[b:b + 1].contiguous(), bits[b:b + 1].contiguous(), keep[b:b + 1].contiguous(), 1, H, scale, 0.1)
        assert torch.equal(d1, dqkv[rows])
    # <dout, f(x + eps d) - f(x - eps d)> / (2 eps) == <dqkv, d>  (no dropout: f is deterministic), on the V block where f is linear
    out0, lse0, _ = ops.attn_fwd(qkv, bits, B, H, scale, 0.0)
    dq0 = ops.attn_bwd(dout, qkv, lse0, bits, None, B, H, scale, 0.0)
    d = torch.zeros_like(qkv).view(B * n, 3, H * HD)
    d[:, 2] = torch.randn(B * n, H * HD, generator=torch.Generator().manual_seed(4)).to(BF16).cuda()
    d = d.view(B * n, -1)
    outd, _, _ = ops.attn_fwd(d + qkv * torch.tensor([1.0, 1.0, 0.0], device="cuda").repeat_interleave(H * HD).to(BF16), bits, B, H, scale, 0.0)   # same q, k; V = d
    lhs = (dout.float() * outd.float()).sum().item()
    rhs = (dq0.float() * d.float()).sum().item()
    # both sides are sums of 8.9 M signed products of bf16-rounded factors (relative rounding 2^-9 each): the rounding noise is a random
    # walk of that many steps, far below the bound but far above any relative tolerance on the (heavily cancelled) totals
    noise = 2.0 ** -9 * math.sqrt(((dout.float() * outd.float()) ** 2).sum().item() + ((dq0.float() * d.float()) ** 2).sum().item())
    assert abs(lhs - rhs) <= 6 * noise, (lhs, rhs, noise)
    assert abs(lhs) > 20 * noise                                            # ... and the quantity itself is well above that noise


def test_gemm_full_height_is_row_separable_and_linear():
    ops = _ops()
    from sam_textvqa_amd import _capi as capi
    m, k, nn = B * 182, 768, 3072
    g = torch.Generator().manual_seed(1)
    x = torch.randn(m, k, generator=g).to(BF16).cuda()
    w = (torch.randn(nn, k, generator=g) * 0.05).to(BF16).cuda()
    bias = torch.randn(nn, generator=g).cuda()
    y = ops.gemm(x, w, epilogue=capi.EPI_BIAS, bias=bias)
    # rows are independent: any row block computed alone gives the same bits (tile boundaries move, the k order per element does not)
    for lo, hi in ((0, 182), (5000, 5192), (m - 100, m)):
        assert torch.equal(ops.gemm(x[lo:hi].contiguous(), w, epilogue=capi.EPI_BIAS, bias=bias), y[lo:hi])
    # a 182-row block against fp64 on the host
    ref = x[5000:5182].double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    assert_close_bf16(y[5000:5182], ref.float(), name="gemm rows 5000..5182 of 11648")
    # linearity in x (fp32 accumulate: exact up to the bf16 rounding of the three results)
    x2 = torch.randn(m, k, generator=g).to(BF16).cuda()
    xs = (x.float() + x2.float()).to(BF16)
    exact = xs.float() == x.float() + x2.float()                          # rows where the bf16 sum is exact
    rows = exact.all(dim=1).nonzero().flatten()[:64]
    if rows.numel():
        y2, ys = ops.gemm(x2, w), ops.gemm(xs, w)
        assert_close_bf16(ys[rows], (ops.gemm(x, w)[rows].float() + y2[rows].float()), ulps=3, name="linearity in x")


def test_layernorm_full_height_statistics():
    ops = _ops()
    m, d = B * 182, 768
    x = (torch.randn(m, d, generator=torch.Generator().manual_seed(2)) * 3 + 1.5).to(BF16).cuda()
    ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, ones, zeros, 1e-12)
    yf = y.float()
    assert yf.mean(dim=1).abs().max().item() < 2e-3 and (yf.var(dim=1, unbiased=False) - 1).abs().max().item() < 1e-2
    torch.testing.assert_close(mean, x.float().mean(dim=1), rtol=1e-5, atol=1e-5)
    # idempotence: normalising a normalised row changes nothing beyond bf16 rounding
    y2, _, _ = ops.layernorm_fwd(y, ones, zeros, 1e-12)
    assert (y2.float() - yf).abs().max().item() <= 2 ** -6
    # backward: dx is orthogonal to 1 and to xhat for every row (the two projections LayerNorm's Jacobian removes)
    dy = torch.randn(m, d, generator=torch.Generator().manual_seed(3)).to(BF16).cuda()
    dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    dx, _ = ops.layernorm_bwd(dy, x, mean, rstd, ones, dg, db)
    dxf = dx.float()
    scale = dxf.abs().max().item()
    assert dxf.sum(dim=1).abs().max().item() < 5e-2 * scale * math.sqrt(d) / 8
    assert (dxf * yf).sum(dim=1).abs().max().item() < 5e-2 * scale * math.sqrt(d)
    torch.testing.assert_close(db, dy.float().sum(0), rtol=1e-4, atol=1e-2)
