"""GPU parity: bf16 MFMA GEMM family (forward / dgrad / wgrad layouts, fused epilogues) vs fp32 matmul of
the same bf16-rounded operands."""
import math

import pytest
import torch

from tests.util import assert_close_bf16

pytestmark = pytest.mark.gpu


def _mods():
    from sam_textvqa_amd import _capi, ops
    return ops, _capi


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


SHAPES = [(728, 768, 768), (256, 128, 64), (130, 5000, 768), (200, 768, 3008), (1000, 2304, 768), (48, 768, 8), (1, 8, 8)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_forward_bias_and_f32(M, N, K):
    ops, capi = _mods()
    x, w = rnd((M, K), 1), rnd((N, K), 2, 0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = x.float() @ w.float().t() + b
    y = ops.gemm(x.cuda(), w.cuda(), epilogue=capi.EPI_BIAS, bias=b.cuda())
    assert_close_bf16(y, ref, name="bias bf16")
    y32 = ops.gemm(x.cuda(), w.cuda(), epilogue=capi.EPI_BIAS, bias=b.cuda(), out_dtype=torch.float32)
    assert_close_bf16(y32, ref, ulps=0, name="bias f32")
    y0 = ops.gemm(x.cuda(), w.cuda())
    assert_close_bf16(y0, x.float() @ w.float().t(), name="plain")


def test_forward_gelu_and_dropout_residual():
    ops, capi = _mods()
    M, N, K = 728, 3072, 768
    x, w = rnd((M, K), 4), rnd((N, K), 5, 0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(6)) * 0.1
    pre_ref = x.float() @ w.float().t() + b
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    h = ops.gemm(x.cuda(), w.cuda(), epilogue=capi.EPI_BIAS_GELU, bias=b.cuda(), aux_out=pre)
    assert_close_bf16(pre, pre_ref, name="pre-gelu")
    assert_close_bf16(h, gelu(pre_ref), name="gelu")
    # training form: the derivative is stored instead of the pre-activation (4-wave kernels via force_tile, 8-wave by default) ...
    for ft in (128, 0):
        dact = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        h2 = ops.gemm(x.cuda(), w.cuda(), epilogue=capi.EPI_BIAS_GELU_GRAD, bias=b.cuda(), aux_out=dact, force_tile=ft)
        assert_close_bf16(dact, dgelu(pre_ref), name="gelu'")
        assert_close_bf16(h2, gelu(pre_ref), name="gelu (training form)")
        # ... and the backward multiplies by it
        dyy, wt = rnd((M, K), 44), rnd((K, N), 45, 0.05)
        got = ops.gemm(dyy.cuda(), wt.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=dact, force_tile=ft)
        assert_close_bf16(got, (dyy.float() @ wt.float()) * dact.float().cpu(), name="dgrad * gelu'")
    # dense -> (+bias) -> dropout(p=0) -> + residual
    w2, res = rnd((K, N), 7, 0.03), rnd((M, K), 8)
    b2 = torch.randn(K, generator=torch.Generator().manual_seed(9)) * 0.1
    hh = h.cpu()
    ref = hh.float() @ w2.float().t() + b2 + res.float()
    z = ops.gemm(h, w2.cuda(), epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2.cuda(), residual=res.cuda())
    assert_close_bf16(z, ref, name="bias+res")
    # with dropout: every element is either residual (dropped) or residual + (acc+bias)/(1-p)
    p = 0.1
    zd = ops.gemm(h, w2.cuda(), epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2.cuda(), residual=res.cuda(), p_drop=p, seed=11, offset=3).float().cpu()
    inv = 1.0 / (1.0 - round(p * 65536) / 65536.0)
    dense = hh.float() @ w2.float().t() + b2
    kept_val = dense * inv + res.float()
    is_drop = (zd - res.float()).abs() <= 2.0 ** -7 * res.float().abs() + 1e-6
    is_kept = (zd - kept_val).abs() <= 1e-3 * kept_val.abs().max() + 2.0 ** -8 * kept_val.abs()
    assert (is_drop | is_kept).all()
    frac = 1.0 - (is_drop & ~is_kept).float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    zd2 = ops.gemm(h, w2.cuda(), epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2.cuda(), residual=res.cuda(), p_drop=p, seed=11, offset=3).float().cpu()
    assert torch.equal(zd, zd2)     # counter-based: same (seed, offset) -> same mask


@pytest.mark.parametrize("M,N,K", [(728, 768, 3072), (130, 768, 5000), (256, 64, 128), (200, 3008, 768)])
def test_dgrad_layout(M, N, K):
    """dx[M,N] = dy[M,K] . W[K,N]  (W stored [out=K][in=N]: contraction index is W's ROW index)"""
    ops, capi = _mods()
    dy, w = rnd((M, K), 10), rnd((K, N), 11, 0.05)
    ref = dy.float() @ w.float()
    dx = ops.gemm(dy.cuda(), w.cuda(), b_kcontig=False)
    assert_close_bf16(dx, ref, name="dgrad")
    pre = rnd((M, N), 12)
    dxg = ops.gemm(dy.cuda(), w.cuda(), b_kcontig=False, epilogue=capi.EPI_DGELU, aux_in=pre.cuda())
    xp = pre.float()
    dg = 0.5 * (1 + torch.erf(xp / math.sqrt(2))) + xp * torch.exp(-0.5 * xp * xp) / math.sqrt(2 * math.pi)
    assert_close_bf16(dxg, ref * dg, name="dgrad*gelu'")
    res = rnd((M, N), 13)
    dxr = ops.gemm(dy.cuda(), w.cuda(), b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=res.cuda())
    assert_close_bf16(dxr, ref + res.float(), name="dgrad+res")


@pytest.mark.parametrize("R,M,N", [(728, 768, 768), (182, 2304, 768), (130, 5000, 768), (1000, 768, 3072), (200, 768, 3008), (48, 768, 8)])
def test_wgrad_layout(R, M, N):
    """dW[M,N] = dy[R,M]^T . x[R,N]  (both operands k-strided), fp32 output with accumulate"""
    ops, capi = _mods()
    dy, x = rnd((R, M), 14), rnd((R, N), 15)
    ref = dy.float().t() @ x.float()
    dw = ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out_dtype=torch.float32)
    assert_close_bf16(dw, ref, ulps=0, name="wgrad")
    ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out=dw, accumulate=True)
    assert_close_bf16(dw, 2 * ref, ulps=0, name="wgrad accumulate")


@pytest.mark.parametrize("R,M,N,split", [(11648, 768, 768, -1), (1000, 2304, 768, 4), (182, 768, 3072, -1), (5000, 768, 8, 7), (728, 5000, 768, -1)])
def test_wgrad_split_k_and_fused_bias_grad(R, M, N, split):
    ops, capi = _mods()
    dy, x = rnd((R, M), 18), rnd((R, N), 19)
    ref = dy.float().t() @ x.float()
    base = torch.randn(M, N, generator=torch.Generator().manual_seed(20))
    dw = base.clone().cuda()
    db = torch.full((M,), 0.5, device="cuda")
    ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out=dw, accumulate=True, split_k=split, bias_grad=db)
    assert_close_bf16(dw, base + ref, ulps=0, name="split-k wgrad")
    assert_close_bf16(db, 0.5 + dy.float().sum(0), ulps=0, name="fused bias grad")
    dw2, db2 = base.clone().cuda(), torch.full((M,), 0.5, device="cuda")
    ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out=dw2, accumulate=True, split_k=split, bias_grad=db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed summation order: bit-reproducible
    with pytest.raises(capi.SamHipError):          # a bias gradient only comes out of the accumulate (wgrad) form of split-K
        ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out=dw, accumulate=False, split_k=4, bias_grad=db)
    ops.gemm(dy.cuda(), x.cuda(), a_kcontig=False, b_kcontig=False, out=dw, accumulate=False, split_k=4)      # store form: partials summed, C overwritten
    assert_close_bf16(dw, ref, ulps=0, name="split-k store")


@pytest.mark.parametrize("M,N,K,split", [(1280, 768, 3072, -1), (768, 768, 5000, -1), (100, 72, 1600, 3), (1280, 3072, 768, -1), (64, 8, 4096, 8)])
def test_split_k_with_epilogue_matches_unsplit(M, N, K, split):
    """skinny problems (TextBert's 20 tokens/sample, classifier dgrad) split K and fold the epilogue into the partial-sum reduction:
    every epilogue, forward and dgrad layouts, against the un-split kernel (force_tile=64 disables the automatic split) and torch"""
    ops, capi = _mods()
    x, w, dy = rnd((M, K), 31), rnd((N, K), 32, 0.05), rnd((M, K), 33)
    wT = rnd((K, N), 34, 0.05)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(35)).cuda()
    res, pre = rnd((M, N), 36).cuda(), rnd((M, N), 37).cuda()
    xg, wg, dyg, wTg = x.cuda(), w.cuda(), dy.cuda(), wT.cuda()
    ref = x.float() @ w.float().t()
    got = ops.gemm(xg, wg, split_k=split)
    assert_close_bf16(got, ref, name="split fwd none")
    got = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bias, out_dtype=torch.float32, split_k=split)
    assert_close_bf16(got, ref + bias.cpu(), ulps=0, name="split fwd bias f32")
    aux = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    got = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_GELU, bias=bias, aux_out=aux, split_k=split)
    assert_close_bf16(aux, ref + bias.cpu(), name="split gelu pre-activation")
    assert_close_bf16(got, gelu(ref + bias.cpu()), name="split gelu")
    # dgrad layout: dy[M,K] . W[K,N]
    refd = dy.float() @ wT.float()
    got = ops.gemm(dyg, wTg, b_kcontig=False, split_k=split)
    assert_close_bf16(got, refd, name="split dgrad none")
    got = ops.gemm(dyg, wTg, b_kcontig=False, epilogue=capi.EPI_DGELU, aux_in=pre, split_k=split)
    one = ops.gemm(dyg, wTg, b_kcontig=False, epilogue=capi.EPI_DGELU, aux_in=pre, force_tile=64)
    assert_close_bf16(got, one.float(), ulps=2, name="split dgelu vs unsplit")
    for p in (0.0, 0.1):
        got = ops.gemm(dyg, wTg, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=p, seed=5, offset=9, split_k=split)
        one = ops.gemm(dyg, wTg, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=p, seed=5, offset=9, force_tile=64)
        assert_close_bf16(got, one.float(), ulps=2, name="split dropout+res vs unsplit (same Philox stream) p=%g" % p)
    if p and M * N > 50000:
        dropped = ((got.float() - res.float()).abs() < 1e-6).float().mean().item()
        assert abs(dropped - 0.1) < 0.02, dropped
    # bit-reproducible
    a1 = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bias, split_k=split)
    a2 = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bias, split_k=split)
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("tile", [64, 128, 160, 192, 256])
def test_both_tile_configs_all_layouts(tile):
    ops, capi = _mods()
    M, N, K = 600, 520, 712          # ragged in every dimension for both tile sizes; K % 64 != 0
    x, w, b = rnd((M, K), 21), rnd((N, K), 22, 0.05), torch.randn(N, generator=torch.Generator().manual_seed(23))
    assert_close_bf16(ops.gemm(x.cuda(), w.cuda(), epilogue=capi.EPI_BIAS, bias=b.cuda(), force_tile=tile), x.float() @ w.float().t() + b, name="fwd")
    dy, w2 = rnd((M, K), 24), rnd((K, N), 25, 0.05)
    assert_close_bf16(ops.gemm(dy.cuda(), w2.cuda(), b_kcontig=False, force_tile=tile), dy.float() @ w2.float(), name="dgrad")
    R = 1000
    dyy, xx = rnd((R, 520), 26), rnd((R, 264), 27)
    dw, db = torch.zeros(520, 264, device="cuda"), torch.zeros(520, device="cuda")
    ops.gemm(dyy.cuda(), xx.cuda(), a_kcontig=False, b_kcontig=False, out=dw, accumulate=True, split_k=-1, bias_grad=db, force_tile=tile)
    assert_close_bf16(dw, dyy.float().t() @ xx.float(), ulps=0, name="wgrad")
    assert_close_bf16(db, dyy.float().sum(0), ulps=0, name="bias grad")


@pytest.mark.parametrize("tile", [1192, 3192, 1256, 1448, 1128, 12192, 12448])      # 12192 / 12448: the 12-wave kernels with loader waves (gemm12.hip); 3192: 192x192 with the deferred (sliced, LDS-staged) epilogue (opt-in: measured slower); 1448: 192x256
@pytest.mark.parametrize("M,N,K", [(3500, 3080, 128), (600, 520, 64), (4000, 2304, 192), (256, 256, 704), (11648, 768, 768)])
def test_eight_wave_persistent_kernels(tile, M, N, K):
    """gemm8.hip (256x256 / 192x192 tiles, one block per CU walking several tiles): ragged M and N, one to eleven k-tiles per tile, more tiles
    than CUs (so blocks cross tile boundaries with operands of the next tile in flight), forward and dgrad layouts, every fused epilogue"""
    ops, capi = _mods()
    x, w = rnd((M, K), 61), rnd((N, K), 62, 0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(63)) * 0.1
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    ref = x.float() @ w.float().t()
    assert_close_bf16(ops.gemm(xg, wg, force_tile=tile), ref, name="fwd none")
    assert_close_bf16(ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bg, force_tile=tile), ref + b, name="fwd bias")
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    h = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=bg, aux_out=pre, force_tile=tile)
    assert_close_bf16(pre, dgelu(ref + b), name="fwd gelu'")
    assert_close_bf16(h, gelu(ref + b), name="fwd gelu")
    res = rnd((M, N), 64)
    z = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bg, residual=res.cuda(), force_tile=tile)
    assert_close_bf16(z, ref + b + res.float(), name="fwd bias+res")
    # dropout: the same Philox (row, col/8) stream as the 4-wave kernels -> identical keep pattern, values equal up to the accumulation order
    zd = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bg, residual=res.cuda(), p_drop=0.1, seed=11, offset=3, force_tile=tile)
    z4 = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bg, residual=res.cuda(), p_drop=0.1, seed=11, offset=3, force_tile=128)
    assert_close_bf16(zd, z4.float(), ulps=2, name="fwd dropout+res vs 4-wave kernel")
    # dgrad layout: dy[M,K] . W[K,N]
    wT = rnd((K, N), 65, 0.05)
    refd = x.float() @ wT.float()
    assert_close_bf16(ops.gemm(xg, wT.cuda(), b_kcontig=False, force_tile=tile), refd, name="dgrad none")
    prev = rnd((M, N), 66)
    assert_close_bf16(ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=prev.cuda(), force_tile=tile), refd * prev.float(), name="dgrad * aux")
    assert_close_bf16(ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=res.cuda(), force_tile=tile), refd + res.float(), name="dgrad+res")
    # strided operand views (leading dimensions larger than the logical width), as the attention block hands them over
    big = rnd((M, K + 64), 67).cuda()
    assert_close_bf16(ops.gemm(big[:, 64:], wg, force_tile=tile), big[:, 64:].float().cpu() @ w.float().t(), name="lda view")
    # bit-reproducible
    assert torch.equal(ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bg, force_tile=tile), ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bg, force_tile=tile))
    if tile >= 12000:
        # the loader-wave kernels compute every tile exactly as the 8-wave kernels of the same tile shape do (same fragments, same MFMA order, same epilogue):
        # bit-identical outputs, several times over (a race between the loader waves' DMA and the compute waves' reads would show as a rare wrong tile)
        same = {12192: 1192, 12448: 1448}[tile]
        for _ in range(4):
            assert torch.equal(ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bg, force_tile=tile), ops.gemm(xg, wg, epilogue=capi.EPI_BIAS, bias=bg, force_tile=same))
            assert torch.equal(ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=prev.cuda(), force_tile=tile),
                               ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=prev.cuda(), force_tile=same))
            assert torch.equal(ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bg, residual=res.cuda(), p_drop=0.1, seed=11, offset=3, force_tile=tile),
                               ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bg, residual=res.cuda(), p_drop=0.1, seed=11, offset=3, force_tile=same))


def test_short_k_problem_that_fills_the_chip_once_takes_the_two_block_configuration():
    """TextBert's FFN1 forward / FFN2 dgrad (1280 x 3072 x 768: 240 tiles of 128 x 128) are sent to the 8-wave kernel's 128 x 128 two-blocks-per-CU configuration by
    the default picker (gemm8.hip): same results as the forced configuration bit for bit, and as the 4-wave 64 x 64 kernel up to the accumulation order"""
    ops, capi = _mods()
    M, N, K = 1280, 3072, 768
    x, w = rnd((M, K), 71), rnd((N, K), 72, 0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(73)) * 0.1
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    outs = {}
    for tile in (0, 1128, 64):
        pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        h = ops.gemm(xg, wg, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=bg, aux_out=pre, force_tile=tile)
        outs[tile] = (h, pre)
    assert torch.equal(outs[0][0], outs[1128][0]) and torch.equal(outs[0][1], outs[1128][1])
    assert_close_bf16(outs[0][0], outs[64][0].float(), ulps=2, name="128x128 vs 64x64 gelu")
    ref = x.float() @ w.float().t() + b
    assert_close_bf16(outs[0][0], gelu(ref), name="fwd gelu"); assert_close_bf16(outs[0][1], dgelu(ref), name="fwd gelu'")
    wT, aux = rnd((K, N), 74, 0.05), rnd((M, N), 75)
    d0 = ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=aux.cuda())
    d1 = ops.gemm(xg, wT.cuda(), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=aux.cuda(), force_tile=1128)
    assert torch.equal(d0, d1)
    assert_close_bf16(d0, (x.float() @ wT.float()) * aux.float(), name="dgrad * aux")


def test_eight_wave_kernels_decline_what_they_cannot_do():
    ops, capi = _mods()
    x, w = rnd((512, 72), 71).cuda(), rnd((512, 72), 72).cuda()
    with pytest.raises(capi.SamHipError):
        ops.gemm(x, w, force_tile=1192)              # K % 64 != 0: no partial k-tiles there
    y = ops.gemm(x, w)                               # the heuristic path falls back to the 4-wave kernels
    assert_close_bf16(y, x.float().cpu() @ w.float().cpu().t(), name="fallback")


def test_grouped_wgrad_matches_individual():
    ops, capi = _mods()
    R = 1456
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    jobs, refs = [], []
    for j, (m, n) in enumerate(shapes):
        dy, x = rnd((R, m), 30 + j), rnd((R, n), 40 + j)
        base = torch.randn(m, n, generator=torch.Generator().manual_seed(50 + j))
        dw = base.clone().cuda()
        db = torch.full((m,), 0.25, device="cuda") if j % 2 else None
        jobs.append((dy.cuda(), x.cuda(), dw, db))
        refs.append((base + dy.float().t() @ x.float(), None if db is None else 0.25 + dy.float().sum(0)))
    ops.wgrad_grouped(jobs)
    for (dy, x, dw, db), (rw, rb) in zip(jobs, refs):
        assert_close_bf16(dw, rw, ulps=0, name="grouped wgrad")
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="grouped bias grad")
    # twelve problems in one launch (three layers' weight gradients: autograd.DeferredWgrads): every gradient gets its own three contributions
    for _, _, dw, db in jobs:
        dw.zero_()
        if db is not None:
            db.zero_()
    trip = [(dy, x, dw.clone(), None if db is None else db.clone()) for _ in range(3) for dy, x, dw, db in jobs]
    ops.wgrad_grouped(trip)
    for k, (dy, x, dw, db) in enumerate(trip):
        assert_close_bf16(dw, dy.float().t() @ x.float(), ulps=0, name="12-problem grouped wgrad %d" % k)
        if db is not None:
            assert_close_bf16(db, dy.float().sum(0), ulps=0, name="12-problem grouped bias grad %d" % k)
    with pytest.raises(capi.SamHipError):
        ops.wgrad_grouped(jobs * 6)         # more than 20 problems


def _wgrad_jobs(R, shapes, seed0=0, bias=(1, 3), want_ref=True):
    """(jobs, refs): refs = the plain PyTorch fp32 products of the same bf16 operands (torch.matmul in fp32 on the device, TF32 off: on the pool's slower hosts the
    CPU matmuls of these 8-20 problem sets were most of the tests' wall time); None when the caller only wants the jobs"""
    jobs, refs = [], []
    torch.backends.cuda.matmul.allow_tf32 = False
    for j, (m, n) in enumerate(shapes):
        dy, x = rnd((R, m), seed0 + 30 + j).cuda(), rnd((R, n), seed0 + 40 + j).cuda()
        base = torch.randn(m, n, generator=torch.Generator().manual_seed(seed0 + 50 + j)).cuda()
        db = torch.full((m,), 0.25, device="cuda") if j in bias else None
        jobs.append((dy, x, base.clone(), db))
        if want_ref:
            refs.append((base + dy.float().t() @ x.float(), None if db is None else 0.25 + dy.float().sum(0)))
    return jobs, (refs if want_ref else None)


@pytest.mark.parametrize("R", [1024, 1088, 4480])
def test_grouped_wgrad_eight_wave_pair_exchange(R):
    """the encoder layer's four weight gradients on the 8-wave kernel: 108 tiles of 256 x 256, every tile's K range split over a PAIR of blocks that
    swap accumulator halves through the workspace inside the launch.  Against an fp32 reference, against the 4-wave kernel, bit-identical
    between runs, and the pair flags are back at zero after every launch (R = 1088: an odd number of k-tiles, the halves are 9 + 8)."""
    ops, capi = _mods()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    jobs, refs = _wgrad_jobs(R, shapes)
    ops.wgrad_grouped(jobs, force_tile=1256)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(jobs, refs):
        assert_close_bf16(dw, rw, ulps=0, name="8-wave grouped wgrad")
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="8-wave grouped bias grad")
    ws = ops._grouped_ws(jobs[0][0].device, 0)
    assert int(ws[:216].view(torch.int32).abs().sum()) == 0, "pair flags must be consumed"
    first = [(j[2].clone(), None if j[3] is None else j[3].clone()) for j in jobs]
    for _ in range(3):                                   # same inputs, fresh accumulators: bit-identical results, launch after launch
        again, _ = _wgrad_jobs(R, shapes, want_ref=False)
        ops.wgrad_grouped(again, force_tile=1256)
        for (a, b), j in zip(first, again):
            assert torch.equal(a, j[2])
            assert b is None or torch.equal(b, j[3])
    old, _ = _wgrad_jobs(R, shapes, want_ref=False)
    ops.wgrad_grouped(old, force_tile=128)               # the 4-wave kernel: same products, fp32 sums in another order
    for (a, b), j in zip(first, old):
        assert float((a - j[2]).abs().max()) <= 2e-4 * (1.0 + float(j[2].abs().max()))
        assert b is None or float((b - j[3]).abs().max()) <= 2e-4 * (1.0 + float(j[3].abs().max()))


@pytest.mark.parametrize("R,force", [(1024, 1256), (1024, 128), (1456, 0)])
def test_grouped_wgrad_overwrite_equals_accumulate_into_zero(R, force):
    """accumulate = False: dW and the fused bias gradient are overwritten -- bit-identical to accumulating into zeroed buffers, whatever was there before
    (the Trainer leaves the encoder layers' gradients un-zeroed and lets their backward overwrite them).  Both grouped kernels."""
    ops, capi = _mods()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    jobs, _ = _wgrad_jobs(R, shapes, want_ref=False)
    zero = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for dy, x, dw, db in jobs]
    ops.wgrad_grouped(zero, force_tile=force)
    junk = [(dy, x, torch.full_like(dw, 7.5), None if db is None else torch.full_like(db, -3.0)) for dy, x, dw, db in jobs]
    ops.wgrad_grouped(junk, force_tile=force, accumulate=False)
    for a, b in zip(zero, junk):
        assert torch.equal(a[2], b[2])
        assert a[3] is None or torch.equal(a[3], b[3])


def test_grouped_wgrad_mixed_depths_in_one_launch():
    """an MMT layer pair (8 problems, 216 tiles, deep K) and TextBert's three layers (12 problems, 324 tiles, shallow K) as ONE launch of 20 problems: the deep
    tiles are dispatched first, one per CU, the shallow ones fill the idle CUs and follow as they finish (gemm8w.hip, n_long).  Against fp32, with fused bias
    gradients, overwrite mode, and bit-identical to the two separate launches (every tile sums its own K range whole, in the same order)."""
    ops, capi = _mods()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    deep, deep_ref = _wgrad_jobs(2048, shapes * 2, seed0=100, bias=(0, 3, 5))
    shal, shal_ref = _wgrad_jobs(320, shapes * 3, seed0=200, bias=(1, 2, 7, 11))
    mixed = shal[:5] + deep + shal[5:]                    # any order: the launch sorts by depth
    ops.wgrad_grouped(mixed, force_tile=1256)
    for (dy, x, dw, db), (rw, rb) in zip(deep + shal, deep_ref + shal_ref):
        assert_close_bf16(dw, rw, ulps=0, name="mixed-depth grouped wgrad")
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="mixed-depth grouped bias grad")
    d2, _ = _wgrad_jobs(2048, shapes * 2, seed0=100, bias=(0, 3, 5), want_ref=False)
    s2, _ = _wgrad_jobs(320, shapes * 3, seed0=200, bias=(1, 2, 7, 11), want_ref=False)
    ops.wgrad_grouped(d2, force_tile=1256)
    ops.wgrad_grouped(s2, force_tile=1256)
    for a, b in zip(deep + shal, d2 + s2):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))
    junk = [(dy, x, torch.full_like(dw, 3.5), None if db is None else torch.full_like(db, -1.0)) for dy, x, dw, db in mixed]
    ops.wgrad_grouped(junk, force_tile=1256, accumulate=False)
    zero = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for dy, x, dw, db in mixed]
    ops.wgrad_grouped(zero, force_tile=1256)
    for a, b in zip(zero, junk):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))


def test_grouped_wgrad_more_than_twelve_problems_the_eight_wave_kernel_declines():
    """20 problems whose depths are not multiples of 64 (a batch that is not a multiple of 32): the 8-wave grouped kernel declines, the set goes
    out as chunks of at most 12 on the 4-wave kernel instead of raising (whether the call succeeds must not depend on K % 64 or the CU count); fp32 reference,
    fused bias gradients, per-job accumulate flags, and the forced 4-wave route (force_tile = 128) the same way"""
    ops, capi = _mods()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    for force in (0, 128):
        deep, deep_ref = _wgrad_jobs(1096, shapes * 2, seed0=300, bias=(0, 3, 5))        # 6 * 182 + 4: K % 64 = 8
        shal, shal_ref = _wgrad_jobs(120, shapes * 3, seed0=400, bias=(1, 2, 7, 11))     # 6 * 20
        ops.wgrad_grouped(deep + shal, force_tile=force)
        for (dy, x, dw, db), (rw, rb) in zip(deep + shal, deep_ref + shal_ref):
            assert_close_bf16(dw, rw, ulps=0, name="20-problem chunked grouped wgrad")
            if db is not None:
                assert_close_bf16(db, rb, ulps=0, name="20-problem chunked grouped bias grad")
    junk = [(dy, x, torch.full_like(dw, 3.5), None if db is None else torch.full_like(db, -1.0)) for dy, x, dw, db in deep + shal]
    ops.wgrad_grouped(junk, accumulate=False)
    zero = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for dy, x, dw, db in deep + shal]
    ops.wgrad_grouped(zero)
    for a, b in zip(zero, junk):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))


@pytest.mark.parametrize("R", [1472, 2048])
def test_grouped_wgrad_loader_wave_kernel_pair_slices_and_shallow_tiles(R):
    """gemm12w.hip: an MMT layer pair is 288 tiles of 192 x 256 -- 256 whole tiles (one per CU), then the K range of the 32 left-over tiles in 8 slices each, summed inside
    the launch in a fixed order; TextBert-size and head-size problems follow as whole shallow tiles.  Against fp32, fused bias gradients (whole tiles AND sliced
    tiles carry them), accumulate and overwrite, bit-identical run to run and equal (to fp32 summation order) to the 8-wave kernel; the exchange's counters are
    back at zero and its error word clear after every launch.  R = 1472: 23 k-tiles per tile, seven slices of 3 and one of 2."""
    ops, capi = _mods()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    deep, deep_ref = _wgrad_jobs(R, shapes * 2, seed0=500, bias=(0, 1, 3, 5, 6))          # (3072 x 768 with bias: its first tile column includes sliced tiles)
    ops.wgrad_grouped(deep, force_tile=12448)
    ops.grouped_ws_check()
    for (dy, x, dw, db), (rw, rb) in zip(deep, deep_ref):
        assert_close_bf16(dw, rw, ulps=0, name="loader-wave grouped wgrad")
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="loader-wave grouped bias grad")
    first = [(j[2].clone(), None if j[3] is None else j[3].clone()) for j in deep]
    for _ in range(3):                                    # fresh accumulators, same inputs: bit-identical, launch after launch (also: counters really returned to zero)
        again, _ = _wgrad_jobs(R, shapes * 2, seed0=500, bias=(0, 1, 3, 5, 6), want_ref=False)
        ops.wgrad_grouped(again, force_tile=12448)
        for (a, b), j in zip(first, again):
            assert torch.equal(a, j[2]) and (b is None or torch.equal(b, j[3]))
    ops.grouped_ws_check()
    old, _ = _wgrad_jobs(R, shapes * 2, seed0=500, bias=(0, 1, 3, 5, 6), want_ref=False)
    ops.wgrad_grouped(old, force_tile=1256)               # the 8-wave kernel: same products, fp32 sums in another order
    for (a, b), j in zip(first, old):
        assert float((a - j[2]).abs().max()) <= 2e-4 * (1.0 + float(j[2].abs().max()))
        assert b is None or float((b - j[3]).abs().max()) <= 2e-4 * (1.0 + float(j[3].abs().max()))
    # with shallow problems behind the pair (TextBert's three layers at 320 rows, a ragged head-size problem), any order; overwrite == accumulate into zero
    shal, shal_ref = _wgrad_jobs(320, shapes * 2 + [(5000, 768), (768, 768)], seed0=600, bias=(1, 2, 8))
    d2, d2_ref = _wgrad_jobs(R, shapes * 2, seed0=700, bias=(2, 7))
    mixed = shal[:3] + d2 + shal[3:]
    ops.wgrad_grouped(mixed, force_tile=12448)
    for (dy, x, dw, db), (rw, rb) in zip(d2 + shal, d2_ref + shal_ref):
        assert_close_bf16(dw, rw, ulps=0, name="loader-wave mixed-depth wgrad")
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="loader-wave mixed-depth bias grad")
    junk = [(dy, x, torch.full_like(dw, 3.5), None if db is None else torch.full_like(db, -1.0)) for dy, x, dw, db in mixed]
    ops.wgrad_grouped(junk, force_tile=12448, accumulate=False)
    zero = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for dy, x, dw, db in mixed]
    ops.wgrad_grouped(zero, force_tile=12448)
    for a, b in zip(zero, junk):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))
    ops.grouped_ws_check()
    # a set with fewer deep tiles than CUs is not one for this kernel: forcing it is an error, the default goes to the 8-wave kernel
    one, _ = _wgrad_jobs(R, shapes, want_ref=False)
    with pytest.raises(capi.SamHipError):
        ops.wgrad_grouped(one, force_tile=12448)


def test_grouped_wgrad_eight_wave_ragged_and_unsplit():
    """tile edges (M, N multiples of 8 but not of 256), a single problem, and a problem set with too many tiles for pairs (one block per tile)"""
    ops, capi = _mods()
    for shapes, R, bias in [([(200, 328), (520, 264)], 512, (0,)), ([(768, 768)], 2048, (0,)), ([(3072, 3072)], 512, (0,))]:
        jobs, refs = _wgrad_jobs(R, shapes, seed0=7, bias=bias)
        ops.wgrad_grouped(jobs, force_tile=1256)
        for (dy, x, dw, db), (rw, rb) in zip(jobs, refs):
            assert_close_bf16(dw, rw, ulps=0, name="8-wave grouped wgrad %s" % (shapes,))
            if db is not None:
                assert_close_bf16(db, rb, ulps=0, name="8-wave grouped bias grad %s" % (shapes,))
    jobs, _ = _wgrad_jobs(1456, [(768, 768)], want_ref=False)           # K % 64 != 0: not a problem for this kernel; forcing it is an error, the default falls back
    with pytest.raises(capi.SamHipError):
        ops.wgrad_grouped(jobs, force_tile=1256)
    ops.wgrad_grouped(jobs)


def test_strided_views_and_errors():
    ops, capi = _mods()
    big = rnd((300, 2304), 16).cuda()
    w = rnd((768, 768), 17, 0.05).cuda()
    y = ops.gemm(big[:, 768:1536], w)          # leading dimension 2304, K = 768
    assert_close_bf16(y, big[:, 768:1536].float().cpu() @ w.float().cpu().t(), name="lda view")
    with pytest.raises(capi.SamHipError):
        ops.gemm(rnd((16, 12), 1).cuda(), rnd((16, 12), 2).cuda())      # K % 8 != 0
    with pytest.raises(capi.SamHipError):
        ops.gemm(big, w, out=torch.empty(300, 768, dtype=torch.bfloat16, device="cuda"), accumulate=True)


def test_c_abi_from_a_plain_cxx_host_program(tmp_path):
    """no Python between the caller and the library: LayerNorm, bias GEMM and prefix-LM attention through include/sam_hip.h from C++,
    each checked against a double-precision host computation inside the program"""
    import subprocess
    from tests.cabi_host import build_host_smoke
    exe = build_host_smoke(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_OK" in r.stdout, r.stdout[-3000:]


def test_layernorm_inside_the_splitk_reduction_is_bit_identical_to_two_launches():
    """sam_ln_fuse: LayerNorm(dropout(x W^T + b) + residual) (BertSelfOutput / BertOutput, sa_m4c.py:653,680,1016-1028).  When the library splits K (skinny M,
    long K: TextBert's 1280 rows, the 320 / 64 rows of a decoding step) the pass that sums the partials normalises the rows it owns: z, y, mean, rstd must be
    the bits of gemm + layernorm_fwd; when it does not split, `done` stays 0 and gemm_ln runs the separate LayerNorm"""
    import torch
    from sam_textvqa_amd import _capi as capi, ops
    g = torch.Generator().manual_seed(12)
    for (m, n, k, p_drop, expect_fused) in ((1280, 768, 3072, 0.1, True), (320, 768, 3072, 0.0, True), (64, 768, 3072, 0.1, True), (1283, 512, 2304, 0.1, True),
                                           (1280, 768, 768, 0.1, False), (11648, 768, 3072, 0.1, False)):
        a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
        bias = torch.randn(n, generator=g).cuda()
        res = torch.randn(m, n, generator=g).to(torch.bfloat16).cuda()
        gamma, beta = (1 + 0.1 * torch.randn(n, generator=g)).cuda(), (0.1 * torch.randn(n, generator=g)).cuda()
        kw = dict(epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=p_drop, seed=7, offset=3)
        z0 = ops.gemm(a, w, **kw)
        y0, mean0, rstd0 = ops.layernorm_fwd(z0, gamma, beta, 1e-12)
        ln = capi.LnFuse()
        y = torch.empty_like(z0); mean = torch.empty(m, device="cuda"); rstd = torch.empty(m, device="cuda")
        ln.gamma, ln.beta, ln.eps, ln.y, ln.ldy, ln.mean, ln.rstd, ln.done = gamma.data_ptr(), beta.data_ptr(), 1e-12, y.data_ptr(), y.stride(0), mean.data_ptr(), rstd.data_ptr(), 5
        z1 = ops.gemm(a, w, ln=ln, **kw)
        assert bool(ln.done) == expect_fused, (m, n, k, ln.done)
        assert torch.equal(z0, z1)
        if ln.done:
            assert torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0), (m, n, k)
        z2, y2, mean2, rstd2 = ops.gemm_ln(a, w, gamma, beta, 1e-12, **kw)
        assert torch.equal(z2, z0) and torch.equal(y2, y0) and torch.equal(mean2, mean0) and torch.equal(rstd2, rstd0)


@pytest.mark.parametrize("m,n,k,p_drop", [(11648, 768, 768, 0.1), (11648, 768, 3072, 0.1), (11200, 768, 768, 0.0), (4000, 768, 768, 0.1), (11648, 1536, 768, 0.1), (13000, 768, 768, 0.1)])
def test_layernorm_inside_the_mmt_size_launch(m, n, k, p_drop, monkeypatch):
    """VERDICT r5 missing #1 / SURVEY 8(b) `linear_bias_dropout_residual_ln` as ONE kernel at MMT size (sa_m4c.py:653, 680, 1016-1028): a product the loader-wave
    kernel covers in one round of tiles (11648 x 768: 244 tiles of 192 x 192) normalises its rows inside the launch -- the waves and blocks that share a row
    exchange (mean, M2) pairs (gemm_common.h: gemm_ln_pass).  z is the two-launch form's bit for bit; mean / rstd agree to fp32 rounding, y to one bf16 ulp of the
    largest output; launch after launch the same bits; the workspace's counters are back at zero; shapes that need more than one round (N = 1536; 13000 rows)
    fall back to the separate LayerNorm and say so (`done` = 0).  Opt-in (SAM_GEMM_LN_FUSE=1): measured slower than the two launches (profiles/r6_gemm_experiments.txt #16)."""
    import torch
    from sam_textvqa_amd import _capi as capi, ops
    monkeypatch.setenv("SAM_GEMM_LN_FUSE", "1")
    g = torch.Generator().manual_seed(21)
    a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(n, generator=g).cuda()
    res = torch.randn(m, n, generator=g).to(torch.bfloat16).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(n, generator=g)).cuda(), (0.1 * torch.randn(n, generator=g)).cuda()
    kw = dict(epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=bias, residual=res, p_drop=p_drop, seed=7, offset=3)
    z0 = ops.gemm(a, w, **kw)
    y0, mean0, rstd0 = ops.layernorm_fwd(z0, gamma, beta, 1e-12)
    expect = n == 768 and 7680 <= m <= 12288          # the loader-wave kernel's range (>= 160 tiles of 192 x 192) and ONE round on 256 CUs (<= 64 row tiles x 4)
    ln = capi.LnFuse()
    y = torch.full_like(z0, 7.0); mean = torch.full((m,), 7.0, device="cuda"); rstd = torch.full((m,), 7.0, device="cuda")
    xws = ops._ln_xws(a.device, int(capi.call("sam_gemm_ln_ws_bytes", m, n)))
    ln.gamma, ln.beta, ln.eps, ln.y, ln.ldy, ln.mean, ln.rstd, ln.done = gamma.data_ptr(), beta.data_ptr(), 1e-12, y.data_ptr(), y.stride(0), mean.data_ptr(), rstd.data_ptr(), 5
    ln.xws, ln.xws_bytes = xws.data_ptr(), xws.numel() * 4
    z1 = ops.gemm(a, w, ln=ln, **kw)
    torch.cuda.synchronize()
    assert bool(ln.done) == expect, (m, n, k, ln.done)
    assert torch.equal(z0, z1)
    if not expect:
        z2, y2, mean2, rstd2 = ops.gemm_ln(a, w, gamma, beta, 1e-12, **kw)          # the caller-facing form: falls back, same bits as two launches
        assert torch.equal(y2, y0) and torch.equal(mean2, mean0)
        return
    assert int(xws[:ops.capi_ln_words()].view(torch.int32).abs().sum()) == 0, "error word / counters must be zero after the launch"
    assert torch.allclose(mean, mean0, rtol=0, atol=2e-6 * float(z0.float().abs().max())) and torch.allclose(rstd, rstd0, rtol=2e-6, atol=0)
    ulp = 2.0 ** -8 * float(y0.float().abs().max())
    d = (y.float() - y0.float()).abs()
    assert float(d.max()) <= ulp and float((d > 0).float().mean()) < 0.02, (float(d.max()), ulp, float((d > 0).float().mean()))
    for _ in range(3):          # bit-reproducible, and through the caller-facing form
        z2, y2, mean2, rstd2 = ops.gemm_ln(a, w, gamma, beta, 1e-12, **kw)
        assert torch.equal(z2, z0) and torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    ops.ln_xws_check()


@pytest.mark.parametrize("reserve", [16, 32, 40, 64])
def test_persistent_grids_with_cus_withheld_compute_the_same(reserve):
    """sam_set_cu_reserve (VERDICT r5 missing #3): the persistent grids sized for (CUs - reserve).  A tile's value does not depend on which block computes it:
    every forward / dgrad GEMM and the one-pass attention backward are BIT-identical to the full-grid launch; the grouped weight gradient keeps every tile's
    K range whole (or split over the same pair) as long as its blocks still fit one round -- bit-identical up to a reserve of 40 for a layer pair -- and matches
    the fp32 reference beyond that."""
    ops, capi = _mods()
    M = 11648
    x, w1, w2 = rnd((M, 768), 1).cuda(), rnd((3072, 768), 2, 0.05).cuda(), rnd((768, 3072), 3, 0.05).cuda()
    b1, b2 = torch.randn(3072, device="cuda") * 0.1, torch.randn(768, device="cuda") * 0.1
    h_in, res = rnd((M, 3072), 4).cuda(), rnd((M, 768), 5).cuda()
    dy = rnd((M, 3072), 6).cuda()
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)] * 2          # a layer PAIR: 216 whole-K tiles

    def run():
        out = []
        dact = torch.empty(M, 3072, dtype=torch.bfloat16, device="cuda")
        out.append(ops.gemm(x, w1, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=b1, aux_out=dact))
        out.append(dact)
        out.append(ops.gemm(h_in, w2, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b2, residual=res, p_drop=0.1, seed=11, offset=3))
        out.append(ops.gemm(dy, w1, b_kcontig=False))                                  # dgrad layout, K = 3072
        out.append(ops.gemm(x, w1[:2304].contiguous(), epilogue=capi.EPI_BIAS, bias=b1[:2304].contiguous()))        # QKV: 256 x 256 tiles
        jobs, refs = _wgrad_jobs(1024, shapes)
        ops.wgrad_grouped(jobs)
        out += [j[2] for j in jobs] + [j[3] for j in jobs if j[3] is not None]
        torch.cuda.synchronize()
        return [o.clone() for o in out], refs, jobs

    assert ops.set_cu_reserve(0) == 0
    full, refs, _ = run()
    try:
        assert ops.set_cu_reserve(reserve) == reserve
        part, _, jobs = run()
    finally:
        ops.set_cu_reserve(0)
    for i, (a, b) in enumerate(zip(full[:5], part[:5])):
        assert torch.equal(a, b), "GEMM %d differs under a reserve of %d CUs" % (i, reserve)
    for (dyj, xj, dw, db), (rw, rb) in zip(jobs, refs):
        assert_close_bf16(dw, rw, ulps=0, name="grouped wgrad, reserve %d" % reserve)
        if db is not None:
            assert_close_bf16(db, rb, ulps=0, name="grouped bias grad, reserve %d" % reserve)
    if reserve <= 40:
        for a, b in zip(full[5:], part[5:]):
            assert torch.equal(a, b), "grouped wgrad differs under a reserve of %d CUs" % reserve
