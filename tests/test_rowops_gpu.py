"""GPU parity: layernorm fwd/bwd (+ fused dropout-masked gradient and bias/gain reductions), colsum, masked BCE
loss, pointer-net scores, gradient-norm + Adam — each against the oracle / torch fp32 on the same inputs."""
import math

import numpy as np
import pytest
import torch

from oracle import sa_m4c_oracle as O
from tests import oracle_cases as OC
from tests.golden import common as C
from tests.util import assert_close_bf16

pytestmark = pytest.mark.gpu


def _mods():
    from sam_textvqa_amd import _capi, ops
    return ops, _capi


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def test_layernorm_golden_primitives():
    """the reference's own BertLayerNorm numbers (tests/golden/primitives.npz), D=96 fp32 input"""
    ops, _ = _mods()
    g = OC.load("primitives")
    x = torch.from_numpy(C.det_uniform("prim.x", (7, 96), -4, 4))
    w = torch.from_numpy(C.det_param("prim.LayerNorm.weight", (96,), 0.1))
    b = torch.from_numpy(C.det_param("prim.LayerNorm.bias", (96,), 0.1))
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-12)
    assert_close_bf16(y, torch.from_numpy(g["ln_out"]), name="LN golden fwd")
    gy = torch.from_numpy(C.det_uniform("prim.gy", (7, 96))).to(torch.bfloat16)
    dg, db = torch.zeros(96, device="cuda"), torch.zeros(96, device="cuda")
    dx, _ = ops.layernorm_bwd(gy.cuda(), x.cuda(), mean, rstd, w.cuda(), dg, db)
    # golden used the fp32 gy; ours is bf16-rounded -> compare against the oracle on the rounded gy as well
    xo = x.clone().requires_grad_(True)
    ln = O.BertLayerNorm(96); ln.weight.data.copy_(w); ln.bias.data.copy_(b)
    (ln(xo) * gy.float()).sum().backward()
    assert_close_bf16(dx, xo.grad, name="LN dx")
    assert_close_bf16(dg, ln.weight.grad, ulps=0, name="LN dgamma"); assert_close_bf16(db, ln.bias.grad, ulps=0, name="LN dbeta")
    assert_close_bf16(dx, torch.from_numpy(g["ln_dx"]), frac=6e-3, name="LN dx vs golden (bf16 gy)")


@pytest.mark.parametrize("M,D,in_dtype", [(1000, 768, torch.bfloat16), (37, 768, torch.float32), (5, 3072, torch.bfloat16), (130, 256, torch.bfloat16),
                                         (1, 8, torch.bfloat16), (3, 2048, torch.bfloat16), (2049, 768, torch.bfloat16),      # one row, the widest row, one row past 512 x 4
                                         (13001, 768, torch.bfloat16), (9000, 1024, torch.bfloat16)])     # two full trips of the backward's 3-slot (2-slot) row ring + a ragged rest
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_layernorm_fwd_bwd(M, D, in_dtype, p_drop):
    ops, capi = _mods()
    if D > 2048:
        with pytest.raises(capi.SamHipError):
            ops.layernorm_fwd(rnd((M, D), 1).cuda(), torch.ones(D).cuda(), torch.zeros(D).cuda(), 1e-12)
        return
    x = rnd((M, D), 1, 2.0, in_dtype)
    w = 1 + 0.1 * rnd((D,), 2, dtype=torch.float32); b = 0.1 * rnd((D,), 3, dtype=torch.float32)
    dy = rnd((M, D), 4)
    ln = O.BertLayerNorm(D); ln.weight.data.copy_(w); ln.bias.data.copy_(b)
    xo = x.float().requires_grad_(True)
    yo = ln(xo); (yo * dy.float()).sum().backward()
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-12)
    assert_close_bf16(y, yo, name="LN fwd")
    dg, db, dbias = (torch.zeros(D, device="cuda") for _ in range(3))
    dx, dxd = ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, w.cuda(), dg, db, dbias, want_dropped=True, p_drop=p_drop, seed=5, offset=9)
    assert_close_bf16(dx, xo.grad, name="LN dx")
    assert_close_bf16(dg, ln.weight.grad, ulps=0, name="dgamma"); assert_close_bf16(db, ln.bias.grad, ulps=0, name="dbeta")
    if p_drop == 0:
        assert dxd.data_ptr() == dx.data_ptr()
        assert_close_bf16(dbias, xo.grad.sum(0), ulps=0, frac=2e-3, name="dbias")
    else:
        # the dropped copy must use exactly the mask of the forward GEMM epilogue with the same (seed, offset)
        ones = torch.ones(M, 8, dtype=torch.bfloat16)
        wsel = torch.zeros(D, 8, dtype=torch.bfloat16); wsel[:, 0] = 1       # acc = 1 everywhere
        z = ops.gemm(ones.cuda(), wsel.cuda(), epilogue=capi.EPI_BIAS_DROPOUT_RES, p_drop=p_drop, seed=5, offset=9).float().cpu()
        keep = z > 0
        inv = 1.0 / (1.0 - round(p_drop * 65536) / 65536.0)
        assert M * D < 4096 or abs(keep.float().mean().item() - (1 - p_drop)) < 0.02          # (a statistic: not for the 8-element case)
        ref_dxd = torch.where(keep, xo.grad * inv, torch.zeros(()))
        assert_close_bf16(dxd, ref_dxd, name="dropped dx")
        assert_close_bf16(dbias, ref_dxd.sum(0), ulps=0, frac=2e-3, name="dbias (dropped)")
    # accumulate semantics
    dg2 = dg.clone()
    ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, w.cuda(), dg2, db.clone())
    assert torch.allclose(dg2, 2 * dg, rtol=1e-5, atol=1e-5)


def test_wave_sum_is_the_xor_butterfly_bit_for_bit():
    """wave_sum (common.h) runs the xor-butterfly 32, 16, 8, 4, 2, 1 through v_permlane swaps and DPP instead of ds_bpermute: the LayerNorm forward's row mean,
    which is wave_sum(lane partials) / D, must equal an fp32 emulation of exactly that addition order (lane l owns the 4-element chunks l, l + 64, l + 128)"""
    ops, _ = _mods()
    M, D = 64, 768
    x = rnd((M, D), 21, 3.0, torch.float32)
    _, mean, _ = ops.layernorm_fwd(x.cuda(), torch.ones(D).cuda(), torch.zeros(D).cuda(), 1e-12)
    v = x.numpy().reshape(M, 3, 64, 4)                       # [row, j, lane, e]
    s = np.zeros((M, 64), np.float32)
    for j in range(3):
        s = s + ((v[:, j, :, 0] + v[:, j, :, 1]) + (v[:, j, :, 2] + v[:, j, :, 3]))
    lanes = np.arange(64)
    for o in (32, 16, 8, 4, 2, 1):
        s = s + s[:, lanes ^ o]
    assert (s == s[:, :1]).all()                             # every lane ends with the same bits
    want = s[:, 0] / np.float32(D)
    assert np.array_equal(mean.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_colsum():
    ops, _ = _mods()
    x = rnd((1000, 2304), 7)
    out = torch.zeros(2304, device="cuda")
    ops.colsum(x.cuda()[:, :768], out[:768])
    assert_close_bf16(out[:768], x[:, :768].float().sum(0), ulps=0, name="colsum view")
    ops.colsum(x.cuda(), out, accumulate=False)
    assert_close_bf16(out, x.float().sum(0), ulps=0, name="colsum")


@pytest.mark.parametrize("R,V,No", [(36, 5000, 50), (24, 301, 7), (12, 40, 6)])      # the bench's sizes; odd widths (element-wise path); the golden's
def test_bce_loss_matches_reference_loss(R, V, No):
    ops, _ = _mods()
    fixed, ocr = rnd((R, V), 8, 3.0, torch.float32), rnd((R, No), 9, 3.0, torch.float32)
    ocr[:, No - No // 5:] = -10000.0                               # padded OCR columns carry the literal -10000
    t = (torch.rand(R, V + No, generator=torch.Generator().manual_seed(10)) > 0.98).float()
    mask = (torch.arange(R) % 12 < 5).float()
    s = torch.cat([fixed, ocr], 1).view(R // 12, 12, V + No).requires_grad_(True)
    ref = O.m4c_decoding_bce_with_mask_loss(s, t.view(R // 12, 12, -1), mask.view(R // 12, 12))
    ref.backward()
    loss, dfix, docr = ops.bce_loss(fixed.cuda(), ocr.cuda(), t.cuda(), mask.cuda(), grad_scale=0.5)
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
    g = s.grad.view(R, -1) * 0.5
    assert_close_bf16(dfix, g[:, :V], name="d fixed"); assert_close_bf16(docr, g[:, V:], ulps=0, name="d ocr")
    # all-zero mask: count clamps to 1, loss 0, grads 0 (task_utils.py:28)
    loss0, d0, _ = ops.bce_loss(fixed.cuda(), ocr.cuda(), t.cuda(), torch.zeros(R).cuda())
    assert loss0.item() == 0.0 and (d0 == 0).all()


@pytest.mark.parametrize("S,No,D", [(12, 50, 768), (30, 100, 768), (5, 7, 64)])      # the model's shape (matrix-core kernel); the stress shape and a small one (dot-product kernel)
def test_ptr_scores_golden_shapes(S, No, D):
    ops, _ = _mods()
    B = 3
    q, k = rnd((B, S, D), 11), rnd((B, No, D), 12)
    mask = torch.from_numpy(C.pad_mask([No, 0, No // 3], No))
    scale = 1.0 / math.sqrt(D)
    qo, ko = q.float().requires_grad_(True), k.float().requires_grad_(True)
    ref = qo @ ko.transpose(-1, -2) * scale + ((1.0 - mask.float()) * -10000.0).unsqueeze(1)     # sa_m4c.py:891-893
    out = ops.ptr_scores_fwd(q.cuda(), k.cuda(), mask.to(torch.uint8).cuda(), scale)
    assert_close_bf16(out, ref, ulps=0, frac=1e-6, name="ptr scores")
    assert (out.cpu()[1] < -9000).all()                       # fully padded sample: every column carries -10000
    ds = rnd((B, S, No), 13, dtype=torch.float32)
    (ref * ds).sum().backward()
    dq, dk = ops.ptr_scores_bwd(ds.cuda(), q.cuda(), k.cuda(), scale)
    assert_close_bf16(dq, qo.grad, name="ptr dq"); assert_close_bf16(dk, ko.grad, name="ptr dk")


def test_gradnorm_and_adam_match_torch():
    ops, _ = _mods()
    n1, n2 = 1000, 3000                   # two param groups with different lr
    g = torch.Generator().manual_seed(14)
    p0 = torch.randn(n1 + n2, generator=g)
    ref_p = [p0[:n1].clone().requires_grad_(True), p0[n1:].clone().requires_grad_(True)]
    opt = torch.optim.Adam([{"params": [ref_p[0]]}, {"params": [ref_p[1]], "lr": 3e-4}], lr=1e-3)
    p, m, v = p0.clone().cuda(), torch.zeros(n1 + n2).cuda(), torch.zeros(n1 + n2).cuda()
    pb = torch.empty(n1 + n2, dtype=torch.bfloat16, device="cuda")
    nsq = torch.zeros(1, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n1 + n2, generator=g) * (10.0 if step == 2 else 0.01)
        ref_p[0].grad, ref_p[1].grad = grad[:n1].clone(), grad[n1:].clone()
        total = torch.nn.utils.clip_grad_norm_(ref_p, 0.25)
        opt.step()
        gd = grad.cuda()
        ops.sumsq(gd, nsq)
        assert abs(math.sqrt(nsq.item()) - total.item()) <= 1e-5 * total.item()
        ops.adam_step(p, gd, m, v, pb, [n1, n1 + n2], [1e-3, 3e-4], step, gnorm_sq=nsq, max_norm=0.25)
        ref = torch.cat([ref_p[0].detach(), ref_p[1].detach()])
        assert torch.allclose(p.cpu(), ref, rtol=1e-5, atol=1e-6), (p.cpu() - ref).abs().max()
        assert torch.equal(pb.cpu(), p.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("route", ["0", "1"])
def test_row_sparse_region_is_bit_identical_to_the_dense_pass(route, monkeypatch):
    """sam_sparse_rows: a table region whose untouched rows (g = m = v = 0) are skipped by the norm and by Adam gives the same norm and, bit for bit, the same
    parameters / moments / bf16 shadows as the dense pass over several steps with changing touched sets; the touched rows' gradient is cleared by the
    optimizer, the flags are set by the embedding backward; through ctypes and through torch.ops"""
    monkeypatch.setenv("SAM_COARSE_OPS", route)
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, d, head, tail = 301, 768, 4096, 1000 * 4
    lo, hi = head, head + rows * d
    n = hi + tail
    p0 = torch.randn(n, generator=g)
    state = {}
    for mode in ("dense", "sparse"):
        p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
        pb = p0.to(torch.bfloat16).cuda()                 # (the shadows exist before the first step: rows that are never updated keep them)
        grad = torch.zeros(n).cuda()
        table_g = grad[lo:hi].view(rows, d)
        touched = torch.zeros(rows, dtype=torch.uint8).cuda()
        if mode == "sparse":
            table_g._sam_touched = touched
        sp = (lo, hi, d, touched) if mode == "sparse" else None
        nsq = torch.zeros(1).cuda()
        gg = torch.Generator().manual_seed(4)
        norms = []
        for step in range(1, 5):
            if mode == "dense":
                grad[lo:hi].zero_()
            grad[:lo] = torch.randn(lo, generator=gg).cuda() * 0.1
            grad[hi:] = torch.randn(tail, generator=gg).cuda() * 0.1
            ids = torch.randint(0, rows, (40,), generator=gg)
            ids[:5] = 0                                                              # padding row: never receives a gradient
            dy = torch.randn(40, d, generator=gg).to(torch.bfloat16).cuda()
            order = torch.sort(ids, stable=True)
            ops.embedding_bwd_sorted(dy[order.indices.cuda()].contiguous(), order.values.cuda(), table_g, padding_idx=0)
            ops.sumsq(grad, nsq, sparse=sp)
            norms.append(nsq.item())
            if mode == "sparse":                 # the clip factor from the DENSE norm in both runs: the two norms differ in their last bits (another
                nsq.fill_(state["dense"][4][step - 1])      # partition of the same addends), which would move every parameter by an ulp
            ops.adam_step(p, grad, m, v, pb, [lo, n], [1e-3, 3e-4], step, gnorm_sq=nsq, max_norm=0.25, sparse=sp)
            if mode == "sparse":
                assert (table_g == 0).all()                                           # cleared row by row, no per-step fill
                assert touched[0].item() == 0 and touched.sum().item() > 30
        state[mode] = (p.cpu(), m.cpu(), v.cpu(), pb.cpu(), norms)
    for a, b in zip(state["dense"][:4], state["sparse"][:4]):
        assert torch.equal(a, b)
    for a, b in zip(state["dense"][4], state["sparse"][4]):
        assert abs(a - b) <= 1e-6 * abs(a)                                            # (same addends, another partition of the partial sums)
    never = (state["sparse"][1][lo:hi].view(rows, d) == 0).all(1)
    assert never.sum() > 100 and torch.equal(state["sparse"][0][lo:hi].view(rows, d)[never], p0[lo:hi].view(rows, d)[never])


def test_adam_in_gated_pieces_is_bit_identical_to_one_launch():
    """sam_adam_step_range: the update applied as two pieces (the row-sparse table inside the first) gives, bit for bit, the parameters / moments / bf16
    shadows of one sam_adam_step_dev launch; zero_grad clears exactly the gradient it has used; a closed gate (device word 0) leaves everything alone"""
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(8)
    rows, d, head, mid, tail = 64, 768, 4096, 5000 * 4, 7001 * 4
    lo, hi = head, head + rows * d
    n = hi + mid + tail
    split = hi + mid
    p0, m0, v0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 1e-4
    grad0 = torch.randn(n, generator=g) * 0.05
    touched = (torch.rand(rows, generator=g) > 0.5).to(torch.uint8)
    grad0[lo:hi].view(rows, d)[touched == 0] = 0
    m0[lo:hi].view(rows, d)[touched == 0] = 0
    v0[lo:hi].view(rows, d)[touched == 0] = 0
    sched = torch.tensor([1e-3, 3e-4, 1 - 0.9 ** 3, 1 - 0.999 ** 3], dtype=torch.float32).cuda()
    nsq = (grad0.double() ** 2).sum().float().reshape(1).cuda()
    seg = [split, n]
    res = {}
    for mode in ("one", "pieces", "closed"):
        p, m, v, gr = p0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda(), grad0.clone().cuda()
        pb = p0.to(torch.bfloat16).cuda()
        sp = (lo, hi, d, touched.cuda())
        if mode == "one":
            ops.adam_step_dev(p, gr, m, v, pb, seg, sched, gnorm_sq=nsq, max_norm=0.25, sparse=sp)
        else:
            gate = torch.tensor([0 if mode == "closed" else 1], dtype=torch.int32).cuda()
            ops.adam_step_range(p, gr, m, v, pb, seg, sched, 0, split, gnorm_sq=nsq, max_norm=0.25, sparse=sp, zero_grad=True, gate=gate)
            if mode == "pieces":
                assert (gr[:split] == 0).all() and torch.equal(gr[split:].cpu(), grad0[split:])         # only the piece's own gradient is cleared
            ops.adam_step_range(p, gr, m, v, pb, seg, sched, split, n, gnorm_sq=nsq, max_norm=0.25, sparse=sp, zero_grad=True, gate=gate)
        res[mode] = tuple(t.cpu() for t in (p, m, v, pb, gr))
    for a, b in zip(res["one"][:4], res["pieces"][:4]):
        assert torch.equal(a, b)
    assert (res["pieces"][4] == 0).all()
    for a, b in zip(res["closed"], (p0, m0, v0, p0.to(torch.bfloat16), grad0)):
        assert torch.equal(a, b)
    with pytest.raises(Exception):                      # a piece may not cut the row-sparse region
        ops.adam_step_range(p, gr, m, v, pb, seg, sched, 0, lo + d, sparse=(lo, hi, d, touched.cuda()))


def test_embedding_backward_without_atomics_is_deterministic_and_equals_index_add(monkeypatch):
    """sam_embedding_bwd (TextBert's word table; the data-parallel row-sparse scatter): the block of a row's first occurrence is its only writer and adds the
    duplicates in list order -- bit-identical run to run, equal to the exact sum in that order, heavy duplication (one row 300 times, spread over several
    256-position scan chunks), padding / out-of-range / negative ids skipped, touched flags set, an accumulate into a non-zero table; and equal (to fp32
    rounding) to the atomic kernel it replaces (SAM_EMBED_BWD_ATOMIC=1 is read once per process: checked through index_add here)."""
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(5)
    for T, D, rows in ((1280, 768, 30522), (777, 768, 97), (5, 256, 11), (2600, 1024, 64)):
        ids = torch.randint(0, rows, (T,), generator=g)
        ids[: T // 20] = 0                            # padding rows
        if T > 600:
            ids[torch.randperm(T, generator=g)[:300]] = 7                 # one row 300 times
            ids[3], ids[T - 1] = -1, rows + 5                             # (the reducer pads short lists with -1)
        dy = torch.randn(T, D, generator=g).to(torch.bfloat16)
        base = torch.randn(rows, D, generator=g)
        outs = []
        for _ in range(2):
            tab = base.clone().cuda()
            touched = torch.zeros(rows, dtype=torch.uint8).cuda()
            tab._sam_touched = touched
            ops.embedding_bwd(dy.cuda(), ids.cuda(), tab, padding_idx=0)
            outs.append((tab.cpu(), touched.cpu()))
        assert torch.equal(outs[0][0], outs[1][0])
        keep = (ids != 0) & (ids >= 0) & (ids < rows)
        # exact model of the kernel: fp32 sum of the row's dy in list order, then ONE fp32 add into the table
        ref = base.clone()
        acc = {}
        for t in torch.nonzero(keep).flatten().tolist():
            r = int(ids[t])
            acc[r] = dy[t].float() if r not in acc else acc[r] + dy[t].float()
        for r, a in acc.items():
            ref[r] = base[r] + a
        assert torch.equal(outs[0][0], ref), (T, D, rows, (outs[0][0] - ref).abs().max().item())
        exact = base.clone().double()
        exact.index_add_(0, ids[keep], dy[keep].double())
        assert (outs[0][0].double() - exact).abs().max().item() < 2e-4 and torch.equal(outs[0][0][0], base[0])
        flags = torch.zeros(rows, dtype=torch.uint8)
        flags[ids[keep]] = 1
        assert torch.equal(outs[0][1], flags)


def test_sorted_embedding_backward_is_deterministic_and_equals_index_add():
    """sam_embedding_bwd_sorted (data-parallel row-sparse exchange): one writer per table row, duplicates summed in list order"""
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(5)
    T, D, rows = 777, 768, 97
    ids = torch.randint(0, rows, (T,), generator=g)
    ids[:40] = 0                              # padding rows
    dy = torch.randn(T, D, generator=g).to(torch.bfloat16)
    order = torch.sort(ids, stable=True)
    base = torch.randn(rows, D, generator=g)
    outs = []
    for _ in range(2):
        tab = base.clone().cuda()
        ops.embedding_bwd_sorted(dy[order.indices].cuda().contiguous(), order.values.cuda(), tab, padding_idx=0)
        outs.append(tab.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = base.clone().double()
    keep = ids != 0
    ref.index_add_(0, ids[keep], dy[keep].double())
    assert (outs[0].double() - ref).abs().max().item() < 1e-4 and torch.equal(outs[0][0], base[0])


def test_add_dropout_kernel_mask_scale_and_backward_consistency():
    """sam_add_dropout_bf16 (object / OCR input-encoder dropout, sa_m4c.py:224,263): p = 0 is the bf16 sum; p > 0 drops ~p of the elements, scales
    the kept ones by 1 / keep (keep probability quantised to 16 bits), and the backward (same kernel on dy, b = NULL) regenerates the SAME mask"""
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1500, 768, generator=g).to(torch.bfloat16).cuda()
    b = torch.randn(1500, 768, generator=g).to(torch.bfloat16).cuda()
    ref = (a.float() + b.float())
    out0 = ops.add_dropout(a, b, 0.0)
    assert torch.equal(out0, ref.to(torch.bfloat16))
    p = 0.1
    out = ops.add_dropout(a, b, p, seed=11, offset=5)
    kept = out != 0
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - p) < 0.003, frac
    keep_q = 1.0 - round(p * 65536) / 65536.0
    assert torch.equal(out[kept], (ref / keep_q).to(torch.bfloat16)[kept])
    assert torch.equal(ops.add_dropout(a, b, p, seed=11, offset=5), out)                       # counter-based: same (seed, offset) -> same bits
    assert not torch.equal(ops.add_dropout(a, b, p, seed=11, offset=6) != 0, kept)
    ones = torch.ones_like(a)
    mask = ops.add_dropout(ones, None, p, seed=11, offset=5)                                   # the backward pass applied to dy = 1
    live = ref.to(torch.bfloat16) != 0
    assert torch.equal((mask != 0) & live, kept & live)
    assert torch.equal(mask[mask != 0].float(), torch.full_like(mask[mask != 0].float(), float(torch.tensor(1.0 / keep_q).to(torch.bfloat16))))
    # strided rows (a column block of a wider buffer)
    wide = torch.randn(64, 2304, generator=g).to(torch.bfloat16).cuda()
    v = wide[:, 768:1536]
    assert torch.equal(ops.add_dropout(v, None, 0.0), v)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_input_encoder_node_matches_fp32_reference(p_drop):
    """InputEncoderFn = dropout(LN_a(feat W_a^T + b_a) + LN_b(bbox W_b^T + b_b)), sa_m4c.py:213-224 / 252-263: the wide projection as a GEMM, the rest in
    sam_input_encoder_fwd / _bwd.  Output and all eight parameter gradients against fp32 autograd on the same bf16-rounded operands (z_a rounded to
    bf16 where the GEMM stores it); with dropout the mask is read off the output (zeros) and must be the hidden-state stream's mask for (seed, offset)"""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd import ops
    from sam_textvqa_amd.autograd import InputEncoderFn, dropout_clock
    from sam_textvqa_amd.params import prepare
    import torch.nn as nn

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.la, self.na = nn.Linear(2048, 768), M.BertLayerNorm(768, eps=1e-12)
            self.lb, self.nb = nn.Linear(4, 768), M.BertLayerNorm(768, eps=1e-12)
    torch.manual_seed(0)
    enc = Enc().cuda()
    with torch.no_grad():
        for ln in (enc.na, enc.nb):
            ln.weight.add_(0.1 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
        enc.lb.bias.add_(0.05 * torch.randn_like(enc.lb.bias))
    fp = prepare(enc)
    g = torch.Generator().manual_seed(1)
    R = 1003                                                                         # not a multiple of the rows per block
    feat = torch.randn(R, 2048, generator=g).to(torch.bfloat16).cuda()
    boxes = torch.rand(R, 5, generator=g).cuda()                                     # [x1, y1, x2, y2, area] rows as the batch carries them: 4 are read
    gout = torch.randn(R, 768, generator=g).to(torch.bfloat16).cuda()
    fp.zero_grad()
    dropout_clock.manual_seed(77)
    y = InputEncoderFn.apply(enc.la.weight, feat, boxes, enc.la, enc.na, enc.lb, enc.nb, p_drop, enc.la)
    y.backward(gout)
    seed, offset = dropout_clock.seed, dropout_clock.offset
    # fp32 reference
    wa = enc.la.weight.detach().to(torch.bfloat16).float().requires_grad_(True); ba = enc.la.bias.detach().clone().requires_grad_(True)
    wb = enc.lb.weight.detach().to(torch.bfloat16).float().requires_grad_(True); bb = enc.lb.bias.detach().clone().requires_grad_(True)
    ga, bta = enc.na.weight.detach().clone().requires_grad_(True), enc.na.bias.detach().clone().requires_grad_(True)
    gb, btb = enc.nb.weight.detach().clone().requires_grad_(True), enc.nb.bias.detach().clone().requires_grad_(True)

    def ln(x, gam, bet):
        mu = x.mean(-1, keepdim=True)
        return (x - mu) / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + 1e-12) * gam + bet
    za = feat.float() @ wa.t() + ba
    # the GEMM stores z_a in bf16: its OWN rounding is used (another summation order flips bf16 ties: one ulp of |z_a| ~ 4 is 0.03), straight-through for the gradient
    from sam_textvqa_amd.autograd import _padded_views
    wv, _, bv, _, _, _ = _padded_views(enc.la.weight, enc.la.bias)
    za_gpu = ops.gemm(feat, wv, epilogue=1, bias=bv).float()
    assert (za_gpu - za.detach()).abs().max().item() <= 2.0 ** -7 * za.detach().abs().max().item()
    za = za + (za_gpu - za.detach())
    zb = boxes[:, :4].to(torch.bfloat16).float() @ wb.t() + bb
    pre = ln(za, ga, bta) + ln(zb, gb, btb)
    if p_drop > 0:
        keep_q = 1.0 - round(p_drop * 65536) / 65536.0
        mask = (ops.add_dropout(torch.ones(R, 768, dtype=torch.bfloat16, device="cuda"), None, p_drop, seed, offset) != 0).float()
        assert abs(mask.mean().item() - (1 - p_drop)) < 0.01
        ref = pre * mask / keep_q
        assert torch.equal((y != 0).float(), mask) or ((y == 0).float() - (1 - mask)).abs().sum() <= 3          # (an exact 0.0 before the mask is possible)
    else:
        ref = pre
    assert_close_bf16(y, ref.detach(), name="encoder out")
    ref.backward(gout.float())

    def close(got, want, name, frac=2e-3):
        err = (got.float() - want).abs().max().item()
        assert err <= frac * want.abs().max().item() + 1e-6, (name, err, want.abs().max().item())
    close(enc.na.weight.grad, ga.grad, "d gamma_a"); close(enc.na.bias.grad, bta.grad, "d beta_a")
    close(enc.nb.weight.grad, gb.grad, "d gamma_b"); close(enc.nb.bias.grad, btb.grad, "d beta_b")
    close(enc.lb.bias.grad, bb.grad, "d b_b"); close(enc.lb.weight.grad, wb.grad, "d W_b")
    close(enc.la.bias.grad, ba.grad, "d b_a", 4e-3); close(enc.la.weight.grad, wa.grad, "d W_a", 4e-3)           # (through d z_a rounded to bf16)
    # a second backward pass accumulates (the Trainer zero-fills these gradients per step)
    before = enc.lb.weight.grad.clone()
    dropout_clock.manual_seed(77)
    y2 = InputEncoderFn.apply(enc.la.weight, feat, boxes, enc.la, enc.na, enc.lb, enc.nb, p_drop, enc.la)
    y2.backward(gout)
    assert torch.equal(y2, y)
    close(enc.lb.weight.grad, 2 * before, "accumulated d W_b", 1e-5)


def test_copy_blocks_and_ge_u8():
    """sam_copy_blocks: strided slices, casts both ways, accumulation, zero-fill, several blocks per launch -- against torch"""
    from sam_textvqa_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    seq = torch.randn(5, 182, 768, device="cuda", generator=g).bfloat16()
    a, b = torch.empty(5, 50, 768, device="cuda", dtype=torch.bfloat16), torch.empty(5, 12, 768, device="cuda", dtype=torch.float32)
    z = torch.full((5, 20, 768), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.copy_blocks([(seq[:, 120:170], a), (seq[:, 170:], b), (None, z)])
    assert torch.equal(a, seq[:, 120:170]) and torch.equal(b, seq[:, 170:].float()) and not z.any()
    out = torch.zeros(5, 182, 768, device="cuda", dtype=torch.bfloat16)
    ops.copy_blocks([(a, out[:, 120:170]), (b, out[:, 170:])])                      # fp32 -> bf16 into a strided destination
    assert torch.equal(out[:, 120:], seq[:, 120:]) and not out[:, :120].any()
    acc = torch.randn(1, 300, 768, device="cuda", generator=g)
    add = torch.randn(1, 300, 768, device="cuda", generator=g).bfloat16()
    want = acc + add.float()
    wide = torch.zeros(1, 300, 776, device="cuda")                                   # padded row stride, as the classifier's gradient view
    wide[:, :, :768] = acc
    ops.copy_blocks([(add, wide[:, :, :768], True)])
    assert torch.equal(wide[:, :, :768], want) and not wide[:, :, 768:].any()
    many = [torch.randn(2, 3, 8, device="cuda", generator=g) for _ in range(11)]     # more than 8 blocks: split over launches
    outs = [torch.empty(2, 3, 8, device="cuda", dtype=torch.bfloat16) for _ in many]
    ops.copy_blocks(list(zip(many, outs)))
    assert all(torch.equal(o, m.bfloat16()) for o, m in zip(outs, many))
    with pytest.raises(Exception):
        ops.copy_blocks([(seq[:, :, :6], torch.empty(5, 182, 6, device="cuda", dtype=torch.bfloat16))])      # width not a multiple of 4
    ids = torch.randint(0, 5050, (64, 12), device="cuda", generator=g)
    assert torch.equal(ops.ge_u8(ids, 5000), ids.ge(5000).view(torch.uint8).reshape(-1))
