"""GPU parity of the front-end kernels (csrc/embed.hip): feature normalise + pack, embedding sums, previous-prediction gather —
each against the torch formulation of the reference lines it replaces (sam/sa_m4c.py:217-253, 921-948; BertEmbeddings)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close_bf16

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ops():
    from sam_textvqa_amd import ops
    return ops


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("normalize", [True, False])
def test_l2norm_pack_matches_normalize_cat(normalize):
    """the OCR row of forward_ocr_encoding: FastText 300 | PHOC 604 | FRCN 2048 | 50 zeros, K padded 3002 -> 3008"""
    ops = _ops()
    m = 37
    parts = [rnd((m, 300), 1), rnd((m, 604), 2, 3.0), rnd((m, 2048), 3, 0.2)]
    parts[1][5] = 0.0                      # an all-zero row: x / max(0, eps) = 0, no NaN
    out = torch.full((m, 3008), 7.0, dtype=BF16, device="cuda")
    col = 0
    for i, p in enumerate(parts):
        ops.l2norm_pack(p.cuda(), out, col, normalize, zero_upto=3008 if i == 2 else 0)
        col += p.shape[1]
    ref = torch.cat([F.normalize(p, dim=-1) if normalize else p for p in parts] + [torch.zeros(m, 56)], dim=-1)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got[:, 2952:] == 0).all()
    assert_close_bf16(got, ref, ulps=1, name="l2norm_pack")
    assert (got[5, 300:904] == 0).all()


def test_l2norm_pack_unaligned_rows_and_bad_shapes():
    """box coordinates: 4 columns sliced out of [.., 5] fp32 rows (not 16-byte aligned) go through the scalar twin; K padded 4 -> 8"""
    ops = _ops()
    from sam_textvqa_amd._capi import SamHipError
    boxes = rnd((6, 50, 5), 9).cuda()
    x = boxes[:, :, :-1].flatten(0, 1)
    assert x.stride() == (5, 1)
    out = torch.full((300, 8), 3.0, dtype=BF16, device="cuda")
    ops.l2norm_pack(x, out, 0, normalize=False, zero_upto=8)
    assert torch.equal(out[:, :4], x.to(BF16)) and (out[:, 4:] == 0).all()
    out2 = torch.empty((300, 8), dtype=BF16, device="cuda")
    ops.l2norm_pack(x, out2, 2, normalize=True)                           # odd column offset, normalised
    assert_close_bf16(out2[:, 2:6], F.normalize(x, dim=-1), ulps=1, name="scalar normalize")
    with pytest.raises(SamHipError):
        ops.l2norm_pack(torch.zeros(4, 8, device="cuda"), torch.empty((4, 16), dtype=BF16, device="cuda"), col0=12)      # col0 + D > ldo


@pytest.mark.parametrize("with_table,with_types", [(True, False), (False, True), (True, True)])
def test_embed_sum_fwd_bwd(with_table, with_types):
    ops = _ops()
    b, s, d, rows_tab = 5, 12, 96, 40
    pos, tt = rnd((20, d), 1).cuda(), rnd((3, d), 2).cuda()
    table = rnd((rows_tab, d), 3, dtype=BF16).cuda() if with_table else None
    ids = torch.randint(0, rows_tab, (b * s,), generator=torch.Generator().manual_seed(4)).cuda() if with_table else None
    types = torch.randint(0, 3, (b * s,), generator=torch.Generator().manual_seed(5)).to(torch.uint8).cuda() if with_types else None
    e = ops.embed_sum_fwd(pos, tt, b * s, s, table=table, ids=ids, type_ids=types)
    ref = pos[:s].repeat(b, 1) + (tt[types.long()] if with_types else tt[0])
    if with_table:
        ref = (table[ids].float() + pos[:s].repeat(b, 1)) + (tt[types.long()] if with_types else tt[0])
    assert torch.equal(e, ref)
    # backward: d_pos / d_tt accumulate INTO existing gradients
    de = rnd((b * s, d), 6, dtype=BF16).cuda()
    d_pos, d_tt = torch.ones(20, d, device="cuda"), torch.ones(3, d, device="cuda")
    ops.embed_sum_bwd(de, s, d_pos, d_tt, types, n_types=3 if with_types else 1)
    ref_pos = torch.ones(20, d, device="cuda")
    ref_pos[:s] += de.float().view(b, s, d).sum(0)
    ref_tt = torch.ones(3, d, device="cuda")
    if with_types:
        ref_tt.index_add_(0, types.long(), de.float())
    else:
        ref_tt[0] += de.float().sum(0)
    torch.testing.assert_close(d_pos, ref_pos, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d_tt, ref_tt, rtol=1e-5, atol=1e-5)


def test_embed_sum_bwd_is_deterministic():
    ops = _ops()
    de = rnd((64 * 20, 768), 7, dtype=BF16).cuda()
    outs = []
    for _ in range(3):
        d_pos, d_tt = torch.zeros(512, 768, device="cuda"), torch.zeros(2, 768, device="cuda")
        ops.embed_sum_bwd(de, 20, d_pos, d_tt, None, n_types=1)
        outs.append((d_pos.clone(), d_tt.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    torch.testing.assert_close(outs[0][1][0], de.float().sum(0), rtol=1e-5, atol=1e-4)


def _gather_ref(ans, ocr, inds, n_ocr):
    b, s = inds.shape
    v = ans.shape[0]
    table = torch.cat([ans.unsqueeze(0).expand(b, -1, -1), ocr.view(b, n_ocr, -1)], dim=1)      # the reference's [B, V+n_ocr, D] table
    return torch.gather(table, 1, inds.unsqueeze(-1).expand(-1, -1, table.shape[-1])).reshape(b * s, -1)


def test_gather2_add_fwd_bwd_no_dropout():
    ops = _ops()
    b, s, d, v, n_ocr = 6, 12, 768, 300, 50
    ans, ocr = rnd((v, d), 1, dtype=BF16).cuda(), rnd((b * n_ocr, d), 2, dtype=BF16).cuda()
    emb = rnd((b * s, d), 3, dtype=BF16).cuda()
    inds = torch.randint(0, v + n_ocr, (b, s), generator=torch.Generator().manual_seed(4))
    inds[:, 0] = 1
    inds[0, 5:] = 0                                   # repeated rows (padding index) -> the atomics path
    inds = inds.cuda()
    out = ops.gather2_add_fwd(ans, ocr, inds, n_ocr, emb)
    ref = (_gather_ref(ans, ocr, inds, n_ocr).float() + emb.float()).to(BF16)
    assert torch.equal(out, ref)
    dy = rnd((b * s, d), 5, dtype=BF16).cuda()
    d_ans, d_ocr, d_emb = ops.gather2_add_bwd(dy, inds, v, n_ocr)
    assert torch.equal(d_emb, dy)
    flat = inds.reshape(-1)
    is_ocr = flat >= v
    ref_ans = torch.zeros(v, d, device="cuda").index_add_(0, flat[~is_ocr], dy.float()[~is_ocr])
    rows = (torch.arange(b, device="cuda").repeat_interleave(s) * n_ocr + flat - v)[is_ocr]
    ref_ocr = torch.zeros(b * n_ocr, d, device="cuda").index_add_(0, rows, dy.float()[is_ocr])
    torch.testing.assert_close(d_ans, ref_ans, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d_ocr, ref_ocr, rtol=1e-5, atol=1e-5)


def test_gather2_add_dropout_mask_shared_by_fwd_and_bwd():
    ops = _ops()
    b, s, d, v, n_ocr = 16, 12, 768, 64, 50
    ans, ocr = torch.zeros(v, d, dtype=BF16, device="cuda"), torch.zeros(b * n_ocr, d, dtype=BF16, device="cuda")
    inds = torch.randint(0, v + n_ocr, (b, s), generator=torch.Generator().manual_seed(1)).cuda()
    ones = torch.ones(b * s, d, dtype=BF16, device="cuda")
    p = 0.1
    out = ops.gather2_add_fwd(ans, ocr, inds, n_ocr, ones, p, seed=11, offset=3)
    keep_f = out.float() != 0
    _, _, d_emb = ops.gather2_add_bwd(ones, inds, v, n_ocr, True, p, seed=11, offset=3)
    assert torch.equal(keep_f, d_emb.float() != 0)
    rate = keep_f.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01
    kept = out.float()[keep_f]
    assert torch.allclose(kept, torch.full_like(kept, 1 / (1 - p)), rtol=1e-2)
    out2 = ops.gather2_add_fwd(ans, ocr, inds, n_ocr, ones, p, seed=11, offset=4)
    assert not torch.equal(out2, out)


def test_gather2_add_clamps_out_of_range_indices():
    ops = _ops()
    b, s, d, v, n_ocr = 2, 4, 64, 10, 5
    ans, ocr = rnd((v, d), 1, dtype=BF16).cuda(), rnd((b * n_ocr, d), 2, dtype=BF16).cuda()
    inds = torch.tensor([[-3, 0, 14, 99], [9, 10, 15, 2]]).cuda()
    out = ops.gather2_add_fwd(ans, ocr, inds, n_ocr)
    ref = _gather_ref(ans, ocr, inds.clamp(0, v + n_ocr - 1), n_ocr)
    assert torch.equal(out, ref)


def test_prev_pred_embeddings_module_matches_oracle():
    """PrevPredEmbeddings (sam/sa_m4c.py:900-948) through the fused path vs the oracle, forward and all gradients"""
    from oracle import sa_m4c_oracle as O
    from sam_textvqa_amd import modules as M
    torch.manual_seed(0)
    cfg = O.BertConfig(hidden_size=768, num_attention_heads=12, num_hidden_layers=1, hidden_dropout_prob=0.0)
    ref = O.PrevPredEmbeddings(cfg)
    mine = M.PrevPredEmbeddings(M.BertConfig(hidden_size=768, num_attention_heads=12, num_hidden_layers=1, hidden_dropout_prob=0.0))
    with torch.no_grad():
        for n, p in ref.named_parameters():
            p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("layer_norm.weight") else 0.0))
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    b, s, v, n_ocr = 4, 12, 200, 50
    ans = (torch.randn(v, 768) * 0.5).requires_grad_(True)
    ocr = (torch.randn(b, n_ocr, 768) * 0.5).to(BF16).float().requires_grad_(True)
    inds = torch.randint(0, v + n_ocr, (b, s))
    out_ref = ref(ans, ocr, inds)
    gy = torch.randn_like(out_ref)
    out_ref.backward(gy)
    ans_g = ans.detach().cuda().requires_grad_(True)
    ocr_g = ocr.detach().to(BF16).cuda().requires_grad_(True)
    mine.train()
    out = mine(ans_g, ocr_g, inds.cuda())
    # the two summands are LayerNorm outputs held in bf16 (as every activation of the path): their half-ulp roundings add, so the
    # bound is one bf16 ulp (2^-7) of the largest operand rather than of the (possibly cancelled) sum
    assert_close_bf16(out.float().cpu(), out_ref.detach(), frac=2.0 ** -7, ulps=2, name="prev_pred fwd")
    out.backward(gy.to(BF16).cuda())
    assert_close_bf16(ans_g.grad.float().cpu(), ans.grad, frac=2.0 ** -7, ulps=4, name="d ans_emb")
    assert_close_bf16(ocr_g.grad.float().cpu(), ocr.grad, frac=2.0 ** -7, ulps=4, name="d ocr_emb")
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert_close_bf16(q.grad.float().cpu(), p.grad, frac=2.0 ** -7, ulps=4, name="d " + n)
