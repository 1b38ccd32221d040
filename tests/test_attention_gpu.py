"""GPU parity: mask bit packers (bit-exact) and fused attention fwd/bwd vs the fp32 oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import sa_m4c_oracle as O
from oracle import spatial_graph as SG
from tests.golden import common as C
from tests.util import assert_close_bf16, unpack_bits

pytestmark = pytest.mark.gpu


def _ops():
    from sam_textvqa_amd import ops
    return ops


def make_problem(B, T, n_obj, n_ocr, n_dec, H=12, ctx=3, seed=0, full_valid=False):
    rng = np.random.RandomState(seed)
    n_oo = n_obj + n_ocr
    n_txt_valid = [int(rng.randint(1, T + 1)) if T else 0 for _ in range(B)]
    n_obj_valid = [n_obj if full_valid else int(rng.randint(1, n_obj + 1)) for _ in range(B)]
    n_ocr_valid = [int(rng.randint(0, n_ocr + 1)) for _ in range(B)]
    if B > 1:
        n_ocr_valid[1] = 0          # a sample whose OCR tokens are all padding
    kv = np.concatenate([C.pad_mask(n_txt_valid, T), C.pad_mask(n_obj_valid, n_obj), C.pad_mask(n_ocr_valid, n_ocr)], axis=1)
    adj = []
    for b in range(B):
        boxes = np.concatenate([C.det_boxes("p%d.%d.obj" % (seed, b), n_obj_valid[b], n_obj, 0.21),
                                C.det_boxes("p%d.%d.ocr" % (seed, b), n_ocr_valid[b], n_ocr, 0.08)], axis=0)
        with np.errstate(all="ignore"):
            adj.append(SG.compose(SG.relation_codes(boxes, 0.5), ctx))
    adj = torch.from_numpy(np.stack(adj))[..., :H].contiguous()
    return dict(B=B, T=T, n_oo=n_oo, n_dec=n_dec, N=T + n_oo + n_dec, H=H, key_valid=torch.from_numpy(kv), adj=adj)


def oracle_attention(qkv, allow, B, H, scale, keep=None, inv_keep=1.0):
    """fp32 restatement of sa_m4c.py:563-598 on a boolean allow mask; qkv [B*N, 3*H*64] float (requires_grad ok)."""
    rows, three_d = qkv.shape
    N, Dm = rows // B, three_d // 3
    x = qkv.view(B, N, 3, H, Dm // H).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * scale
    s = s.masked_fill(~allow, -10000.0)          # same additive value as the reference
    alive = allow.any(-1, keepdim=True).float()
    p = torch.softmax(s, dim=-1) * alive
    lse = torch.logsumexp(s.masked_fill(~allow, float("-inf")), dim=-1)
    if keep is not None:
        p = p * keep.float() * inv_keep
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(rows, Dm)
    return ctx, lse


@pytest.mark.parametrize("shape", [(3, 20, 100, 50, 12), (2, 4, 10, 6, 3), (2, 20, 200, 100, 30)])
@pytest.mark.parametrize("quadrants", [(1, 2), (4, 7, 8, 9), ()])
def test_mask_bits_bit_exact(shape, quadrants):
    ops = _ops()
    pr = make_problem(*shape, seed=1)
    dev = "cuda"
    kv = pr["key_valid"].to(torch.uint8).to(dev)
    base = ops.mask_bits_prefix_lm(kv, pr["n_dec"])
    ref_plain = O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], None, (), 1)
    assert torch.equal(unpack_bits(base, pr["N"]), ref_plain)
    assert (unpack_bits(base, base.shape[-1] * 32)[..., pr["N"]:] == 0).all()      # keys >= N read 0
    sp = ops.mask_bits_spatial(base, pr["adj"].to(dev), pr["T"], pr["H"], quadrants)
    ref = O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], pr["adj"], quadrants, pr["H"])
    assert torch.equal(unpack_bits(sp, pr["N"]), ref)
    # additive-mask entry point (module-level drop-in API) gives the same base bits
    ext = O.MMT.extended_attention_mask(pr["key_valid"][:, :pr["T"]], pr["key_valid"][:, pr["T"]:pr["T"] + shape[2]],
                                        pr["key_valid"][:, pr["T"] + shape[2]:], pr["n_dec"]).float().contiguous()
    assert torch.equal(ops.mask_bits_from_additive(ext.to(dev)).cpu(), base.cpu())


def test_mask_bits_from_int8_bhnn_format():
    """the [B,H,N,N] int8 relation-type layout named by the north star gives the same bits as the dataset layout"""
    ops = _ops()
    pr = make_problem(2, 20, 100, 50, 12, seed=4)
    base = ops.mask_bits_prefix_lm(pr["key_valid"].to(torch.uint8).cuda(), pr["n_dec"])
    want = ops.mask_bits_spatial(base, pr["adj"].cuda(), pr["T"], pr["H"], (1, 2))
    allow = O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], pr["adj"], (1, 2), pr["H"])     # bool [B,H,N,N]
    got = ops.mask_bits_from_int8_bhnn(allow.to(torch.int8).cuda())
    assert torch.equal(got.cpu(), want.cpu())
    sp_only = O.allow_mask(torch.ones_like(pr["key_valid"]), pr["T"], pr["n_oo"], 0, pr["adj"], (1, 2), pr["H"])
    pad = torch.zeros(2, pr["H"], pr["N"], pr["N"], dtype=torch.int8)
    pad[:, :, : pr["T"] + pr["n_oo"], : pr["T"] + pr["n_oo"]] = sp_only.to(torch.int8)
    pad[:, :, pr["T"] + pr["n_oo"]:, :] = 1
    got2 = ops.mask_bits_from_int8_bhnn(pad.cuda(), base)
    assert torch.equal(got2.cpu(), want.cpu())


def test_spatial_graph_kernel_matches_reference_goldens():
    from tests import oracle_cases as OC
    ops = _ops()
    g = OC.load("spatial_graph")
    for nm in ("known6", "grid", "rnd60", "cross"):
        bx = g[nm + ".boxes"]
        boxes = torch.from_numpy(bx)[None].cuda()
        # Pairs whose centre direction lies EXACTLY on a sector boundary (dx = 0, dy = 0 or |dx| = |dy|; only the synthetic
        # `grid` case has them) are classified by the last ulp of libm's asin/acos in the reference itself; there either adjacent
        # sector is accepted.  Everywhere else the kernel must be bit-identical to the reference goldens.
        cx, cy = 0.5 * (bx[:, 0] + bx[:, 2]), 0.5 * (bx[:, 1] + bx[:, 3])
        dx, dy = np.abs(cx[:, None] - cx[None, :]), np.abs(cy[:, None] - cy[None, :])
        on_boundary = (dx < 1e-12) | (dy < 1e-12) | (np.abs(dx - dy) < 1e-12)
        for ctx in (1, 3, 5, 7, 9):
            got = ops.spatial_relation_tensor(boxes, ctx)[0].cpu().numpy()
            want = g["%s.ctx%d" % (nm, ctx)]
            diff = (got != want).any(-1)
            assert not (diff & ~on_boundary).any(), "%s ctx%d: mismatch off the sector boundaries" % (nm, ctx)
            if nm != "grid":
                np.testing.assert_array_equal(got, want, err_msg="%s ctx%d" % (nm, ctx))
            else:       # boundary pairs: same number of channels set, and they overlap the reference's (adjacent sector)
                assert (got.sum(-1) == want.sum(-1)).all()
                assert ((got & want).sum(-1)[diff] >= (ctx - 1)).all() if ctx > 1 else True
    # batched, at the c3 size, against the torch-vectorised builder
    from sam_textvqa_amd.spatial_graph import relation_tensor
    from sam_textvqa_amd.synthetic import make_batch
    bd = make_batch(4, device="cuda", seed=9)
    boxes = torch.cat([bd["pad_obj_bboxes"][..., :4], bd["pad_ocr_bboxes"][..., :4]], 1).double()
    assert torch.equal(ops.spatial_relation_tensor(boxes, 5), relation_tensor(boxes, 5))


def test_mask_bits_rejects_bad_quadrant():
    ops = _ops()
    pr = make_problem(1, 4, 10, 6, 3)
    base = ops.mask_bits_prefix_lm(pr["key_valid"].to(torch.uint8).cuda(), pr["n_dec"])
    with pytest.raises(ValueError):
        ops.mask_bits_spatial(base, pr["adj"].cuda(), pr["T"], 12, (3,))


@pytest.mark.parametrize("shape,spatial", [((3, 20, 100, 50, 12), True), ((3, 20, 100, 50, 12), False), ((4, 20, 0, 0, 0), False),
                                           ((2, 20, 200, 100, 30), True), ((2, 5, 30, 20, 7), True)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_fwd_bwd(shape, spatial, p_drop):
    ops = _ops()
    dev = "cuda"
    B, T, n_obj, n_ocr, n_dec = shape
    H, hd = 12, 64
    if n_obj + n_ocr == 0:      # TextBert-style: key padding only
        rng = np.random.RandomState(3)
        kvm = torch.from_numpy(C.pad_mask([int(rng.randint(1, T + 1)) for _ in range(B)], T))
        pr = dict(B=B, T=T, n_oo=0, n_dec=0, N=T, H=H, key_valid=kvm, adj=None)
    else:
        pr = make_problem(*shape, seed=2)
    N = pr["N"]
    base = ops.mask_bits_prefix_lm(pr["key_valid"].to(torch.uint8).to(dev), pr["n_dec"])
    if spatial:
        allow_bits = ops.mask_bits_spatial(base, pr["adj"].to(dev), pr["T"], H, (1, 2))
        allow = O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], pr["adj"], (1, 2), H)
    else:
        allow_bits = base
        allow = O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], None, (), H)
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * N, 3 * H * hd, generator=g) * 1.5).to(torch.bfloat16)
    dout = torch.randn(B * N, H * hd, generator=g).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(hd)

    out, lse2, keep_bits = ops.attn_fwd(qkv.to(dev), allow_bits, B, H, scale, p_drop, seed=1234, offset=7)
    keep, inv_keep = None, 1.0
    if p_drop > 0:
        keep = unpack_bits(keep_bits, N)
        frac = keep[allow].float().mean().item()
        assert abs(frac - (1 - p_drop)) < 0.01, frac
        inv_keep = 1.0 / (1.0 - round(p_drop * 65536) / 65536.0)
    qkv_ref = qkv.float().requires_grad_(True)
    ref_out, ref_lse = oracle_attention(qkv_ref, allow, B, H, scale, keep, inv_keep)
    assert_close_bf16(out, ref_out, name="attn out")
    alive = allow.any(-1)
    got_lse = lse2.cpu() * math.log(2.0)
    assert torch.isinf(got_lse[~alive]).all() and (got_lse[~alive] > 0).all()
    assert torch.allclose(got_lse[alive], ref_lse[alive], atol=2e-3, rtol=1e-4)
    # fully masked rows (text rows of spatial layers) are EXACT zeros, as in the reference
    dead_rows = (~alive).permute(0, 2, 1).reshape(B * N, H)
    assert (out.cpu().float().view(B * N, H, hd)[dead_rows] == 0).all()

    (ref_out * dout.float()).sum().backward()
    dqkv = ops.attn_bwd(dout.to(dev), qkv.to(dev), lse2, allow_bits, keep_bits, B, H, scale, p_drop)
    assert_close_bf16(dqkv, qkv_ref.grad, name="attn dqkv")


@pytest.mark.parametrize("T", [1, 17, 193, 257, 384])
def test_attention_sequence_length_edges(T):
    """lengths around the template boundaries: one token, one key past a 16-key tile, the first lengths that need 16 and 24 key tiles (two-pass dQ,
    8 waves per block), and the 384-key limit itself; key-padding mask with a fully padded sample, dropout on"""
    ops = _ops()
    B, H, hd = 3, 12, 64
    valid = [T, max(1, T // 2), 1]
    kvm = torch.from_numpy(C.pad_mask(valid, T))
    base = ops.mask_bits_prefix_lm(kvm.to(torch.uint8).cuda(), 0)
    allow = O.allow_mask(kvm, T, 0, 0, None, (), H)
    g = torch.Generator().manual_seed(9)
    qkv = (torch.randn(B * T, 3 * H * hd, generator=g) * 1.5).to(torch.bfloat16)
    dout = torch.randn(B * T, H * hd, generator=g).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(hd)
    out, lse2, keep_bits = ops.attn_fwd(qkv.cuda(), base, B, H, scale, 0.1, seed=77, offset=3)
    keep = unpack_bits(keep_bits, T)
    inv_keep = 1.0 / (1.0 - round(0.1 * 65536) / 65536.0)
    qkv_ref = qkv.float().requires_grad_(True)
    ref_out, _ = oracle_attention(qkv_ref, allow, B, H, scale, keep, inv_keep)
    assert_close_bf16(out, ref_out, name="attn out T=%d" % T)
    (ref_out * dout.float()).sum().backward()
    dqkv = ops.attn_bwd(dout.cuda(), qkv.cuda(), lse2, base, keep_bits, B, H, scale, 0.1)
    assert_close_bf16(dqkv, qkv_ref.grad, name="attn dqkv T=%d" % T)


def test_attention_error_paths():
    ops = _ops()
    from sam_textvqa_amd._capi import SamHipError
    qkv = torch.zeros(8 * 4, 3 * 12 * 32, dtype=torch.bfloat16, device="cuda")      # head_dim 32: unsupported
    allow = torch.zeros(4, 1, 8, 1, dtype=torch.int32, device="cuda")
    with pytest.raises(SamHipError):
        ops.attn_fwd(qkv, allow, 4, 12, 0.1)
    with pytest.raises(SamHipError):
        ops.attn_fwd(qkv.cpu(), allow, 4, 12, 0.1)
    with pytest.raises(SamHipError):                     # 385 keys: past the single-pass limit of the fused kernel
        ops.mask_bits_prefix_lm(torch.ones(2, 385, dtype=torch.uint8, device="cuda"), 0)


def test_pack_masks_one_launch_equals_cat_and_cast():
    """sam_pack_masks_u8: the batch's question / object / OCR padding masks (int64 as the reference's collate emits them, or any other dtype) -> the three
    uint8 forms the kernels read"""
    from sam_textvqa_amd import ops
    g = torch.Generator().manual_seed(0)
    q = (torch.rand(7, 20, generator=g) > 0.3).long().cuda()
    o = (torch.rand(7, 100, generator=g) > 0.1).long().cuda()
    c = (torch.rand(7, 50, generator=g) > 0.5).long().cuda()
    c[3] = 0                                                             # a sample without OCR tokens
    kv, q8, c8 = ops.pack_masks(q, o, c)
    assert kv.dtype == torch.uint8 and torch.equal(kv, torch.cat([q, o, c], 1).to(torch.uint8))
    assert torch.equal(q8, q.to(torch.uint8)) and torch.equal(c8, c.to(torch.uint8))
    kv2, _, _ = ops.pack_masks(q.float() * 3.0, o.bool(), c.int())      # other dtypes, non-0/1 values: non-zero = valid
    assert torch.equal(kv2, kv)
