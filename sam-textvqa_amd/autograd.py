"""torch.autograd.Function wrappers chaining the C-ABI kernels.

Parameter gradients are NOT returned to autograd: wgrad GEMMs and the fused reductions accumulate straight into the
flat gradient buffer (params.py).  Each Function receives one parameter tensor as an `anchor` input only so that its
output requires grad even when the activation input does not."""
import math
import os

import torch
from torch.autograd import Function

from . import _capi as capi
from . import ops, parallel, torchops
from .params import FlatParams, flat_of

BF16 = torch.bfloat16


class DropoutClock:
    """Philox (seed, offset) source: every dropout site of every step gets a fresh offset; forward and backward of
    one site share it.  The Trainer folds the rank into the seed (seed ^ rank << 32) so data-parallel replicas draw different masks."""

    def __init__(self):
        self.seed = 0x5A4D3443  # "SAM4C"
        self.offset = 0

    def manual_seed(self, seed):
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, 0

    def next(self):
        self.offset += 1
        return self.seed, self.offset


dropout_clock = DropoutClock()


def _w(p):
    """bf16 shadow of a prepared parameter"""
    s = getattr(p, "_sam_bf16", None)
    if s is None:
        flat_of(p)
    return s


def _fused_qkv(att):
    """(wqkv bf16 [3D,D], bqkv f32 [3D], dwqkv f32 [3D,D], dbqkv f32 [3D]) as single views over q|k|v"""
    cache = getattr(att, "_sam_qkv", None)
    if cache is None:
        ws = [att.query.weight, att.key.weight, att.value.weight]
        bs = [att.query.bias, att.key.bias, att.value.bias]
        views = (FlatParams.adjacent(ws, "_sam_bf16"), FlatParams.adjacent(bs, None), FlatParams.adjacent(ws, "grad"), FlatParams.adjacent(bs, "grad"))
        if any(v is None for v in views):
            raise RuntimeError("query/key/value parameters are not adjacent in flat storage; prepare() the enclosing module")
        cache = att._sam_qkv = views
    return cache


# ------------------------------------------------------------------------------------------------ linear
def _pad8(n):
    return (n + 7) // 8 * 8


def _padded_views(weight, bias):
    """GEMM-shaped views over a prepared nn.Linear's storage: rows padded to a multiple of 8 (zero rows live behind the
    parameter in flat storage), row stride already a multiple of 8.  -> (w bf16, dW f32, b f32 | None, db f32 | None, n_pad, k_pad)"""
    w, g = _w(weight), weight.grad
    n, k_pad = weight.shape[0], w.stride(0)
    n_pad = _pad8(n)
    wv = torch.as_strided(w, (n_pad, k_pad), (k_pad, 1))
    gv = torch.as_strided(g, (n_pad, k_pad), (k_pad, 1))
    bv = dbv = None
    if bias is not None:
        bv = torch.as_strided(bias.data, (n_pad,), (1,))
        dbv = torch.as_strided(bias.grad, (n_pad,), (1,))
    return wv, gv, bv, dbv, n_pad, k_pad


class LinearFn(Function):
    """y = x W^T + b  (nn.Linear sites outside the encoder layers: input projections, classifier, pointer q/k).
    Any in_features / out_features: both are zero-padded to multiples of 8 (in storage for the weights, on the fly for x / dy)."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_f32, side=None):
        ctx.weight, ctx.bias, ctx.side = weight, bias, side
        wv, _, bv, _, n_pad, k_pad = _padded_views(weight, bias)
        n, k = weight.shape
        x2 = x.reshape(-1, x.shape[-1])
        pre_padded = x2.shape[1] == k_pad           # caller already laid the operand out K-padded with zero tail columns (ops.l2norm_pack)
        if x2.shape[1] != k and not pre_padded:
            raise capi.SamHipError("LinearFn: input width %d != in_features %d" % (x2.shape[1], k))
        if (k_pad != k and not pre_padded) or x2.stride(1) != 1 or x2.stride(0) % 8 or x2.dtype != BF16 or x2.data_ptr() % 16:
            xp = torch.zeros((x2.shape[0], k_pad), dtype=BF16, device=x.device)   # zero-padded K (e.g. 3002 -> 3008, 4 -> 8)
            xp[:, :k] = x2
            x2 = xp
        y = ops.gemm(x2, wv, epilogue=capi.EPI_BIAS, bias=bv, out_dtype=torch.float32 if out_f32 else BF16)
        ctx.save_for_backward(x2)
        ctx.in_shape, ctx.in_dtype = x.shape, x.dtype
        return y[:, :n].view(*x.shape[:-1], n) if n_pad == n else y[:, :n].reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        wv, gv, _, dbv, n_pad, k_pad = _padded_views(weight, bias)
        n, k = weight.shape
        if ctx.side is not None and "dy_bf16" in ctx.side:
            dy = ctx.side.pop("dy_bf16")                         # the loss handed its bf16 gradient over directly (see BceLossFn.backward)
        dy2 = dy.reshape(-1, n)
        if n_pad == n and dy2.dtype != BF16 and dy2.is_contiguous():
            dy2 = dy2.to(BF16)                                   # fp32 scores (classifier): one cast, no padding needed
        elif n_pad != n or dy2.dtype != BF16 or dy2.stride(1) != 1 or dy2.stride(0) % 8 or dy2.data_ptr() % 16:
            dp = torch.zeros((dy2.shape[0], n_pad), dtype=BF16, device=dy.device)
            dp[:, :n] = dy2
            dy2 = dp
        if not DeferredWgrads.add_extra((dy2, x2, gv, dbv, True)):
            ops.gemm(dy2, x2, a_kcontig=False, b_kcontig=False, out=gv, accumulate=True, split_k=-1, bias_grad=dbv)   # dW += dy^T x ; db += colsum(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy2, wv, b_kcontig=False)[:, :ctx.in_shape[-1]]                    # dx = dy W
            dx = dx.reshape(ctx.in_shape).to(ctx.in_dtype)
        return dx, None, None, None, None


def linear(x, lin, out_f32=False):
    if out_f32 and torch.is_grad_enabled():
        # fp32 scores (the classifier): autograd insists on an fp32 gradient for them, which the loss would have to produce from its bf16 one and
        # the backward GEMMs cast straight back (two passes over [B*12, V]).  The output carries a side channel the loss may use instead.
        side = {}
        y = LinearFn.apply(x, lin.weight, lin.bias, out_f32, side)
        y._sam_grad_side = side
        return y
    return LinearFn.apply(x, lin.weight, lin.bias, out_f32)


# ------------------------------------------------------------------------------------------------ layernorm
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype not in (BF16, torch.float32) or x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, weight, bias, eps)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.weight, ctx.bias, ctx.in_shape, ctx.in_dtype = weight, bias, x.shape, x.dtype
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16 or not dy2.is_contiguous():
            dy2 = dy2.to(BF16).contiguous()
        dx, _ = ops.layernorm_bwd(dy2, x2, mean, rstd, ctx.weight, ctx.weight.grad, ctx.bias.grad)
        return (dx.view(ctx.in_shape).to(ctx.in_dtype) if ctx.needs_input_grad[0] else None), None, None, None


def layer_norm(x, ln):
    return LayerNormFn.apply(x, ln.weight, ln.bias, ln.variance_epsilon)


class EmbeddingFn(Function):
    """table[idx] with the gradient scattered straight into the flat gradient buffer (word embeddings: 30522 x 768 table,
    B*20 mostly-distinct rows per step; torch's dense embedding backward costs ~70 us per table)"""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx):
        ctx.weight, ctx.padding_idx = weight, padding_idx
        ctx.save_for_backward(idx)
        return torch.nn.functional.embedding(idx, _w(weight))          # bf16 rows

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16 or not dy2.is_contiguous():
            dy2 = dy2.to(BF16).contiguous()
        ops.embedding_bwd(dy2, idx.reshape(-1).contiguous(), ctx.weight.grad, ctx.padding_idx)
        return None, None, None


def embedding(idx, emb):
    return EmbeddingFn.apply(idx, emb.weight, -1 if emb.padding_idx is None else emb.padding_idx)


class EmbedLayerNormFn(Function):
    """LN(table[ids] + pos[r % seq] + tt[type_ids[r]]) -> bf16 [rows, D]: BertEmbeddings.forward up to the dropout (pytorch-transformers;
    TextBert, sam/sa_m4c.py:377) and the position / token-type half of PrevPredEmbeddings.forward (sam/sa_m4c.py:932-945; table = None).
    Two launches forward, three or four backward; the four parameter gradients go straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, anchor, ids, table_w, pos_w, tt_w, type_ids, rows, seq, ln, padding_idx):
        ids = None if ids is None else ids.reshape(-1).contiguous()
        e = ops.embed_sum_fwd(pos_w.data, tt_w.data, rows, seq, table=None if table_w is None else _w(table_w), ids=ids, type_ids=type_ids)
        y, mean, rstd = ops.layernorm_fwd(e, ln.weight, ln.bias, ln.variance_epsilon)
        ctx.save_for_backward(e, mean, rstd, ids, type_ids)
        ctx.params, ctx.seq, ctx.padding_idx = (table_w, pos_w, tt_w, ln), seq, padding_idx
        return y

    @staticmethod
    def backward(ctx, dy):
        e, mean, rstd, ids, type_ids = ctx.saved_tensors
        table_w, pos_w, tt_w, ln = ctx.params
        DeferredWgrads.flush()                                  # the encoder layers above this embedding block have all run their backward
        if dy.dtype != BF16 or not dy.is_contiguous():
            dy = dy.to(BF16).contiguous()
        d_e, _ = ops.layernorm_bwd(dy, e, mean, rstd, ln.weight, ln.weight.grad, ln.bias.grad)
        ops.embed_sum_bwd(d_e, ctx.seq, pos_w.grad, tt_w.grad, type_ids, n_types=1 if type_ids is None else min(4, tt_w.shape[0]))
        rid = getattr(ln, "_sam_region_id", None)
        if rid is not None and parallel.active_reducer is not None:
            parallel.active_reducer.mark_done(rid)            # position / token-type / LayerNorm gradients of this embedding block are final
        if table_w is not None:
            red = parallel.active_reducer
            if red is not None and getattr(table_w, "_sam_sparse_reduce", False):
                red.sparse_rows(table_w.grad, ids, d_e, ctx.padding_idx)      # data parallel: every rank's rows, table left out of the dense all-reduce
            else:
                ops.embedding_bwd(d_e, ids, table_w.grad, ctx.padding_idx)
        return (None,) * 10


class PrevPredFn(Function):
    """PrevPredEmbeddings.forward (sam/sa_m4c.py:900-948) as ONE autograd node:
        LN_ans(ans_emb)[ind] or LN_ocr(ocr_emb)[b, ind - V]  +  dropout(LN_emb(pos[s] + type[ind >= V]))
    Seven launches forward; backward writes every parameter gradient itself (ans-embedding rows go straight into the classifier weight
    gradient: ans_emb IS the classifier weight, sa_m4c.py:273-274) and returns only d ocr_emb.  Being a single node matters for data
    parallelism: when the gradient of `ocr_emb` arrives upstream, every gradient this module contributes is final (parallel.GradBarrierFn)."""

    @staticmethod
    def forward(ctx, anchor, ans_emb, ocr_emb, prev_inds, mod, p_drop):
        b, s = prev_inds.shape
        n_ocr, d = ocr_emb.shape[1], ocr_emb.shape[2]
        ans_x = ans_emb.data if ans_emb.is_contiguous() or ans_emb.stride(1) == 1 else ans_emb.contiguous()
        ans, m_a, r_a = ops.layernorm_fwd(ans_x, mod.ans_layer_norm.weight, mod.ans_layer_norm.bias, mod.ans_layer_norm.variance_epsilon)
        ocr_x = ocr_emb.reshape(b * n_ocr, d)
        if ocr_x.dtype not in (BF16, torch.float32) or ocr_x.stride(1) != 1 or ocr_x.stride(0) % 4:
            ocr_x = ocr_x.contiguous()
        ocr, m_o, r_o = ops.layernorm_fwd(ocr_x, mod.ocr_layer_norm.weight, mod.ocr_layer_norm.bias, mod.ocr_layer_norm.variance_epsilon)
        inds = prev_inds.contiguous()
        is_ocr = ops.ge_u8(inds, ans_emb.shape[0])                               # token type 1 for copied OCR tokens, sa_m4c.py:936
        e = ops.embed_sum_fwd(mod.position_embeddings.weight.data, mod.token_type_embeddings.weight.data, b * s, s, type_ids=is_ocr)
        emb, m_e, r_e = ops.layernorm_fwd(e, mod.emb_layer_norm.weight, mod.emb_layer_norm.bias, mod.emb_layer_norm.variance_epsilon)
        ctx.seed = dropout_clock.next()
        out = ops.gather2_add_fwd(ans, ocr, inds, n_ocr, emb, p_drop, *ctx.seed)
        ctx.save_for_backward(ans_x, m_a, r_a, ocr_x, m_o, r_o, e, m_e, r_e, inds, is_ocr)
        ctx.mod, ctx.ans_param, ctx.p_drop, ctx.shapes = mod, ans_emb, p_drop, (ocr_emb.shape, ocr_emb.dtype, n_ocr)
        return out.view(b, s, d)

    @staticmethod
    def backward(ctx, dy):
        ans_x, m_a, r_a, ocr_x, m_o, r_o, e, m_e, r_e, inds, is_ocr = ctx.saved_tensors
        mod, ans_param = ctx.mod, ctx.ans_param
        ocr_shape, ocr_dtype, n_ocr = ctx.shapes
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16 or not dy2.is_contiguous():
            dy2 = dy2.to(BF16).contiguous()
        d_ans, d_ocr, d_emb = ops.gather2_add_bwd(dy2, inds, ans_x.shape[0], n_ocr, True, ctx.p_drop, *ctx.seed)
        ln = mod.emb_layer_norm
        d_e, _ = ops.layernorm_bwd(d_emb, e, m_e, r_e, ln.weight, ln.weight.grad, ln.bias.grad)
        tt = mod.token_type_embeddings.weight
        ops.embed_sum_bwd(d_e, inds.shape[1], mod.position_embeddings.weight.grad, tt.grad, is_ocr, n_types=min(4, tt.shape[0]))
        d_ans16 = torch.empty(d_ans.shape, dtype=BF16, device=d_ans.device)
        d_ocr16 = torch.empty(d_ocr.shape, dtype=BF16, device=d_ocr.device)
        ops.copy_blocks([(d_ans.unsqueeze(0), d_ans16.unsqueeze(0)), (d_ocr.unsqueeze(0), d_ocr16.unsqueeze(0))])      # both fp32 -> bf16 casts: one launch
        ln = mod.ans_layer_norm
        dx_ans, _ = ops.layernorm_bwd(d_ans16, ans_x, m_a, r_a, ln.weight, ln.weight.grad, ln.bias.grad)
        ln = mod.ocr_layer_norm
        dx_ocr, _ = ops.layernorm_bwd(d_ocr16, ocr_x, m_o, r_o, ln.weight, ln.weight.grad, ln.bias.grad)
        g_ans = None
        if getattr(ans_param, "_sam_flat", None) is not None and ans_param.grad is not None:
            # prepared parameter (the classifier weight): accumulate here, not through autograd -- bf16 rows added into the fp32 gradient in one launch
            g = ans_param.grad
            if g.dim() == 2 and g.stride(1) == 1 and g.stride(0) % 4 == 0 and g.shape[1] % 4 == 0:
                ops.copy_blocks([(dx_ans.unsqueeze(0), g.unsqueeze(0), True)])
            else:
                g.add_(dx_ans.float())
        elif ctx.needs_input_grad[1]:
            g_ans = dx_ans.to(ans_param.dtype)
        g_ocr = dx_ocr.view(ocr_shape).to(ocr_dtype) if ctx.needs_input_grad[2] else None
        return None, g_ans, g_ocr, None, None, None


class SeqRowsFn(Function):
    """(OCR rows, decoder rows) of the MMT output [B, N, D] as two contiguous tensors -- what the classifier and the pointer network consume
    (sa_m4c.py:270-278).  As plain slices autograd copies the decoder rows once per consumer and builds the gradient of `seq` from two zero-filled
    [B, N, D] tensors and an add; here: two copies forward, one zero-fill and two copies backward."""

    @staticmethod
    def forward(ctx, seq, ocr0, n_ocr, n_dec):
        ctx.cfg = (seq.shape, seq.dtype, ocr0, n_ocr, n_dec)
        a, b = seq[:, ocr0: ocr0 + n_ocr], seq[:, seq.shape[1] - n_dec:]
        if _glue_ok(seq):
            oa, ob = torch.empty(a.shape, dtype=seq.dtype, device=seq.device), torch.empty(b.shape, dtype=seq.dtype, device=seq.device)
            ops.copy_blocks([(a, oa), (b, ob)])                 # one launch
            return oa, ob
        return a.contiguous(), b.contiguous()

    @staticmethod
    def backward(ctx, d_ocr, d_dec):
        shape, dtype, ocr0, n_ocr, n_dec = ctx.cfg
        ref = d_ocr if d_ocr is not None else d_dec
        if d_ocr is not None and d_dec is not None and ocr0 + n_ocr + n_dec == shape[1] and _glue_ok(d_ocr) and _glue_ok(d_dec) and dtype in (BF16, torch.float32):
            d = torch.empty(shape, dtype=dtype, device=ref.device)
            blocks = [(d_ocr, d[:, ocr0: ocr0 + n_ocr]), (d_dec, d[:, shape[1] - n_dec:])]
            if ocr0:
                blocks.append((None, d[:, :ocr0]))
            ops.copy_blocks(blocks)                             # zero rows + both slices: one launch
            return d, None, None, None
        d = torch.zeros(shape, dtype=dtype, device=ref.device)
        if d_ocr is not None:
            d[:, ocr0: ocr0 + n_ocr] = d_ocr
        if d_dec is not None:
            d[:, shape[1] - n_dec:] = d_dec
        return d, None, None, None


def _glue_ok(t):
    """can sam_copy_blocks address this [B, rows, cols] tensor?"""
    return (t is not None and t.is_cuda and t.dim() == 3 and t.dtype in (BF16, torch.float32) and t.stride(2) == 1 and t.shape[2] % 4 == 0 and
            t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0)


class CatRowsFn(Function):
    """torch.cat of token groups along dim 1 into the [B, N, D] bf16 sequence the MMT consumes (sam/sa_m4c.py:814-818): one launch forward; the backward
    hands every group its contiguous gradient slice from ONE launch (autograd's cat backward returns strided views, which every consumer then copies)"""

    @staticmethod
    def forward(ctx, *groups):
        b, d = groups[0].shape[0], groups[0].shape[2]
        ctx.rows, ctx.dtypes = [g.shape[1] for g in groups], [g.dtype for g in groups]
        out = torch.empty((b, sum(ctx.rows), d), dtype=BF16, device=groups[0].device)
        blocks, r0 = [], 0
        for g in groups:
            blocks.append((g, out[:, r0: r0 + g.shape[1]]))
            r0 += g.shape[1]
        ops.copy_blocks(blocks)
        return out

    @staticmethod
    def backward(ctx, dx):
        if not _glue_ok(dx):
            dx = dx.contiguous()
        outs, blocks, r0 = [], [], 0
        for n, dt in zip(ctx.rows, ctx.dtypes):
            o = torch.empty((dx.shape[0], n, dx.shape[2]), dtype=dt, device=dx.device)
            blocks.append((dx[:, r0: r0 + n], o))
            outs.append(o)
            r0 += n
        ops.copy_blocks(blocks)
        return tuple(outs)


def cat_rows(groups):
    """the [B, N, D] bf16 concatenation of the token groups; CatRowsFn when every group is addressable by the block-copy kernel, torch.cat otherwise"""
    if all(_glue_ok(g) for g in groups) and len(groups) <= 8 and len({(g.shape[0], g.shape[2]) for g in groups}) == 1:
        return CatRowsFn.apply(*groups)
    return torch.cat([g.to(BF16) for g in groups], dim=1)


class DropoutFn(Function):
    """element-wise dropout on bf16 rows with the library's counter-based stream (ops.add_dropout): forward and backward regenerate the same
    mask from (seed, offset), nothing is stored.  Replaces F.dropout (torch's own Philox stream + a saved mask) on the path."""

    @staticmethod
    def forward(ctx, x, p_drop):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16 or x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.to(BF16).contiguous()
        ctx.seed, ctx.p_drop = dropout_clock.next(), p_drop
        return ops.add_dropout(x2, None, p_drop, *ctx.seed).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16 or dy2.stride(1) != 1 or dy2.stride(0) % 8 or dy2.data_ptr() % 16:
            dy2 = dy2.to(BF16).contiguous()
        return ops.add_dropout(dy2, None, ctx.p_drop, *ctx.seed).view(dy.shape), None


def dropout(x, p_drop, training):
    return DropoutFn.apply(x, float(p_drop)) if training and p_drop > 0 else x


class InputEncoderFn(Function):
    """dropout(LN_a(feat W_a^T + b_a) + LN_b(bbox W_b^T + b_b)): the object / OCR input encoder (sam/sa_m4c.py:213-224, 252-263) as ONE autograd
    node of two launches forward (the wide projection as a GEMM; the 4 -> 768 box projection, both LayerNorms, the sum and the dropout in
    sam_input_encoder_fwd) and three backward (sam_input_encoder_bwd = one row pass + a fixed-order finalize of the eight small gradients, then the
    wide weight gradient with its bias gradient fused); the inputs are features, nothing flows further upstream.  A single node so that, for data
    parallelism, the end of its backward IS the point at which the eight parameters' gradients are final (`owner._sam_region_id`).
    feat: bf16 [R, K_pad] (ops.l2norm_pack); bbox: fp32 [R, >= 4], read in place from the batch (row stride free)."""

    @staticmethod
    def forward(ctx, anchor, feat, bbox, lin_a, ln_a, lin_b, ln_b, p_drop, owner):
        wa, _, ba, _, na, _ = _padded_views(lin_a.weight, lin_a.bias)
        wb, _, bb, _, nb, _ = _padded_views(lin_b.weight, lin_b.bias)
        if feat.shape[1] != wa.shape[1] or na != lin_a.weight.shape[0] or nb != lin_b.weight.shape[0] or lin_b.weight.shape[1] != 4:
            raise capi.SamHipError("InputEncoderFn: the feature operand must arrive K-padded (ops.l2norm_pack), out_features must be a multiple of 8, the box projection 4 -> D")
        if bbox.dtype != torch.float32 or bbox.stride(1) != 1:
            bbox = bbox.float().contiguous()
        za = ops.gemm(feat, wa, epilogue=capi.EPI_BIAS, bias=ba)
        ctx.seed = dropout_clock.next() if p_drop > 0 else (0, 0)
        out, stats = ops.input_encoder_fwd(za, bbox, wb, bb, ln_a, ln_b, p_drop, *ctx.seed)
        ctx.save_for_backward(feat, bbox, za, stats)
        ctx.mods, ctx.p_drop, ctx.owner = (lin_a, ln_a, lin_b, ln_b), p_drop, owner
        return out

    @staticmethod
    def backward(ctx, dy):
        feat, bbox, za, stats = ctx.saved_tensors
        lin_a, ln_a, lin_b, ln_b = ctx.mods
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16 or dy2.stride(1) != 1 or dy2.stride(0) % 4 or dy2.data_ptr() % 8:
            dy2 = dy2.to(BF16).contiguous()
        _, gva, _, dba, _, _ = _padded_views(lin_a.weight, lin_a.bias)
        wb, gvb, bb, dbb, _, _ = _padded_views(lin_b.weight, lin_b.bias)
        dza = ops.input_encoder_bwd(dy2, za, bbox, wb, bb, ln_a, ln_b, stats, gvb, dbb, ctx.p_drop, *ctx.seed)
        ops.gemm(dza, feat, a_kcontig=False, b_kcontig=False, out=gva, accumulate=True, split_k=-1, bias_grad=dba)        # dW_a += dza^T feat ; db_a += colsum(dza)
        rid = getattr(ctx.owner, "_sam_region_id", None)
        if rid is not None and parallel.active_reducer is not None:
            parallel.active_reducer.mark_done(rid)
        return (None,) * 9


class GradBarrierFn(Function):
    """identity; its backward runs when the COMPLETE gradient of `x` has arrived, i.e. after every consumer of x has run its backward
    (autograd's dependency counting), and then reports `name` to the active gradient reducer.  SAM4C puts one on each of the three
    encoder outputs that feed the MMT (question, objects, OCR): once all three have fired, everything downstream of them -- the MMT
    layers, PrevPredEmbeddings, the classifier and the pointer network -- has finished its backward, so those gradients are final."""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        red = parallel.active_reducer
        if red is not None:
            DeferredWgrads.flush_extras()        # (the heads' weight gradients must be final before their regions close)
            red.barrier_hit(ctx.name)
        return g, None


def region_done(rid):
    """report gradient region `rid` final to the active data-parallel reducer.  Deferred LayerNorm finalizes (the dgamma / dbeta / dbias partial
    sums of the layer's two LayerNorm backwards) belong to the region: they are reduced first, in one batched launch."""
    red = parallel.active_reducer
    if rid is None or red is None:
        return
    if ops.LnFinalizeQueue.defer:
        ops.LnFinalizeQueue.flush()
        if torchops.enabled():
            torchops.ns().ln_finalize_flush()
    red.mark_done(rid)


class DeferredWgrads:
    """weight gradients of layers marked `_sam_defer_wgrad` (TextBert's three 1280-row layers: a grouped launch of four 18-GFLOP problems fills 216 of
    512 block slots for 38 us, three times in a row on the tail's critical chain) are queued by EncoderLayerFn.backward and run as ONE grouped launch of up
    to 12 problems when the embedding block below them starts its backward (EmbedLayerNormFn) -- or, failing that, when the Trainer joins the backward.
    Their data-parallel regions are reported done after that launch, in backward order.  Without a gradient reducer the MMT's last layer group waits for them
    (`held`) and the two go out as one launch of 20 problems of mixed depth (flush)."""
    jobs, layers, acc = [], [], None
    late = []                   # [(jobs, stream)]: operand sets a launch on `stream` is (or may still be) reading; join() / clear() wait for each stream named here
    side_stream = None
    held = []                   # [(jobs, layers, acc, event)]: the MMT's last group, waiting for TextBert's problems (flush)
    extra_events = []           # one event per extra, recorded where its operands were produced
    extras = []                 # weight gradients of nn.Linear sites outside the encoder layers (classifier, pointer-net q / k), waiting for the next MMT group
    late_armed = False          # set by whoever will call join() before it reads the gradients (Trainer._eager_step); plain autograd use never leaves the issuing stream

    @classmethod
    def add(cls, layer, jobs, acc):
        if any(l is layer for l in cls.layers):
            cls.clear()          # a backward pass that never reached its flush (it raised): its operands are stale, this pass recomputes them
        if cls.jobs and (cls.acc != acc or len(cls.jobs) + len(jobs) > 12):
            cls.flush()
        cls.jobs, cls.acc = cls.jobs + list(jobs), acc
        cls.layers.append(layer)
        # MMT layers go out in PAIRS (modules.SAM4C marks them `_sam_defer_flush_at = 2`): two layers are 216 tiles of 256 x 256 -- one launch round with
        # the K range of every tile whole, i.e. no pair exchange and no second block per tile, instead of twice 108 tiles x 2 k-halves
        limit = getattr(layer, "_sam_defer_flush_at", None)
        if limit and len(cls.layers) >= limit:
            cls.flush(late=bool(getattr(layer, "_sam_wgrad_late", False)))

    @classmethod
    def add_extra(cls, job):
        """the classifier's and the pointer network's weight gradients (three GEMMs of 9-60 tiles x 12-50 k-tiles + their split-K reductions: ~60 us of launches at
        the head of the backward's chain) ride on the 40 CUs the MMT's next layer-pair launch leaves idle (gemm8w.hip: shallow problems behind the deep ones).
        Only inside Trainer._eager_step (which flushes whatever is left), on the device, for shapes the grouped kernels take; False = the caller launches it itself.
        Their data-parallel regions are closed by the encoder-output barriers, i.e. after the MMT's whole backward: the launch they join is long done by then."""
        dy, x, dw, db, _ = job
        if not (cls.late_armed and dy.is_cuda and os.environ.get("SAM_DEFER_HEAD_WGRAD", "1") != "0" and defer_mmt_pairs_active()):
            return False
        if dy.shape[0] % 64 or dy.shape[1] % 8 or x.shape[1] % 8 or dy.stride(0) % 8 or x.stride(0) % 8 or dw.stride(0) % 4 or len(cls.extras) >= 6:
            return False
        if dy.data_ptr() % 16 or x.data_ptr() % 16 or dw.data_ptr() % 16:
            return False
        ev = torch.cuda.Event()
        ev.record()                         # the operands were produced on THIS stream (the pointer net's run on the model's side stream): the launch that takes
        cls.extra_events.append(ev)         # them -- on whatever stream it goes out -- waits for the event, not for the autograd engine's implicit ordering
        cls.extras.append(job)
        return True

    @classmethod
    def flush(cls, late=False):
        if not cls.jobs and not cls.held and not (cls.extras and not late):
            return
        jobs, layers, acc = cls.jobs, cls.layers, cls.acc
        cls.jobs, cls.layers, cls.acc = [], [], None
        armed = late and cls.late_armed and jobs and jobs[0][0].is_cuda and wgrad_late_enabled()
        # (not under a gradient reducer: the held group's buckets would leave only after the launch of 20, at the very end of the backward, instead of
        # underneath TextBert's chain -- ~60 MB more of exposed all-reduce per step for 0.02 ms of compute)
        if armed and wgrad_merge_enabled() and not cls.held and parallel.active_reducer is None:
            # the group that closes the MMT's backward is HELD: TextBert's three 1280-row layers (12 shallow problems: 324 tiles of 20 k-tiles, a 90 us launch
            # that fills the chip 1.3 times) join it in ONE launch of 20 problems -- the pair's 216 deep tiles take 216 CUs for ~300 us, TextBert's tiles
            # run on the 40 CUs that round leaves idle (csrc/gemm8w.hip, n_long).  The launch goes out where TextBert's flush happens (its embedding
            # block's backward, on the side stream), after an event recorded here; Trainer joins that stream before it reads a gradient.
            ev = torch.cuda.Event()
            ev.record()
            cls.held = [(jobs, layers, acc, ev)]
            return
        if cls.held:
            hjobs, hlayers, hacc, ev = cls.held[0]
            cls.held = []
            if jobs and hacc == acc and len(hjobs) + len(jobs) <= 20 and os.environ.get("SAM_WGRAD_MERGE_STREAM", "1") != "0":
                # the launch of 20 goes to a stream of its own: what follows on the issuing stream (TextBert's embedding block: five small kernels, 45 us)
                # runs beside it instead of behind it; Trainer joins before the gradient norm
                cur = torch.cuda.current_stream()
                if cls.side_stream is None:
                    cls.side_stream = torch.cuda.Stream()
                cls.side_stream.wait_stream(cur)
                cls.side_stream.wait_event(ev)
                with torch.cuda.stream(cls.side_stream):
                    ops.wgrad_grouped(hjobs + jobs, accumulate=acc)
                    for layer in hlayers + layers:
                        region_done(getattr(layer, "_sam_region_id", None))
                cls.late.append((hjobs + jobs, cls.side_stream))
                return
            torch.cuda.current_stream().wait_event(ev)
            if jobs and hacc == acc and len(hjobs) + len(jobs) <= 20:
                jobs, layers = hjobs + jobs, hlayers + layers
            else:                                   # (nothing to merge with, or not mergeable: the held group goes out by itself)
                ops.wgrad_grouped(hjobs, accumulate=hacc)
                for layer in hlayers:
                    region_done(getattr(layer, "_sam_region_id", None))
            cls.late.append((hjobs, torch.cuda.current_stream()))      # operands of the held group were allocated on another stream: alive until join()
            if not jobs:
                return
        elif armed:
            # the group that closes the MMT's backward (its first layers): nothing on the rest of the backward path reads these weight gradients, and what
            # follows on the issuing stream is the tail's chain of small kernels (embedding blocks, object / OCR encoders, TextBert's 1280-row layers).
            # The ~0.3 ms launch runs on a stream of its own next to that chain instead of in front of it; Trainer joins it before the gradient norm.
            if parallel.active_reducer is not None and ops.LnFinalizeQueue.defer:     # (the layers' LayerNorm partial sums: finalized where they were written)
                ops.LnFinalizeQueue.flush()
                if torchops.enabled():
                    torchops.ns().ln_finalize_flush()
            cur = torch.cuda.current_stream()
            if cls.side_stream is None:
                cls.side_stream = torch.cuda.Stream()
            cls.side_stream.wait_stream(cur)
            with torch.cuda.stream(cls.side_stream):
                ops.wgrad_grouped(jobs, accumulate=acc)
                for layer in layers:
                    region_done(getattr(layer, "_sam_region_id", None))
            cls.late.append((jobs, cls.side_stream))       # operands were allocated on the issuing stream: alive until join()
            return
        if cls.extras and len(jobs) + len(cls.extras) <= 20:
            cls._wait_extras()
            cls.late.append((cls.extras, torch.cuda.current_stream()))     # (allocated on another stream: alive until join(); the launch below reads them HERE)
            jobs, cls.extras = list(jobs) + cls.extras, []
        if jobs:
            ops.wgrad_grouped(jobs, accumulate=bool(acc))
        for layer in layers:
            region_done(getattr(layer, "_sam_region_id", None))

    @classmethod
    def flush_extras(cls):
        """extras that found no MMT group to ride on (a model whose only group is the late one): launched by themselves, here"""
        if cls.extras:
            cls._wait_extras()
            jobs, cls.extras = cls.extras, []
            ops.wgrad_grouped(jobs, accumulate=True)
            cls.late.append((jobs, torch.cuda.current_stream()))

    @classmethod
    def _wait_extras(cls):
        cur = torch.cuda.current_stream()
        for ev in cls.extra_events:
            cur.wait_event(ev)
        cls.extra_events = []

    @classmethod
    def join(cls):
        """the current stream waits for every weight-gradient launch that went to the side (flush(late=True)) or took operands of another stream: each entry
        names the stream ITS launch was issued on (round 5 kept one class-wide `late_stream` that was never reset: from the second step on it could name last
        step's stream while the extras had gone out on another)"""
        cur = torch.cuda.current_stream()
        for st in cls._late_streams():
            if st != cur:
                cur.wait_stream(st)
        cls.late = []

    @classmethod
    def _late_streams(cls):
        out = []
        for _, st in cls.late:
            if st is not None and all(st != o for o in out):
                out.append(st)
        return out

    @classmethod
    def clear(cls):
        cls.jobs, cls.layers, cls.acc = [], [], None
        cls.held, cls.extras, cls.extra_events = [], [], []
        for st in cls._late_streams():      # a launch may still be reading its operands on its stream: the issuing stream waits before they are dropped
            try:
                if st != torch.cuda.current_stream():
                    torch.cuda.current_stream().wait_stream(st)
            except RuntimeError:
                pass
        cls.late = []


def defer_mmt_pairs_active():
    """are MMT layers' weight gradients going out in grouped launches at all (modules.SAM4C marks the layers)?"""
    return os.environ.get("SAM_DEFER_MMT_WGRAD", "2") not in ("0", "1")


def wgrad_late_enabled():
    return os.environ.get("SAM_WGRAD_LATE", "1") != "0"


def wgrad_merge_enabled():
    return os.environ.get("SAM_WGRAD_MERGE_TB", "1") != "0"


def defer_wgrad_enabled():
    return os.environ.get("SAM_DEFER_TB_WGRAD", "1") != "0"


# ------------------------------------------------------------------------------------------------ encoder layer
class EncoderLayerFn(Function):
    """One BERT-style encoder layer (spatial or plain — the difference is entirely in `allow`), forward and backward,
    13 kernel launches forward / 21 backward, nothing but these kernels touches the activations.
    Reference: SpatialBertLayer.forward sam/sa_m4c.py:670-684 (and pytorch-transformers BertLayer for 'n' layers)."""

    @staticmethod
    def forward(ctx, x, anchor, layer, allow, batch, p_attn, p_hid):
        att, so, inter, out = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        wqkv, bqkv, _, _ = _fused_qkv(att)
        heads = att.num_attention_heads
        scale = 1.0 / math.sqrt(att.attention_head_size)
        seeds = [dropout_clock.next() for _ in range(3)]
        ctx.layer, ctx.batch, ctx.p_attn, ctx.p_hid, ctx.seeds, ctx.scale = layer, batch, p_attn, p_hid, seeds, scale
        if coarse_ops_active() and x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1:
            # one custom-op call enqueues the whole layer from C++ (csrc_torch/sam_torch_ops.cpp): same kernels, same order, a tenth of the host time
            outs = torchops.ns().encoder_layer_fwd(x, allow, _layer_params(layer), batch, heads, scale, p_attn, p_hid, [v for sd in seeds for v in sd],
                                                   so.LayerNorm.variance_epsilon, out.LayerNorm.variance_epsilon)
            ctx.save_for_backward(*outs[1:], allow)
            ctx.coarse = True
            return outs[0]
        ctx.coarse = False
        qkv = ops.gemm(x, wqkv, epilogue=capi.EPI_BIAS, bias=bqkv)
        fused_bwd = x.shape[0] // batch <= ops.attn_bwd_fused_max_n() and os.environ.get("SAM_ATTN_BWD_FUSED", "1") != "0"     # (the residual is only written for a backward that reads it)
        if fused_bwd:       # the one-pass attention backward takes delta from the output and its rounding residual
            ctxv, lse2, keep, ctx_lo = ops.attn_fwd(qkv, allow, batch, heads, scale, p_attn, *seeds[0], want_residual=True)
        else:
            (ctxv, lse2, keep), ctx_lo = ops.attn_fwd(qkv, allow, batch, heads, scale, p_attn, *seeds[0]), None
        z1, a, mean1, rstd1 = ops.gemm_ln(ctxv, _w(so.dense.weight), so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.variance_epsilon,
                                          epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=so.dense.bias, residual=x, p_drop=p_hid, seed=seeds[1][0], offset=seeds[1][1])
        pre = torch.empty((x.shape[0], inter.dense.weight.shape[0]), dtype=BF16, device=x.device)
        h = ops.gemm(a, _w(inter.dense.weight), epilogue=capi.EPI_BIAS_GELU_GRAD, bias=inter.dense.bias, aux_out=pre)     # pre := gelu'(a W1^T + b1)
        z2, y, mean2, rstd2 = ops.gemm_ln(h, _w(out.dense.weight), out.LayerNorm.weight, out.LayerNorm.bias, out.LayerNorm.variance_epsilon,
                                          epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=out.dense.bias, residual=a, p_drop=p_hid, seed=seeds[2][0], offset=seeds[2][1])
        ctx.save_for_backward(x, qkv, ctxv, lse2, keep, z1, mean1, rstd1, a, pre, h, z2, mean2, rstd2, allow, ctx_lo)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer, p_hid, seeds = ctx.layer, ctx.p_hid, ctx.seeds
        att, so, inter, out = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        # the Trainer marks a layer's twelve gradients "fresh" (not zeroed, nothing accumulated yet) at the start of a step: the first -- and in this
        # model only -- backward through the layer then OVERWRITES them, which saves the zero-fill and the read half of every weight gradient's
        # read-modify-write; without the mark (plain autograd use, gradient accumulation) everything accumulates as usual
        acc = not getattr(layer, "_sam_grad_fresh", False)
        layer._sam_grad_fresh = False
        defer = bool(getattr(layer, "_sam_defer_wgrad", False)) and defer_wgrad_enabled()
        if ctx.coarse:
            *saved, allow = ctx.saved_tensors
            if defer:
                dx, dy2, dpre, dy1, dqkv = torchops.ns().encoder_layer_bwd_nowgrad(dy, saved, allow, _layer_params(layer), _layer_grads(layer), ctx.batch,
                                                                                   att.num_attention_heads, ctx.scale, ctx.p_attn, p_hid,
                                                                                   [v for sd in seeds for v in sd], bool(ctx.needs_input_grad[0]), acc)
                _, _, dwqkv, dbqkv = _fused_qkv(att)
                x_, ctx_, a_, h_ = saved[0], saved[2], saved[8], saved[10]
                DeferredWgrads.add(layer, [(dy2, h_, out.dense.weight.grad, None), (dpre, a_, inter.dense.weight.grad, inter.dense.bias.grad),
                                           (dy1, ctx_, so.dense.weight.grad, None), (dqkv, x_, dwqkv, dbqkv)], acc)
            else:
                dx = torchops.ns().encoder_layer_bwd(dy, saved, allow, _layer_params(layer), _layer_grads(layer), ctx.batch, att.num_attention_heads, ctx.scale,
                                                     ctx.p_attn, p_hid, [v for sd in seeds for v in sd], bool(ctx.needs_input_grad[0]), acc)
                region_done(getattr(layer, "_sam_region_id", None))
            return (dx if ctx.needs_input_grad[0] else None), None, None, None, None, None, None
        x, qkv, ctxv, lse2, keep, z1, mean1, rstd1, a, pre, h, z2, mean2, rstd2, allow, ctx_lo = ctx.saved_tensors
        wqkv, _, dwqkv, dbqkv = _fused_qkv(att)
        if dy.dtype != BF16 or not dy.is_contiguous():
            dy = dy.to(BF16).contiguous()
        # ---- output block: y = LN(dropout(h W2^T + b2) + a)
        dz2, dy2 = ops.layernorm_bwd(dy, z2, mean2, rstd2, out.LayerNorm.weight, out.LayerNorm.weight.grad, out.LayerNorm.bias.grad,
                                     dbias=out.dense.bias.grad, want_dropped=True, p_drop=p_hid, seed=seeds[2][0], offset=seeds[2][1], accumulate=acc, may_defer=True)
        wgrads = [(dy2, h, out.dense.weight.grad, None)]          # the four weight gradients go out as ONE grouped launch at the end
        dpre = ops.gemm(dy2, _w(out.dense.weight), b_kcontig=False, epilogue=capi.EPI_MUL_AUX, aux_in=pre)
        # ---- intermediate: h = gelu(a W1^T + b1)
        wgrads.append((dpre, a, inter.dense.weight.grad, inter.dense.bias.grad))          # bias gradient fused into the wgrad
        da = ops.gemm(dpre, _w(inter.dense.weight), b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=dz2)   # + residual path
        # ---- attention output block: a = LN(dropout(ctx Wo^T + bo) + x)
        dz1, dy1 = ops.layernorm_bwd(da, z1, mean1, rstd1, so.LayerNorm.weight, so.LayerNorm.weight.grad, so.LayerNorm.bias.grad,
                                     dbias=so.dense.bias.grad, want_dropped=True, p_drop=p_hid, seed=seeds[1][0], offset=seeds[1][1], accumulate=acc, may_defer=True)
        wgrads.append((dy1, ctxv, so.dense.weight.grad, None))
        dctx = ops.gemm(dy1, _w(so.dense.weight), b_kcontig=False)
        # ---- attention core + fused QKV projection
        dqkv = ops.attn_bwd(dctx, qkv, lse2, allow, keep, ctx.batch, att.num_attention_heads, ctx.scale, ctx.p_attn, out=ctxv, out_lo=ctx_lo)
        wgrads.append((dqkv, x, dwqkv, dbqkv))
        if defer:
            DeferredWgrads.add(layer, wgrads, acc)
        else:
            ops.wgrad_grouped(wgrads, accumulate=acc)
        dx = ops.gemm(dqkv, wqkv, b_kcontig=False, epilogue=capi.EPI_BIAS_DROPOUT_RES, residual=dz1) if ctx.needs_input_grad[0] else None
        if not defer:
            region_done(getattr(layer, "_sam_region_id", None))     # this layer's gradients are final: its bucket may go out now
        return dx, None, None, None, None, None, None


def _rows_bf16(t):
    x2 = t.reshape(-1, t.shape[-1])
    if x2.dtype != BF16 or x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
        x2 = x2.to(BF16).contiguous()
    return x2


class DenseDropoutResLnFn(Function):
    """BertSelfOutput / BertOutput called as stand-alone modules (sam/sa_m4c.py:653, 680): LayerNorm(dropout(x W^T + b) + residual) through the SAME kernels the
    fused encoder layer uses -- GEMM with the bias + dropout + residual epilogue, sam_layernorm_fwd; backward: sam_layernorm_bwd (which regenerates the
    dropout mask and emits the masked gradient for the dense + its bias gradient), weight-gradient GEMM, dgrad GEMM.  No torch arithmetic."""

    @staticmethod
    def forward(ctx, x, residual, mod, p_drop):
        dense, ln = mod.dense, mod.LayerNorm
        n, k = dense.weight.shape
        if n % 8 or k % 8:
            raise capi.SamHipError("%s: hidden sizes must be multiples of 8 (got %d -> %d)" % (type(mod).__name__, k, n))
        x2, r2 = _rows_bf16(x), _rows_bf16(residual)
        seed = dropout_clock.next()
        z, y, mean, rstd = ops.gemm_ln(x2, _w(dense.weight), ln.weight, ln.bias, ln.variance_epsilon, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=dense.bias,
                                       residual=r2, p_drop=p_drop, seed=seed[0], offset=seed[1])
        ctx.save_for_backward(x2, z, mean, rstd)
        ctx.mod, ctx.p_drop, ctx.seed, ctx.shapes = mod, p_drop, seed, (x.shape, x.dtype, residual.shape, residual.dtype)
        return y.view(*residual.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, z, mean, rstd = ctx.saved_tensors
        dense, ln = ctx.mod.dense, ctx.mod.LayerNorm
        xs, xd, rs, rd = ctx.shapes
        dy2 = _rows_bf16(dy)
        dz, dyd = ops.layernorm_bwd(dy2, z, mean, rstd, ln.weight, ln.weight.grad, ln.bias.grad, dbias=dense.bias.grad, want_dropped=True, p_drop=ctx.p_drop,
                                    seed=ctx.seed[0], offset=ctx.seed[1], accumulate=True)
        ops.gemm(dyd, x2, a_kcontig=False, b_kcontig=False, out=dense.weight.grad, accumulate=True, split_k=-1)          # dW += dyd^T x
        dx = ops.gemm(dyd, _w(dense.weight), b_kcontig=False).view(xs).to(xd) if ctx.needs_input_grad[0] else None       # dx = dyd W
        return dx, (dz.view(rs).to(rd) if ctx.needs_input_grad[1] else None), None, None


class DenseGeluFn(Function):
    """BertIntermediate as a stand-alone module (sa_m4c.py:678, 985-991): erf-GELU(x W^T + b) in the GEMM's epilogue, which also stores gelu'(pre) when a
    backward will follow; backward: dpre = dy * gelu' (sam_rowvec_bf16), weight gradient with the bias gradient fused, dgrad."""

    @staticmethod
    def forward(ctx, x, mod, want_grad):
        dense = mod.dense
        n, k = dense.weight.shape
        if n % 8 or k % 8:
            raise capi.SamHipError("BertIntermediate: sizes must be multiples of 8 (got %d -> %d)" % (k, n))
        x2 = _rows_bf16(x)
        dact = torch.empty((x2.shape[0], n), dtype=BF16, device=x2.device) if want_grad else None
        h = ops.gemm(x2, _w(dense.weight), epilogue=capi.EPI_BIAS_GELU_GRAD, bias=dense.bias, aux_out=dact)
        if want_grad:
            ctx.save_for_backward(x2, dact)
        ctx.mod, ctx.shape, ctx.dtype = mod, x.shape, x.dtype
        return h.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, dact = ctx.saved_tensors
        dense = ctx.mod.dense
        dpre = ops.rowvec("mul", _rows_bf16(dy), dact)
        ops.gemm(dpre, x2, a_kcontig=False, b_kcontig=False, out=dense.weight.grad, accumulate=True, split_k=-1, bias_grad=dense.bias.grad)
        dx = ops.gemm(dpre, _w(dense.weight), b_kcontig=False).view(ctx.shape).to(ctx.dtype) if ctx.needs_input_grad[0] else None
        return dx, None, None


def coarse_ops_active():
    """the C++ per-layer ops are used unless switched off (SAM_COARSE_OPS=0) or bench.py's per-kernel event profiler is recording"""
    return torchops.enabled() and capi.profiler is None


def _layer_params(layer):
    """the 12 operands of one encoder layer in the order csrc_torch/sam_torch_ops.cpp expects (cached: the views never move)"""
    c = getattr(layer, "_sam_coarse_params", None)
    if c is None:
        att, so, inter, out = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        wqkv, bqkv, _, _ = _fused_qkv(att)
        c = layer._sam_coarse_params = [wqkv, bqkv, _w(so.dense.weight), so.dense.bias.data, so.LayerNorm.weight.data, so.LayerNorm.bias.data,
                                        _w(inter.dense.weight), inter.dense.bias.data, _w(out.dense.weight), out.dense.bias.data,
                                        out.LayerNorm.weight.data, out.LayerNorm.bias.data]
    return c


def _layer_grads(layer):
    c = getattr(layer, "_sam_coarse_grads", None)
    if c is None:
        att, so, inter, out = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        _, _, dwqkv, dbqkv = _fused_qkv(att)
        c = layer._sam_coarse_grads = [dwqkv, dbqkv, so.dense.weight.grad, so.dense.bias.grad, so.LayerNorm.weight.grad, so.LayerNorm.bias.grad,
                                       inter.dense.weight.grad, inter.dense.bias.grad, out.dense.weight.grad, out.dense.bias.grad,
                                       out.LayerNorm.weight.grad, out.LayerNorm.bias.grad]
    return c


def encoder_layer(x2d, layer, allow, batch, training):
    cfg_p_attn = layer.attention.self.dropout_p if training else 0.0
    cfg_p_hid = layer.output.dropout_p if training else 0.0
    return EncoderLayerFn.apply(x2d, layer.output.LayerNorm.weight, layer, allow, batch, cfg_p_attn, cfg_p_hid)


class AttentionFn(Function):
    """stand-alone fused attention (module-level SpatialBertSelfAttention / BertSelfAttention API)"""

    @staticmethod
    def forward(ctx, qkv, allow, batch, heads, scale, p_drop, side=None):
        qkv = qkv.contiguous()
        seed = dropout_clock.next()
        if qkv.shape[0] // batch <= ops.attn_bwd_fused_max_n() and os.environ.get("SAM_ATTN_BWD_FUSED", "1") != "0":
            out, lse2, keep, out_lo = ops.attn_fwd(qkv, allow, batch, heads, scale, p_drop, *seed, want_residual=True)
        else:
            (out, lse2, keep), out_lo = ops.attn_fwd(qkv, allow, batch, heads, scale, p_drop, *seed), None
        ctx.save_for_backward(qkv, lse2, keep, allow, out, out_lo)
        ctx.cfg = (batch, heads, scale, p_drop)
        if side is not None:         # `output_attentions`: what ops.attn_probs rebuilds the probabilities from
            side.update(qkv=qkv, lse2=lse2, keep=keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, lse2, keep, allow, out, out_lo = ctx.saved_tensors
        batch, heads, scale, p_drop = ctx.cfg
        return ops.attn_bwd(dout.to(BF16).contiguous(), qkv, lse2, allow, keep, batch, heads, scale, p_drop, out=out, out_lo=out_lo), None, None, None, None, None, None


class RowScaleFn(Function):
    """ctx * vec over the model dimension (head_mask as one factor per head, broadcast over the head's columns: probs * head_mask before P V is the same as scaling the
    head's context columns, sa_m4c.py:591-594); backward: the same scaling of the gradient.  vec: fp32 [D], no gradient."""

    @staticmethod
    def forward(ctx, x, vec):
        ctx.save_for_backward(vec)
        return ops.rowvec("mul_vec", x.contiguous(), vec=vec)

    @staticmethod
    def backward(ctx, dy):
        (vec,) = ctx.saved_tensors
        return ops.rowvec("mul_vec", dy.to(BF16).contiguous(), vec=vec), None


class HeadBiasFn(Function):
    """context_layer + biases(0) (`use_bias`, sa_m4c.py:439-443, 600-603): one fp32 row added to every context row; backward: the gradient passes through,
    the bias row receives its column sums (sam_colsum_bf16, accumulated into the parameter's .grad in flat storage)."""

    @staticmethod
    def forward(ctx, x, table):
        ctx.table = table
        return ops.rowvec("add_vec", x.contiguous(), vec=table.detach().reshape(-1).float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.to(BF16).contiguous()
        ops.colsum(dy, ctx.table.grad.view(-1), accumulate=True)
        return dy, None


# ------------------------------------------------------------------------------------------------ pointer net / loss
class PtrScoresFn(Function):
    @staticmethod
    def forward(ctx, q, k, ocr_mask_u8, scale):
        q, k = q.contiguous(), k.contiguous()
        ctx.save_for_backward(q, k)
        ctx.scale = scale
        return ops.ptr_scores_fwd(q, k, ocr_mask_u8, scale)

    @staticmethod
    def backward(ctx, ds):
        q, k = ctx.saved_tensors
        dq, dk = ops.ptr_scores_bwd(ds.float().contiguous(), q, k, ctx.scale)
        return dq, dk, None, None


_PLACEHOLDERS = {}


def _placeholder(device):
    t = _PLACEHOLDERS.get(device)
    if t is None:
        t = _PLACEHOLDERS[device] = torch.zeros((), dtype=torch.float32, device=device)
    return t


class BceLossFn(Function):
    """M4CDecodingBCEWithMaskLoss (sam/task_utils.py:19-30) on the two score blocks; gradient computed in the forward pass"""

    @staticmethod
    def forward(ctx, fixed, ocr, targets, loss_mask, grad_scale, unit_grad=False, global_count=None):
        r = fixed.shape[0] * fixed.shape[1]
        f2, o2 = fixed.reshape(r, -1), ocr.reshape(r, -1)
        # data parallel: global_count = all-reduced number of unmasked decoding steps (device scalar, no host sync); the kernel normalises
        # by max(global_count, 1) in fp32, so loss and both gradient blocks carry exactly this rank's share of the global mean
        loss, d_fixed, d_ocr = ops.bce_loss(f2, o2, targets.reshape(r, -1), loss_mask.reshape(r).contiguous(), grad_scale, global_count)
        ctx.save_for_backward(d_fixed, d_ocr)
        ctx.shapes, ctx.unit_grad = (fixed.shape, ocr.shape), unit_grad
        ctx.side = getattr(fixed, "_sam_grad_side", None) if fixed.dtype == torch.float32 else None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d_fixed, d_ocr = ctx.saved_tensors
        if ctx.unit_grad:      # the caller (Trainer.step) runs loss.backward() itself: upstream gradient is exactly 1, skip three elementwise passes
            if ctx.side is not None:
                # the producer of `fixed` (LinearFn via linear(..., out_f32=True)) takes the bf16 gradient through its side channel; what goes through
                # autograd is a zero-stride fp32 placeholder of the right shape (no kernel, and nothing reads it)
                ctx.side["dy_bf16"] = d_fixed.view(ctx.shapes[0])
                return _placeholder(d_ocr.device).expand(ctx.shapes[0]), d_ocr.view(ctx.shapes[1]), None, None, None, None, None
            return d_fixed.view(ctx.shapes[0]), d_ocr.view(ctx.shapes[1]), None, None, None, None, None
        return (d_fixed.view(ctx.shapes[0]) * g).to(torch.float32), (d_ocr.view(ctx.shapes[1]) * g), None, None, None, None, None
