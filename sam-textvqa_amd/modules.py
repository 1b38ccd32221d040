"""Host-side mirror of the reference's nn.Module surface for the SA-M4C hot path (sam/sa_m4c.py).

Same class names, constructor arguments, config keys, forward signatures, batch_dict side effects and state_dict
keys as the reference, so a reference checkpoint loads unchanged and callers (train.py:92-94,133;
task_utils.py:118; evaluator.py:168) can switch imports.  All arithmetic runs in the HIP kernels of
libsam_hip.so through autograd.py; there is no eager fallback: the modules raise if the library is missing or
the tensors are not on the GPU.  Activations are bf16 between kernels, parameters are fp32 masters with bf16
shadows (params.py)."""
import logging
import math
import os
from collections import Counter

import torch
from torch import nn

from . import _capi as capi
from . import ops
from .autograd import (BF16, AttentionFn, DenseDropoutResLnFn, DenseGeluFn, EmbedLayerNormFn, HeadBiasFn, RowScaleFn, GradBarrierFn, InputEncoderFn, PrevPredFn, SeqRowsFn, cat_rows, dropout, PtrScoresFn, _fused_qkv, _w, encoder_layer, layer_norm,
                       linear)
from .params import prepare
from .registry import registry


class BertConfig:
    """pytorch-transformers BertConfig defaults; from_dict copies EVERY key into the object (train.py:92-93)."""

    DEFAULTS = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
                    type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, output_attentions=False, output_hidden_states=False)

    def __init__(self, **kw):
        self.__dict__.update(self.DEFAULTS)
        self.__dict__.update(kw)

    @classmethod
    def from_dict(cls, d):
        return cls(**dict(d))


# ------------------------------------------------------------------------------------------ mask plumbing
class AllowBits:
    """Per-batch allow-bit masks: `base` [B,1,N,NW] (key padding + prefix-LM/causal) and a cache of the per-head
    spatial masks derived from it, keyed by the relation tensor they were built from.  Built once per batch, shared
    by all layers' forward and backward."""

    def __init__(self, base):
        self.base = base
        self._spatial = {}

    def spatial(self, adj, n_txt, n_heads, quadrants):
        key = (adj.data_ptr(), adj._version, n_txt, n_heads, tuple(quadrants))
        bits = self._spatial.get(key)
        if bits is None:
            if adj.dtype != torch.int8:
                adj = adj.to(torch.int8)
            adj = adj.to(self.base.device, non_blocking=True).contiguous()
            bits = self._spatial[key] = ops.mask_bits_spatial(self.base, adj, n_txt, n_heads, quadrants)
        return bits


def as_allow(attention_mask):
    """accept either AllowBits (internal callers) or the reference's additive [B,1,N,N] float mask (module-level API)"""
    if isinstance(attention_mask, AllowBits):
        return attention_mask
    cached = getattr(attention_mask, "_sam_allow", None)
    if cached is None or cached[0] != attention_mask._version:
        m = attention_mask
        if m.dim() == 4 and m.shape[2] == 1:      # TextBert-style [B,1,1,N] key mask
            m = m.expand(-1, -1, m.shape[3], -1)
        bits = ops.mask_bits_from_additive(m.to(device="cuda", dtype=torch.float32).contiguous())
        cached = (attention_mask._version, AllowBits(bits))
        try:
            attention_mask._sam_allow = cached
        except Exception:
            pass
    return cached[1]


def _to_rows(hidden_states):
    b, n, d = hidden_states.shape
    x = hidden_states.reshape(b * n, d)
    if x.dtype != BF16 or not x.is_contiguous():
        x = x.to(BF16).contiguous()
    return x, b, n


def _head_scale(head_mask, n_heads, device):
    """head_mask (sa_m4c.py:591-592: attention_probs * head_mask) as ONE factor per head -> fp32 [H] on the device, or None.  The reference never passes one
    (`head_mask = [None] * num_hidden_layers`, :846; BertSpatialEncoder does not even hand it to its layers); the module-level API accepts what the BERT code
    base uses: a tensor of H values in any broadcast shape ([H], [1, H, 1, 1], [H, 1, 1]).  Per-sample or per-position masks are refused."""
    if head_mask is None:
        return None
    if isinstance(head_mask, (list, tuple)):
        raise capi.SamHipError("head_mask: pass ONE layer's mask (a tensor of %d per-head factors), not the per-layer list" % n_heads)
    hm = torch.as_tensor(head_mask)
    if hm.numel() != n_heads:
        raise NotImplementedError("head_mask of shape %s: only one factor per head (%d values, e.g. [H] or [1, H, 1, 1]) is supported; per-sample / per-position "
                                  "masks would need the probabilities materialised" % (tuple(hm.shape), n_heads))
    return hm.reshape(n_heads).to(device=device, dtype=torch.float32).contiguous()


class _HipModule(nn.Module):
    """base: lazily moves this subtree into flat GPU storage on first use"""

    def _ready(self):
        fp = self.__dict__.get("_sam_flat_params")
        if fp is None:
            first = next(self.parameters(), None)
            if first is None or getattr(first, "_sam_flat", None) is not None:
                return                      # parameter-free, or a child of an already prepared root
            fp = prepare(self)              # this module becomes a storage root
        fp.ensure_fresh()                   # only roots pay for the version scan


# ------------------------------------------------------------------------------------------ leaf blocks
class BertLayerNorm(_HipModule):
    """sam/sa_m4c.py:1016-1028"""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        self._ready()
        return layer_norm(x, self)


class BertSelfOutput(_HipModule):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout_p = config.hidden_dropout_prob

    def forward(self, hidden_states, input_tensor):
        """LayerNorm(dropout(dense(hidden_states)) + input_tensor): one GEMM with the bias + dropout + residual epilogue and the LayerNorm kernel -- the
        fused encoder layer's own launches (autograd.DenseDropoutResLnFn); nothing here is computed by torch"""
        self._ready()
        return DenseDropoutResLnFn.apply(hidden_states, input_tensor, self, float(self.dropout_p) if self.training else 0.0)


class BertIntermediate(_HipModule):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act not in ("gelu",):
            raise NotImplementedError("only erf-GELU (hidden_act='gelu') is implemented, as used by every shipped config")
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        self._ready()
        return DenseGeluFn.apply(hidden_states, self, torch.is_grad_enabled())          # erf-GELU in the GEMM's epilogue (autograd.DenseGeluFn)


class BertOutput(_HipModule):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout_p = config.hidden_dropout_prob

    forward = BertSelfOutput.forward


class BertSelfAttention(_HipModule):
    """plain additive-mask MHA of the 'n' layers / TextBert (pytorch-transformers BertSelfAttention)"""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = config.hidden_size
        self.output_attentions = config.output_attentions
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout_p = config.attention_probs_dropout_prob

    def _allow_bits(self, attention_mask, spatial_adj_matrix=None):
        return as_allow(attention_mask).base

    def forward(self, hidden_states, attention_mask, head_mask=None):
        return self._attend(hidden_states, self._allow_bits(attention_mask), head_mask)

    def _attend(self, hidden_states, allow, head_mask):
        """-> (context [B, N, D],) or (context, attention_probs fp32 [B, H, N, N]) under `output_attentions`.  The fused kernel never writes the probabilities:
        they are rebuilt afterwards from its saved rows (ops.attn_probs) -- after dropout and head mask, as the reference returns them (sa_m4c.py:586-609),
        detached from the graph (nothing upstream differentiates through them)."""
        self._ready()
        x, b, n = _to_rows(hidden_states)
        hs = _head_scale(head_mask, self.num_attention_heads, x.device)
        qkv = torch.cat([linear(x, self.query), linear(x, self.key), linear(x, self.value)], dim=1)
        p = self.dropout_p if self.training else 0.0
        scale = 1.0 / math.sqrt(self.attention_head_size)
        side = {} if self.output_attentions else None
        ctx = AttentionFn.apply(qkv, allow, b, self.num_attention_heads, scale, p, side)
        if hs is not None:
            ctx = RowScaleFn.apply(ctx, hs.repeat_interleave(self.attention_head_size).contiguous())
        if getattr(self, "use_bias", False):
            ctx = HeadBiasFn.apply(ctx, self.biases.weight)
        out = (ctx.view(b, n, -1).to(hidden_states.dtype),)
        if self.output_attentions:
            with torch.no_grad():
                out += (ops.attn_probs(side["qkv"], allow, side["lse2"], side["keep"], b, self.num_attention_heads, scale, p, head_scale=hs),)
        return out


class SpatialBertSelfAttention(BertSelfAttention):
    """sam/sa_m4c.py:399-610: one head per spatial relation; mask = min(attention_mask, relation mask), dead rows -> 0"""

    def __init__(self, config, use_implicit=False):
        assert hasattr(config, "num_spatial_relations")
        nn.Module.__init__(self)
        self.num_attention_heads = config.num_spatial_relations
        self.num_spatial_relations = config.num_spatial_relations
        if hasattr(config, "num_implicit_relations") and use_implicit:
            self.num_attention_heads += config.num_implicit_relations
            self.num_implicit_relations = config.num_implicit_relations
        if config.hidden_size % self.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, self.num_attention_heads))
        self.output_attentions = config.output_attentions
        self.max_seq_len = config.max_seq_length
        self.mask_quadrants = list(config.attention_mask_quadrants)
        self.max_decoding_steps = config.num_decoding_steps
        self.attention_head_size = config.hidden_size // self.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout_p = 0.0 if getattr(config, "no_drop", False) else config.attention_probs_dropout_prob
        self.use_bias = bool(getattr(config, "use_bias", False))
        if self.use_bias:
            # sa_m4c.py:439-443: one learned row added to the merged context (state_dict key `biases.weight` [1, hidden]).  Off in every shipped config: such a
            # layer runs through the module-by-module composition (_FusedLayer._run), not the single fused-layer node, and the persistent decoding kernel declines it
            logging.getLogger(__name__).info("using head biases")
            self.biases = nn.Embedding(1, config.hidden_size)

    def _allow_bits(self, attention_mask, spatial_adj_matrix=None):
        return as_allow(attention_mask).spatial(spatial_adj_matrix, self.max_seq_len, self.num_attention_heads, self.mask_quadrants)

    def forward(self, hidden_states, attention_mask, spatial_adj_matrix, head_mask=None):
        return self._attend(hidden_states, self._allow_bits(attention_mask, spatial_adj_matrix), head_mask)


class BertAttention(_HipModule):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None):
        so = self.self(input_tensor, attention_mask, head_mask)
        return (self.output(so[0], input_tensor),) + so[1:]


class SpatialBertAttention(_HipModule):
    """sam/sa_m4c.py:613-657 (prune_heads omitted: it references an undefined symbol upstream and is never called)"""

    def __init__(self, config, use_implicit=False):
        super().__init__()
        self.self = SpatialBertSelfAttention(config, use_implicit)
        self.output = BertSelfOutput(config)
        self.pruned_heads = set()

    def forward(self, input_tensor, attention_mask, spatial_adj_matrix, head_mask=None):
        so = self.self(input_tensor, attention_mask, spatial_adj_matrix, head_mask)
        return (self.output(so[0], input_tensor),) + so[1:]


class _FusedLayer(_HipModule):
    def _run(self, hidden_states, allow, head_mask):
        self._ready()
        att = self.attention.self
        if head_mask is not None or att.output_attentions or getattr(att, "use_bias", False):
            # the switches no shipped config turns on (off the path): module by module -- attention core (+ head factors, + head biases, + the
            # probabilities), then BertSelfOutput / BertIntermediate / BertOutput, each on the library's kernels (autograd.DenseDropoutResLnFn / DenseGeluFn)
            so = att._attend(hidden_states, allow, head_mask)
            a = self.attention.output(so[0], hidden_states)
            y = self.output(self.intermediate(a), a)
            return (y.to(hidden_states.dtype),) + so[1:]
        x, b, n = _to_rows(hidden_states)
        y = encoder_layer(x, self, allow, b, self.training)
        return (y.view(b, n, -1).to(hidden_states.dtype),)


# ------------------------------------------------------------------------------------------ inference (greedy decoding)
def _layer_tail(layer, ctx, x):
    """everything of an encoder layer after the attention core, eval mode (no dropout), on a row subset: O-proj + LN, FFN + LN"""
    so, inter, out = layer.attention.output, layer.intermediate, layer.output
    if getattr(layer.attention.self, "use_bias", False):          # context_layer + biases(0), sa_m4c.py:600-603
        ctx = ops.rowvec("add_vec", ctx.contiguous(), vec=layer.attention.self.biases.weight.detach().reshape(-1).float().contiguous())
    # (gemm_ln: the LayerNorm rides on the GEMM's split-K reduction pass when there is one -- the few-row steps of the decoding loops -- and is its own launch otherwise)
    _, a, _, _ = ops.gemm_ln(ctx, _w(so.dense.weight), so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.variance_epsilon,
                             epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=so.dense.bias, residual=x)
    h = ops.gemm(a, _w(inter.dense.weight), epilogue=capi.EPI_BIAS_GELU_GRAD, bias=inter.dense.bias)      # (the 8-wave kernels carry this form; no aux_out: the derivative is not stored)
    return ops.gemm_ln(h, _w(out.dense.weight), out.LayerNorm.weight, out.LayerNorm.bias, out.LayerNorm.variance_epsilon,
                       epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=out.dense.bias, residual=a)[1]


def layer_infer_full(layer, x, allow, batch):
    """eval-mode layer forward over all rows that also returns what the decoding cache keeps: (y, qkv [B*N,3D], ctx [B*N,D], lse2)"""
    att = layer.attention.self
    wqkv, bqkv, _, _ = _fused_qkv(att)
    qkv = ops.gemm(x, wqkv, epilogue=capi.EPI_BIAS, bias=bqkv)
    ctx, lse2, _ = ops.attn_fwd(qkv, allow, batch, att.num_attention_heads, 1.0 / math.sqrt(att.attention_head_size))
    return _layer_tail(layer, ctx, x), qkv, ctx, lse2


def layer_infer_decode(layer, x_dec, cache, allow, batch, n, n_dec):
    """re-run one layer for the decoder rows only (x_dec [B*n_dec, D]) against the cached keys/values of the other rows.
    Under the prefix-LM mask (sa_m4c.py:834-844) no encoder row sees a decoder key, so every non-decoder row of every layer is
    identical in all 12 greedy steps of sa_m4c.py:294-302: only B*n_dec of B*N rows are recomputed per step."""
    att = layer.attention.self
    wqkv, bqkv, _, _ = _fused_qkv(att)
    qkv, ctx, lse2 = cache
    d3 = qkv.shape[1]
    qkv_dec = ops.gemm(x_dec, wqkv, epilogue=capi.EPI_BIAS, bias=bqkv)
    qkv.view(batch, n, d3)[:, n - n_dec:] = qkv_dec.view(batch, n_dec, d3)
    ops.attn_fwd_rows(qkv, allow, batch, att.num_attention_heads, 1.0 / math.sqrt(att.attention_head_size), n - n_dec, ctx, lse2)
    ctx_dec = ctx.view(batch, n, -1)[:, n - n_dec:].reshape(batch * n_dec, -1)
    return _layer_tail(layer, ctx_dec, x_dec)


class BertLayer(_FusedLayer):
    """pytorch-transformers BertLayer ('n' layers, sa_m4c.py:718-722,741-743; TextBert layers)"""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None):
        return self._run(hidden_states, self.attention.self._allow_bits(attention_mask), head_mask)


class SpatialBertLayer(_FusedLayer):
    """sam/sa_m4c.py:660-684"""

    def __init__(self, config, use_implicit=False):
        super().__init__()
        self.attention = SpatialBertAttention(config, use_implicit)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, spatial_adj_matrix, head_mask=None):
        return self._run(hidden_states, self.attention.self._allow_bits(attention_mask, spatial_adj_matrix), head_mask)


class BertEncoder(_HipModule):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask, head_mask=None):
        allow = as_allow(attention_mask)
        all_att = ()
        for i, layer in enumerate(self.layer):
            outs = layer(hidden_states, allow, None if head_mask is None else head_mask[i])
            hidden_states = outs[0]
            all_att += outs[1:]
        return (hidden_states,) + ((all_att,) if all_att else ())


class BertEmbeddings(_HipModule):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout_p = config.hidden_dropout_prob

    def forward(self, input_ids):
        self._ready()
        n = input_ids.size(1)
        # positions are 0..n-1 and the token type is 0 for every token: plain slices (their backward is a batch reduction, not a
        # scatter); the word rows are gathered from the bf16 shadow table and their gradient is scattered by sam_embedding_bwd
        b = input_ids.size(0)
        pad = self.word_embeddings.padding_idx
        y = EmbedLayerNormFn.apply(self.LayerNorm.weight, input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                                   self.token_type_embeddings.weight, None, b * n, n, self.LayerNorm, -1 if pad is None else pad)
        return dropout(y.view(b, n, -1), self.dropout_p, self.training)


def _bert_init_weights(module, initializer_range):
    """BertPreTrainedModel.init_weights(): N(0, range) Linear/Embedding weights, LN = 1/0, Linear bias 0"""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=initializer_range)
        elif isinstance(m, BertLayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()


class TextBert(_HipModule):
    """sam/sa_m4c.py:374-396"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        _bert_init_weights(self, config.initializer_range)

    def forward(self, batch_dict):
        self._ready()
        x = self.embeddings(batch_dict["question_indices"])
        m8 = batch_dict.get("_sam_masks_u8")          # (key_valid, question, ocr) packed once per forward by SAM4C.forward
        q8 = m8[1] if m8 is not None else batch_dict["question_mask"].to(torch.uint8).contiguous()
        allow = AllowBits(ops.mask_bits_prefix_lm(q8, 0))
        return self.encoder(x, allow, head_mask=[None] * self.config.num_hidden_layers)[0]


MATRIX_TYPE_MAP = {"none": "1", "share3": "3", "share5": "5", "share7": "7", "share9": "9"}


class BertSpatialEncoder(_HipModule):
    """sam/sa_m4c.py:687-770"""

    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer_type_list = list(config.layer_type_list)
        cnt = Counter(self.layer_type_list)
        self.num_spatial_layers, self.num_normal_layers, self.num_implicit_layers = cnt["s"], cnt["n"], cnt["i"]
        mix = getattr(config, "mix_list", None)
        self.mix_list = ["none"] * len(self.layer_type_list) if mix is None else list(mix)
        assert len(self.mix_list) == len(self.layer_type_list)
        self.matrix_type_map = dict(MATRIX_TYPE_MAP)
        self.normal_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_normal_layers)])
        self.spatial_layers = nn.ModuleList([SpatialBertLayer(config) for _ in range(self.num_spatial_layers)])
        self.implicit_layers = nn.ModuleList([SpatialBertLayer(config, True) for _ in range(self.num_implicit_layers)])

    def forward(self, hidden_states, attention_mask, batch_dict, head_mask=None):
        allow = as_allow(attention_mask)
        normal, spatial = iter(self.normal_layers), iter(self.spatial_layers)
        all_hidden, all_att = (), ()
        for kind, mix in zip(self.layer_type_list, self.mix_list):
            if self.output_hidden_states:
                all_hidden += (hidden_states,)
            if kind == "n":
                outs = next(normal)(hidden_states, allow)
            elif kind == "s":
                outs = next(spatial)(hidden_states, allow, self._adjacency_for(batch_dict, mix))
            else:
                raise ValueError   # 'i' layers are rejected by the reference as well (sa_m4c.py:751-752)
            hidden_states = outs[0]
            if self.output_attentions:
                all_att += (outs[1],)
        assert next(normal, None) is None and next(spatial, None) is None
        outputs = (hidden_states,)
        if self.output_hidden_states:
            outputs += (all_hidden + (hidden_states,),)
        if self.output_attentions:
            outputs += (all_att,)          # last-layer hidden state, (all hidden states), (all attentions): sa_m4c.py:765-770
        return outputs

    def _adjacency_for(self, batch_dict, mix):
        """relation tensor of spatial context `mix` (sa_m4c.py:746-747)"""
        key = self.matrix_type_map[mix]
        mats = batch_dict["spatial_adj_matrices"]
        if key not in mats:
            raise KeyError("SA-M4C.mix_list asks for %r heads (relation tensor %r) but the batch only carries contexts %s: the dataset-level "
                           "mix_list and the model's mix_list disagree (as in the shipped train-tvqa-eval-tvqa-c5.yml)" % (mix, key, sorted(mats)))
        return mats[key]

    def _layer_plan(self, allow, batch_dict, n_txt):
        """[(layer, allow bits)] in execution order"""
        normal, spatial, plan = iter(self.normal_layers), iter(self.spatial_layers), []
        for kind, mix in zip(self.layer_type_list, self.mix_list):
            if kind == "n":
                plan.append((next(normal), allow.base))
            elif kind == "s":
                layer = next(spatial)
                plan.append((layer, layer.attention.self._allow_bits(allow, self._adjacency_for(batch_dict, mix))))
            else:
                raise ValueError
        return plan

    def infer_full(self, x, allow, batch_dict, batch):
        """eval forward over all rows; returns (hidden [B*N,D], per-layer decoding caches)"""
        caches = []
        for layer, bits in self._layer_plan(allow, batch_dict, None):
            x, qkv, ctx, lse2 = layer_infer_full(layer, x, bits, batch)
            caches.append((qkv, ctx, lse2))
        return x, caches

    def infer_decode(self, x_dec, allow, batch_dict, batch, n, n_dec, caches):
        for (layer, bits), cache in zip(self._layer_plan(allow, batch_dict, None), caches):
            x_dec = layer_infer_decode(layer, x_dec, cache, bits, batch, n, n_dec)
        return x_dec


class PrevPredEmbeddings(_HipModule):
    """sam/sa_m4c.py:900-948 — without materialising the [B, V+n_ocr, D] table (15.5 MB/sample upstream): the two
    sources are gathered separately and selected."""

    def __init__(self, config):
        super().__init__()
        h, eps = config.hidden_size, config.layer_norm_eps
        self.position_embeddings = nn.Embedding(100, h)
        self.token_type_embeddings = nn.Embedding(5, h)
        self.ans_layer_norm = BertLayerNorm(h, eps=eps)
        self.ocr_layer_norm = BertLayerNorm(h, eps=eps)
        self.emb_layer_norm = BertLayerNorm(h, eps=eps)
        self.dropout_p = config.hidden_dropout_prob

    def forward(self, ans_emb, ocr_emb, prev_inds):
        self._ready()
        assert prev_inds.dim() == 2 and prev_inds.dtype == torch.long and ans_emb.dim() == 2
        b, s = prev_inds.shape
        n_ans, n_ocr = ans_emb.size(0), ocr_emb.size(1)
        out = PrevPredFn.apply(self.emb_layer_norm.weight, ans_emb, ocr_emb, prev_inds, self, self.dropout_p if self.training else 0.0)
        return out


class MMT(_HipModule):
    """sam/sa_m4c.py:773-863"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.prev_pred_embeddings = PrevPredEmbeddings(config)
        self.encoder = BertSpatialEncoder(config)
        _bert_init_weights(self, config.initializer_range)

    def forward(self, batch_dict, fixed_ans_emb):
        self._ready()
        ev = batch_dict.pop("_sam_upd_event", None)
        if ev is not None:                                # the MMT's piece of the previous step's update ran on its own stream (Trainer._issue_pending_update)
            torch.cuda.current_stream().wait_event(ev)
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, batch_dict["ocr_mmt_in"], batch_dict["train_prev_inds"])
        ev = batch_dict.pop("_sam_tb_event", None)
        if ev is not None:                                # TextBert ran on a side stream (SAM4C.forward): join it here, as late as possible
            torch.cuda.current_stream().wait_event(ev)
            batch_dict["text_bert_emb"].record_stream(torch.cuda.current_stream())
        x = cat_rows([batch_dict["text_bert_emb"], batch_dict["obj_mmt_in"], batch_dict["ocr_mmt_in"], dec_emb])
        n_txt = batch_dict["question_mask"].size(-1)
        n_obj = batch_dict["pad_obj_mask"].size(-1)
        n_ocr = batch_dict["pad_ocr_mask"].size(-1)
        n_dec = dec_emb.size(1)
        m8 = batch_dict.get("_sam_masks_u8")
        if m8 is not None:
            key_valid = m8[0]
        else:
            key_valid = torch.cat([batch_dict["question_mask"], batch_dict["pad_obj_mask"], batch_dict["pad_ocr_mask"]], dim=1).to(device=x.device, dtype=torch.uint8).contiguous()
        allow = AllowBits(ops.mask_bits_prefix_lm(key_valid, n_dec))
        cache = batch_dict.get("_sam_decode_cache")
        if cache is not None and not torch.is_grad_enabled():
            return self._forward_cached(batch_dict, cache, x, allow, n_txt, n_obj, n_ocr, n_dec)
        seq = self.encoder(x, allow, batch_dict, head_mask=[None] * self.config.num_hidden_layers)[0]
        ocr0 = n_txt + n_obj
        return {"mmt_seq_output": seq, "mmt_txt_output": seq[:, :n_txt], "mmt_ocr_output": seq[:, ocr0: ocr0 + n_ocr],
                "mmt_dec_output": seq[:, -n_dec:]}

    def _forward_cached(self, batch_dict, cache, x, allow, n_txt, n_obj, n_ocr, n_dec):
        """greedy decoding with encoder-row caching.  First call: full eval pass that records each layer's
        q|k|v and attention output; later calls: only the n_dec decoder rows go through the layers."""
        b, n, d = x.shape
        if "layers" not in cache:
            seq2d, cache["layers"] = self.encoder.infer_full(x.reshape(b * n, d).contiguous(), allow, batch_dict, b)
            cache["seq"] = seq2d.view(b, n, d)
        else:
            x_dec = x[:, n - n_dec:].reshape(b * n_dec, d).contiguous()
            y_dec = self.encoder.infer_decode(x_dec, allow, batch_dict, b, n, n_dec, cache["layers"])
            cache["seq"][:, n - n_dec:] = y_dec.view(b, n_dec, d)
        seq = cache["seq"]
        ocr0 = n_txt + n_obj
        return {"mmt_seq_output": seq, "mmt_txt_output": seq[:, :n_txt], "mmt_ocr_output": seq[:, ocr0: ocr0 + n_ocr],
                "mmt_dec_output": seq[:, -n_dec:]}


class OcrPtrNet(_HipModule):
    """sam/sa_m4c.py:866-897"""

    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.query_key_size = hidden_size if query_key_size is None else query_key_size
        self.query = nn.Linear(hidden_size, self.query_key_size)
        self.key = nn.Linear(hidden_size, self.query_key_size)

    def forward(self, query_inputs, key_inputs, attention_mask):
        self._ready()
        assert attention_mask.dim() == 2
        squeeze = query_inputs.dim() == 2
        if squeeze:
            query_inputs = query_inputs.unsqueeze(1)
        q = linear(query_inputs.to(BF16), self.query)
        k = linear(key_inputs.to(BF16), self.key)
        mask = attention_mask if (attention_mask.dtype == torch.uint8 and attention_mask.is_cuda and attention_mask.is_contiguous()) \
            else attention_mask.to(device=q.device).ne(0).to(torch.uint8).contiguous()
        s = PtrScoresFn.apply(q, k, mask, 1.0 / math.sqrt(self.query_key_size))
        return s.squeeze(1) if squeeze else s


def _pack_features(parts, normalize, n_zero_cols):
    """[B, n, D_i] fp32 feature blocks -> one bf16 [B, n, pad8(sum D_i + n_zero_cols)] GEMM operand: each block L2-normalised along its own
    last dim (F.normalize, sa_m4c.py:219,232-234) and written at its column offset; trailing columns zero.  One launch per block."""
    b, n = parts[0].shape[:2]
    k = sum(p.shape[-1] for p in parts) + n_zero_cols
    k_pad = (k + 7) // 8 * 8
    out = torch.empty((b * n, k_pad), dtype=BF16, device=parts[0].device)
    col = 0
    for i, p in enumerate(parts):
        last = i == len(parts) - 1
        p2 = p.float().flatten(0, 1)                      # [B*n, D] view (a column slice of wider rows stays a strided view: no copy)
        ops.l2norm_pack(p2 if p2.stride(1) == 1 else p2.contiguous(), out, col, normalize, zero_upto=k_pad if last else 0)
        col += p.shape[-1]
    return out.view(b, n, k_pad)


def load_bert_base_into(text_bert, path):
    """copy the embeddings and the first num_hidden_layers encoder layers of a bert-base-uncased state dict into `text_bert`
    (what TextBert.from_pretrained("bert-base-uncased", config=...) does upstream, sa_m4c.py:76-78, minus the download)"""
    if os.path.isdir(path):
        cands = [os.path.join(path, f) for f in ("model.safetensors", "pytorch_model.bin")]
        path = next((c for c in cands if os.path.exists(c)), cands[-1])
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    own = text_bert.state_dict()
    got = {}
    for k, v in sd.items():
        k = k[5:] if k.startswith("bert.") else k
        k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
        if k in own and own[k].shape == v.shape:
            got[k] = v
    missing = [k for k in own if k not in got]
    if missing:
        raise RuntimeError("bert-base weights at %s lack %d TextBert tensors (first: %s)" % (path, len(missing), missing[0]))
    text_bert.load_state_dict(got, strict=True)
    return text_bert


class SAM4C(_HipModule):
    """sam/sa_m4c.py:20-371 (aux heads, beam search and the fc7-finetune image encoder are out of scope: disabled /
    dead upstream)."""

    def __init__(self, mmt_config, text_bert_config, num_answers=None, bos_idx=None):
        super().__init__()
        self.mmt_config, self.text_bert_config = mmt_config, text_bert_config
        self.normalize = mmt_config.normalize
        if getattr(mmt_config, "use_aux_heads", False):
            raise NotImplementedError("use_aux_heads is absent from every shipped config; not implemented")
        self.finetune_modules = []
        h = mmt_config.hidden_size
        self.text_bert = TextBert(text_bert_config)
        for layer in self.text_bert.encoder.layer:
            layer._sam_defer_wgrad = True               # their weight gradients go out together, in one grouped launch (autograd.DeferredWgrads)
        self._mmt_wgrad_pairs = int(os.environ.get("SAM_DEFER_MMT_WGRAD", "2"))
        if getattr(text_bert_config, "text_bert_init_from_bert_base", False):
            # sa_m4c.py:74-85: TextBert starts from bert-base-uncased and trains at lr_scale_text_bert x the base rate (its own optimizer group,
            # BEFORE the MMT group).  The reference downloads the weights; here they come from a local file / directory
            # (config key `text_bert_pretrained_path` or $SAM_BERT_BASE: a state dict with `bert.embeddings.*` / `bert.encoder.layer.N.*`
            # keys, .bin / .pt / .safetensors), or later from a checkpoint -- without one the layers keep their N(0, 0.02) init.
            src = getattr(text_bert_config, "text_bert_pretrained_path", None) or os.environ.get("SAM_BERT_BASE")
            if src:
                load_bert_base_into(self.text_bert, src)
            else:
                logging.getLogger(__name__).warning("text_bert_init_from_bert_base: no local bert-base-uncased weights given "
                                                    "(text_bert_pretrained_path / $SAM_BERT_BASE); TextBert keeps its random init until a checkpoint is loaded")
            self.finetune_modules.append({"module": self.text_bert, "lr_scale": getattr(text_bert_config, "lr_scale_text_bert", 1.0)})
        self.text_bert_out_linear = nn.Identity() if h == 768 else nn.Linear(768, h)
        self.linear_obj_feat_to_mmt_in = nn.Linear(mmt_config.obj_feature_size, h)
        self.linear_obj_bbox_to_mmt_in = nn.Linear(4, h)
        self.obj_feat_layer_norm = BertLayerNorm(h)
        self.obj_bbox_layer_norm = BertLayerNorm(h)
        self.obj_drop_p = mmt_config.obj_drop
        self.linear_ocr_feat_to_mmt_in = nn.Linear(mmt_config.ocr_feature_size, h)
        self.linear_ocr_bbox_to_mmt_in = nn.Linear(4, h)
        self.ocr_feat_layer_norm = BertLayerNorm(h)
        self.ocr_bbox_layer_norm = BertLayerNorm(h)
        self.ocr_drop_p = mmt_config.ocr_drop
        self.mmt = MMT(mmt_config)
        if self._mmt_wgrad_pairs >= 2:
            # weight gradients of the MMT layers in groups of `_mmt_wgrad_pairs` layers per launch (backward order); a last incomplete group is flushed by the Trainer's join
            enc = self.mmt.encoder
            for layer in list(getattr(enc, "normal_layers", [])) + list(getattr(enc, "spatial_layers", [])):
                layer._sam_defer_wgrad = True
                layer._sam_defer_flush_at = self._mmt_wgrad_pairs
            if len(enc.layer_type_list) % self._mmt_wgrad_pairs == 0:
                # the group that the FIRST layer (last in the backward) closes runs beside the tail of the backward, not in front of it (DeferredWgrads.flush)
                first = (enc.normal_layers if enc.layer_type_list[0] == "n" else enc.spatial_layers)[0]
                first._sam_wgrad_late = True
        self.finetune_modules.append({"module": self.mmt, "lr_scale": mmt_config.lr_scale_mmt})
        self.ocr_ptr_net = OcrPtrNet(hidden_size=h, query_key_size=mmt_config.ptr_query_size)
        n_out = num_answers if num_answers is not None else len(registry.answer_vocab)
        self.bos_idx = bos_idx if bos_idx is not None else registry.BOS_IDX
        self.classifier = nn.Linear(h, n_out)
        self.overlap_text_bert = __import__("os").environ.get("SAM_NO_TB_OVERLAP") != "1"
        self._side_stream = None
        self.decode_cache = True      # eval-mode greedy loop re-runs only the decoder rows (set False for the reference's 12 full passes)

    def _sam_param_rank(self, name):
        """address order of the parameters inside their optimizer group (params.FlatParams): ascending address = LATER gradient, so that
        the data-parallel buckets, walked from the end of the buffer, can leave in backward order.  word-embedding table (row-sparse
        exchange) | object / OCR encoders (their backward runs last) | TextBert | pointer net, classifier | MMT."""
        if name.startswith("text_bert.embeddings.word_embeddings"):
            return 0
        if name.startswith(("linear_obj", "obj_")):
            return 1
        if name.startswith(("linear_ocr", "ocr_feat", "ocr_bbox")):
            return 2
        if name.startswith("text_bert"):
            return 3
        if name.startswith("ocr_ptr_net"):
            return 4
        if name.startswith("classifier"):
            return 5
        return 6

    def _input_encoder(self, feat, bbox, lin_a, ln_a, lin_b, ln_b, p_drop, n):
        """dropout(LN(feat W^T + b) + LN(bbox W^T + b)) -> [B, n, D]: one autograd node, HIP kernels only (autograd.InputEncoderFn)"""
        b = feat.shape[0]
        out = InputEncoderFn.apply(lin_a.weight, feat.flatten(0, 1), bbox.flatten(0, 1), lin_a, ln_a, lin_b, ln_b,
                                   float(p_drop) if self.training else 0.0, lin_a)
        return out.view(b, n, -1)

    def _forward_obj_encoding(self, bd):
        feat = _pack_features([bd["pad_obj_features"]], self.normalize, 0)
        bbox = bd["pad_obj_bboxes"]                       # [B, n, 5]: the first four columns are read in place by the fused encoder tail
        x = self._input_encoder(feat, bbox, self.linear_obj_feat_to_mmt_in, self.obj_feat_layer_norm, self.linear_obj_bbox_to_mmt_in, self.obj_bbox_layer_norm,
                                self.obj_drop_p, feat.shape[1])
        bd["obj_mmt_in"] = GradBarrierFn.apply(x, "obj") if self.training and torch.is_grad_enabled() else x

    def _forward_ocr_encoding(self, bd):
        ft, ph, fc = bd["ocr_fasttext"], bd["ocr_phoc"], bd["pad_ocr_features"]
        assert ft.size(-1) == 300 and ph.size(-1) == 604
        # FastText | PHOC | FRCN | 50 legacy all-zero order columns (sa_m4c.py:242), normalised and packed into the K-padded GEMM operand
        feat = _pack_features([ft, ph, fc] if self.mmt_config.use_phoc_fasttext else [fc], self.normalize, 50)
        bbox = bd["pad_ocr_bboxes"]
        x = self._input_encoder(feat, bbox, self.linear_ocr_feat_to_mmt_in, self.ocr_feat_layer_norm, self.linear_ocr_bbox_to_mmt_in, self.ocr_bbox_layer_norm,
                                self.ocr_drop_p, feat.shape[1])
        bd["ocr_mmt_in"] = GradBarrierFn.apply(x, "ocr") if self.training and torch.is_grad_enabled() else x

    def _forward_text_bert(self, bd):
        t = self.text_bert(bd)
        t = t if isinstance(self.text_bert_out_linear, nn.Identity) else linear(t, self.text_bert_out_linear)
        bd["text_bert_emb"] = GradBarrierFn.apply(t, "txt") if self.training and torch.is_grad_enabled() else t

    def _forward_mmt(self, bd):
        cache = bd.get("_sam_decode_cache")
        if cache is not None and "text_bert_emb" in cache:
            bd["text_bert_emb"] = cache["text_bert_emb"]
        elif not bd.pop("_sam_tb_ready", False):       # (already computed on the side stream by forward())
            self._forward_text_bert(bd)
            if cache is not None:
                cache["text_bert_emb"] = bd["text_bert_emb"]      # question encoding does not depend on the decoding step
        bd.update(self.mmt(bd, fixed_ans_emb=self.classifier.weight))

    def _forward_output(self, bd):
        dec, ocr_rows = bd["mmt_dec_output"], bd["mmt_ocr_output"]
        if self.training and torch.is_grad_enabled():
            seq = bd["mmt_seq_output"]
            ocr_rows, dec = SeqRowsFn.apply(seq, seq.shape[1] - dec.shape[1] - ocr_rows.shape[1], ocr_rows.shape[1], dec.shape[1])
        m8 = bd.get("_sam_masks_u8")
        ocr_mask = m8[2] if m8 is not None else bd["pad_ocr_mask"]
        if self.training and self.overlap_text_bert and torch.is_grad_enabled():
            # the two heads are independent chains of small kernels (768 decoder rows): the pointer network runs on the side stream next to
            # the classifier, forward and (autograd replays the streams) backward
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dyn = self.ocr_ptr_net(dec, ocr_rows, ocr_mask)
            bd["fixed_scores"] = linear(dec, self.classifier, out_f32=True)
            main.wait_stream(side)
            dyn.record_stream(main)
            bd["dynamic_ocr_scores"] = dyn
            if bd.get("_sam_want_scores", True):       # (the Trainer's loss reads the two blocks; it asks for no concatenated copy)
                bd["scores"] = torch.cat([bd["fixed_scores"], bd["dynamic_ocr_scores"]], dim=-1)
            return
        bd["fixed_scores"] = linear(dec, self.classifier, out_f32=True)
        bd["dynamic_ocr_scores"] = self.ocr_ptr_net(dec, ocr_rows, ocr_mask)
        if bd.get("_sam_want_scores", True) or not self.training:
            bd["scores"] = torch.cat([bd["fixed_scores"], bd["dynamic_ocr_scores"]], dim=-1)

    def set_beam_size(self, beam_size):
        """sa_m4c.py:53-56"""
        from .decoder import BeamSearch
        self.beam_size = beam_size
        self.bsdecoder = BeamSearch(self.beam_size, bos_idx=self.bos_idx)
        logging.getLogger(__name__).info("Using beam size: %s", self.beam_size)

    def _forward_beam_search(self, batch_dict):
        """sa_m4c.py:304-314 + sam/beam_search.py: the batch is expanded beam_size times, then ONE full pass and n_dec - 1 captured decoding steps
        (decoder.DecodeSession) with sam_beam_step choosing the surviving beams on the device"""
        from .decoder import session_for, shared_beams_enabled
        if self.training:
            raise RuntimeError("beam search runs in eval mode (evaluator.py:137-160 calls model.eval() first); call .eval()")
        if getattr(self, "bsdecoder", None) is None:
            raise RuntimeError("call set_beam_size(k) before forward(..., use_beam_search=True) (sa_m4c.py:53-56, evaluator.py:139)")
        bs = self.bsdecoder
        if shared_beams_enabled() and bs._decode_size > 1:
            # the beams of a sample share its encoder rows: the batch is NOT repeated beam_size times (beam_search.py:31-82), the session keeps one
            # copy per sample (decoder.DecodeSession, shared=True); what the reference's expansion leaves in batch_dict for its callers -- the
            # decoder state, the scores and question_id, beam_size rows per sample -- is produced all the same
            k = bs._decode_size
            bs.completed_ids, bs._batch_size = None, batch_dict["train_prev_inds"].shape[0]
            ses = session_for(self, batch_dict, beam=k, eos_idx=bs._EOS_IDX, shared=True)
            ses.run(batch_dict)
            if "question_id" in batch_dict:
                batch_dict["question_id"] = batch_dict["question_id"].repeat_interleave(k, dim=0)
            return batch_dict
        batch_dict = bs.init_batch(batch_dict)
        batch_dict.pop("_beam_done", None)
        ses = session_for(self, batch_dict, beam=bs._decode_size, eos_idx=bs._EOS_IDX)
        ses.run(batch_dict)
        return batch_dict

    def forward(self, batch_dict, use_beam_search=False):
        self._ready()
        if use_beam_search:
            bd = self._forward_beam_search(batch_dict)
            if bd is not batch_dict:
                batch_dict.update(bd)
            res = {"textvqa_scores": batch_dict["scores"], "complete_seqs": batch_dict["complete_seqs"].squeeze(), "topkscores": batch_dict["topkscores"].squeeze()}
            if "question_id" in batch_dict:
                res["question_id"] = batch_dict["question_id"].squeeze()
            return res           # sa_m4c.py:192-202
        if not self.training and self.decode_cache and not torch.is_grad_enabled() and os.environ.get("SAM_DECODE_SESSION", "1") != "0":
            # greedy decoding (sa_m4c.py:285-302) as one full pass + n_dec - 1 captured decoding steps over static buffers (decoder.DecodeSession);
            # the eager loop in _forward_impl is the same arithmetic launch by launch (SAM_DECODE_SESSION=0, decode_cache=False, or with autograd on)
            from .decoder import session_for
            session_for(self, batch_dict).run(batch_dict)
            return {"textvqa_scores": batch_dict["scores"]}
        if all(k in batch_dict for k in ("question_mask", "pad_obj_mask", "pad_ocr_mask")) and batch_dict["question_mask"].is_cuda:
            batch_dict["_sam_masks_u8"] = ops.pack_masks(batch_dict["question_mask"], batch_dict["pad_obj_mask"], batch_dict["pad_ocr_mask"])
        try:
            return self._forward_impl(batch_dict)
        finally:
            batch_dict.pop("_sam_masks_u8", None)

    def _forward_impl(self, batch_dict):
        if self.training and self.overlap_text_bert:
            # TextBert (20 tokens/sample: 240-block grids, latency-bound) runs on a side stream underneath the object / OCR encoders; autograd
            # replays each node's backward on its forward stream, so its backward overlaps theirs too
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._forward_text_bert(batch_dict)
            batch_dict["_sam_tb_ready"] = True
            ev = torch.cuda.Event()
            ev.record(side)
            batch_dict["_sam_tb_event"] = ev             # joined by MMT.forward right before it concatenates the four token groups
            self._forward_obj_encoding(batch_dict)
            self._forward_ocr_encoding(batch_dict)
        else:
            self._forward_obj_encoding(batch_dict)
            self._forward_ocr_encoding(batch_dict)
        if self.training:
            self._forward_mmt(batch_dict)
            self._forward_output(batch_dict)
        else:   # greedy decoding, sa_m4c.py:285-302
            steps = batch_dict["train_prev_inds"].size(1)
            batch_dict["train_prev_inds"] = torch.zeros_like(batch_dict["train_prev_inds"])
            batch_dict["train_prev_inds"][:, 0] = self.bos_idx
            if self.decode_cache and not torch.is_grad_enabled():
                batch_dict["_sam_decode_cache"] = {}      # encoder rows are step-invariant: computed once, decoder rows 12x
            for _ in range(steps):
                self._forward_mmt(batch_dict)
                self._forward_output(batch_dict)
                batch_dict["train_prev_inds"][:, 1:] = batch_dict["scores"].argmax(dim=-1)[:, :-1]
            batch_dict.pop("_sam_decode_cache", None)
        return {"textvqa_scores": batch_dict.get("scores")}

    def get_optimizer_parameters(self, base_lr):
        """sa_m4c.py:349-371"""
        groups, special = [], set()
        for m in self.finetune_modules:
            ps = list(m["module"].parameters())
            groups.append({"params": ps, "lr": base_lr * m["lr_scale"]})
            special.update(ps)
        groups.insert(0, {"params": [p for p in self.parameters() if p not in special]})
        return groups
