"""fp32 PyTorch-CPU restatement of the SA-M4C hot path (SURVEY.md §8a rows a-1 … a-18).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Every class keeps the reference's attribute
names so that a reference ``state_dict`` loads unchanged; every function cites the
reference lines (relative to /root/reference) whose arithmetic it restates.

Pinning status:
  * pinned by golden vectors generated from the reference's OWN code (tests/golden/*.npz,
    generator tests/golden/make_golden.py): SpatialBertSelfAttention, SpatialBertAttention /
    SpatialBertLayer wiring, BertSpatialEncoder, MMT, PrevPredEmbeddings, OcrPtrNet,
    BertLayerNorm (fallback class), gelu, SAM4C forward, spatial graph (oracle/spatial_graph.py).
  * "parity unpinned" against the pinned wheel: the arithmetic of BertSelfOutput,
    BertIntermediate, BertOutput, BertLayer/BertSelfAttention, BertEmbeddings, BertEncoder lives
    in the third-party dependency pytorch-transformers (requirements.txt:1, ==1.0.0; effective
    >=1.1.0 because sa_m4c.py:380,780 call init_weights() with no argument), which is absent
    from /root/reference and from this image.  They are restated here from that library's
    public API and cross-checked in make_golden.py against the same-named classes of the
    installed `transformers` (its direct descendant) where the signatures still agree.
"""
import math
from bisect import bisect
from collections import Counter

import torch
import torch.nn.functional as F
from torch import nn

NEG = -10000.0  # the reference's additive "masked" value (sa_m4c.py:387,551,844,879)


class BertConfig:
    """Attribute bag with pytorch-transformers' BertConfig defaults; from_dict copies EVERY key
    (train.py:92-93 relies on that to smuggle max_seq_length, layer_type_list, mix_list, ...)."""

    DEFAULTS = dict(
        vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
        initializer_range=0.02, layer_norm_eps=1e-12, output_attentions=False,
        output_hidden_states=False,
    )

    def __init__(self, **kw):
        for k, v in self.DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        return cls(**dict(d))


def gelu(x):
    """erf-GELU, sa_m4c.py:985-991."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class BertLayerNorm(nn.Module):
    """TF-style LN, eps inside the sqrt, biased variance — sa_m4c.py:1016-1028."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        mu = x.mean(-1, keepdim=True)
        var = (x - mu).pow(2).mean(-1, keepdim=True)
        return self.weight * ((x - mu) / torch.sqrt(var + self.variance_epsilon)) + self.bias


# --------------------------------------------------------------------------------------
# third-party (pytorch-transformers) blocks, restated from the public API
# --------------------------------------------------------------------------------------
class BertSelfOutput(nn.Module):
    """dense -> dropout -> +residual -> LayerNorm (used via sa_m4c.py:617-619,653)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = BertLayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class BertIntermediate(nn.Module):
    """dense -> erf-GELU (used via sa_m4c.py:663-667,678)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)

    def forward(self, hidden_states):
        return gelu(self.dense(hidden_states))


class BertOutput(nn.Module):
    """dense -> dropout -> +residual -> LayerNorm (used via sa_m4c.py:668,680)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = BertLayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


def _split_heads(x, n_heads):
    b, n, d = x.shape
    return x.view(b, n, n_heads, d // n_heads).permute(0, 2, 1, 3)


def _merge_heads(x):
    b, h, n, d = x.shape
    return x.permute(0, 2, 1, 3).contiguous().view(b, n, h * d)


class BertSelfAttention(nn.Module):
    """Plain additive-mask MHA of the 'n' layers and TextBert (third-party; sa_m4c.py:718-722)."""

    def __init__(self, cfg):
        super().__init__()
        self.num_attention_heads = cfg.num_attention_heads
        self.attention_head_size = cfg.hidden_size // cfg.num_attention_heads
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.value = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.dropout = nn.Dropout(cfg.attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask, head_mask=None):
        h = self.num_attention_heads
        q = _split_heads(self.query(hidden_states), h)
        k = _split_heads(self.key(hidden_states), h)
        v = _split_heads(self.value(hidden_states), h)
        s = q @ k.transpose(-1, -2) / math.sqrt(self.attention_head_size) + attention_mask
        p = self.dropout(torch.softmax(s, dim=-1))
        if head_mask is not None:
            p = p * head_mask
        return (_merge_heads(p @ v),)


class BertAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = BertSelfAttention(cfg)
        self.output = BertSelfOutput(cfg)

    def forward(self, x, attention_mask, head_mask=None):
        return (self.output(self.self(x, attention_mask, head_mask)[0], x),)


class BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = BertAttention(cfg)
        self.intermediate = BertIntermediate(cfg)
        self.output = BertOutput(cfg)

    def forward(self, x, attention_mask, head_mask=None):
        a = self.attention(x, attention_mask, head_mask)[0]
        return (self.output(self.intermediate(a), a),)


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])

    def forward(self, x, attention_mask, head_mask=None):
        for i, layer in enumerate(self.layer):
            x = layer(x, attention_mask, None if head_mask is None else head_mask[i])[0]
        return (x,)


class BertEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = BertLayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, input_ids):
        n = input_ids.size(1)
        pos = torch.arange(n, dtype=torch.long, device=input_ids.device).unsqueeze(0).expand_as(input_ids)
        e = (self.word_embeddings(input_ids) + self.position_embeddings(pos)
             + self.token_type_embeddings(torch.zeros_like(input_ids)))
        return self.dropout(self.LayerNorm(e))


def bert_init_weights(module, initializer_range=0.02):
    """BertPreTrainedModel.init_weights(): N(0, range) for Linear/Embedding weights, LN = 1/0,
    Linear bias 0 (sa_m4c.py:380,780 call it)."""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=initializer_range)
        elif isinstance(m, BertLayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()


class TextBert(nn.Module):
    """sa_m4c.py:374-396."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = BertEmbeddings(cfg)
        self.encoder = BertEncoder(cfg)
        bert_init_weights(self, cfg.initializer_range)

    def forward(self, batch_dict):
        x = self.embeddings(batch_dict["question_indices"])
        ext = (1.0 - batch_dict["question_mask"].unsqueeze(1).unsqueeze(2)) * NEG
        return self.encoder(x, ext, head_mask=[None] * self.config.num_hidden_layers)[0]


# --------------------------------------------------------------------------------------
# the reference's own hot-path classes
# --------------------------------------------------------------------------------------
_QUADRANT_SLICES = {  # quadrant id -> (row region, col region); sa_m4c.py:505-549
    1: ("txt", "txt"), 2: ("txt", "oo"), 4: ("oo", "txt"), 7: ("dec", "txt"), 8: ("dec", "oo"), 9: ("dec", "dec"),
}


class SpatialBertSelfAttention(nn.Module):
    """Relation-masked MHA, one head per spatial relation — sa_m4c.py:399-610.

    ``faithful=True`` keeps the reference's per-call fp32 mask materialisation *and* its debug
    ``torch.unique`` (sa_m4c.py:569), i.e. what the reference really costs on a CPU;
    ``faithful=False`` computes the same numbers without the debug sort.
    """

    faithful = False

    def __init__(self, cfg, use_implicit=False):
        super().__init__()
        assert hasattr(cfg, "num_spatial_relations")
        self.num_attention_heads = cfg.num_spatial_relations
        self.num_spatial_relations = cfg.num_spatial_relations
        if hasattr(cfg, "num_implicit_relations") and use_implicit:
            self.num_attention_heads += cfg.num_implicit_relations
            self.num_implicit_relations = cfg.num_implicit_relations
        if cfg.hidden_size % self.num_attention_heads != 0:
            raise ValueError("hidden size %d not a multiple of the number of heads %d"
                             % (cfg.hidden_size, self.num_attention_heads))
        self.output_attentions = cfg.output_attentions
        self.max_seq_len = cfg.max_seq_length
        self.mask_quadrants = cfg.attention_mask_quadrants
        self.max_decoding_steps = cfg.num_decoding_steps
        self.attention_head_size = cfg.hidden_size // self.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(cfg.hidden_size, self.all_head_size)
        self.key = nn.Linear(cfg.hidden_size, self.all_head_size)
        self.value = nn.Linear(cfg.hidden_size, self.all_head_size)
        p = 0.0 if getattr(cfg, "no_drop", False) else cfg.attention_probs_dropout_prob
        self.dropout = nn.Dropout(p)
        self.use_bias = bool(getattr(cfg, "use_bias", False))
        if self.use_bias:
            self.biases = nn.Embedding(1, cfg.hidden_size)

    def build_spatial_mask(self, attention_mask, spatial_adj_matrix, n):
        """Additive [B,H,N,N] mask from the multi-hot relation tensor — sa_m4c.py:470-552."""
        b, n_oo, _, n_rel = spatial_adj_matrix.shape
        t = self.max_seq_len
        region = {"txt": slice(0, t), "oo": slice(t, t + n_oo), "dec": slice(t + n_oo, n)}
        m = attention_mask.new_ones((b, n, n, n_rel))
        m[:, region["oo"], region["oo"], :] = spatial_adj_matrix
        if self.num_attention_heads != self.num_spatial_relations:
            m = torch.cat([m, attention_mask.new_ones((b, n, n, self.num_implicit_relations))], dim=-1)
        for quad in self.mask_quadrants:
            if quad not in _QUADRANT_SLICES:
                raise ValueError
            rows, cols = _QUADRANT_SLICES[quad]
            m[:, region[rows], region[cols], : self.num_spatial_relations] = 0
        return ((1.0 - m) * NEG).permute(0, 3, 1, 2)

    def forward(self, hidden_states, attention_mask, spatial_adj_matrix, head_mask=None):
        n = hidden_states.size(1)
        spatial_mask = self.build_spatial_mask(attention_mask, spatial_adj_matrix, n)
        h = self.num_attention_heads
        q = _split_heads(self.query(hidden_states), h)
        k = _split_heads(self.key(hidden_states), h)
        v = _split_heads(self.value(hidden_states), h)
        scores = q @ k.transpose(-1, -2) / math.sqrt(self.attention_head_size)      # :563-564
        combined = torch.min(attention_mask, spatial_mask)                           # :568
        if self.faithful:
            assert len(torch.unique(combined)) <= 2                                  # :569 (debug sort)
        row_alive = ((combined.max(dim=-1)[0] - NEG) / -NEG).unsqueeze(-1)          # :574-575
        probs = torch.softmax(scores + combined, dim=-1) * row_alive                 # :578-584
        probs = self.dropout(probs)                                                  # :588
        if head_mask is not None:
            probs = probs * head_mask
        ctx = _merge_heads(probs @ v)                                                # :594-598
        if self.use_bias:
            ctx = ctx + self.biases(ctx.new_zeros(1).long())
        return (ctx, probs) if self.output_attentions else (ctx,)


class SpatialBertAttention(nn.Module):
    """sa_m4c.py:613-657 (prune_heads omitted: references an undefined symbol, never called)."""

    def __init__(self, cfg, use_implicit=False):
        super().__init__()
        self.self = SpatialBertSelfAttention(cfg, use_implicit)
        self.output = BertSelfOutput(cfg)

    def forward(self, input_tensor, attention_mask, spatial_adj_matrix, head_mask=None):
        so = self.self(input_tensor, attention_mask, spatial_adj_matrix, head_mask)
        return (self.output(so[0], input_tensor),) + so[1:]


class SpatialBertLayer(nn.Module):
    """sa_m4c.py:660-684."""

    def __init__(self, cfg, use_implicit=False):
        super().__init__()
        self.attention = SpatialBertAttention(cfg, use_implicit)
        self.intermediate = BertIntermediate(cfg)
        self.output = BertOutput(cfg)

    def forward(self, hidden_states, attention_mask, spatial_adj_matrix, head_mask=None):
        ao = self.attention(hidden_states, attention_mask, spatial_adj_matrix, head_mask)
        return (self.output(self.intermediate(ao[0]), ao[0]),) + ao[1:]


MATRIX_TYPE_MAP = {"none": "1", "share3": "3", "share5": "5", "share7": "7", "share9": "9"}  # :710-716


class BertSpatialEncoder(nn.Module):
    """'n'/'s' layer dispatcher — sa_m4c.py:687-770."""

    def __init__(self, cfg):
        super().__init__()
        self.output_attentions = cfg.output_attentions
        self.output_hidden_states = cfg.output_hidden_states
        self.layer_type_list = list(cfg.layer_type_list)
        cnt = Counter(self.layer_type_list)
        self.num_spatial_layers, self.num_normal_layers, self.num_implicit_layers = cnt["s"], cnt["n"], cnt["i"]
        mix = getattr(cfg, "mix_list", None)
        self.mix_list = ["none"] * len(self.layer_type_list) if mix is None else list(mix)
        assert len(self.mix_list) == len(self.layer_type_list)
        self.matrix_type_map = dict(MATRIX_TYPE_MAP)
        self.normal_layers = nn.ModuleList([BertLayer(cfg) for _ in range(self.num_normal_layers)])
        self.spatial_layers = nn.ModuleList([SpatialBertLayer(cfg) for _ in range(self.num_spatial_layers)])
        self.implicit_layers = nn.ModuleList([SpatialBertLayer(cfg, True) for _ in range(self.num_implicit_layers)])

    def forward(self, hidden_states, attention_mask, batch_dict, head_mask=None):
        normal, spatial = iter(self.normal_layers), iter(self.spatial_layers)
        all_hidden, all_att = (), ()
        for kind, mix in zip(self.layer_type_list, self.mix_list):
            if self.output_hidden_states:
                all_hidden += (hidden_states,)
            if kind == "n":
                out = next(normal)(hidden_states, attention_mask)
            elif kind == "s":
                adj = batch_dict["spatial_adj_matrices"][self.matrix_type_map[mix]]
                out = next(spatial)(hidden_states, attention_mask, adj)
            else:
                raise ValueError  # 'i' layers are unreachable in the reference too (:751-752)
            hidden_states = out[0]
            if self.output_attentions:
                all_att += (out[1],)
        assert next(normal, None) is None and next(spatial, None) is None
        outputs = (hidden_states,)
        if self.output_hidden_states:
            outputs += (all_hidden + (hidden_states,),)
        if self.output_attentions:
            outputs += (all_att,)
        return outputs


def causal_mask(n, device=None):
    """lower-triangular ones — sa_m4c.py:960-967."""
    return torch.tril(torch.ones(n, n, device=device))


def batch_gather(x, inds):
    """x[b, inds[b, s], :] — sa_m4c.py:970-982."""
    b, length, dim = x.shape
    flat = x.reshape(b * length, dim)
    off = (torch.arange(b, device=inds.device) * length).unsqueeze(-1)
    return F.embedding(off + inds, flat)


class PrevPredEmbeddings(nn.Module):
    """sa_m4c.py:900-948."""

    def __init__(self, cfg):
        super().__init__()
        h, eps = cfg.hidden_size, cfg.layer_norm_eps
        self.position_embeddings = nn.Embedding(100, h)
        self.token_type_embeddings = nn.Embedding(5, h)
        self.ans_layer_norm = BertLayerNorm(h, eps=eps)
        self.ocr_layer_norm = BertLayerNorm(h, eps=eps)
        self.emb_layer_norm = BertLayerNorm(h, eps=eps)
        self.emb_dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, ans_emb, ocr_emb, prev_inds):
        assert prev_inds.dim() == 2 and prev_inds.dtype == torch.long and ans_emb.dim() == 2
        b, s = prev_inds.shape
        n_ans = ans_emb.size(0)
        table = torch.cat([self.ans_layer_norm(ans_emb).unsqueeze(0).expand(b, -1, -1),
                           self.ocr_layer_norm(ocr_emb)], dim=1)
        raw = batch_gather(table, prev_inds)
        pos = torch.arange(s, dtype=torch.long, device=ocr_emb.device).unsqueeze(0).expand(b, s)
        typ = prev_inds.ge(n_ans).long()
        emb = self.emb_dropout(self.emb_layer_norm(self.position_embeddings(pos) + self.token_type_embeddings(typ)))
        return raw + emb


class MMT(nn.Module):
    """sa_m4c.py:773-863."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.prev_pred_embeddings = PrevPredEmbeddings(cfg)
        self.encoder = BertSpatialEncoder(cfg)
        bert_init_weights(self, cfg.initializer_range)

    @staticmethod
    def extended_attention_mask(question_mask, obj_mask, ocr_mask, n_dec):
        """[B,1,N,N] additive prefix-LM mask — sa_m4c.py:805-844."""
        b = question_mask.size(0)
        dec = torch.zeros(b, n_dec, dtype=torch.long, device=question_mask.device)
        m = torch.cat([question_mask, obj_mask, ocr_mask, dec], dim=1)
        n = m.size(1)
        ext = m.unsqueeze(1).unsqueeze(2).repeat(1, 1, n, 1)
        ext[:, :, -n_dec:, -n_dec:] = causal_mask(n_dec, m.device)
        return (1.0 - ext) * NEG

    def forward(self, batch_dict, fixed_ans_emb):
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, batch_dict["ocr_mmt_in"], batch_dict["train_prev_inds"])
        x = torch.cat([batch_dict["text_bert_emb"], batch_dict["obj_mmt_in"], batch_dict["ocr_mmt_in"], dec_emb], dim=1)
        n_txt = batch_dict["question_mask"].size(-1)
        n_obj = batch_dict["pad_obj_mask"].size(-1)
        n_ocr = batch_dict["pad_ocr_mask"].size(-1)
        n_dec = dec_emb.size(1)
        ext = self.extended_attention_mask(batch_dict["question_mask"], batch_dict["pad_obj_mask"],
                                           batch_dict["pad_ocr_mask"], n_dec)
        seq = self.encoder(x, ext, batch_dict, head_mask=[None] * self.config.num_hidden_layers)[0]
        ocr0 = n_txt + n_obj
        return {
            "mmt_seq_output": seq,
            "mmt_txt_output": seq[:, :n_txt],
            "mmt_ocr_output": seq[:, ocr0: ocr0 + n_ocr],
            "mmt_dec_output": seq[:, -n_dec:],
        }


class OcrPtrNet(nn.Module):
    """Bilinear pointer scores over OCR tokens — sa_m4c.py:866-897."""

    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.query_key_size = hidden_size if query_key_size is None else query_key_size
        self.query = nn.Linear(hidden_size, self.query_key_size)
        self.key = nn.Linear(hidden_size, self.query_key_size)

    def forward(self, query_inputs, key_inputs, attention_mask):
        ext = ((1.0 - attention_mask) * NEG).unsqueeze(1)
        assert ext.dim() == 3
        q = self.query(query_inputs)
        squeeze = q.dim() == 2
        if squeeze:
            q = q.unsqueeze(1)
        s = q @ self.key(key_inputs).transpose(-1, -2) / math.sqrt(self.query_key_size) + ext
        return s.squeeze(1) if squeeze else s


class SAM4C(nn.Module):
    """Model shell — sa_m4c.py:20-371 (aux heads, beam search, fc7 finetune branch out of scope)."""

    def __init__(self, mmt_config, text_bert_config, num_answers=5000, bos_idx=1):
        super().__init__()
        self.mmt_config, self.text_bert_config = mmt_config, text_bert_config
        self.normalize = mmt_config.normalize
        self.bos_idx = bos_idx
        self.finetune_modules = []
        h = mmt_config.hidden_size
        assert not text_bert_config.text_bert_init_from_bert_base, "no network: random-init TextBert only"
        self.text_bert = TextBert(text_bert_config)
        self.text_bert_out_linear = nn.Identity() if h == 768 else nn.Linear(768, h)
        self.linear_obj_feat_to_mmt_in = nn.Linear(mmt_config.obj_feature_size, h)
        self.linear_obj_bbox_to_mmt_in = nn.Linear(4, h)
        self.obj_feat_layer_norm = BertLayerNorm(h)
        self.obj_bbox_layer_norm = BertLayerNorm(h)
        self.obj_drop = nn.Dropout(mmt_config.obj_drop)
        self.linear_ocr_feat_to_mmt_in = nn.Linear(mmt_config.ocr_feature_size, h)
        self.linear_ocr_bbox_to_mmt_in = nn.Linear(4, h)
        self.ocr_feat_layer_norm = BertLayerNorm(h)
        self.ocr_bbox_layer_norm = BertLayerNorm(h)
        self.ocr_drop = nn.Dropout(mmt_config.ocr_drop)
        self.mmt = MMT(mmt_config)
        self.finetune_modules.append({"module": self.mmt, "lr_scale": mmt_config.lr_scale_mmt})
        self.ocr_ptr_net = OcrPtrNet(hidden_size=h, query_key_size=mmt_config.ptr_query_size)
        self.classifier = nn.Linear(h, num_answers)

    # sa_m4c.py:204-219
    def _forward_obj_encoding(self, bd):
        feat = bd["pad_obj_features"]
        if self.normalize:
            feat = F.normalize(feat, dim=-1)
        x = (self.obj_feat_layer_norm(self.linear_obj_feat_to_mmt_in(feat))
             + self.obj_bbox_layer_norm(self.linear_obj_bbox_to_mmt_in(bd["pad_obj_bboxes"][:, :, :-1])))
        bd["obj_mmt_in"] = self.obj_drop(x)

    # sa_m4c.py:221-257
    def _forward_ocr_encoding(self, bd):
        ft, ph, fc = bd["ocr_fasttext"], bd["ocr_phoc"], bd["pad_ocr_features"]
        if self.normalize:
            ft, ph, fc = F.normalize(ft, dim=-1), F.normalize(ph, dim=-1), F.normalize(fc, dim=-1)
        order = fc.new_zeros((ph.size(0), ph.size(1), 50))
        parts = [ft, ph, fc, order] if self.mmt_config.use_phoc_fasttext else [fc, order]
        feat = torch.cat(parts, dim=-1)
        x = (self.ocr_feat_layer_norm(self.linear_ocr_feat_to_mmt_in(feat))
             + self.ocr_bbox_layer_norm(self.linear_ocr_bbox_to_mmt_in(bd["pad_ocr_bboxes"][:, :, :-1])))
        bd["ocr_mmt_in"] = self.ocr_drop(x)

    # sa_m4c.py:259-278
    def _forward_mmt(self, bd):
        bd["text_bert_emb"] = self.text_bert_out_linear(self.text_bert(bd))
        bd.update(self.mmt(bd, fixed_ans_emb=self.classifier.weight))

    def _forward_output(self, bd):
        fixed = self.classifier(bd["mmt_dec_output"])
        dyn = self.ocr_ptr_net(bd["mmt_dec_output"], bd["mmt_ocr_output"], bd["pad_ocr_mask"])
        bd["scores"] = torch.cat([fixed, dyn], dim=-1)

    # sa_m4c.py:179-202, 280-302
    def forward(self, batch_dict, use_beam_search=False):
        assert not use_beam_search, "beam search is out of scope (disabled upstream, train.py:222-225)"
        self._forward_obj_encoding(batch_dict)
        self._forward_ocr_encoding(batch_dict)
        if self.training:
            self._forward_mmt(batch_dict)
            self._forward_output(batch_dict)
        else:
            steps = batch_dict["train_prev_inds"].size(1)
            batch_dict["train_prev_inds"] = torch.zeros_like(batch_dict["train_prev_inds"])
            batch_dict["train_prev_inds"][:, 0] = self.bos_idx
            for _ in range(steps):
                self._forward_mmt(batch_dict)
                self._forward_output(batch_dict)
                batch_dict["train_prev_inds"][:, 1:] = batch_dict["scores"].argmax(dim=-1)[:, :-1]
        return {"textvqa_scores": batch_dict["scores"]}

    # sa_m4c.py:349-371
    def get_optimizer_parameters(self, base_lr):
        groups, special = [], set()
        for m in self.finetune_modules:
            ps = list(m["module"].parameters())
            groups.append({"params": ps, "lr": base_lr * m["lr_scale"]})
            special.update(ps)
        groups.insert(0, {"params": [p for p in self.parameters() if p not in special]})
        return groups


# --------------------------------------------------------------------------------------
# train-step harness semantics (row a-18)
# --------------------------------------------------------------------------------------
def m4c_decoding_bce_with_mask_loss(scores, targets, loss_mask):
    """task_utils.py:19-30."""
    assert scores.dim() == 3 and loss_mask.dim() == 2
    losses = F.binary_cross_entropy_with_logits(scores, targets, reduction="none") * loss_mask.unsqueeze(-1)
    count = torch.clamp(loss_mask.sum(), min=1.0)
    return losses.sum() / count


def lr_lambda(it, warmup_iters=1000, warmup_factor=0.2, lr_decay_iters=(14000, 19000), lr_decay=0.1):
    """task_utils.py:48-54."""
    if it <= warmup_iters:
        a = float(it) / float(warmup_iters)
        return warmup_factor * (1.0 - a) + a
    return pow(lr_decay, bisect(list(lr_decay_iters), it))


def make_optimizer(model, base_lr=1e-4, **sched):
    """Adam + LambdaLR exactly as task_utils.py:37-57."""
    opt = torch.optim.Adam(model.get_optimizer_parameters(base_lr), lr=base_lr)
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: lr_lambda(it, **sched))


def train_step(model, batch, opt, sched, max_grad_norm=0.25):
    """train.py:133-144 (metric/string code excluded)."""
    out = model(batch)
    loss = m4c_decoding_bce_with_mask_loss(out["textvqa_scores"], batch["targets"], batch["train_loss_mask"])
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    opt.step()
    sched.step()
    model.zero_grad()
    return loss.detach()


# --------------------------------------------------------------------------------------
# boolean allow-mask truth table — what the bit packer must reproduce
# --------------------------------------------------------------------------------------
def allow_mask(key_valid, n_txt, n_oo, n_dec, adj=None, quadrants=(1, 2), n_heads=12):
    """bool [B,H,N,N]: True where the reference's combined additive mask is 0.

    key_valid: [B, n_txt+n_oo] (question/obj/ocr pad masks concatenated); adj: int8
    [B,n_oo,n_oo,R] multi-hot or None for the plain ('n') layers.  Derived from
    sa_m4c.py:475-552 (spatial part) ∧ sa_m4c.py:805-844 (key-valid / prefix-LM / causal)."""
    b = key_valid.size(0)
    n = n_txt + n_oo + n_dec
    base = torch.zeros(b, n, n, dtype=torch.bool)
    base[:, :, : n_txt + n_oo] = key_valid.bool().unsqueeze(1)
    base[:, n - n_dec:, n - n_dec:] = causal_mask(n_dec).bool()
    base = base.unsqueeze(1).expand(b, n_heads, n, n).clone()
    if adj is None:
        return base
    sp = torch.ones(b, n_heads, n, n, dtype=torch.bool)
    oo = slice(n_txt, n_txt + n_oo)
    r = adj.size(-1)
    sp[:, :r, oo, oo] = adj.permute(0, 3, 1, 2) != 0
    region = {"txt": slice(0, n_txt), "oo": oo, "dec": slice(n_txt + n_oo, n)}
    for quad in quadrants:
        rows, cols = _QUADRANT_SLICES[quad]
        sp[:, :r, region[rows], region[cols]] = False
    return base & sp
