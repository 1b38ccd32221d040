#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (/root/reference) on CPU.

Runs ONLY in the build container (the reference never travels to the GPU box).  The reference
imports `easydict` and `pytorch_transformers`, neither of which is installed; per SURVEY.md
appendix B two `sys.modules` stand-ins are injected:
  * easydict.EasyDict  — attribute dict;
  * pytorch_transformers.modeling_bert — BertSelfOutput / BertIntermediate / BertOutput /
    BertEmbeddings are the SAME-NAMED CLASSES OF THE INSTALLED `transformers` (direct descendant of
    pytorch-transformers, identical parameter names); BertLayer/BertEncoder/BertSelfAttention
    (signature changed upstream) and BertConfig/BertPreTrainedModel are restated from the 1.x API.
So the vectors pin the reference's own arithmetic (sam/sa_m4c.py, sam/spatial_utils.py) bit for
bit, and the third-party blocks against transformers' implementation of them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import common as C  # noqa: E402

REF = "/root/reference"


# ------------------------------------------------------------------ shims
def install_shims():
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            super().__setitem__(k, EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def update(self, *a, **kw):
            for k, v in dict(*a, **kw).items():
                self[k] = v

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    from transformers.models.bert import modeling_bert as hf

    class BertConfig:
        def __init__(self, vocab_size_or_config_json_file=30522, hidden_size=768, num_hidden_layers=12,
                     num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                     initializer_range=0.02, layer_norm_eps=1e-12):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size, self.num_hidden_layers = hidden_size, num_hidden_layers
            self.num_attention_heads, self.intermediate_size = num_attention_heads, intermediate_size
            self.hidden_act, self.hidden_dropout_prob = hidden_act, hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings, self.type_vocab_size = max_position_embeddings, type_vocab_size
            self.initializer_range, self.layer_norm_eps = initializer_range, layer_norm_eps
            self.output_attentions = self.output_hidden_states = False
            self.pad_token_id = 0
            self.position_embedding_type = "absolute"

        @classmethod
        def from_dict(cls, d):
            cfg = cls(vocab_size_or_config_json_file=-1)
            for k, v in d.items():
                cfg.__dict__[k] = v
            return cfg

    class BertLayerNorm(nn.Module):
        def __init__(self, hidden_size, eps=1e-12):
            super().__init__()
            self.weight = nn.Parameter(torch.ones(hidden_size))
            self.bias = nn.Parameter(torch.zeros(hidden_size))
            self.variance_epsilon = eps

        def forward(self, x):
            u = x.mean(-1, keepdim=True)
            s = (x - u).pow(2).mean(-1, keepdim=True)
            return self.weight * ((x - u) / torch.sqrt(s + self.variance_epsilon)) + self.bias

    class BertSelfAttention(nn.Module):  # 1.x semantics: softmax(QK^T/sqrt(d)+mask) -> dropout -> V
        def __init__(self, config):
            super().__init__()
            self.h = config.num_attention_heads
            self.d = config.hidden_size // self.h
            self.query = nn.Linear(config.hidden_size, config.hidden_size)
            self.key = nn.Linear(config.hidden_size, config.hidden_size)
            self.value = nn.Linear(config.hidden_size, config.hidden_size)
            self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

        def forward(self, x, mask, head_mask=None):
            sp = lambda t: t.view(t.size(0), t.size(1), self.h, self.d).permute(0, 2, 1, 3)
            q, k, v = sp(self.query(x)), sp(self.key(x)), sp(self.value(x))
            p = self.dropout(nn.Softmax(dim=-1)(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.d) + mask))
            if head_mask is not None:
                p = p * head_mask
            c = torch.matmul(p, v).permute(0, 2, 1, 3).contiguous()
            return (c.view(c.size(0), c.size(1), self.h * self.d),)

    class BertAttention(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.self = BertSelfAttention(config)
            self.output = hf.BertSelfOutput(config)

        def forward(self, x, mask, head_mask=None):
            return (self.output(self.self(x, mask, head_mask)[0], x),)

    class BertLayer(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.attention = BertAttention(config)
            self.intermediate = hf.BertIntermediate(config)
            self.output = hf.BertOutput(config)

        def forward(self, x, mask, head_mask=None):
            a = self.attention(x, mask, head_mask)[0]
            return (self.output(self.intermediate(a), a),)

    class BertEncoder(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

        def forward(self, x, mask, head_mask=None):
            for i, l in enumerate(self.layer):
                x = l(x, mask, head_mask[i])[0]
            return (x,)

    class BertPreTrainedModel(nn.Module):
        def __init__(self, config, *a, **kw):
            super().__init__()
            self.config = config

        def _init_weights(self, m):
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            elif isinstance(m, (BertLayerNorm, nn.LayerNorm)):
                m.bias.data.zero_(); m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

        def init_weights(self):
            self.apply(self._init_weights)

    pt = types.ModuleType("pytorch_transformers")
    mb = types.ModuleType("pytorch_transformers.modeling_bert")
    for k, v in dict(BertConfig=BertConfig, BertLayerNorm=BertLayerNorm, BertEmbeddings=hf.BertEmbeddings,
                     BertEncoder=BertEncoder, BertLayer=BertLayer, BertSelfOutput=hf.BertSelfOutput,
                     BertIntermediate=hf.BertIntermediate, BertOutput=hf.BertOutput,
                     BertPreTrainedModel=BertPreTrainedModel).items():
        setattr(mb, k, v)
    pt.modeling_bert = mb
    sys.modules["pytorch_transformers"] = pt
    sys.modules["pytorch_transformers.modeling_bert"] = mb
    sys.path.insert(0, REF)
    return BertConfig


def ref_adjacency(ref_su, boxes, ctx):
    """relation tensor exactly as the reference dataset builds it (textvqa_dataset.py:378-409)."""
    out = []
    for b in range(boxes.shape[0]):
        shared = ref_su.build_graph_using_normalized_boxes(boxes[b], distance_threshold=0.5)
        mats = {"1": ref_su.torch_broadcast_adj_matrix(torch.from_numpy(shared["1"]))}
        for c, (base, p, m) in {"3": ("1", "31", "32"), "5": ("3", "51", "52"),
                                "7": ("5", "71", "72"), "9": ("7", "91", "92")}.items():
            mats[c] = torch.max(torch.max(mats[base], ref_su.torch_broadcast_adj_matrix(torch.from_numpy(shared[p]))),
                                ref_su.torch_broadcast_adj_matrix(torch.from_numpy(shared[m])))
        out.append(mats[str(ctx)])
    return torch.stack(out)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-22s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024, len(arrays)))


def ext_mask_np(dims):
    """an [B,1,N,N] additive mask of the MMT kind, built independently in numpy (layer-case INPUT)."""
    kv = np.concatenate([C.pad_mask(dims["n_txt_valid"], dims["T"]), C.pad_mask(dims["n_obj_valid"], dims["n_obj"]),
                         C.pad_mask(dims["n_ocr_valid"], dims["n_ocr"])], axis=1)
    n_dec = dims["n_dec"]
    n = kv.shape[1] + n_dec
    allow = np.zeros((dims["B"], n, n), dtype=np.float32)
    allow[:, :, : n - n_dec] = kv[:, None, :]
    allow[:, n - n_dec:, n - n_dec:] = np.tril(np.ones((n_dec, n_dec), dtype=np.float32))
    return ((1.0 - allow) * -10000.0)[:, None]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    BertConfig = install_shims()
    import sam.sa_m4c as ref
    import sam.spatial_utils as ref_su
    from tools.registry import registry

    # ---------------- spatial graph ----------------
    g = {}
    boxes6 = np.array([[.1, .1, .5, .5], [.2, .2, .3, .3], [.6, .1, .8, .3], [.1, .6, .3, .9], [.12, .12, .5, .5], [0, 0, 0, 0]])
    grid = np.array([[x, y, x + .1, y + .1] for x in (0., .3, .6) for y in (0., .3, .6)] + [[.3, .3, .4, .4], [.25, .3, .45, .4]])
    rnd = C.case_boxes("graph_rnd", dict(B=1, n_obj=40, n_ocr=20, n_obj_valid=[33], n_ocr_valid=[20]))[0]
    cross = np.array([[.4, .1, .6, .9], [.1, .4, .9, .6], [.45, .45, .55, .55], [0, 0, 0, 0], [.45, .45, .55, .55]])
    for nm, bx in (("known6", boxes6), ("grid", grid), ("rnd60", rnd), ("cross", cross)):
        shared = ref_su.build_graph_using_normalized_boxes(bx, distance_threshold=0.5)
        g[nm + ".boxes"] = bx
        for k, v in shared.items():
            g["%s.code%s" % (nm, k)] = v
        for ctx in (1, 3, 5, 7, 9):
            g["%s.ctx%d" % (nm, ctx)] = np_(ref_adjacency(ref_su, bx[None], ctx)[0])
    save("spatial_graph", **g)

    # ---------------- elementwise primitives ----------------
    x = torch.from_numpy(C.det_uniform("prim.x", (7, 96), -4, 4)).requires_grad_(True)
    ln = ref.BertLayerNorm(96, eps=1e-12)
    C.fill_state_dict(ln, 0.1, prefix="prim.LayerNorm.")
    y = ln(x); gy = torch.from_numpy(C.det_uniform("prim.gy", (7, 96)))
    (y * gy).sum().backward()
    x2 = torch.from_numpy(C.det_uniform("prim.x", (7, 96), -4, 4)).requires_grad_(True)
    z = ref.gelu(x2); (z * gy).sum().backward()
    save("primitives", ln_out=np_(y), ln_dx=np_(x.grad), ln_dw=np_(ln.weight.grad), ln_db=np_(ln.bias.grad),
         gelu_out=np_(z), gelu_dx=np_(x2.grad))

    # ---------------- SpatialBertLayer ----------------
    for name, case in C.LAYER_CASES.items():
        d = case["dims"]
        cfg = BertConfig.from_dict(C.mmt_config_dict(d, ["s"], case["ctx"], case["quadrants"]))
        layer = ref.SpatialBertLayer(cfg).eval()
        C.fill_state_dict(layer, d["ws"], prefix=name + ".")
        n = d["T"] + d["n_obj"] + d["n_ocr"] + d["n_dec"]
        hidden = torch.from_numpy(C.det_uniform(name + ".hidden", (d["B"], n, d["D"]))).requires_grad_(True)
        ext = torch.from_numpy(ext_mask_np(d))
        adj = ref_adjacency(ref_su, C.case_boxes(name, d), case["ctx"])
        out = layer(hidden, ext, adj)[0]
        gout = torch.from_numpy(C.det_uniform(name + ".gout", tuple(out.shape)))
        (out * gout).sum().backward()
        arrays = dict(adj=np_(adj), out=np_(out), d_hidden=np_(hidden.grad))
        ctx_only = layer.attention.self(hidden.detach(), ext, adj)[0]
        arrays["ctx"] = np_(ctx_only)
        for pn, p in layer.named_parameters():
            if d is C.SMALL or pn in ("attention.self.query.weight", "attention.self.value.bias",
                                      "attention.output.LayerNorm.weight", "output.dense.bias"):
                arrays["grad." + pn] = np_(p.grad)
        save(name, **arrays)

    # ---------------- SpatialBertLayer with the switches no shipped config turns on: use_bias (sa_m4c.py:439-443, 600-603), output_attentions (:604-609), head_mask (:591-592) ----
    name, case = "layer_small_switches", C.LAYER_CASES["layer_small_c3"]
    d = case["dims"]
    cfg = BertConfig.from_dict(C.mmt_config_dict(d, ["s"], case["ctx"], case["quadrants"], use_bias=True, output_attentions=True))
    layer = ref.SpatialBertLayer(cfg).eval()
    C.fill_state_dict(layer, d["ws"], prefix=name + ".")
    n = d["T"] + d["n_obj"] + d["n_ocr"] + d["n_dec"]
    hidden = torch.from_numpy(C.det_uniform(name + ".hidden", (d["B"], n, d["D"]))).requires_grad_(True)
    ext = torch.from_numpy(ext_mask_np(d))
    adj = ref_adjacency(ref_su, C.case_boxes(name, d), case["ctx"])
    head_mask = torch.from_numpy(C.det_uniform(name + ".head_mask", (1, d["H"], 1, 1), 0.25, 1.5))
    head_mask[0, 3] = 0.0                                   # one head switched off
    out, probs = layer(hidden, ext, adj, head_mask)
    gout = torch.from_numpy(C.det_uniform(name + ".gout", tuple(out.shape)))
    (out * gout).sum().backward()
    arrays = dict(adj=np_(adj), head_mask=np_(head_mask), out=np_(out), probs=np_(probs), d_hidden=np_(hidden.grad))
    for pn, p in layer.named_parameters():
        arrays["grad." + pn] = np_(p.grad)
    save(name, **arrays)

    # ---------------- OcrPtrNet ----------------
    d = C.SMALL
    ptr = ref.OcrPtrNet(d["D"], d["D"])
    C.fill_state_dict(ptr, d["ws"], prefix="ptr.")
    qi = torch.from_numpy(C.det_uniform("ptr.q", (d["B"], d["n_dec"], d["D"]))).requires_grad_(True)
    ki = torch.from_numpy(C.det_uniform("ptr.k", (d["B"], d["n_ocr"], d["D"]))).requires_grad_(True)
    om = torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"]))
    sc = ptr(qi, ki, om)
    gs = torch.from_numpy(C.det_uniform("ptr.gs", tuple(sc.shape)))
    (sc * gs).sum().backward()
    save("ptr_net", scores=np_(sc), d_q=np_(qi.grad), d_k=np_(ki.grad),
         **{"grad." + n: np_(p.grad) for n, p in ptr.named_parameters()})

    # ---------------- MMT ----------------
    for name, case in C.MMT_CASES.items():
        d = case["dims"]
        cfg = BertConfig.from_dict(C.mmt_config_dict(d, case["layers"], case["ctx"], case["quadrants"]))
        mmt = ref.MMT(cfg).eval()
        C.fill_state_dict(mmt, d["ws"], prefix=name + ".")
        bd, leaves = mmt_batch(name, d, case["ctx"], ref_adjacency(ref_su, C.case_boxes(name, d), case["ctx"]))
        res = mmt(bd, fixed_ans_emb=leaves["fixed_ans_emb"])
        seq = res["mmt_seq_output"]
        gout = torch.from_numpy(C.det_uniform(name + ".gout", tuple(seq.shape)))
        (seq * gout).sum().backward()
        arrays = dict(adj=np_(bd["spatial_adj_matrices"][str(case["ctx"])]), seq=np_(seq))
        for k, v in leaves.items():
            arrays["d_" + k] = np_(v.grad)
        for pn, p in mmt.named_parameters():
            if d is C.SMALL or pn.endswith("attention.self.key.weight") or "prev_pred" in pn:
                if p.grad is not None:
                    g = np_(p.grad)
                    arrays["grad." + pn] = g if (d is C.SMALL or g.ndim < 2) else g[:8].copy()   # row slice at full size
        save(name, **arrays)

    # ---------------- SAM4C ----------------
    for name, case in C.SAM4C_CASES.items():
        d = case["dims"]
        registry.answer_vocab = list(range(d["V"]))
        registry.BOS_IDX, registry.EOS_IDX = 1, 2
        ocr_feat = 300 + 604 + case["ocr_fc"] + 50
        mcfg = BertConfig.from_dict(C.mmt_config_dict(d, case["layers"], case["ctx"], case["quadrants"],
                                                      obj_feature_size=case["obj_feat"], ocr_feature_size=ocr_feat))
        tcfg = BertConfig.from_dict(dict(num_hidden_layers=case["txt_layers"], text_bert_init_from_bert_base=False,
                                         vocab_size=case["txt_vocab"], max_position_embeddings=32, intermediate_size=128,
                                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, lr_scale_text_bert=0.1))
        model = ref.SAM4C(mcfg, tcfg)
        C.fill_state_dict(model, d["ws"], prefix=name + ".")
        model.train()  # all dropout probabilities are 0 in this config -> deterministic
        bd = sam4c_batch(name, d, case, ref_adjacency(ref_su, C.case_boxes(name, d), case["ctx"]))
        scores = model(bd)["textvqa_scores"]
        losses = torch.nn.functional.binary_cross_entropy_with_logits(scores, bd["targets"], reduction="none")
        losses = losses * bd["train_loss_mask"].unsqueeze(-1)
        loss = losses.sum() / torch.max(bd["train_loss_mask"].sum(), torch.tensor(1.0))   # task_utils.py:24-30
        loss.backward()
        arrays = dict(adj=np_(bd["spatial_adj_matrices"][str(case["ctx"])]), scores=np_(scores), loss=np_(loss))
        for pn, p in model.named_parameters():   # big text_bert / projection matrices: keep a row slice only
            if p.grad is not None:
                g = np_(p.grad)
                arrays["grad." + pn] = g if g.size <= 40000 else g.reshape(g.shape[0], -1)[:8].copy()
        groups = model.get_optimizer_parameters(1e-4)
        arrays["group_sizes"] = np.array([len(gr["params"]) for gr in groups])
        model.eval()
        bd2 = sam4c_batch(name, d, case, bd["spatial_adj_matrices"][str(case["ctx"])])
        with torch.no_grad():
            arrays["greedy_scores"] = np_(model(bd2)["textvqa_scores"])
            arrays["greedy_prev_inds"] = np_(bd2["train_prev_inds"])
        # ---- beam search (sam/beam_search.py driven by sa_m4c.py:304-314), the reference's own BeamSearch class.  Two runs: the model's real EOS
        # index, and an EOS index chosen among the tokens the first run actually emitted so that beams COMPLETE early (the completed-beam
        # branch, beam_search.py:89-93,140-158, is otherwise never taken by a random-weight model).
        for tag, beam, eos in beam_cases(model, name, d, case, bd):
            registry.EOS_IDX = eos
            model.set_beam_size(beam)
            bd3 = sam4c_batch(name, d, case, bd["spatial_adj_matrices"][str(case["ctx"])])
            bd3["train_prev_inds"] = torch.zeros_like(bd3["train_prev_inds"]); bd3["train_prev_inds"][:, 0] = registry.BOS_IDX
            bd3["question_id"] = torch.arange(d["B"]) + 100
            trace = []
            orig_decode = model.bsdecoder.decode

            def traced(batch_dict, t, _orig=orig_decode, _trace=trace):
                sc = batch_dict["scores"][:, t, :].clone()
                out = _orig(batch_dict, t)
                _trace.append((sc, out[1]["train_prev_inds"].clone(), out[1]["topkscores"].clone()))
                return out
            model.bsdecoder.decode = traced
            with torch.no_grad(), legacy_integer_division():
                res = model(bd3, use_beam_search=True)
            arrays["beam.%s.cfg" % tag] = np.array([beam, eos, len(trace)])
            arrays["beam.%s.complete_seqs" % tag] = np_(res["complete_seqs"])
            arrays["beam.%s.topkscores" % tag] = np_(res["topkscores"])
            arrays["beam.%s.question_id" % tag] = np_(res["question_id"])
            arrays["beam.%s.final_scores" % tag] = np_(res["textvqa_scores"])
            for t, (sc, pi, tk) in enumerate(trace):
                arrays["beam.%s.step%d.scores" % (tag, t)] = np_(sc)
                arrays["beam.%s.step%d.prev_inds" % (tag, t)] = np_(pi)
                arrays["beam.%s.step%d.topkscores" % (tag, t)] = np_(tk)
        registry.EOS_IDX = 2
        save(name, **arrays)


class legacy_integer_division:
    """`indices / vocab_size` in sam/beam_search.py:113 is integer division: the reference targets torch <= 1.4, where `/` on integer tensors
    floors (true division for them arrived in torch 1.5/1.6).  For the span of the reference's beam search the old operator is restored for
    INTEGER tensor / INTEGER operands only; everything else goes to the current implementation."""

    def __enter__(self):
        self._orig = torch.Tensor.__truediv__
        orig = self._orig

        def div(a, b):
            b_int = isinstance(b, int) or (torch.is_tensor(b) and not b.is_floating_point() and not b.is_complex())
            if torch.is_tensor(a) and not a.is_floating_point() and not a.is_complex() and a.dtype != torch.bool and b_int:
                return torch.div(a, b, rounding_mode="floor")
            return orig(a, b)
        torch.Tensor.__truediv__ = div
        return self

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self._orig


def beam_cases(model, name, d, case, bd):
    """(tag, beam size, EOS index): `early` takes as EOS the token the greedy decoder emits for sample 0 at step 2, so that beams complete
    early and keep re-emitting EOS"""
    bd2 = sam4c_batch(name, d, case, bd["spatial_adj_matrices"][str(case["ctx"])])
    with torch.no_grad():
        model(bd2)
    tok = int(bd2["train_prev_inds"][0, 2])
    return [("k3", 3, 2), ("early", 4, tok)]


def mmt_batch(name, d, ctx, adj):
    mk = lambda k, shape: torch.from_numpy(C.det_uniform("%s.%s" % (name, k), shape)).requires_grad_(True)
    leaves = dict(text_bert_emb=mk("text_bert_emb", (d["B"], d["T"], d["D"])),
                  obj_mmt_in=mk("obj_mmt_in", (d["B"], d["n_obj"], d["D"])),
                  ocr_mmt_in=mk("ocr_mmt_in", (d["B"], d["n_ocr"], d["D"])),
                  fixed_ans_emb=mk("fixed_ans_emb", (d["V"], d["D"])))
    bd = dict(leaves)
    del bd["fixed_ans_emb"]
    bd.update(question_mask=torch.from_numpy(C.pad_mask(d["n_txt_valid"], d["T"])),
              pad_obj_mask=torch.from_numpy(C.pad_mask(d["n_obj_valid"], d["n_obj"])),
              pad_ocr_mask=torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"])),
              train_prev_inds=torch.from_numpy(C.det_int(name + ".prev", (d["B"], d["n_dec"]), 0, d["V"] + d["n_ocr"])),
              spatial_adj_matrices={str(ctx): adj})
    return bd, leaves


def sam4c_batch(name, d, case, adj):
    B = d["B"]
    t = lambda k, shape, lo=-1.0, hi=1.0: torch.from_numpy(C.det_uniform("%s.%s" % (name, k), shape, lo, hi))
    boxes = C.case_boxes(name, d)
    area = ((boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1]))[..., None]
    b5 = torch.from_numpy(np.concatenate([boxes, area], axis=-1).astype(np.float32))
    targets = (C.det_uniform(name + ".targets", (B, d["n_dec"], d["V"] + d["n_ocr"]), 0, 1) > 0.97).astype(np.float32)
    lm = np.zeros((B, d["n_dec"]), dtype=np.float32); lm[0, :2] = 1; lm[1, :d["n_dec"]] = 1
    return dict(
        pad_obj_features=t("obj_feat", (B, d["n_obj"], case["obj_feat"])),
        pad_obj_bboxes=b5[:, : d["n_obj"]].contiguous(), pad_ocr_bboxes=b5[:, d["n_obj"]:].contiguous(),
        pad_obj_mask=torch.from_numpy(C.pad_mask(d["n_obj_valid"], d["n_obj"])),
        pad_ocr_mask=torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"])),
        pad_ocr_features=t("ocr_fc", (B, d["n_ocr"], case["ocr_fc"])),
        ocr_fasttext=t("ocr_ft", (B, d["n_ocr"], 300)), ocr_phoc=t("ocr_phoc", (B, d["n_ocr"], 604), 0, 1),
        question_indices=torch.from_numpy(C.det_int(name + ".qidx", (B, d["T"]), 1, case["txt_vocab"])),
        question_mask=torch.from_numpy(C.pad_mask(d["n_txt_valid"], d["T"])),
        train_prev_inds=torch.from_numpy(C.det_int(name + ".prev", (B, d["n_dec"]), 0, d["V"] + d["n_ocr"])),
        targets=torch.from_numpy(targets), train_loss_mask=torch.from_numpy(lm),
        spatial_adj_matrices={str(case["ctx"]): adj},
    )


if __name__ == "__main__":
    main()
