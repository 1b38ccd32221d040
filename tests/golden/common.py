"""Shared definitions for the golden fixtures: deterministic tensors and the case table.

Used by tests/golden/make_golden.py (runs ONLY in the build container, imports the reference
from /root/reference) and by the tests (which never touch /root/reference).  Nothing here
depends on torch's or numpy's RNG streams: values come from an integer hash of (name, index),
so both sides regenerate bit-identical inputs and weights on any machine.
"""
import zlib

import numpy as np

GOLDEN_DIR = __import__("os").path.dirname(__import__("os").path.abspath(__file__))


def det_uniform(name, shape, lo=-1.0, hi=1.0):
    """float32 array in [lo, hi): splitmix64 hash of (crc32(name), flat index) — version independent."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode())) * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits -> exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_int(name, shape, lo, hi):
    """int64 array in [lo, hi)."""
    u = det_uniform(name, shape, 0.0, 1.0).astype(np.float64)
    return np.minimum(lo + np.floor(u * (hi - lo)), hi - 1).astype(np.int64)


def det_param(name, shape, weight_scale):
    """deterministic parameter by role: LayerNorm gains near 1, biases small, weights uniform."""
    leaf = name.split(".")[-1]
    is_ln = "LayerNorm" in name or "layer_norm" in name
    if is_ln and leaf == "weight":
        return 1.0 + 0.1 * det_uniform(name, shape)
    if leaf == "bias":
        return 0.05 * det_uniform(name, shape)
    return weight_scale * det_uniform(name, shape)


def fill_state_dict(module, weight_scale, prefix=""):
    """overwrite every parameter of a torch module in place with det_param(name)."""
    import torch
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.from_numpy(det_param(prefix + name, tuple(p.shape), weight_scale)))


def det_boxes(name, n_valid, n_total, wh_hi):
    """normalised xyxy boxes: xy~U(0,0.8), wh~U(0.01,wh_hi); rows >= n_valid are all-zero padding."""
    xy = det_uniform(name + ".xy", (n_total, 2), 0.0, 0.8).astype(np.float64)
    wh = det_uniform(name + ".wh", (n_total, 2), 0.01, wh_hi).astype(np.float64)
    b = np.concatenate([xy, np.minimum(xy + wh, 1.0)], axis=1)
    b[n_valid:] = 0.0
    return b


# ------------------------------------------------------------------------------------------
# case table.  T = question tokens, n_obj / n_ocr / n_dec, D hidden, I FFN width, ctx = spatial context
# ------------------------------------------------------------------------------------------
SMALL = dict(B=2, T=4, n_obj=10, n_ocr=6, n_dec=3, D=96, I=384, H=12, V=40, ws=0.12,
             n_txt_valid=[3, 4], n_obj_valid=[10, 7], n_ocr_valid=[4, 0])
FULL = dict(B=2, T=20, n_obj=100, n_ocr=50, n_dec=12, D=768, I=3072, H=12, V=64, ws=0.04,
            n_txt_valid=[9, 20], n_obj_valid=[100, 63], n_ocr_valid=[17, 50])

LAYER_CASES = {  # one SpatialBertLayer, fwd + bwd
    "layer_small_c3": dict(dims=SMALL, ctx=3, quadrants=[1, 2]),
    "layer_small_c5": dict(dims=SMALL, ctx=5, quadrants=[1, 2]),
    "layer_small_c1_q": dict(dims=SMALL, ctx=1, quadrants=[4, 7, 8, 9]),
    "layer_full_c3": dict(dims=FULL, ctx=3, quadrants=[1, 2]),
}
MMT_CASES = {  # whole MMT (PrevPredEmbeddings + n/s encoder), fwd + bwd
    "mmt_small_c3": dict(dims=SMALL, ctx=3, layers=["n", "s", "s"], quadrants=[1, 2]),
    "mmt_small_c5": dict(dims=SMALL, ctx=5, layers=["n", "n", "s"], quadrants=[1, 2]),
    "mmt_full_c3": dict(dims=FULL, ctx=3, layers=["n", "n", "s", "s", "s", "s"], quadrants=[1, 2]),
    "mmt_full_c5": dict(dims=FULL, ctx=5, layers=["n", "n", "s", "s", "s", "s"], quadrants=[1, 2]),   # configs/train-tvqa-eval-tvqa-c5.yml:52-54
}
SAM4C_CASES = {  # whole model incl. TextBert, input encoders, pointer net, loss
    "sam4c_small_c3": dict(dims=dict(SMALL, n_ocr=50, n_ocr_valid=[11, 0]), ctx=3, layers=["n", "s"],
                           quadrants=[1, 2], obj_feat=24, ocr_fc=16, txt_vocab=30, txt_layers=1),
}


def mmt_config_dict(dims, layers, ctx, quadrants, **extra):
    mix = {1: "none", 3: "share3", 5: "share5", 7: "share7", 9: "share9"}[ctx]
    d = dict(hidden_size=dims["D"], intermediate_size=dims["I"], num_attention_heads=dims["H"],
             num_spatial_relations=dims["H"], max_seq_length=dims["T"], num_decoding_steps=dims["n_dec"],
             attention_mask_quadrants=list(quadrants), layer_type_list=list(layers),
             mix_list=[("none" if k == "n" else mix) for k in layers],
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, no_drop=True,
             num_hidden_layers=len(layers), ptr_query_size=dims["D"], lr_scale_mmt=1.0,
             obj_drop=0.0, ocr_drop=0.0, normalize=True, use_phoc_fasttext=True)
    d.update(extra)
    return d


def pad_mask(valid, total):
    m = np.zeros((len(valid), total), dtype=np.int64)
    for b, v in enumerate(valid):
        m[b, :v] = 1
    return m


def case_boxes(case_name, dims):
    """per-sample (150-style) box table: objects then OCR tokens; float64 [B, n_obj+n_ocr, 4]."""
    out = []
    for b in range(dims["B"]):
        obj = det_boxes("%s.obj%d" % (case_name, b), dims["n_obj_valid"][b], dims["n_obj"], 0.21)
        ocr = det_boxes("%s.ocr%d" % (case_name, b), dims["n_ocr_valid"][b], dims["n_ocr"], 0.08)
        out.append(np.concatenate([obj, ocr], axis=0))
    return np.stack(out)
