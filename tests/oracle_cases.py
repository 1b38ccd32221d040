"""Rebuild the golden cases on the oracle side (no reference import): modules, inputs, and the
expected arrays from tests/golden/*.npz.  Shared by the CPU oracle tests and the GPU parity tests."""
import os

import numpy as np
import torch

from oracle import sa_m4c_oracle as O
from tests.golden import common as C


def load(name):
    return dict(np.load(os.path.join(C.GOLDEN_DIR, name + ".npz")))


def ext_mask(d):
    return O.MMT.extended_attention_mask(torch.from_numpy(C.pad_mask(d["n_txt_valid"], d["T"])),
                                         torch.from_numpy(C.pad_mask(d["n_obj_valid"], d["n_obj"])),
                                         torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"])), d["n_dec"])


def key_valid(d):
    return np.concatenate([C.pad_mask(d["n_txt_valid"], d["T"]), C.pad_mask(d["n_obj_valid"], d["n_obj"]),
                           C.pad_mask(d["n_ocr_valid"], d["n_ocr"])], axis=1)


def layer_case(name, module_cls=None):
    """-> (module, hidden[requires_grad], ext_mask, adj, gout, golden)."""
    case = C.LAYER_CASES[name]
    d = case["dims"]
    cfg = O.BertConfig.from_dict(C.mmt_config_dict(d, ["s"], case["ctx"], case["quadrants"]))
    layer = (module_cls or O.SpatialBertLayer)(cfg).eval()
    C.fill_state_dict(layer, d["ws"], prefix=name + ".")
    n = d["T"] + d["n_obj"] + d["n_ocr"] + d["n_dec"]
    g = load(name)
    hidden = torch.from_numpy(C.det_uniform(name + ".hidden", (d["B"], n, d["D"]))).requires_grad_(True)
    gout = torch.from_numpy(C.det_uniform(name + ".gout", (d["B"], n, d["D"])))
    return layer, hidden, ext_mask(d), torch.from_numpy(g["adj"]), gout, g


def mmt_batch(name, d, ctx, adj):
    mk = lambda k, shape: torch.from_numpy(C.det_uniform("%s.%s" % (name, k), shape)).requires_grad_(True)
    leaves = dict(text_bert_emb=mk("text_bert_emb", (d["B"], d["T"], d["D"])),
                  obj_mmt_in=mk("obj_mmt_in", (d["B"], d["n_obj"], d["D"])),
                  ocr_mmt_in=mk("ocr_mmt_in", (d["B"], d["n_ocr"], d["D"])),
                  fixed_ans_emb=mk("fixed_ans_emb", (d["V"], d["D"])))
    bd = {k: v for k, v in leaves.items() if k != "fixed_ans_emb"}
    bd.update(question_mask=torch.from_numpy(C.pad_mask(d["n_txt_valid"], d["T"])),
              pad_obj_mask=torch.from_numpy(C.pad_mask(d["n_obj_valid"], d["n_obj"])),
              pad_ocr_mask=torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"])),
              train_prev_inds=torch.from_numpy(C.det_int(name + ".prev", (d["B"], d["n_dec"]), 0, d["V"] + d["n_ocr"])),
              spatial_adj_matrices={str(ctx): adj})
    return bd, leaves


def mmt_case(name, module_cls=None):
    case = C.MMT_CASES[name]
    d = case["dims"]
    cfg = O.BertConfig.from_dict(C.mmt_config_dict(d, case["layers"], case["ctx"], case["quadrants"]))
    mmt = (module_cls or O.MMT)(cfg).eval()
    C.fill_state_dict(mmt, d["ws"], prefix=name + ".")
    g = load(name)
    bd, leaves = mmt_batch(name, d, case["ctx"], torch.from_numpy(g["adj"]))
    n = d["T"] + d["n_obj"] + d["n_ocr"] + d["n_dec"]
    gout = torch.from_numpy(C.det_uniform(name + ".gout", (d["B"], n, d["D"])))
    return mmt, bd, leaves, gout, g


def sam4c_configs(name, cfg_cls=None):
    case = C.SAM4C_CASES[name]
    d = case["dims"]
    cfg_cls = cfg_cls or O.BertConfig
    ocr_feat = 300 + 604 + case["ocr_fc"] + 50
    mcfg = cfg_cls.from_dict(C.mmt_config_dict(d, case["layers"], case["ctx"], case["quadrants"],
                                               obj_feature_size=case["obj_feat"], ocr_feature_size=ocr_feat))
    tcfg = cfg_cls.from_dict(dict(num_hidden_layers=case["txt_layers"], text_bert_init_from_bert_base=False,
                                  vocab_size=case["txt_vocab"], max_position_embeddings=32, intermediate_size=128,
                                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, lr_scale_text_bert=0.1))
    return mcfg, tcfg


def sam4c_batch(name, adj):
    case = C.SAM4C_CASES[name]
    d = case["dims"]
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
