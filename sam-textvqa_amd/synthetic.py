"""Synthetic SA-M4C batches with the reference's batch_dict schema and the statistics of §8(d):
n_txt~U{5..T}, all objects valid, n_ocr~U{1..n_ocr}; boxes xy~U(0,0.8), wh~U(0.01,0.21) (OCR 0.01..0.08); relation
tensor built from those boxes by spatial_graph.relation_tensor (so head densities are realistic, not Bernoulli noise);
features N(0,1); targets multi-hot with 1-3 ones per step; loss mask = first k~U{1..n_dec} steps."""
import torch

from .spatial_graph import relation_tensor

SHAPES = {   # name -> (T, n_obj, n_ocr, n_dec)
    "c3": (20, 100, 50, 12), "c5": (20, 100, 50, 12), "stress": (20, 200, 100, 30),
}


def mmt_config_dict(context=3, layers=("n", "n", "s", "s", "s", "s"), n_dec=12, T=20, n_obj=100, n_ocr=50):
    """the SA-M4C block of configs/train-tvqa-eval-tvqa-c{3,5}.yml (model keys only)"""
    mix = {1: "none", 3: "share3", 5: "share5", 7: "share7", 9: "share9"}[context]
    return dict(num_hidden_layers=2, num_spatial_layers=4, heads_type="mix", layer_type_list=list(layers),
                mix_list=[("none" if k == "n" else mix) for k in layers], obj_drop=0.1, ocr_drop=0.1, hidden_size=768,
                num_spatial_relations=12, type_vocab_size=2, vocab_size=30522, ptr_query_size=768, ocr_feature_size=3002,
                obj_feature_size=2048, use_phoc_fasttext=True, normalize=True, lr_scale_mmt=1.0, num_decoding_steps=n_dec,
                max_obj_num=n_obj, max_ocr_num=n_ocr, max_seq_length=T, attention_mask_quadrants=[1, 2])


def text_bert_config_dict():
    """TextBERT block of the ymls, with random init (no network for bert-base-uncased)"""
    return dict(lr_scale_text_bert=0.1, num_hidden_layers=3, text_bert_init_from_bert_base=False, vocab_size=30522)


def make_batch(batch_size, T=20, n_obj=100, n_ocr=50, n_dec=12, vocab=5000, context=3, device="cuda", seed=1234, ocr_feature_fc=2048):
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = batch_size
    n_txt_valid = torch.randint(5, T + 1, (B,), generator=g)
    n_ocr_valid = torch.randint(1, n_ocr + 1, (B,), generator=g)
    ar = lambda n: torch.arange(n).unsqueeze(0)
    question_mask = (ar(T) < n_txt_valid.unsqueeze(1)).long()
    obj_mask = torch.ones(B, n_obj, dtype=torch.long)
    ocr_mask = (ar(n_ocr) < n_ocr_valid.unsqueeze(1)).long()

    def boxes(n, wh_hi, mask):
        xy = torch.rand(B, n, 2, generator=g, dtype=torch.float64) * 0.8
        wh = 0.01 + torch.rand(B, n, 2, generator=g, dtype=torch.float64) * (wh_hi - 0.01)
        b = torch.cat([xy, (xy + wh).clamp(max=1.0)], dim=-1)
        return b * mask.unsqueeze(-1).double()

    obj_b, ocr_b = boxes(n_obj, 0.21, obj_mask), boxes(n_ocr, 0.08, ocr_mask)
    all_b = torch.cat([obj_b, ocr_b], dim=1).to(device)
    if all_b.is_cuda:                                                            # int8 [B, n_oo, n_oo, 12] built on the device
        from . import ops
        adj = ops.spatial_relation_tensor(all_b.contiguous(), context)            # HIP kernel, one thread per box pair
    else:
        adj = relation_tensor(all_b, context)                                     # same semantics, vectorised torch (CPU baseline)
    area = lambda b: ((b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])).unsqueeze(-1)
    steps = torch.randint(1, n_dec + 1, (B,), generator=g)
    loss_mask = (ar(n_dec) < steps.unsqueeze(1)).float()
    W = vocab + n_ocr
    targets = torch.zeros(B, n_dec, W)
    k = torch.randint(1, 4, (B, n_dec), generator=g)
    for j in range(3):
        idx = torch.randint(0, W, (B, n_dec, 1), generator=g)
        targets.scatter_(2, idx, (k > j).float().unsqueeze(-1))
    prev = torch.randint(0, vocab, (B, n_dec), generator=g)
    use_ocr = torch.rand(B, n_dec, generator=g) < 0.3
    prev = torch.where(use_ocr, vocab + (torch.rand(B, n_dec, generator=g) * n_ocr_valid.unsqueeze(1)).long(), prev)
    prev[:, 0] = 1                                                               # BOS
    rn = lambda *s: torch.randn(*s, generator=g)
    bd = dict(
        pad_obj_features=rn(B, n_obj, 2048), pad_obj_bboxes=torch.cat([obj_b, area(obj_b)], -1).float(), pad_obj_mask=obj_mask,
        pad_ocr_features=rn(B, n_ocr, ocr_feature_fc), pad_ocr_bboxes=torch.cat([ocr_b, area(ocr_b)], -1).float(), pad_ocr_mask=ocr_mask,
        ocr_fasttext=rn(B, n_ocr, 300), ocr_phoc=torch.rand(B, n_ocr, 604, generator=g),
        question_indices=torch.randint(1, 30522, (B, T), generator=g) * question_mask, question_mask=question_mask,
        train_prev_inds=prev, targets=targets, train_loss_mask=loss_mask)
    bd = {k_: v.to(device) for k_, v in bd.items()}
    bd["spatial_adj_matrices"] = {str(context): adj, "1": adj if context == 1 else None}
    if context != 1:
        del bd["spatial_adj_matrices"]["1"]
    return bd


def clone_batch(bd):
    """SAM4C.forward mutates batch_dict (adds obj_mmt_in, scores, ...): give every step a fresh shallow copy of the inputs"""
    out = {k: v for k, v in bd.items() if k != "spatial_adj_matrices"}
    out["spatial_adj_matrices"] = dict(bd["spatial_adj_matrices"])
    return out
