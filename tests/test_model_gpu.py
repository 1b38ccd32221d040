"""GPU parity of the drop-in modules against golden vectors produced by the REFERENCE at full size
(tests/golden/layer_full_c3.npz, mmt_full_c3.npz) and against the fp32 oracle on seeded inputs.

The HIP path computes in bf16 (fp32 accumulate); the goldens are fp32.  Two checks per tensor:
  * vs the oracle run on the bf16-rounded weights/inputs (isolates kernel arithmetic from input rounding) and
  * vs the fp32 golden itself,
both relative to max |ref|.  Every comparison prints the error it achieved (PARITY lines, `pytest -s`) and is bounded at about twice that
(table `L` below): ~0.7 % for one layer, 1-2 % through six; the per-kernel 1e-3 bound is enforced in the kernel tests."""
import os
import numpy as np
import pytest
import torch

from oracle import sa_m4c_oracle as O
from tests import oracle_cases as OC
from tests.golden import common as C

pytestmark = pytest.mark.gpu

# Limits of the model-level comparisons (relative to max |ref| unless noted): about 2x the error the HIP path achieves (run with -s for the
# PARITY lines).  The per-kernel 1e-3 bound lives in the kernel tests; these bound the accumulation of bf16 storage roundings through depth.
L = dict(layer_out_o=0.013, layer_dh_o=0.016, layer_out_g=0.014, layer_dh_g=0.016, layer_ctx_g=0.009, layer_pgrad_g=0.012,   # achieved 0.4-0.8 %
         mmt_seq_g=0.03, mmt_dleaf_g=0.035, mmt_pgrad_g=0.04,                      # six layers, c=3 and c=5: achieved 1.0-2.0 %
         alone_vs_batch=0.007,                                                    # other tile shapes / split-K: achieved 0.35 %
         sam4c_scores=0.004, sam4c_loss=5e-5, sam4c_pgrad=0.017,                   # achieved 0.19 %, 6e-6, 0.85 %
         stress_seq=0.03, stress_grad=0.026, greedy_agree=0.95)                    # achieved 1.4 %, 1.3 %, 1.00


def rel_err(got, ref):
    got, ref = got.detach().float().cpu().double(), torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def within(name, err, limit):
    """parity bound with the achieved error on record: limits are set at about twice what the kernels reach (printed with -s), so a
    mid-size defect in one layer of six cannot hide under a depth-scaled allowance"""
    print("PARITY %-58s achieved %.3e  limit %.3e" % (name, err, limit))
    assert err < limit, "%s: error %.3e exceeds %.3e" % (name, err, limit)


def score_err(out, ref):
    """relative-to-max error of the answer scores over the entries that are not the literal -10000 of padded OCR columns (sa_m4c.py:893)"""
    out, ref = out.detach().float().cpu().double(), ref.detach().double()
    live = ref > -9000
    assert ((out < -9000) == ~live).all()
    return ((out - ref).abs()[live].max() / ref[live].abs().max()).item()


def bf16_round_module(m):
    """round every >=2-D parameter to bf16 in place (what the HIP path multiplies with)"""
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    return m


def test_spatial_layer_full_size_vs_reference_golden():
    import sam_textvqa_amd.modules as M
    name = "layer_full_c3"
    o_layer, hidden, ext, adj, gout, g = OC.layer_case(name)
    case = C.LAYER_CASES[name]
    d = case["dims"]
    cfg = M.BertConfig.from_dict(C.mmt_config_dict(d, ["s"], case["ctx"], case["quadrants"]))
    layer = M.SpatialBertLayer(cfg).eval()
    layer.load_state_dict(o_layer.state_dict())          # identical keys: a reference checkpoint is a drop-in
    layer.cuda()
    h = hidden.detach().to(torch.bfloat16).cuda().requires_grad_(True)
    out = layer(h, ext.cuda(), adj.cuda())[0]            # module-level API: additive fp32 mask + int8 relation tensor
    (out.float() * gout.cuda()).sum().backward()
    # oracle on bf16-rounded weights + input
    bf16_round_module(o_layer)
    hb = hidden.detach().to(torch.bfloat16).float().requires_grad_(True)
    oo = o_layer(hb, ext, adj)[0]
    (oo * gout).sum().backward()
    within("layer out vs oracle(bf16 weights)", rel_err(out, oo), L["layer_out_o"])
    within("layer d_hidden vs oracle(bf16 weights)", rel_err(h.grad, hb.grad), L["layer_dh_o"])
    within("layer out vs reference golden", rel_err(out, g["out"]), L["layer_out_g"])
    within("layer d_hidden vs reference golden", rel_err(h.grad, g["d_hidden"]), L["layer_dh_g"])
    # text rows of a spatial layer: context is exactly 0 (sa_m4c.py:574-584)
    ctx = layer.attention.self(h.detach(), ext.cuda(), adj.cuda())[0]
    assert (ctx[:, : d["T"]] == 0).all()
    within("layer attention context vs reference golden", rel_err(ctx, g["ctx"]), L["layer_ctx_g"])
    for pn, p in layer.named_parameters():
        if "grad." + pn in g:
            within("layer grad " + pn, rel_err(p.grad, g["grad." + pn]), L["layer_pgrad_g"])


@pytest.mark.parametrize("name", ["mmt_full_c3", "mmt_full_c5"])
def test_mmt_full_size_vs_reference_golden(name):
    """the shipped c=3 and c=5 encoders (configs/train-tvqa-eval-tvqa-c{3,5}.yml: n,n,s,s,s,s) at full size against the REFERENCE's outputs"""
    import sam_textvqa_amd.modules as M
    o_mmt, bd, leaves, gout, g = OC.mmt_case(name)
    case = C.MMT_CASES[name]
    d = case["dims"]
    cfg = M.BertConfig.from_dict(C.mmt_config_dict(d, case["layers"], case["ctx"], case["quadrants"]))
    mmt = M.MMT(cfg).eval()
    mmt.load_state_dict(o_mmt.state_dict())
    mmt.cuda()
    gbd = {k: (v.detach().cuda() if torch.is_tensor(v) else v) for k, v in bd.items()}
    gbd["spatial_adj_matrices"] = {k: v.cuda() for k, v in bd["spatial_adj_matrices"].items()}
    gl = {k: v.detach().cuda().requires_grad_(True) for k, v in leaves.items()}
    for k in ("text_bert_emb", "obj_mmt_in", "ocr_mmt_in"):
        gbd[k] = gl[k]
    seq = mmt(gbd, fixed_ans_emb=gl["fixed_ans_emb"])["mmt_seq_output"]
    (seq.float() * gout.cuda()).sum().backward()
    within(name + " seq vs reference golden", rel_err(seq, g["seq"]), L["mmt_seq_g"])
    for k in leaves:
        within(name + " d_" + k, rel_err(gl[k].grad, g["d_" + k]), L["mmt_dleaf_g"])
    for pn, p in mmt.named_parameters():
        if "grad." + pn in g:
            ref = g["grad." + pn]
            within(name + " grad " + pn, rel_err(p.grad[: ref.shape[0]] if ref.shape != tuple(p.shape) else p.grad, ref), L["mmt_pgrad_g"])


def test_state_dict_keys_match_oracle_and_roundtrip():
    import sam_textvqa_amd.modules as M
    mcfg, tcfg = OC.sam4c_configs("sam4c_small_c3")
    mcfg_h = M.BertConfig.from_dict(dict(mcfg.__dict__, hidden_size=768, intermediate_size=256, ptr_query_size=768))
    tcfg_h = M.BertConfig.from_dict(tcfg.__dict__)
    model = M.SAM4C(mcfg_h, tcfg_h, num_answers=40, bos_idx=1)
    ref = O.SAM4C(O.BertConfig.from_dict(mcfg_h.__dict__), O.BertConfig.from_dict(tcfg_h.__dict__), num_answers=40)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    assert [tuple(v.shape) for v in model.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    model.load_state_dict({"module." [7:] + k: v for k, v in ref.state_dict().items()})
    model.cuda()
    from sam_textvqa_amd.params import prepare
    prepare(model)
    sd = model.state_dict()
    for k, v in ref.state_dict().items():
        assert torch.equal(sd[k].cpu(), v), k
    assert [len(gr["params"]) for gr in model.get_optimizer_parameters(1e-4)] == [len(gr["params"]) for gr in ref.get_optimizer_parameters(1e-4)]


def _small_full_model(ctx, layers, shapes, vocab=300, seed=0, ffn=None):
    """(hip model, oracle model) with identical weights; dropout off; TextBert 1 layer; ffn: intermediate_size of every layer (default: the configs' 3072)"""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import mmt_config_dict, text_bert_config_dict
    T, n_obj, n_ocr, n_dec = shapes
    md = mmt_config_dict(ctx, layers, n_dec=n_dec, T=T, n_obj=n_obj, n_ocr=n_ocr)
    md.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, obj_drop=0.0, ocr_drop=0.0)
    td = dict(text_bert_config_dict(), num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=500)
    if ffn:
        md["intermediate_size"] = td["intermediate_size"] = ffn
    torch.manual_seed(seed)
    ref = O.SAM4C(O.BertConfig.from_dict(md), O.BertConfig.from_dict(td), num_answers=vocab)
    with torch.no_grad():           # spread the LayerNorm gains / biases so that nothing is hidden by the 1/0 init
        for n_, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    model = M.SAM4C(M.BertConfig.from_dict(md), M.BertConfig.from_dict(td), num_answers=vocab, bos_idx=1)
    model.load_state_dict(ref.state_dict())
    return model, ref


@pytest.mark.parametrize("ctx,layers,shapes", [(3, ("n", "s"), (20, 100, 50, 12)), (5, ("s", "n", "s"), (20, 100, 50, 12)),
                                               (3, ("n", "s", "s"), (20, 160, 70, 12)), (3, ("s", "n"), (13, 37, 21, 5)), (5, ("s", "s"), (20, 120, 60, 12)),
                                               (3, ("n", "s"), (7, 20, 9, 3)), (3, ("s", "n"), (20, 200, 70, 30))])
def test_sam4c_train_forward_backward_vs_oracle(ctx, layers, shapes, batch=3, vocab=300):
    """whole model (input encoders, TextBert, MMT, classifier, pointer net, masked BCE): loss and gradients vs the fp32 oracle.  Besides the two
    BASELINE shapes: sequence lengths in every other class of the attention kernels' key tiling and of the masks' row stride (39, 76, 212, 262
    and 320 tokens -- round 5 found the spatial mask packer wrong for 257..320 keys, which neither BASELINE shape touches)"""
    from sam_textvqa_amd.params import prepare
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import masked_bce_loss
    model, ref = _small_full_model(ctx, layers, shapes, vocab=vocab)
    bd_cpu = make_batch(batch, *shapes, vocab=vocab, context=ctx, device="cpu", seed=11)
    bd_cpu["question_indices"] = (bd_cpu["question_indices"] % 499 + 1) * bd_cpu["question_mask"]      # padded tokens -> id 0 = padding_idx
    ref.train()
    out_ref = ref(clone_batch(bd_cpu))["textvqa_scores"]
    loss_ref = O.m4c_decoding_bce_with_mask_loss(out_ref, bd_cpu["targets"], bd_cpu["train_loss_mask"])
    loss_ref.backward()
    model.cuda().train()
    fp = prepare(model)
    bd = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd_cpu.items()}
    fp.zero_grad()
    out = model(bd)["textvqa_scores"]
    loss = masked_bce_loss(bd)
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(out_ref.shape)
    within("sam4c scores c=%d" % ctx, score_err(out, out_ref), L["sam4c_scores"])
    # (the loss is a mean over B * n_dec * (vocab + n_ocr) scores: the two 3- and 5-step shapes average a quarter / a half of the terms of the 12-step ones)
    within("sam4c loss c=%d" % ctx, abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), L["sam4c_loss"] * (4 if shapes[3] < 12 else 1))
    # padded OCR columns carry the literal -10000 (sa_m4c.py:893)
    pad = (bd_cpu["pad_ocr_mask"] == 0)
    assert (out.cpu()[:, :, vocab:][pad.unsqueeze(1).expand(-1, out.shape[1], -1)] < -9000).all()
    bad, worst = [], 0.0
    refp = dict(ref.named_parameters())
    biggest = max(p.grad.norm().item() for p in ref.parameters() if p.grad is not None)
    for pn, p in model.named_parameters():
        g_ref = refp[pn].grad
        # d(key.bias) is mathematically 0 (softmax shift invariance): only rounding noise on both sides -> skip ~zero references
        if g_ref is None or g_ref.norm().item() < 1e-5 * biggest:
            assert p.grad.norm().item() < 1e-2 * biggest, pn
            continue
        e = ((p.grad.cpu().double() - g_ref.double()).norm() / g_ref.double().norm()).item()
        worst = max(worst, e)
        if e > L["sam4c_pgrad"]:
            bad.append((pn, round(e, 4)))
    within("sam4c worst parameter-gradient norm error c=%d" % ctx, worst, L["sam4c_pgrad"])
    assert not bad, bad


def test_mmt_stress_shape_forward_backward_vs_oracle():
    """BASELINE config 5 shapes: 200 obj + 100 OCR + 30 dec (+20 text) = 350 tokens (the NKT = 24 attention template, the 128 KB-LDS backward);
    2 samples with different padding, layers n,s,s; output AND the gradients of every input and parameter against the fp32 oracle"""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import make_batch, mmt_config_dict
    shapes = (20, 200, 100, 30)
    md = mmt_config_dict(3, ("n", "s", "s"), n_dec=30, T=20, n_obj=200, n_ocr=100)
    md.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(1)
    ref = O.MMT(O.BertConfig.from_dict(md)).eval()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    mmt = M.MMT(M.BertConfig.from_dict(md)).eval()
    mmt.load_state_dict(ref.state_dict()); mmt.cuda()
    bd = make_batch(2, *shapes, vocab=100, device="cpu", seed=5)
    g = torch.Generator().manual_seed(2)
    leaves = dict(text_bert_emb=torch.randn(2, 20, 768, generator=g), obj_mmt_in=torch.randn(2, 200, 768, generator=g),
                  ocr_mmt_in=torch.randn(2, 100, 768, generator=g), fixed_ans_emb=torch.randn(100, 768, generator=g))
    gout = torch.randn(2, 350, 768, generator=g)
    rl = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
    rbd = dict(bd); rbd.update({k: rl[k] for k in ("text_bert_emb", "obj_mmt_in", "ocr_mmt_in")})
    seq_ref = ref(rbd, fixed_ans_emb=rl["fixed_ans_emb"])["mmt_seq_output"]
    (seq_ref * gout).sum().backward()
    gl = {k: v.clone().cuda().requires_grad_(True) for k, v in leaves.items()}
    gbd = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd.items()}
    gbd.update({k: gl[k] for k in ("text_bert_emb", "obj_mmt_in", "ocr_mmt_in")})
    seq = mmt(gbd, fixed_ans_emb=gl["fixed_ans_emb"])["mmt_seq_output"]
    (seq.float() * gout.cuda()).sum().backward()
    assert seq.shape == (2, 350, 768)
    within("stress seq", rel_err(seq, seq_ref.detach()), L["stress_seq"])
    for k in leaves:
        within("stress d_" + k, rel_err(gl[k].grad, rl[k].grad), L["stress_grad"])
    refp = dict(ref.named_parameters())
    biggest = max(p.grad.norm().item() for p in ref.parameters() if p.grad is not None)
    worst = 0.0
    for pn, p in mmt.named_parameters():
        g_ref = refp[pn].grad
        if g_ref is None or g_ref.norm().item() < 1e-5 * biggest:
            continue
        worst = max(worst, ((p.grad.cpu().double() - g_ref.double()).norm() / g_ref.double().norm()).item())
    within("stress worst parameter-gradient norm error", worst, L["stress_grad"])


@pytest.mark.parametrize("ctx", [3, 5])
def test_spatial_layer_is_sample_separable_inside_the_full_batch(ctx):
    """B = 64 at full size (11648 token rows), spatial context c = 3 and c = 5 (BASELINE configs 2 and 3): any sample computed inside the batch --
    output, input gradient -- is bit-identical to the same sample computed alone (no cross-sample leakage through tiles, masks or reductions),
    and the lone sample matches the oracle"""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import make_batch, mmt_config_dict
    md = mmt_config_dict(ctx, ("s",))
    md.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(3)
    o_layer = O.SpatialBertLayer(O.BertConfig.from_dict(md)).eval()
    layer = M.SpatialBertLayer(M.BertConfig.from_dict(md)).eval()
    layer.load_state_dict(o_layer.state_dict()); layer.cuda()
    B, N = 64, 182
    bd = make_batch(B, vocab=100, context=ctx, device="cuda", seed=9)
    mask = torch.cat([bd["question_mask"], bd["pad_obj_mask"], bd["pad_ocr_mask"], torch.ones(B, 12, dtype=torch.long, device="cuda")], 1).float()
    ext = ((1.0 - mask) * -10000.0)[:, None, None, :].expand(B, 1, N, N).contiguous()
    adj = bd["spatial_adj_matrices"][str(ctx)]
    g = torch.Generator(device="cuda").manual_seed(4)
    hid = torch.randn(B, N, 768, device="cuda", generator=g).to(torch.bfloat16)
    gout = torch.randn(B, N, 768, device="cuda", generator=g).to(torch.bfloat16)
    h = hid.clone().requires_grad_(True)
    out = layer(h, ext, adj)[0]
    out.backward(gout)
    # (i) same launch shapes, every OTHER sample replaced: the kept samples must not change by a bit
    keep = (0, 37, 63)
    sel = torch.zeros(B, dtype=torch.bool, device="cuda"); sel[list(keep)] = True
    hid2 = torch.where(sel[:, None, None], hid, torch.randn(B, N, 768, device="cuda", generator=g).to(torch.bfloat16))
    gout2 = torch.where(sel[:, None, None], gout, torch.randn(B, N, 768, device="cuda", generator=g).to(torch.bfloat16))
    perm = torch.arange(B, device="cuda"); others = perm[~sel]; perm[~sel] = others.flip(0)
    perm[list(keep)] = torch.tensor(keep, device="cuda")
    h2 = hid2.clone().requires_grad_(True)
    out2 = layer(h2, ext[perm].contiguous().clone().index_copy_(0, torch.tensor(keep, device="cuda"), ext[list(keep)]), adj[perm].contiguous())[0]
    out2.backward(gout2)
    for b in keep:
        assert torch.equal(out2[b], out[b]) and torch.equal(h2.grad[b], h.grad[b]), b
    assert not torch.equal(out2[1], out[1])
    # (ii) the sample alone: other tile shapes (and split-K for the 182-row FFN2), so equal up to summation order
    for b in keep:
        h1 = hid[b:b + 1].clone().requires_grad_(True)
        o1 = layer(h1, ext[b:b + 1].contiguous(), adj[b:b + 1].contiguous())[0]
        o1.backward(gout[b:b + 1])
        within("sample %d alone vs inside the batch of 64: out" % b, rel_err(o1[0], out[b].float().cpu()), L["alone_vs_batch"])
        within("sample %d alone vs inside the batch of 64: d_hidden" % b, rel_err(h1.grad[0], h.grad[b].float().cpu()), L["alone_vs_batch"])
    bf16_round_module(o_layer)
    hb = hid[37:38].float().cpu().requires_grad_(True)
    oo = o_layer(hb, ext[37:38].cpu(), adj[37:38].cpu())[0]
    oo.backward(gout[37:38].float().cpu())
    within("separable layer sample 37 out vs oracle", rel_err(out[37:38], oo), L["layer_out_o"])
    within("separable layer sample 37 d_hidden vs oracle", rel_err(h.grad[37:38], hb.grad), L["layer_dh_o"])


_STRESS_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["SAM_REPO"])
os.environ["SAM_FORCE_DIST"] = "1"; os.environ["SAM_REDUCER_CHECK"] = "1"
from sam_textvqa_amd import parallel
import sam_textvqa_amd.modules as M
from sam_textvqa_amd.synthetic import clone_batch, make_batch, mmt_config_dict, text_bert_config_dict
from sam_textvqa_amd.trainer import Trainer
parallel.init_distributed()
shapes = (20, 200, 100, 30)
md = mmt_config_dict(3, ("n", "n") + ("s",) * 10, n_dec=30, T=20, n_obj=200, n_ocr=100)
torch.manual_seed(0)
model = M.SAM4C(M.BertConfig.from_dict(md), M.BertConfig.from_dict(text_bert_config_dict()), num_answers=5000, bos_idx=1)
tr = Trainer(model, base_lr=1e-4, seed=1)
assert tr.reducer is not None and tr.reducer.check and tr.reducer.comm is not None
batch = make_batch(32, *shapes, vocab=5000, device="cuda", seed=2)
losses = [tr.step(clone_batch(batch)).item() for _ in range(3)]
torch.cuda.synchronize()
print("LOSSES", losses)
assert all(l == l and l < 1e4 for l in losses) and losses[-1] < losses[0], losses
assert all(tr.reducer.done)
assert torch.isfinite(tr.flat.flat).all()
print("STRESS_OK")
"""


@pytest.mark.transport
def test_stress_shape_train_step_b32_twelve_layers_with_reducer_check():
    """BASELINE config 5 as the bench runs it: B = 32, 350 tokens (M = 11200 rows), 12 layers, dropout on, the data-parallel reducer active in a
    1-rank RCCL group with SAM_REDUCER_CHECK=1 (every bucket re-verified at finish()); three steps, finite and decreasing loss"""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAM_REPO=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from tests.util import run_child
    run_child([sys.executable, "-c", _STRESS_SCRIPT], env, "STRESS_OK", "stress_train_step", timeout=900)


@pytest.mark.parametrize("batch", [6, 48])
def test_trainer_full_depth_batches_the_eight_wave_wgrad_declines(batch, monkeypatch):
    """without a gradient reducer the MMT's last layer pair is held and merged with TextBert's 12 problems into ONE grouped wgrad call of 20.
    The 8-wave grouped kernel declines a set when a K = B * N is not a multiple of 64 (B = 6: both; B = 48: the MMT's 48 * 182) -- any batch that is not a
    multiple of 32, e.g. the last partial batch of an epoch (the reference's loader keeps it: sam/task_utils.py:163).  Sets above 12 problems must then
    go out as chunks the 4-wave kernel takes (csrc/gemm.hip: sam_gemm_bf16_grouped), and train like the unmerged path."""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import clone_batch, make_batch, mmt_config_dict, text_bert_config_dict
    from sam_textvqa_amd.trainer import Trainer
    shapes = (20, 100, 50, 12)
    md = mmt_config_dict(3, ("n", "n", "s", "s"), n_dec=12, T=20, n_obj=100, n_ocr=50)
    md.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, obj_drop=0.0, ocr_drop=0.0)
    td = dict(text_bert_config_dict(), num_hidden_layers=3, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=500)
    data = make_batch(batch, *shapes, vocab=300, device="cuda", seed=5)
    data["question_indices"] = data["question_indices"] % 500
    res = []
    for merge in ("1", "0"):
        monkeypatch.setenv("SAM_WGRAD_MERGE_TB", merge)
        torch.manual_seed(0)
        model = M.SAM4C(M.BertConfig.from_dict(md), M.BertConfig.from_dict(td), num_answers=300, bos_idx=1)
        tr = Trainer(model, base_lr=1e-3, seed=3)
        assert tr.reducer is None
        losses = [tr.step(clone_batch(data)).item() for _ in range(3)]
        torch.cuda.synchronize()
        assert all(np.isfinite(losses)), losses
        res.append((losses, tr.flat.flat.clone()))
    (l1, p1), (l0, p0) = res
    assert all(abs(a - b) <= 2e-3 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
    assert (p1 - p0).abs().max().item() < 5e-3


def test_trainer_steps_reduce_loss_and_checkpoint_roundtrip():
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    model, ref = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
    tr = Trainer(model, base_lr=1e-3, seed=3)
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    losses = [tr.step(clone_batch(batch)).item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    assert tr.global_step == 8 and abs(tr.current_lrs()[0] - 1e-3 * (0.2 + 0.8 * 8 / 1000)) < 1e-9
    sd = tr.state_dict()
    assert list(sd["model_state_dict"]) == list(ref.state_dict())
    ref.load_state_dict(sd["model_state_dict"])                   # a checkpoint written here loads into the reference layout
    tr.load_model_state_dict({"module." + k: v for k, v in sd["model_state_dict"].items()})


def test_overwritten_layer_gradients_equal_zero_fill_plus_accumulate(monkeypatch):
    """the Trainer does not zero the encoder layers' gradients: their backward overwrites them (SAM_GRAD_OVERWRITE, default on).  Same gradients,
    bit for bit, as zero-fill + accumulate on every encoder-layer parameter; same trajectory"""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    runs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SAM_GRAD_OVERWRITE", mode)
        model, _ = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
        tr = Trainer(model, base_lr=1e-3, seed=3, use_graph=False)
        assert bool(tr._fresh_ranges) == (mode == "1")
        if mode == "1":
            assert len(tr._fresh_layers) == 3                       # 2 MMT layers + 1 TextBert layer
            tr.flat.grad.fill_(123.0)                                # whatever is in the buffer before the first step must not matter
            if tr.sparse is not None:                                 # (except in the row-sparse table: rows without a gradient are zero by contract)
                tr.flat.grad[tr.sparse[0]: tr.sparse[1]].zero_()
        losses = [tr.step(clone_batch(batch)).item() for _ in range(2)]
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        runs.append((losses, grads, tr.flat.flat.clone(), [id(l) for l in tr._fresh_layers]))
    (l1, g1, p1, _), (l0, g0, p0, _) = runs
    for n in g1:
        if ".encoder.layer." in n or "_layers." in n:              # TextBert / MMT encoder layers: exact
            assert torch.equal(g1[n], g0[n]), n
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
    assert (p1 - p0).abs().max().item() < 1e-5


@pytest.mark.parametrize("coarse", ["1", "0"])
def test_deferred_layernorm_finalize_is_bit_identical(monkeypatch, coarse):
    """SAM_LN_DEFER_FINALIZE (default on): the encoder layers' LayerNorm backwards leave their partial sums in place and one batched launch reduces them
    after the backward pass -- same summation order, so every gradient is bit-identical to the immediate finalize.  Both launch routes (C++ ops, ctypes)."""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    monkeypatch.setenv("SAM_COARSE_OPS", coarse)
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    grads = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SAM_LN_DEFER_FINALIZE", mode)
        model, _ = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
        tr = Trainer(model, base_lr=1e-3, seed=3, use_graph=False)
        assert tr.defer_ln == (mode == "1")
        tr.step(clone_batch(batch))
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if "LayerNorm" in n or n.endswith("dense.bias")})
    assert len(grads[0]) > 10
    for n in grads[0]:
        if ".encoder.layer." in n or "_layers." in n:
            assert torch.equal(grads[0][n], grads[1][n]), n


@pytest.mark.parametrize("from_bert_base", [False, True])
def test_checkpoint_dict_is_the_reference_layout_and_resumes_exactly(from_bert_base, tmp_path):
    """train.py:177-187: model_state_dict, optimizer_state_dict (torch.optim.Adam layout over the reference's param groups), warmup_scheduler_state_dict
    (LambdaLR), global_step, current_val_score, epoch_id.  The dict loads into a plain torch Adam / LambdaLR built the reference's way
    (task_utils.py:37-57), and a fresh trainer resumed from the file continues bit-identically to the uninterrupted run.
    from_bert_base: the shipped YAMLs' three optimizer groups (sa_m4c.py:74-85)."""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import clone_batch, make_batch, mmt_config_dict, text_bert_config_dict
    from sam_textvqa_amd.trainer import Trainer, lr_lambda

    def build():
        torch.manual_seed(5)
        md = mmt_config_dict(3, ("n", "s"))
        md.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, obj_drop=0.0, ocr_drop=0.0)
        td = dict(text_bert_config_dict(), num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=500,
                  text_bert_init_from_bert_base=from_bert_base, lr_scale_text_bert=0.1)
        return M.SAM4C(M.BertConfig.from_dict(md), M.BertConfig.from_dict(td), num_answers=300, bos_idx=1)

    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    tr = Trainer(build(), base_lr=1e-3, seed=3)
    assert len(tr.group_lr) == (3 if from_bert_base else 2)
    for _ in range(3):
        tr.step(clone_batch(batch))
    path = str(tmp_path / "best_model.tar")
    tr.save_checkpoint(path, current_val_score=0.42, epoch_id=7)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "warmup_scheduler_state_dict", "global_step", "current_val_score", "epoch_id"}
    assert ck["global_step"] == 3 and ck["current_val_score"] == 0.42 and ck["epoch_id"] == 7
    # the reference's own objects accept it
    plain = build()
    plain.load_state_dict(ck["model_state_dict"])
    groups = plain.get_optimizer_parameters(1e-3)
    opt = torch.optim.Adam(groups, lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lr_lambda)
    opt.load_state_dict(ck["optimizer_state_dict"])
    sched.load_state_dict(ck["warmup_scheduler_state_dict"])
    assert [len(g["params"]) for g in opt.param_groups] == [len(g["params"]) for g in groups] and sched.last_epoch == 3
    want = [1e-3 * lr_lambda(3) * (0.1 if (from_bert_base and i == 1) else 1.0) for i in range(len(groups))]
    assert all(abs(g["lr"] - w) < 1e-12 for g, w in zip(opt.param_groups, want))
    names = {id(p): n for n, p in plain.named_parameters()}
    mine = dict(tr.model.named_parameters())
    for g in opt.param_groups:
        for p in g["params"]:
            st, q = opt.state[p], mine[names[id(p)]]
            assert int(st["step"]) == 3 and st["exp_avg"].shape == p.shape
            assert torch.equal(st["exp_avg"], tr.flat._view(tr.exp_avg, q._sam_index, q).cpu())
            assert torch.equal(st["exp_avg_sq"], tr.flat._view(tr.exp_avg_sq, q._sam_index, q).cpu())
    # resume: same trajectory as the uninterrupted run, bit for bit (dropout is off; the sparse-table scatter has a fixed order only
    # under a reducer, the single-process word-embedding scatter uses atomics -> compare with a tolerance of a few ulps there)
    tr2 = Trainer(build(), base_lr=1e-3, seed=3).load_checkpoint(path)
    assert tr2.global_step == 3 and tr2.epoch_id == 7 and tr2.current_val_score == 0.42
    assert torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.flat.flat, tr.flat.flat)
    a = [tr.step(clone_batch(batch)).item() for _ in range(2)]
    b = [tr2.step(clone_batch(batch)).item() for _ in range(2)]
    assert all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(a, b)), (a, b)
    assert (tr.flat.flat - tr2.flat.flat).abs().max().item() < 1e-5


def test_greedy_decode_eval_matches_oracle():
    """eval(): BOS-seeded greedy loop, 12 re-forwards (sa_m4c.py:285-302); scores close, decoded indices (argmax of near-equal
    logits may flip under bf16) agree on the overwhelming majority of steps"""
    from sam_textvqa_amd.params import prepare
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    shapes = (20, 100, 50, 12)
    model, ref = _small_full_model(3, ("n", "s"), shapes)
    bd_cpu = make_batch(3, *shapes, vocab=300, context=3, device="cpu", seed=13)
    bd_cpu["question_indices"] = (bd_cpu["question_indices"] % 499 + 1) * bd_cpu["question_mask"]
    ref.eval()
    with torch.no_grad():
        ref_bd = clone_batch(bd_cpu)
        ref_scores = ref(ref_bd)["textvqa_scores"]
    model.cuda().eval()
    prepare(model)
    bd = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd_cpu.items()}
    with torch.no_grad():
        scores = model(bd)["textvqa_scores"]
    assert (bd["train_prev_inds"][:, 0] == 1).all()                     # BOS
    agree = (bd["train_prev_inds"].cpu() == ref_bd["train_prev_inds"]).float().mean().item()
    print("PARITY greedy step agreement %.4f (limit %.2f)" % (agree, L["greedy_agree"]))
    assert agree >= L["greedy_agree"], agree
    same = (bd["train_prev_inds"].cpu() == ref_bd["train_prev_inds"]).all(dim=1)       # samples whose whole decode path agrees
    if same.any():
        assert rel_err(scores[same.cuda()], ref_scores[same]) < 0.05


def test_cached_greedy_decode_equals_full_recompute(monkeypatch):
    """encoder-row caching (12x fewer rows per greedy step) must give the same scores / indices as 12 full forwards (the per-kernel decoding step:
    same kernels as the full pass, bit-identical; the persistent kernel has its own tests in test_decode_gpu.py)"""
    monkeypatch.setenv("SAM_DECODE_FUSED", "0")
    from sam_textvqa_amd.params import prepare
    from sam_textvqa_amd.synthetic import make_batch
    shapes = (20, 100, 50, 12)
    model, _ = _small_full_model(3, ("n", "s", "s"), shapes)
    model.cuda().eval()
    prepare(model)
    outs = []
    for cached in (False, True):
        model.decode_cache = cached
        bd = make_batch(3, *shapes, vocab=300, context=3, device="cuda", seed=17)
        bd["question_indices"] = (bd["question_indices"] % 499 + 1) * bd["question_mask"]
        with torch.no_grad():
            outs.append((model(bd)["textvqa_scores"].float().cpu(), bd["train_prev_inds"].cpu(), bd["mmt_seq_output"].float().cpu()))
        assert "_sam_decode_cache" not in bd
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])      # same kernels, same k order: bit-identical


_DIST_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["SAM_REPO"])
from tests.test_model_gpu import _small_full_model
from sam_textvqa_amd import parallel
from sam_textvqa_amd.synthetic import clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer
os.environ["SAM_FORCE_DIST"] = "1"
os.environ["SAM_REDUCER_CHECK"] = "1"                   # every released bucket is re-checked at finish(): a premature release raises
parallel.init_distributed()                               # 1-rank RCCL group: all-reduce / all-gather really go through RCCL
res = []
three = os.environ.get("SAM_TEST_THREE_GROUPS") == "1"
for dist_on in (True, False):
    os.environ["SAM_FORCE_DIST"] = "1" if dist_on else "0"
    model, _ = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
    if three:        # the shipped YAMLs: TextBert is its own optimizer group (sa_m4c.py:74-85) -> [default | text_bert | mmt] in flat storage
        model.finetune_modules.insert(0, {"module": model.text_bert, "lr_scale": 0.1})
    tr = Trainer(model, base_lr=1e-3, seed=3)
    assert (tr.reducer is not None) == dist_on
    if dist_on:
        assert tr.reducer.comm is not None, "RCCL group: the reducer must enqueue its collectives directly (rccl.py), not through ProcessGroupNCCL Work objects"
    if dist_on and three:
        w = model.text_bert.embeddings.word_embeddings.weight
        red, flat = tr.reducer, tr.flat
        lo_tb = flat.range_of(model.text_bert)[0]
        assert len(tr.group_lr) == 3 and (red.sparse_lo, red.sparse_hi) == (lo_tb, flat.layout[w._sam_index + 1][0]) and red.sparse_lo > 0 and w._sam_sparse_reduce
        assert red.sparse_hi - red.sparse_lo >= w.numel()
        assert all(hi <= red.sparse_lo or lo >= red.sparse_hi for lo, hi in red.buckets)            # the table is in no dense bucket
        # regions from the end: 2 MMT layers | PrevPredEmbeddings (barrier) | 1 TextBert layer | its embeddings | [table] | classifier + pointer net (barrier)
        # | OCR input encoder | object input encoder -- down to offset 0: nothing is left for finish()
        assert len(red.regions) == 8 and red.barrier_regions == [2, 5] and red.regions[4][0] == red.sparse_hi and red.regions[5][1] == red.sparse_lo
        assert red.regions[5][0] == flat.range_of(model.ocr_ptr_net)[0] == red.regions[6][1]
        assert red.regions[6][0] == flat.range_of(model.linear_ocr_feat_to_mmt_in)[0] == red.regions[7][1] and red.regions[7][0] == 0
    if dist_on and not three:
        w = model.text_bert.embeddings.word_embeddings.weight
        assert tr.reducer.dense_lo == tr.flat.layout[1][0] >= w.numel() > 0 and w._sam_sparse_reduce and tr.reducer.overlap
        assert min(lo for lo, _ in tr.reducer.buckets) == tr.reducer.dense_lo and tr.reducer.check
        red = tr.reducer
        # regions, from the end of the buffer down: 2 MMT layers | pointer net + classifier + PrevPredEmbeddings (barrier) | 1 TextBert layer | its embeddings
        # | OCR input encoder | object input encoder, which ends where the row-sparse table begins
        assert len(red.regions) == 7 and red.barrier_region == 2 and red.barrier_names == {"txt", "obj", "ocr"}
        assert red.regions[0][1] == tr.flat.numel and all(a[0] == b[1] for a, b in zip(red.regions[:-1], red.regions[1:]))
        assert red.regions[2] == (tr.flat.range_of(model.ocr_ptr_net)[0], tr.flat.range_of(model.mmt.encoder)[0])
        assert any(lo == red.regions[-1][0] for lo, _ in red.buckets)     # bucket boundary at the low end of the regions
        assert red.regions[-2][0] == tr.flat.range_of(model.linear_ocr_feat_to_mmt_in)[0] and red.regions[-1][0] == tr.flat.range_of(model.linear_obj_feat_to_mmt_in)[0] == red.dense_lo
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    losses = [tr.step(clone_batch(batch)).item() for _ in range(4)]
    if dist_on:
        assert all(tr.reducer.done) and tr.reducer.barrier_seen == {"txt", "obj", "ocr"}      # every finality mark fired during the last backward
        assert tr.reducer.late_buckets == 0                                                   # ... and released every bucket before finish()
        red = tr.reducer                                     # the checker itself: release everything, then write late -> finish() must object
        red.begin_step(); red.region_done(0)
        tr.flat.grad[red.buckets[1][0] + 5] += 1.0
        try:
            red.finish()
            raise SystemExit("SAM_REDUCER_CHECK missed a write after release")
        except RuntimeError as e:
            assert "released before" in str(e), e
        red.begin_step(); tr.flat.zero_grad()
    res.append((losses, tr.flat.flat.clone()))
torch.cuda.synchronize()
(l1, p1), (l0, p0) = res
print("LOSSES", l1, l0)
assert all(abs(a - b) <= 2e-3 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
d = (p1 - p0).abs().max().item()
print("MAXDIFF", d)
assert d < 5e-3, d          # 4 Adam steps at lr 1e-3: a parameter moves <= 4e-3 in total; identical up to atomics order
print("DIST_OK")
"""


@pytest.mark.transport
@pytest.mark.parametrize("three_groups", [False, True])
def test_rccl_path_one_rank_matches_plain_trainer(tmp_path, three_groups):
    """SAM_FORCE_DIST=1: bucketed all-reduce on the side stream + the row-sparse word-embedding exchange run through RCCL in a 1-rank
    group and must train exactly like the reducer-less path.  three_groups: the shipped configs' optimizer layout, where the table sits in
    the middle of the flat buffer and the heads below it"""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAM_REPO=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", SAM_TEST_THREE_GROUPS="1" if three_groups else "0")
    from tests.util import run_child
    run_child([sys.executable, "-c", _DIST_SCRIPT], env, "DIST_OK", "rccl_one_rank_%s" % ("three_groups" if three_groups else "two_groups"))


def test_training_trajectory_matches_oracle_train_step(shapes=(20, 100, 50, 12), layers=("n", "s"), ctx=3, batch=3, check_descent=True):
    """SURVEY §8a a-18 end to end: six optimisation steps (forward, masked BCE, backward, clip 0.25, Adam, LambdaLR warm-up) of the HIP
    Trainer against the oracle's train_step (train.py:133-144 restated) from identical weights on identical batches, dropout off.
    (The arguments are for tools/fuzz_shapes.py, which calls this with random shapes.)"""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    model, ref = _small_full_model(ctx, layers, shapes)
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    tr = Trainer(model, base_lr=1e-3, seed=3)
    opt, sched = O.make_optimizer(ref, base_lr=1e-3)
    ref.train()
    batches = []
    for i in range(3):
        bd = make_batch(batch, *shapes, vocab=300, context=ctx, device="cpu", seed=40 + i)
        bd["question_indices"] = (bd["question_indices"] % 499 + 1) * bd["question_mask"]
        batches.append(bd)
    to_gpu = lambda bd: {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd.items()}
    l_hip, l_ref = [], []
    for step in range(6):
        bd = batches[step % 3]
        l_ref.append(O.train_step(ref, clone_batch(bd), opt, sched).item())
        l_hip.append(tr.step(to_gpu(clone_batch(bd))).item())
    assert all(abs(a - b) <= 0.02 * abs(b) for a, b in zip(l_hip, l_ref)), (l_hip, l_ref)
    assert not check_descent or l_ref[-1] < l_ref[0]          # (a property of the scenario, not of the product: the fuzzer's random shapes switch it off)
    # where the optimiser took the parameters: direction and size of the total update, per tensor family
    sd = tr.state_dict()["model_state_dict"]
    cos, rel = [], []
    for k, v0 in init.items():
        d_ref, d_hip = (ref.state_dict()[k] - v0).flatten().double(), (sd[k].cpu() - v0).flatten().double()
        if d_ref.norm() < 1e-6 or k.endswith(".key.bias"):       # key biases: softmax is invariant to them, their gradient is exactly 0 in
            continue                                               # exact arithmetic and pure rounding noise otherwise (Adam turns noise into +-lr steps)
        cos.append((k, float(torch.dot(d_ref, d_hip) / (d_ref.norm() * d_hip.norm() + 1e-30))))
        rel.append(float((d_ref - d_hip).norm() / d_ref.norm()))
    worst = min(cos, key=lambda t: t[1])
    # Adam normalises every coordinate by its own gradient history: coordinates whose gradient is at the bf16 noise level take
    # different +-lr steps, so agreement is asked of the bulk (median) and of the worst tensor's direction
    assert worst[1] > 0.8, worst
    assert sorted(c for _, c in cos)[len(cos) // 2] > 0.97 and sorted(rel)[len(rel) // 2] < 0.25


def test_sam4c_degenerate_samples_vs_oracle():
    """edge cases the reference's data can produce: an image with no OCR token and no object (everything padded), a one-word question,
    a sample whose decoding steps are all masked out of the loss, and an answer made only of copied OCR tokens"""
    from sam_textvqa_amd.params import prepare
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import masked_bce_loss
    shapes = (20, 100, 50, 12)
    model, ref = _small_full_model(3, ("n", "s", "s"), shapes)
    bd_cpu = make_batch(4, *shapes, vocab=300, context=3, device="cpu", seed=23)
    bd_cpu["pad_obj_mask"][0] = 0
    bd_cpu["pad_ocr_mask"][0] = 0                                       # sample 0: nothing to look at but the question
    bd_cpu["question_mask"][1] = 0
    bd_cpu["question_mask"][1, 0] = 1                                   # sample 1: one-word question
    bd_cpu["train_loss_mask"][2] = 0                                    # sample 2: contributes nothing to the loss
    bd_cpu["train_prev_inds"][3, 1:] = 300 + torch.arange(11) % 50      # sample 3: every previous prediction is a copied OCR token
    bd_cpu["question_indices"] = (bd_cpu["question_indices"] % 499 + 1) * bd_cpu["question_mask"]
    ref.train()
    out_ref = ref(clone_batch(bd_cpu))["textvqa_scores"]
    loss_ref = O.m4c_decoding_bce_with_mask_loss(out_ref, bd_cpu["targets"], bd_cpu["train_loss_mask"])
    loss_ref.backward()
    model.cuda().train()
    fp = prepare(model)
    bd = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd_cpu.items()}
    fp.zero_grad()
    out = model(bd)["textvqa_scores"]
    loss = masked_bce_loss(bd)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(fp.grad).all()
    within("degenerate samples: scores", score_err(out, out_ref), L["sam4c_scores"])
    within("degenerate samples: loss", abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), L["sam4c_loss"])
    assert (out[0, :, 300:] < -9000).all()                               # no OCR token: every pointer score is the literal -10000
    refp = dict(ref.named_parameters())
    biggest = max(p.grad.norm().item() for p in ref.parameters() if p.grad is not None)
    bad, worst = [], 0.0
    for pn, p in model.named_parameters():
        g_ref = refp[pn].grad
        if g_ref is None or g_ref.norm().item() < 1e-5 * biggest:
            continue
        e = ((p.grad.cpu().double() - g_ref.double()).norm() / g_ref.double().norm()).item()
        worst = max(worst, e)
        if e > L["sam4c_pgrad"]:
            bad.append((pn, round(e, 4)))
    within("degenerate samples: worst parameter-gradient norm error", worst, L["sam4c_pgrad"])
    assert not bad, bad


@pytest.mark.parametrize("which", ["BertSelfOutput", "BertOutput", "BertIntermediate"])
def test_stand_alone_sub_module_forward_backward_vs_oracle_class(which):
    """the sub-modules a maintainer may call by themselves -- BertSelfOutput / BertOutput (dense + dropout + residual + LayerNorm,
    sa_m4c.py:653, 680) and BertIntermediate (dense + erf-GELU, :678) -- against the oracle's classes of the same name on the same bf16-rounded weights and
    inputs: output, both input gradients and every parameter gradient (single-kernel outputs at the kernel bound, 1e-3 of max + one bf16 ulp).  Their arithmetic is the library's
    (GEMM epilogues, LayerNorm kernels, sam_rowvec_bf16): the test also asserts that no torch-native kernel ran between the C-ABI calls' inputs and outputs by
    checking bit-equality with the fused encoder layer's own launches on the same operands."""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd import _capi as capi, ops
    from tests.util import assert_close_bf16
    torch.manual_seed(11)
    cfg = dict(hidden_size=768, intermediate_size=3072, hidden_dropout_prob=0.0, layer_norm_eps=1e-12)
    o_mod = getattr(O, which)(O.BertConfig.from_dict(cfg))
    with torch.no_grad():
        for p in o_mod.parameters():
            p.copy_((p + 0.02 * torch.randn_like(p)) if p.dim() == 1 else p.to(torch.bfloat16).float())
    mod = getattr(M, which)(M.BertConfig.from_dict(cfg))
    mod.load_state_dict(o_mod.state_dict())
    mod.cuda().train()
    B, N = 3, 70
    k_in = 3072 if which == "BertOutput" else 768
    x = torch.randn(B, N, k_in).to(torch.bfloat16)
    res = torch.randn(B, N, 768).to(torch.bfloat16)
    n_out = 3072 if which == "BertIntermediate" else 768
    g = torch.randn(B, N, n_out).to(torch.bfloat16)
    xo, ro = x.float().requires_grad_(True), res.float().requires_grad_(True)
    xg, rg = x.cuda().requires_grad_(True), res.cuda().requires_grad_(True)
    if which == "BertIntermediate":
        yo, yg = o_mod(xo), mod(xg)
    else:
        yo, yg = o_mod(xo, ro), mod(xg, rg)
    assert yg.dtype == torch.bfloat16 and yg.shape == yo.shape
    (yo * g.float()).sum().backward()
    (yg.float() * g.cuda().float()).sum().backward()
    # BertIntermediate's forward is ONE kernel: the kernel bound.  Everything else crosses one bf16 intermediate (the pre-LayerNorm sum z, which the backward
    # reads; dy * gelu' before the dgrad GEMM): limits at about twice the achieved error, as in table L
    if which == "BertIntermediate":
        assert_close_bf16(yg, yo, name=which + " out")
        lim = dict(dx=0.006, p=0.006)
    else:
        within(which + " out", rel_err(yg, yo), 0.011)
        within(which + " d residual", rel_err(rg.grad, ro.grad), 0.011)
        lim = dict(dx=0.011, p=0.011)
    within(which + " dx", rel_err(xg.grad, xo.grad), lim["dx"])
    for (k, po), (_, pg) in zip(o_mod.named_parameters(), mod.named_parameters()):
        within("%s grad %s" % (which, k), rel_err(pg.grad, po.grad), lim["p"])
    # the same launches as the fused layer: bit-identical to calling the library ops directly on the same operands
    with torch.no_grad():
        w = mod.dense.weight.detach().to(torch.bfloat16)
        x2 = x.cuda().view(-1, k_in)
        if which == "BertIntermediate":
            direct = ops.gemm(x2, w, epilogue=capi.EPI_BIAS_GELU_GRAD, bias=mod.dense.bias.detach())
        else:
            z = ops.gemm(x2, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=mod.dense.bias.detach(), residual=res.cuda().view(-1, 768), p_drop=0.0)
            direct = ops.layernorm_fwd(z, mod.LayerNorm.weight.detach(), mod.LayerNorm.bias.detach(), 1e-12)[0]
        assert torch.equal(direct.view_as(yg), yg.detach())
    # dropout on: the training form draws from the library's hidden-state stream (keep rate ~0.9 of the dense output), eval is deterministic
    if which != "BertIntermediate":
        mod.dropout_p = 0.1
        y1, y2 = mod(xg.detach(), rg.detach()), mod(xg.detach(), rg.detach())
        assert not torch.equal(y1, y2)
        mod.eval()
        assert torch.equal(mod(xg.detach(), rg.detach()), yg.detach())


def test_long_trajectory_does_not_drift_from_the_oracle():
    """bf16 storage between kernels sets the model-level error (0.4-2 % of max); this checks that it does not GROW with training.  48
    optimisation steps of the HIP Trainer (captured step) and of the oracle's train_step from the same weights on the same 3 rotating batches, dropout off:
    the loss curves stay together (every step within 3 %, the last 20 within 2 % on average), and the relative distance between the two parameter vectors'
    total updates at step 48 is no larger than 1.5x what it was at step 12 -- rounding noise accumulating like a random walk through Adam would grow
    ~2x over that span, a systematic bias 4x."""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    shapes, layers = (6, 14, 10, 4), ("n", "s")
    model, ref = _small_full_model(3, layers, shapes, ffn=1024)          # (a narrower FFN: the host oracle's Adam over the parameters is this test's cost)
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    tr = Trainer(model, base_lr=2e-4, seed=3, use_graph=True)
    opt, sched = O.make_optimizer(ref, base_lr=2e-4)
    ref.train()
    batches = []
    for i in range(3):
        bd = make_batch(4, *shapes, vocab=300, context=3, device="cpu", seed=140 + i)
        bd["question_indices"] = (bd["question_indices"] % 499 + 1) * bd["question_mask"]
        batches.append(bd)
    to_gpu = lambda bd: {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in bd.items()}
    gpu_batches = [to_gpu(b) for b in batches]

    def distance():
        sd = tr.state_dict()["model_state_dict"]
        num = den = 0.0
        for k, v0 in init.items():
            if k.endswith(".key.bias"):
                continue
            d_ref, d_hip = (ref.state_dict()[k] - v0).double(), (sd[k].cpu() - v0).double()
            num += float((d_ref - d_hip).pow(2).sum()); den += float(d_ref.pow(2).sum())
        return (num / den) ** 0.5

    l_hip, l_ref, dist = [], [], {}
    for step in range(48):
        l_ref.append(O.train_step(ref, clone_batch(batches[step % 3]), opt, sched).item())
        l_hip.append(tr.step(clone_batch(gpu_batches[step % 3])).item())
        if step + 1 in (12, 48):
            dist[step + 1] = distance()
    rel = [abs(a - b) / abs(b) for a, b in zip(l_hip, l_ref)]
    print("PARITY 48-step trajectory: loss rel err max %.3e, mean of last 20 %.3e; update distance at 12 / 48 steps %.4f / %.4f; loss %.3f -> %.3f"
          % (max(rel), sum(rel[-20:]) / 20, dist[12], dist[48], l_ref[0], l_ref[-1]))
    assert max(rel) < 0.01 and sum(rel[-20:]) / 20 < 0.005, (max(rel), rel[-5:])          # achieved 5e-4 / 7e-5
    assert dist[48] < 1.5 * dist[12] + 0.005, dist                                          # achieved 0.005 / 0.005: no growth


@pytest.mark.parametrize("train_dropout", [False, True])
def test_spatial_layer_use_bias_head_mask_output_attentions_vs_oracle(train_dropout):
    """VERDICT r5 missing #4: `use_bias` head biases (sa_m4c.py:439-443, 600-603), `head_mask` (:591-592) and `output_attentions` (:604-609) on the module-level
    API of SpatialBertLayer, against the oracle's layer (itself pinned for these switches by tests/golden/layer_small_switches.npz): output, returned
    attention_probs [B, H, N, N], input gradient, every parameter gradient incl. `attention.self.biases.weight`; a checkpoint with the bias row loads.
    train_dropout: attention dropout on -- the returned probabilities are the DROPPED ones (zeros where the kernel's keep bits are clear, 1 / (1 - p) elsewhere)."""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import mmt_config_dict
    torch.manual_seed(5)
    T, n_obj, n_ocr, n_dec, B = 6, 22, 13, 5, 2
    cfgd = mmt_config_dict(3, ("s",), n_dec=n_dec, T=T, n_obj=n_obj, n_ocr=n_ocr)
    cfgd.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.1 if train_dropout else 0.0, use_bias=True, output_attentions=True)
    o_layer = bf16_round_module(O.SpatialBertLayer(O.BertConfig.from_dict(cfgd)))
    with torch.no_grad():
        o_layer.attention.self.biases.weight.copy_(0.3 * torch.randn(1, 768))
    layer = M.SpatialBertLayer(M.BertConfig.from_dict(cfgd))
    assert "attention.self.biases.weight" in layer.state_dict()
    layer.load_state_dict(o_layer.state_dict())
    layer.cuda()
    n = T + n_obj + n_ocr + n_dec
    from oracle import spatial_graph as SG
    rng = np.random.RandomState(1)
    boxes = rng.rand(B, n_obj + n_ocr, 2) * 0.8
    boxes = np.concatenate([boxes, boxes + 0.02 + rng.rand(B, n_obj + n_ocr, 2) * 0.15], -1)
    boxes[1, -3:] = 0
    adj = torch.from_numpy(np.stack([SG.compose(SG.relation_codes(b), 3) for b in boxes]))
    qm, om, cm = torch.ones(B, T, dtype=torch.long), torch.ones(B, n_obj, dtype=torch.long), torch.ones(B, n_ocr, dtype=torch.long)
    qm[0, 4:] = 0; cm[1, -3:] = 0
    ext = O.MMT.extended_attention_mask(qm, om, cm, n_dec)
    head_mask = torch.tensor([1.0, 0.5, 1.25, 0.0, 1.0, 0.75, 1.0, 1.0, 1.5, 1.0, 0.25, 1.0]).view(1, 12, 1, 1)
    x = torch.randn(B, n, 768).to(torch.bfloat16)
    gout = torch.randn(B, n, 768)
    xg = x.cuda().requires_grad_(True)
    if train_dropout:
        layer.train()
        yg, pg = layer(xg, ext.cuda(), adj.cuda(), head_mask.cuda())
        assert pg.shape == (B, 12, n, n) and pg.dtype == torch.float32
        # the oracle cannot draw the library's mask: replay it with the probabilities' own zero pattern (keep = dropped probs != 0 where the undropped are)
        o_layer.eval()
        xo = x.float().requires_grad_(True)
        _, po = o_layer(xo, ext, adj, head_mask)
        live = po > 1e-6                                         # (head 3 is masked off entirely; tiny probabilities are skipped: their kept value may round to 0)
        kept = pg.cpu()[live] != 0
        assert abs(kept.float().mean().item() - 0.9) < 0.02
        inv_keep = 1.0 / (1.0 - round(0.1 * 65536) / 65536.0)
        assert torch.allclose(pg.cpu()[live][kept], po[live][kept] * inv_keep, rtol=2e-2, atol=2e-4)
        assert (pg.cpu()[:, 3] == 0).all()
        return
    layer.eval()
    o_layer.eval()
    xo = x.float().requires_grad_(True)
    yo, po = o_layer(xo, ext, adj, head_mask)
    (yo * gout).sum().backward()
    yg, pg = layer(xg, ext.cuda(), adj.cuda(), head_mask.cuda())
    (yg.float() * gout.cuda()).sum().backward()
    within("switches layer out", rel_err(yg, yo), L["layer_out_o"])
    within("switches layer dx", rel_err(xg.grad, xo.grad), L["layer_dh_o"])
    within("switches attention_probs", rel_err(pg, po), 4e-3)            # (bf16 q, k rows: scores to ~1e-3)
    assert (pg.cpu()[:, 3] == 0).all() and (pg.cpu().sum(-1)[:, 0] <= 1.0 + 1e-3).all()
    for (k, p_o), (_, p_g) in zip(o_layer.named_parameters(), layer.named_parameters()):
        if k.endswith("key.bias"):
            continue                                                        # (softmax is invariant to it: exact 0 in exact arithmetic, rounding noise otherwise)
        within("switches grad " + k, rel_err(p_g.grad, p_o.grad), 0.02)
    # without the switches the same weights run the fused single-node path; the bias row and the head factors are what differs
    plain = layer(xg.detach(), ext.cuda(), adj.cuda())
    assert len(plain) == 2 and plain[1].shape == (B, 12, n, n)           # output_attentions is a config switch: still returned
    with pytest.raises(NotImplementedError):
        layer(xg.detach(), ext.cuda(), adj.cuda(), torch.ones(B, 12, n, n).cuda())
