"""CPU: the oracle restatement vs golden vectors produced by the reference itself
(tests/golden/make_golden.py).  fp32 throughout; tolerance = a few fp32 ulps of accumulated
reordering (1e-5 relative to the tensor's max)."""
import numpy as np
import pytest
import torch

from oracle import sa_m4c_oracle as O
from oracle import spatial_graph as SG
from tests import oracle_cases as OC
from tests.golden import common as C


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = np.abs(b).max()
    err = np.abs(a - b).max()
    # 1e-6 absolute floor: e.g. d(key.bias) is mathematically 0 (softmax shift invariance), only fp32 noise
    assert err <= tol * scale + 1e-6, "abs err %.3g > %.3g*%.3g + 1e-6" % (err, tol, scale)


def test_spatial_graph_matches_reference():
    g = OC.load("spatial_graph")
    for nm in ("known6", "grid", "rnd60", "cross"):
        with np.errstate(all="ignore"):
            codes = SG.relation_codes(g[nm + ".boxes"], 0.5)
        for k in SG.SHARE_KEYS:
            np.testing.assert_array_equal(codes[k], g["%s.code%s" % (nm, k)], err_msg=nm + k)
        for ctx in (1, 3, 5, 7, 9):
            np.testing.assert_array_equal(SG.compose(codes, ctx), g["%s.ctx%d" % (nm, ctx)])


def test_spatial_graph_known_answer():
    # SURVEY.md §8c vector captured from the reference
    boxes = [[.1, .1, .5, .5], [.2, .2, .3, .3], [.6, .1, .8, .3], [.1, .6, .3, .9], [.12, .12, .5, .5], [0, 0, 0, 0]]
    c = SG.relation_codes(boxes)
    assert c["1"].tolist() == [[12, 1, 7, 10, 3, 0], [2, 12, 7, 10, 2, 0], [11, 11, 12, 0, 11, 0],
                               [6, 6, 0, 12, 6, 0], [3, 1, 7, 10, 12, 0], [0, 0, 0, 0, 0, 0]]
    assert c["31"][0].tolist() == [0, 0, 8, 11, 0, 0] and c["32"][2].tolist() == [10, 10, 0, 0, 10, 0]
    assert c["51"][3].tolist() == [8, 8, 0, 0, 8, 0]
    m = SG.replace_maps()
    assert m["31"][11] == 4 and m["32"][4] == 11 and m["51"][10] == 4 and m["52"][5] == 11 and m["91"] == m["92"]


def test_primitives():
    g = OC.load("primitives")
    x = torch.from_numpy(C.det_uniform("prim.x", (7, 96), -4, 4)).requires_grad_(True)
    ln = O.BertLayerNorm(96)
    C.fill_state_dict(ln, 0.1, prefix="prim.LayerNorm.")
    gy = torch.from_numpy(C.det_uniform("prim.gy", (7, 96)))
    y = ln(x); (y * gy).sum().backward()
    close(y.detach(), g["ln_out"]); close(x.grad, g["ln_dx"]); close(ln.weight.grad, g["ln_dw"]); close(ln.bias.grad, g["ln_db"])
    x2 = x.detach().clone().requires_grad_(True)
    z = O.gelu(x2); (z * gy).sum().backward()
    close(z.detach(), g["gelu_out"]); close(x2.grad, g["gelu_dx"])


@pytest.mark.parametrize("name", list(C.LAYER_CASES))
def test_spatial_layer(name):
    layer, hidden, ext, adj, gout, g = OC.layer_case(name)
    out = layer(hidden, ext, adj)[0]
    (out * gout).sum().backward()
    close(out.detach(), g["out"]); close(hidden.grad, g["d_hidden"])
    close(layer.attention.self(hidden.detach(), ext, adj)[0].detach(), g["ctx"])
    for pn, p in layer.named_parameters():
        if "grad." + pn in g:
            close(p.grad, g["grad." + pn], 5e-5)


def test_spatial_layer_with_use_bias_head_mask_and_output_attentions():
    """the switches no shipped config turns on (sa_m4c.py:439-443 / 600-603 head biases, :591-592 head_mask, :604-609 output_attentions), pinned by a golden of the
    reference itself: layer output, the returned attention_probs (after head mask), input gradient and every parameter gradient incl. `biases.weight`"""
    name, case = "layer_small_switches", C.LAYER_CASES["layer_small_c3"]
    d = case["dims"]
    cfg = O.BertConfig.from_dict(C.mmt_config_dict(d, ["s"], case["ctx"], case["quadrants"], use_bias=True, output_attentions=True))
    layer = O.SpatialBertLayer(cfg).eval()
    C.fill_state_dict(layer, d["ws"], prefix=name + ".")
    g = OC.load(name)
    n = d["T"] + d["n_obj"] + d["n_ocr"] + d["n_dec"]
    hidden = torch.from_numpy(C.det_uniform(name + ".hidden", (d["B"], n, d["D"]))).requires_grad_(True)
    out, probs = layer(hidden, OC.ext_mask(d), torch.from_numpy(g["adj"]), torch.from_numpy(g["head_mask"]))
    (out * torch.from_numpy(C.det_uniform(name + ".gout", tuple(out.shape)))).sum().backward()
    close(out.detach(), g["out"]); close(probs.detach(), g["probs"]); close(hidden.grad, g["d_hidden"])
    assert "grad.attention.self.biases.weight" in g and (probs.detach()[:, 3] == 0).all()
    for pn, p in layer.named_parameters():
        close(p.grad, g["grad." + pn], 5e-5)


def test_spatial_layer_faithful_mode_identical():
    layer, hidden, ext, adj, _, g = OC.layer_case("layer_small_c3")
    layer.attention.self.faithful = True
    close(layer(hidden, ext, adj)[0].detach(), g["out"])


def test_allow_mask_equals_additive_masks():
    # boolean truth table == reference's min(attention_mask, spatial_mask) == 0
    for name in ("layer_small_c3", "layer_small_c1_q"):
        layer, hidden, ext, adj, _, _ = OC.layer_case(name)
        d = C.LAYER_CASES[name]["dims"]
        att = layer.attention.self
        sp = att.build_spatial_mask(ext, adj, hidden.size(1))
        combined = torch.min(ext, sp)
        allow = O.allow_mask(torch.from_numpy(OC.key_valid(d)), d["T"], d["n_obj"] + d["n_ocr"], d["n_dec"], adj,
                             C.LAYER_CASES[name]["quadrants"], d["H"])
        assert torch.equal(allow, combined == 0)
        plain = O.allow_mask(torch.from_numpy(OC.key_valid(d)), d["T"], d["n_obj"] + d["n_ocr"], d["n_dec"], None, (), d["H"])
        assert torch.equal(plain, (ext == 0).expand_as(plain))


def test_ptr_net():
    g = OC.load("ptr_net")
    d = C.SMALL
    ptr = O.OcrPtrNet(d["D"], d["D"])
    C.fill_state_dict(ptr, d["ws"], prefix="ptr.")
    qi = torch.from_numpy(C.det_uniform("ptr.q", (d["B"], d["n_dec"], d["D"]))).requires_grad_(True)
    ki = torch.from_numpy(C.det_uniform("ptr.k", (d["B"], d["n_ocr"], d["D"]))).requires_grad_(True)
    sc = ptr(qi, ki, torch.from_numpy(C.pad_mask(d["n_ocr_valid"], d["n_ocr"])))
    (sc * torch.from_numpy(C.det_uniform("ptr.gs", tuple(sc.shape)))).sum().backward()
    close(sc.detach(), g["scores"]); close(qi.grad, g["d_q"]); close(ki.grad, g["d_k"])
    for pn, p in ptr.named_parameters():
        close(p.grad, g["grad." + pn])


@pytest.mark.parametrize("name", list(C.MMT_CASES))
def test_mmt(name):
    mmt, bd, leaves, gout, g = OC.mmt_case(name)
    seq = mmt(bd, fixed_ans_emb=leaves["fixed_ans_emb"])["mmt_seq_output"]
    (seq * gout).sum().backward()
    close(seq.detach(), g["seq"], 5e-5)
    for k, v in leaves.items():
        close(v.grad, g["d_" + k], 1e-4)
    for pn, p in mmt.named_parameters():
        if "grad." + pn in g:
            ref = g["grad." + pn]
            close(p.grad[: ref.shape[0]] if ref.shape != tuple(p.shape) else p.grad, ref, 2e-4)


def test_sam4c_train_and_greedy():
    name = "sam4c_small_c3"
    g = OC.load(name)
    d = C.SAM4C_CASES[name]["dims"]
    mcfg, tcfg = OC.sam4c_configs(name)
    model = O.SAM4C(mcfg, tcfg, num_answers=d["V"], bos_idx=1)
    C.fill_state_dict(model, d["ws"], prefix=name + ".")
    model.train()
    bd = OC.sam4c_batch(name, torch.from_numpy(g["adj"]))
    scores = model(bd)["textvqa_scores"]
    loss = O.m4c_decoding_bce_with_mask_loss(scores, bd["targets"], bd["train_loss_mask"])
    loss.backward()
    close(scores.detach(), g["scores"], 5e-5); close(loss.detach(), g["loss"], 5e-5)
    for pn, p in model.named_parameters():
        if "grad." + pn in g:
            ref = g["grad." + pn]
            mine = p.grad if ref.shape == tuple(p.shape) else p.grad.reshape(p.shape[0], -1)[:8]
            close(mine, ref, 3e-4)
    assert [len(gr["params"]) for gr in model.get_optimizer_parameters(1e-4)] == g["group_sizes"].tolist()
    model.eval()
    bd2 = OC.sam4c_batch(name, torch.from_numpy(g["adj"]))
    with torch.no_grad():
        gs = model(bd2)["textvqa_scores"]
    close(gs, g["greedy_scores"], 5e-5)
    np.testing.assert_array_equal(bd2["train_prev_inds"].numpy(), g["greedy_prev_inds"])


@pytest.mark.parametrize("tag", ["k3", "early"])
def test_beam_search_against_the_reference_beam_search(tag):
    """oracle/beam_search.py == the reference's BeamSearch class driven by SAM4C._forward_beam_search (sam/beam_search.py:6-181,
    sa_m4c.py:304-314): every step's scores, surviving sequences and cumulative scores, and the final results"""
    from oracle import beam_search as BS
    name = "sam4c_small_c3"
    g = OC.load(name)
    d = C.SAM4C_CASES[name]["dims"]
    mcfg, tcfg = OC.sam4c_configs(name)
    model = O.SAM4C(mcfg, tcfg, num_answers=d["V"], bos_idx=1).eval()
    C.fill_state_dict(model, d["ws"], prefix=name + ".")
    beam, eos, nsteps = (int(v) for v in g["beam.%s.cfg" % tag])
    bd = OC.sam4c_batch(name, torch.from_numpy(g["adj"]))
    bd["train_prev_inds"] = torch.zeros_like(bd["train_prev_inds"]); bd["train_prev_inds"][:, 0] = 1
    bd["question_id"] = torch.arange(d["B"]) + 100
    with torch.no_grad():
        res, _, trace = BS.forward_beam_search(model, bd, beam, eos)
    assert len(trace) == nsteps
    for t, (sc, pi, tk) in enumerate(trace):
        close(sc, g["beam.%s.step%d.scores" % (tag, t)], 5e-5)
        np.testing.assert_array_equal(pi.numpy(), g["beam.%s.step%d.prev_inds" % (tag, t)])
        close(tk.float(), g["beam.%s.step%d.topkscores" % (tag, t)], 5e-5)
    np.testing.assert_array_equal(res["complete_seqs"].numpy(), g["beam.%s.complete_seqs" % tag])
    np.testing.assert_array_equal(res["question_id"].numpy(), g["beam.%s.question_id" % tag])
    close(res["topkscores"], g["beam.%s.topkscores" % tag], 5e-5)
    close(res["textvqa_scores"], g["beam.%s.final_scores" % tag], 5e-5)
    if tag == "early":          # the completed-beam branch was really taken: some beam carries EOS before the last step
        assert (g["beam.early.step0.prev_inds"][:, 1] == eos).any()


def test_lr_schedule_and_loss_normaliser():
    # task_utils.py:48-54: warm-up 0.2 -> 1 over 1000 iters, x0.1 at 14k and 19k
    assert O.lr_lambda(0) == pytest.approx(0.2) and O.lr_lambda(500) == pytest.approx(0.6) and O.lr_lambda(1000) == 1.0
    assert O.lr_lambda(1001) == 1.0 and O.lr_lambda(14000) == pytest.approx(0.1) and O.lr_lambda(19000) == pytest.approx(0.01)
    s = torch.zeros(1, 2, 3); t = torch.zeros(1, 2, 3)
    assert O.m4c_decoding_bce_with_mask_loss(s, t, torch.zeros(1, 2)).item() == 0.0   # count clamps to 1
