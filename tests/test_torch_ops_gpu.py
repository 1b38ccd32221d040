"""GPU: the PyTorch-ROCm custom-op layer (torch.ops.sam_hip.*, csrc_torch/sam_torch_ops.cpp) -- fine-grained ops against fp32 references,
and the coarse per-layer ops against the per-kernel ctypes route (same kernels in the same order: bit-identical)."""
import math

import pytest
import torch

from tests.util import assert_close_bf16

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def test_fine_grained_ops_through_torch_ops():
    from sam_textvqa_amd import _capi as capi, ops, torchops
    ns = torchops.ns()
    x, w = rnd((728, 768), 1), rnd((3072, 768), 2, 0.05)
    b = torch.randn(3072, generator=torch.Generator().manual_seed(3)) * 0.1
    y, aux = torch.ops.sam_hip.linear(x.cuda(), w.cuda(), b.cuda(), capi.EPI_BIAS_GELU, None, None, True, 0.0, 0, 0, True, False)
    pre = x.float() @ w.float().t() + b
    assert_close_bf16(aux, pre, name="pre-activation")
    assert_close_bf16(y, 0.5 * pre * (1 + torch.erf(pre / math.sqrt(2))), name="gelu")
    # dgrad layout + residual through the same op
    wT, res = rnd((768, 3072), 4, 0.05), rnd((728, 3072), 5)
    z, none = ns.linear(x.cuda(), wT.cuda(), None, capi.EPI_BIAS_DROPOUT_RES, res.cuda(), None, False, 0.0, 0, 0, False, False)
    assert none.numel() == 0
    assert_close_bf16(z, x.float() @ wT.float() + res.float(), name="dgrad+res")
    # layernorm fwd / bwd
    g, be = torch.rand(768) + 0.5, torch.randn(768) * 0.1
    xf = rnd((300, 768), 6)
    yl, mean, rstd = ns.layernorm_fwd(xf.cuda(), g.cuda(), be.cuda(), 1e-12)
    xr = xf.float().requires_grad_(True)
    mu = xr.mean(-1, keepdim=True)
    ref = (xr - mu) / torch.sqrt(((xr - mu) ** 2).mean(-1, keepdim=True) + 1e-12) * g + be
    assert_close_bf16(yl, ref.detach(), name="ln fwd")
    dy = rnd((300, 768), 7)
    ref.backward(dy.float())
    dx, dg, db = ns.layernorm_bwd(dy.cuda(), xf.cuda(), mean, rstd, g.cuda())
    assert_close_bf16(dx, xr.grad, name="ln bwd dx")
    assert_close_bf16(db, dy.float().sum(0), ulps=0, name="ln bwd dbeta")
    # fused attention: equals the ctypes route bit for bit (same entry point underneath)
    B, N, H = 3, 182, 12
    qkv = rnd((B * N, 3 * 768), 8).cuda()
    kv = torch.ones(B, N - 12, dtype=torch.uint8); kv[1, 100:] = 0
    allow = ops.mask_bits_prefix_lm(kv.cuda(), 12)
    o1, l1, k1 = ns.spatial_attn_fwd(qkv, allow, B, H, 0.125, 0.1, 5, 9)
    o2, l2, k2 = ops.attn_fwd(qkv, allow, B, H, 0.125, 0.1, 5, 9)
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(k1, k2)
    do = rnd((B * N, 768), 9).cuda()
    assert torch.equal(ns.spatial_attn_bwd(do, qkv, l1, allow, k1, B, H, 0.125, 0.1), ops.attn_bwd(do, qkv, l2, allow, k2, B, H, 0.125, 0.1))
    # argument errors surface as RuntimeError, not as a crash
    with pytest.raises(RuntimeError):
        ns.layernorm_fwd(xf, g.cuda(), be.cuda(), 1e-12)          # CPU tensor
    with pytest.raises(RuntimeError):
        ns.linear(x.cuda()[:, :100], w.cuda(), None, 0, None, None, False, 0.0, 0, 0, True, False)        # K mismatch / K % 8


@pytest.mark.parametrize("kind", ["s", "n"])
def test_coarse_encoder_layer_ops_equal_the_per_kernel_route(kind, monkeypatch):
    """EncoderLayerFn through torch.ops.sam_hip.encoder_layer_fwd/_bwd vs through ~35 ctypes launches: outputs, input gradient and every
    parameter gradient bit-identical, dropout on (same counter-based seeds)"""
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd import torchops
    from sam_textvqa_amd.autograd import dropout_clock
    from sam_textvqa_amd.params import prepare
    from sam_textvqa_amd.synthetic import make_batch, mmt_config_dict
    torch.manual_seed(0)
    cfg = M.BertConfig.from_dict(mmt_config_dict(3, (kind,)))
    layer = (M.SpatialBertLayer if kind == "s" else M.BertLayer)(cfg).cuda().train()
    fp = prepare(layer)
    B, N = 4, 182
    bd = make_batch(B, vocab=100, device="cuda", seed=4)
    x = torch.randn(B, N, 768, device="cuda").to(torch.bfloat16)
    key_valid = torch.cat([bd["question_mask"], bd["pad_obj_mask"], bd["pad_ocr_mask"]], 1)
    allow = M.AllowBits(M.ops.mask_bits_prefix_lm(key_valid.to(torch.uint8).contiguous(), 12))
    gout = torch.randn(B, N, 768, device="cuda")
    res = []
    for coarse in (True, False):
        monkeypatch.setattr(torchops, "enabled", lambda c=coarse: c)
        dropout_clock.manual_seed(77)
        fp.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = layer(xi, allow, bd["spatial_adj_matrices"]["3"])[0] if kind == "s" else layer(xi, allow)[0]
        (y.float() * gout).sum().backward()
        res.append((y.detach().clone(), xi.grad.clone(), fp.grad.clone()))
    (y1, dx1, g1), (y0, dx0, g0) = res
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0) and torch.equal(g1, g0)
    assert g1.abs().sum().item() > 0 and dx1.abs().sum().item() > 0


def test_device_side_rng_state_equals_by_value_offsets():
    """sam_set_rng_state: a launch with by-value (seed, offset) under the device state {S, base} draws the masks of the by-value pair
    (S, base + offset) -- forward GEMM epilogue (4-wave and 8-wave kernels), LayerNorm backward (which regenerates that mask), fused attention,
    previous-prediction gather.  This is what lets a captured hipGraph draw fresh, forward/backward-consistent masks on every replay."""
    from sam_textvqa_amd import _capi as capi, ops
    S, base, off = 0x1234567, 5 << 20, 7
    state = torch.tensor([S, base], dtype=torch.int64, device="cuda")
    x, w, res = rnd((1000, 768), 1).cuda(), rnd((768, 768), 2, 0.05).cuda(), rnd((1000, 768), 3).cuda()
    b = torch.randn(768, device="cuda")
    for tile in (128, 1192):
        want = ops.gemm(x, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b, residual=res, p_drop=0.1, seed=S, offset=base + off, force_tile=tile)
        ops.set_rng_state(state)
        try:
            got = ops.gemm(x, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b, residual=res, p_drop=0.1, seed=999, offset=off, force_tile=tile)
        finally:
            ops.set_rng_state(None)
        assert torch.equal(got, want)
        other = ops.gemm(x, w, epilogue=capi.EPI_BIAS_DROPOUT_RES, bias=b, residual=res, p_drop=0.1, seed=999, offset=off, force_tile=tile)
        assert not torch.equal(other, want)                     # back to by-value
    # LayerNorm backward's dropped gradient uses the same stream
    dy, z = rnd((1000, 768), 4).cuda(), rnd((1000, 768), 5).cuda()
    g = torch.ones(768, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(z, g, torch.zeros(768, device="cuda"), 1e-12)
    dg, db = torch.zeros(768, device="cuda"), torch.zeros(768, device="cuda")
    _, want = ops.layernorm_bwd(dy, z, mean, rstd, g, dg, db, want_dropped=True, p_drop=0.1, seed=S, offset=base + off)
    ops.set_rng_state(state)
    try:
        _, got = ops.layernorm_bwd(dy, z, mean, rstd, g, dg, db, want_dropped=True, p_drop=0.1, seed=999, offset=off)
        B, N, H = 2, 182, 12
        qkv = rnd((B * N, 3 * 768), 8).cuda()
        allow = ops.mask_bits_prefix_lm(torch.ones(B, N - 12, dtype=torch.uint8, device="cuda"), 12)
        _, _, keep_got = ops.attn_fwd(qkv, allow, B, H, 0.125, 0.1, 999, off)
    finally:
        ops.set_rng_state(None)
    assert torch.equal(got, want)
    _, _, keep_want = ops.attn_fwd(qkv, allow, B, H, 0.125, 0.1, S, base + off)
    assert torch.equal(keep_got, keep_want)


def test_training_step_captured_as_a_hipgraph():
    """Trainer(use_graph=True): the step is captured once and replayed.  Dropout off: the replayed trajectory equals the eager one (same
    kernels, learning-rate schedule and Adam bias corrections read from device memory); dropout on: every replay draws new masks."""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    from tests.test_model_gpu import _small_full_model
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    runs = []
    for use_graph in (False, True):
        model, _ = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
        tr = Trainer(model, base_lr=1e-3, seed=3, use_graph=use_graph, schedule=dict(warmup_iters=4))      # LR changes every step
        losses = [tr.step(clone_batch(batch)).item() for _ in range(7)]
        assert (tr._graph is not None) == use_graph and tr.global_step == 7
        assert tr._pending == (use_graph and tr.pipeline_update)      # (pipeline_update: the captured step leaves its update to the next replay's head ...
        tr.flush_update()                                              # ... or to whoever needs the parameters first; a no-op otherwise)
        assert not tr._pending
        runs.append((losses, tr.flat.flat.clone(), tr.exp_avg_sq.clone()))
    (l0, p0, v0), (l1, p1, v1) = runs
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(l0, l1)), (l0, l1)      # (two scatter kernels use fp32 atomics: not bit-reproducible)
    assert (p0 - p1).abs().max().item() < 5e-3 and l1[-1] < 0.7 * l1[0]        # 7 Adam steps at lr 1e-3: a parameter moves <= 7e-3 in total
    # a batch of another shape falls back to the eager path without disturbing the graph
    small = make_batch(2, vocab=300, device="cuda", seed=5)
    small["question_indices"] = small["question_indices"] % 500
    assert torch.isfinite(tr.step(clone_batch(small))) and tr.global_step == 8
    assert torch.isfinite(tr.step(clone_batch(batch))) and tr.global_step == 9
    # dropout on: same batch, consecutive replays -> different masks -> different losses; and the loss still goes down
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd.synthetic import mmt_config_dict, text_bert_config_dict
    torch.manual_seed(1)
    model = M.SAM4C(M.BertConfig.from_dict(mmt_config_dict(3, ("n", "s"))), M.BertConfig.from_dict(dict(text_bert_config_dict(), num_hidden_layers=1, vocab_size=500)),
                    num_answers=300, bos_idx=1)
    tr = Trainer(model, base_lr=2e-4, seed=3, use_graph=True)
    before = tr.flat.flat.clone()
    tr.step(clone_batch(batch)); tr.step(clone_batch(batch))
    tr.flush_update()
    frozen = tr.flat.flat.clone()
    a = tr.step(clone_batch(batch)).item()
    tr.flush_update()
    tr.flat.flat.copy_(frozen); tr.flat.refresh_shadows()          # same weights, same batch, next replay: only the masks differ
    b = tr.step(clone_batch(batch)).item()
    assert tr._graph is not None and a != b and abs(a - b) < 0.2 * abs(a), (a, b)
    ls = [tr.step(clone_batch(batch)).item() for _ in range(30)]
    assert ls[-1] < 0.8 * ls[0] and not torch.equal(before, tr.flat.flat)


def test_update_at_the_head_of_the_next_replay_is_the_same_training_run():
    """Trainer(pipeline_update=True): clip + Adam of step k open replay k + 1 (two gated pieces, the MMT's under TextBert's forward) instead of closing
    replay k.  Dropout off: losses, parameters, moments and bf16 shadows after flush_update() equal the closing-update run's (identical kernels on
    identical operands; the two scatter kernels' fp32 atomics are the only noise); state_dict() and an eval-mode forward flush by themselves; an eager
    step of another shape in between and the replay after it (gate closed) keep the trajectory."""
    from sam_textvqa_amd.synthetic import clone_batch, make_batch
    from sam_textvqa_amd.trainer import Trainer
    from tests.test_model_gpu import _small_full_model
    batch = make_batch(4, vocab=300, device="cuda", seed=21)
    batch["question_indices"] = batch["question_indices"] % 500
    small = make_batch(2, vocab=300, device="cuda", seed=5)
    small["question_indices"] = small["question_indices"] % 500
    runs = []
    for pipe in (False, True):
        model, _ = _small_full_model(3, ("n", "s"), (20, 100, 50, 12))
        tr = Trainer(model, base_lr=1e-3, seed=3, use_graph=True, schedule=dict(warmup_iters=4), pipeline_update=pipe)
        losses = [tr.step(clone_batch(batch)).item() for _ in range(5)]
        assert tr._graph is not None and tr._pending == pipe
        if pipe:
            assert tr._update_split() < tr.flat.numel                  # the MMT's parameters close the buffer: two pieces
            stale = tr.flat.flat.clone()
            sd = tr.state_dict()                                        # flushes
            assert not tr._pending and not torch.equal(stale, tr.flat.flat)
            assert sd["global_step"] == 5
        losses.append(tr.step(clone_batch(small)).item())               # another shape: eager, between two replays
        losses += [tr.step(clone_batch(batch)).item() for _ in range(3)]        # the first of these replays runs with its gate closed
        if pipe:
            assert tr._pending
            model.eval()
            with torch.no_grad():
                model(clone_batch(batch))                               # an eval-mode forward applies the pending update first
            assert not tr._pending
            model.train()
        assert tr.global_step == 9
        runs.append((losses, tr.flat.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.flat.bf16.clone(), tr.flat.grad.clone()))
    (l0, p0, m0, v0, b0, g0), (l1, p1, m1, v1, b1, g1) = runs
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
    assert (p0 - p1).abs().max().item() < 5e-3 and (m0 - m1).abs().max().item() < 1e-3 and (v0 - v1).abs().max().item() < 1e-4
    assert (b0.float() - b1.float()).abs().max().item() < 1e-2
    assert torch.equal(b1.float(), p1.to(torch.bfloat16).float())       # the shadows are the rounded masters after the flush
    assert (g1 == 0).all()                                              # ... and the applied update has cleared its gradients


def test_survey_8b_ops_equal_the_cabi_route():
    """masks / pointer scores / loss / optimizer registered under torch.ops.sam_hip (SURVEY 8(b)'s op list): each is the same C entry point
    as the ctypes route underneath, so the results are bit-identical"""
    import os
    from sam_textvqa_amd import _capi as capi, ops, torchops
    ns = torchops.ns()
    g = torch.Generator().manual_seed(11)
    B, T, No, Nc, S, V, D = 3, 20, 100, 50, 12, 500, 768

    def both(fn):
        os.environ["SAM_COARSE_OPS"] = "0"          # ops.* -> ctypes
        try:
            a = fn()
        finally:
            os.environ.pop("SAM_COARSE_OPS", None)
        return a, fn()                               # ops.* -> torch.ops

    qm = (torch.rand(B, T, generator=g) > 0.3).long().cuda()
    om = (torch.rand(B, No, generator=g) > 0.1).long().cuda()
    cm = (torch.rand(B, Nc, generator=g) > 0.5).long().cuda()
    (kv1, q1, c1), (kv2, q2, c2) = both(lambda: ops.pack_masks(qm, om, cm))
    assert torch.equal(kv1, kv2) and torch.equal(q1, q2) and torch.equal(c1, c2)
    base1, base2 = both(lambda: ops.mask_bits_prefix_lm(kv1, S))
    assert torch.equal(base1, base2) and torch.equal(base1, ns.mask_bits_prefix_lm(kv1, S))
    adj = (torch.rand(B, No + Nc, No + Nc, 12, generator=g) > 0.7).to(torch.int8).cuda()
    s1, s2 = both(lambda: ops.mask_bits_spatial(base1, adj, T, 12, [1, 2]))
    assert torch.equal(s1, s2) and torch.equal(s1, ns.pack_relations(base1, adj, T, 12, (1 << 1) | (1 << 2)))
    add = torch.where(torch.rand(B, 1, 182, 182, generator=g) > 0.4, 0.0, -10000.0).cuda()
    a1, a2 = both(lambda: ops.mask_bits_from_additive(add))
    assert torch.equal(a1, a2)
    rel = (torch.rand(B, 12, 182, 182, generator=g) > 0.5).to(torch.int8).cuda()
    r1, r2 = both(lambda: ops.mask_bits_from_int8_bhnn(rel, base1))
    assert torch.equal(r1, r2)
    # pointer scores fwd / bwd
    q, k = rnd((B, S, D), 12).cuda(), rnd((B, Nc, D), 13).cuda()
    p1, p2 = both(lambda: ops.ptr_scores_fwd(q, k, c1, 1.0 / math.sqrt(D)))
    assert torch.equal(p1, p2)
    ds = torch.randn(B, S, Nc, generator=g).cuda()
    (dq1, dk1), (dq2, dk2) = both(lambda: ops.ptr_scores_bwd(ds, q, k, 1.0 / math.sqrt(D)))
    assert torch.equal(dq1, dq2) and torch.equal(dk1, dk2)
    # masked BCE
    fixed, ocr = torch.randn(B * S, V, generator=g).cuda(), torch.randn(B * S, Nc, generator=g).cuda()
    tg = (torch.rand(B * S, V + Nc, generator=g) > 0.98).float().cuda()
    lm = (torch.rand(B * S, generator=g) > 0.4).float().cuda()
    l1, l2 = both(lambda: ops.bce_loss(fixed, ocr, tg, lm, 1.0, None))
    assert torch.allclose(l1[0], l2[0], rtol=1e-6)           # (the loss scalar is summed with atomics: the order of the additions is not fixed)
    assert torch.equal(l1[1], l2[1]) and torch.equal(l1[2], l2[2])
    # sumsq + Adam (host schedule and device schedule)
    n = 4096 * 5
    def adam(dev):
        p = torch.linspace(-1, 1, n).cuda(); gr = torch.sin(torch.arange(n).float()).cuda(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
        sh = torch.empty(n, dtype=torch.bfloat16).cuda(); nsq = torch.zeros(1).cuda()
        ops.sumsq(gr, nsq)
        if dev:
            ops.adam_step_dev(p, gr, m, v, sh, [4096, n], torch.tensor([1e-3, 1e-4, 0.1, 0.001]).cuda(), gnorm_sq=nsq, max_norm=0.25)
        else:
            ops.adam_step(p, gr, m, v, sh, [4096, n], [1e-3, 1e-4], 1, gnorm_sq=nsq, max_norm=0.25)
        return p, m, v, sh, nsq
    for dev in (False, True):
        r1, r2 = both(lambda: adam(dev))
        for x, y in zip(r1, r2):
            assert torch.equal(x, y)
    with pytest.raises(RuntimeError):
        ns.ptr_scores(q.cpu(), k, c1, 1.0)
    with pytest.raises(RuntimeError):
        ns.mask_bits_prefix_lm(kv1.long(), S)


def test_step_advance_matches_the_host_schedule():
    """sam_step_advance (the head node of a captured step): step counter, LambdaLR factor and Adam bias corrections on the device equal the
    host's lr_lambda / 1 - beta^t at every regime of the schedule, through both routes; the RNG offset base advances by the stride"""
    import os
    from sam_textvqa_amd import ops
    from sam_textvqa_amd.trainer import lr_lambda
    base = [1e-4, 1e-5, 1e-4]
    for route in ("0", "1"):
        os.environ["SAM_COARSE_OPS"] = route
        try:
            for it in (0, 1, 2, 499, 999, 1000, 1001, 13999, 14000, 14001, 18999, 19000, 25000):
                step = torch.tensor([it], dtype=torch.int64).cuda()
                rng = torch.tensor([7, 100], dtype=torch.int64).cuda()
                sched = torch.zeros(5).cuda()
                ops.step_advance(rng, 1 << 20, step, base, sched)
                t = it + 1
                want = [l * lr_lambda(it) for l in base] + [1.0 - 0.9 ** t, 1.0 - 0.999 ** t]
                got = sched.cpu().double().tolist()
                assert int(step.item()) == t and rng.tolist() == [7, 100 + (1 << 20)]
                for w_, g_ in zip(want, got):
                    assert abs(w_ - g_) <= 1.2e-7 * abs(w_), (it, want, got)
            step = torch.tensor([5], dtype=torch.int64).cuda(); sched = torch.zeros(3).cuda()
            ops.step_advance(None, 0, step, [1.0], sched, warmup_iters=10, warmup_factor=0.5, lr_decay_iters=(20,), lr_decay=0.5)
            assert abs(sched[0].item() - (0.5 * 0.5 + 0.5)) < 1e-6
        finally:
            os.environ.pop("SAM_COARSE_OPS", None)
