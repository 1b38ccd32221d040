"""How long does the HOST need to enqueue one training step?  Measured as the wall time per step of a run whose GPU work is negligible
(batch 1: the same ~280 launches and the same Python / dispatcher / autograd path as the real step, GPU time far below host time), so the
number is not distorted by the launch queue filling up behind a busy GPU.  SAM_COARSE_OPS=0 selects the per-kernel ctypes route."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from sam_textvqa_amd.synthetic import clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000)
tr = Trainer(model, seed=1)
for b in (1, 64):
    batch = make_batch(b, device="cuda", seed=1)
    for _ in range(5): tr.step(clone_batch(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n): tr.step(clone_batch(batch))
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print("batch %2d: host returns after %.2f ms per step, step incl. GPU %.2f ms  (coarse ops %s)" % (b, t_host * 1e3, t_all * 1e3, os.environ.get("SAM_COARSE_OPS", "1")))
# the captured step (one hipGraph replay per step): what the host pays then
for b in (1, 64):
    model_g = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000)
    trg = Trainer(model_g, seed=1, use_graph=True)
    batch = make_batch(b, device="cuda", seed=1)
    for _ in range(5): trg.step(clone_batch(batch))
    staged = trg.input_buffers() or batch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n): trg.step(clone_batch(staged))
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print("batch %2d, captured step: host returns after %.2f ms per step, step incl. GPU %.2f ms" % (b, t_host * 1e3, t_all * 1e3))
    del trg, model_g
