"""soak: N training steps at the bench configuration; reports loss trajectory, step-time drift and device-memory growth"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sam_textvqa_amd.synthetic import SHAPES, clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
shape = SHAPES["c3"]
model = bench.build_model(3, ("n", "n", "s", "s", "s", "s"), 5000, shape)
tr = Trainer(model, seed=1)
batches = [make_batch(64, *shape, vocab=5000, context=3, device="cuda", seed=100 + i) for i in range(4)]
losses, times, mem = [], [], []
for i in range(n):
    if i % 50 == 0:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    l = tr.step(clone_batch(batches[i % 4]))
    if i % 50 == 49:
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 50 * 1e3)
        losses.append(l.item()); mem.append(torch.cuda.memory_allocated() / 2 ** 20)
        assert torch.isfinite(l).item()
print("loss every 50 steps:", ["%.1f" % x for x in losses])
print("ms/step per block of 50:", ["%.2f" % x for x in times])
print("allocated MiB:", ["%.0f" % x for x in mem], "peak reserved MiB %.0f" % (torch.cuda.max_memory_reserved() / 2 ** 20))
assert torch.isfinite(tr.flat.flat).all() and torch.isfinite(tr.exp_avg_sq).all()
print("SOAK_OK")
