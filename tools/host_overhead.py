"""how long does the HOST need to enqueue one training step (python + ctypes launches), vs the GPU's step time"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from sam_textvqa_amd.synthetic import clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000)
tr = Trainer(model, seed=1)
batch = make_batch(64, device="cuda", seed=1)
for _ in range(3): tr.step(clone_batch(batch))
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e8))      # ~100 ms of GPU spin: the step below is enqueued against a busy GPU = pure host time
    t0 = time.perf_counter()
    tr.step(clone_batch(batch))
    ts.append(time.perf_counter() - t0)
host = sorted(ts)[len(ts) // 2]
torch.cuda.synchronize()
print("host enqueue time per step: %.2f ms" % (host * 1e3))
