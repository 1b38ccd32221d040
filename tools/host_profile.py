"""cProfile of the host side of training steps enqueued against a busy GPU (pure python/ctypes/torch-dispatch time)"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from sam_textvqa_amd.synthetic import clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000)
tr = Trainer(model, seed=1)
batch = make_batch(64, device="cuda", seed=1)
for _ in range(3): tr.step(clone_batch(batch))
torch.cuda.synchronize()
torch.cuda._sleep(int(4e8))
pr = cProfile.Profile()
pr.enable()
for _ in range(5): tr.step(clone_batch(batch))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22); st.sort_stats("cumulative").print_stats(30)
