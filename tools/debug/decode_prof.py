import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.synthetic import clone_batch, make_batch
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
batch = make_batch(64, device="cuda", seed=1)
with torch.no_grad():
    for _ in range(6):
        model(clone_batch(batch))
torch.cuda.synchronize()
