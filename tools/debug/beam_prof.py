"""beam-5 decoding of a batch of 64 under rocprofv3 (tools/debug/beam_trace.sh): which kernels a batch costs"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.registry import registry
from sam_textvqa_amd.synthetic import clone_batch, make_batch
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
batch = make_batch(64, device="cuda", seed=1)
registry.EOS_IDX, registry.BOS_IDX = 2, 1
model.set_beam_size(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
with torch.no_grad():
    for _ in range(8):
        bd = clone_batch(batch)
        bd["train_prev_inds"] = torch.zeros_like(bd["train_prev_inds"]); bd["train_prev_inds"][:, 0] = 1
        model(bd, use_beam_search=True)
torch.cuda.synchronize()
