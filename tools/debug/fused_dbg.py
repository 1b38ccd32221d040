"""debug: fused greedy decoding kernel, graph vs eager, small model"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_decode_gpu import _models, _batch
model, _, shapes = _models()
model.decode_cache = True
for graph in ("0", "1", "1", "0"):
    os.environ["SAM_DECODE_GRAPH"] = graph
    model.__dict__.pop("_sam_decode_sessions", None)
    for seed in (17, 18, 17):
        bd = _batch(3, shapes, 300, seed, "cuda")
        with torch.no_grad():
            sc = model(bd)["textvqa_scores"]
        ses = next(iter(model._sam_decode_sessions.values()))
        print("graph", graph, "seed", seed, "fused", bool(ses.fused), "prev", bd["train_prev_inds"].tolist(), "nan rows", torch.isnan(sc).any(-1).int().tolist(), "err", int(ses._fused_ws[32].item()), "bar", int(ses._fused_ws[0].item()))
