"""greedy-decoding (eval) throughput at the c3 shape, B=64: 12 full forwards (reference behaviour) vs encoder-row caching"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.synthetic import clone_batch, make_batch
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
batch = make_batch(64, device="cuda", seed=1)
for cached in (False, True):
    model.decode_cache = cached
    with torch.no_grad():
        for _ in range(2): model(clone_batch(batch))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): model(clone_batch(batch))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("decode_cache=%s: %.1f ms per batch of 64 (12 greedy steps) = %.0f samples/s" % (cached, dt * 1e3, 64 / dt))
