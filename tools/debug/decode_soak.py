"""soak: the captured greedy decode session replayed many times (the persistent kernel's barriers must never time out, tokens must not drift)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.synthetic import clone_batch, make_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
model.decode_cache = True
batches = [make_batch(64, device="cuda", seed=s) for s in (1, 2, 3)]
ref = []
with torch.no_grad():
    for b in batches:
        bd = clone_batch(b); model(bd); ref.append(bd["train_prev_inds"].clone())
    t0 = time.perf_counter()
    for i in range(n):
        bd = clone_batch(batches[i % 3]); model(bd)
        if i % 50 == 0:
            assert torch.equal(bd["train_prev_inds"], ref[i % 3]), i
    torch.cuda.synchronize()
print("decode soak ok: %d batches, %.2f ms per batch" % (n, (time.perf_counter() - t0) / n * 1e3))
