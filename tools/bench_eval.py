"""evaluation-time decoding throughput at the c3 shape, B=64: 12 full forwards (the reference's loop), the eager encoder-row cache,
the decode session launch by launch, the captured session; then beam search (beam 3 / 5) through the captured session.
usage: python tools/bench_eval.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.registry import registry
from sam_textvqa_amd.synthetic import clone_batch, make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
batch = make_batch(B, device="cuda", seed=1)


def timed(fn, reps=8, warm=3):
    with torch.no_grad():
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
for name, cache, session, graph in (("12 full forwards", False, "0", "0"), ("eager row cache", True, "0", "0"), ("session, eager", True, "1", "0"), ("session, captured", True, "1", "1")):
    model.decode_cache = cache
    os.environ["SAM_DECODE_SESSION"], os.environ["SAM_DECODE_GRAPH"] = session, graph
    model.__dict__.pop("_sam_decode_sessions", None)
    dt = timed(lambda: model(clone_batch(batch)))
    res[name] = dt
    print("greedy %-20s %7.2f ms per batch of %d (12 steps) = %7.0f samples/s" % (name, dt * 1e3, B, B / dt), flush=True)
registry.EOS_IDX, registry.BOS_IDX = 2, 1
for beam in (3, 5):
    model.set_beam_size(beam)
    def run():
        bd = clone_batch(batch)
        bd["train_prev_inds"] = torch.zeros_like(bd["train_prev_inds"]); bd["train_prev_inds"][:, 0] = 1
        return model(bd, use_beam_search=True)
    dt = timed(run, reps=5, warm=2)
    print("beam %d  session, captured   %7.2f ms per batch of %d = %7.0f samples/s" % (beam, dt * 1e3, B, B / dt), flush=True)
