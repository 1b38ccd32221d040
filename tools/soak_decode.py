"""soak of the persistent decoding kernel: N batches of greedy decoding at the stress shape (350 tokens, 100 OCR slots, 30 steps, 12 layers, B = 32 -- the two-chunk attention
phase, 2 494 grid barriers per batch) and at the c3 shape (B = 64), fresh inputs every batch; the session must stay on the persistent kernel (no device-side failure, error
word 0), every 25th batch is decoded again by the captured per-kernel step and must give the same tokens:  python tools/soak_decode.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sam_textvqa_amd.params import prepare  # noqa: E402
from sam_textvqa_amd.synthetic import SHAPES, clone_batch, make_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for tag, layers, shape, bs in (("stress", ("n", "n") + ("s",) * 10, SHAPES["stress"], 32), ("c3", ("n", "n", "s", "s", "s", "s"), SHAPES["c3"], 64)):
    model = bench.build_model(3, layers, 5000, shape).cuda().eval()
    prepare(model)
    t0 = time.perf_counter()
    checked = 0
    with torch.no_grad():
        for k in range(n):
            batch = make_batch(bs, *shape, vocab=5000, context=3, device="cuda", seed=1000 + k)
            os.environ["SAM_DECODE_FUSED"] = "1"
            bd = clone_batch(batch)
            model(bd)
            ses = [v for v in model._sam_decode_sessions.values() if v.beam == 0][0]
            assert ses.fused, "%s: batch %d fell back to the per-kernel step" % (tag, k)
            assert int(ses._fused_ws[256].item()) == 0, "%s: error word set at batch %d" % (tag, k)
            if k % 25 == 0:
                toks = bd["train_prev_inds"].clone()
                os.environ["SAM_DECODE_FUSED"] = "0"
                saved = model.__dict__.pop("_sam_decode_sessions")
                bd2 = clone_batch(batch)
                model(bd2)
                same = (toks == bd2["train_prev_inds"]).all(-1).float().mean().item()
                model.__dict__["_sam_decode_sessions"] = saved
                assert same >= 0.9, "%s: batch %d: only %.2f of the samples decode to the same tokens as the per-kernel step" % (tag, k, same)
                checked += 1
    torch.cuda.synchronize()
    print("%s: %d batches of %d on the persistent kernel, %d cross-checked against the per-kernel step, %.1f s" % (tag, n, bs, checked, time.perf_counter() - t0), flush=True)
    del model
    torch.cuda.empty_cache()
print("SOAK_DECODE_OK")
