"""greedy decoding at the stress shape (350 tokens, 100 OCR, 30 steps, 12 layers, B=32) through the persistent kernel and through the captured
per-kernel step:  python tools/bench_decode_stress.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sam_textvqa_amd.synthetic import SHAPES  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    layers = ("n", "n") + ("s",) * 10
    for fused in ("1", "0"):
        os.environ["SAM_DECODE_FUSED"] = fused
        for b in (32, 64):
            r = bench.eval_decode(3, layers, 5000, SHAPES["stress"], b, dev, reps=4, warmup=2, modes=("greedy",))
            print("SAM_DECODE_FUSED=%s" % fused, json.dumps(r), flush=True)
