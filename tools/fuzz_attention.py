"""tests/test_attention_gpu.py's forward / two-kernel backward / fused backward parity tests (fp32 oracle of sa_m4c.py:563-598 on the same masks and dropout bits)
at random (batch, text, objects, OCR, decoding) shapes, with and without the spatial masks and dropout:  python tools/fuzz_attention.py [count] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_attention_gpu as t  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for _ in range(count):
    while True:
        shape = (rnd.randint(1, 5), rnd.randint(1, 20), rnd.randint(1, 230), rnd.randint(1, 128), rnd.choice([0, 1, 3, 12, 30]))
        if sum(shape[1:]) <= 384:
            break
    spatial, p = rnd.random() < 0.7, rnd.choice([0.0, 0.1, 0.3])
    tag = "shape=%s N=%d spatial=%d p=%.1f" % (shape, sum(shape[1:]), spatial, p)
    for name, fn in (("fwd+bwd", t.test_attention_fwd_bwd), ("fused bwd", t.test_attention_fused_backward)):
        try:
            fn(shape, spatial, p)
            print("ok  ", name, tag, flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("FAIL", name, tag, "::", (str(e).splitlines() or [repr(e)])[0][:220], flush=True)
print("fuzz_attention: %d failures" % bad)
