"""whole-model parity at random sequence shapes: tests/test_model_gpu.py::test_sam4c_train_forward_backward_vs_oracle (loss + every parameter gradient vs
the fp32 oracle) and the greedy decoding comparison of tests/test_decode_gpu.py::test_persistent_decoding_kernel_at_other_sequence_lengths, called with
shapes drawn at random:  python tools/fuzz_shapes.py [count] [seed]"""
import os
import random
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402

from tests import test_decode_gpu as td  # noqa: E402
from tests import test_model_gpu as tm  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(count):
    n_dec = rnd.choice([1, 2, 3, 5, 8, 12, 17, 30])
    while True:          # (T >= 5: the synthetic batch generator draws question lengths from [5, T]; 384 tokens and 128 OCR slots are the kernels' limits)
        shapes = (rnd.randint(5, 20), rnd.randint(1, 230), rnd.randint(1, 128), n_dec)
        if sum(shapes) <= 384:
            break
    ctx = rnd.choice([3, 5])
    layers = tuple(rnd.choice("ns") for _ in range(rnd.randint(1, 3)))
    batch = rnd.choice([1, 2, 3, 3, 5, 9, 16])
    vocab = rnd.choice([300, 300, 77, 1000, 5000, 4999, 16])
    tag = "ctx=%d layers=%s shapes=%s N=%d B=%d V=%d" % (ctx, "".join(layers), shapes, sum(shapes), batch, vocab)
    try:
        tm.test_sam4c_train_forward_backward_vs_oracle(ctx, layers, shapes, batch, vocab)
        print("TRAIN ok   ", tag, flush=True)
    except Exception as e:          # noqa: BLE001
        bad += 1
        print("TRAIN FAIL ", tag, "::", str(e).splitlines()[0][:200] if str(e) else traceback.format_exc()[-300:], flush=True)
    if os.environ.get("FUZZ_TRAINER", "1") != "0" and k % 3 == 0:          # (six oracle optimisation steps on the CPU: every third shape)
        try:
            tm.test_training_trajectory_matches_oracle_train_step(shapes, layers, ctx, batch, False)
            print("TRAINER ok ", tag, flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("TRAINER FAIL", tag, "::", str(e).splitlines()[0][:200] if str(e) else traceback.format_exc()[-300:], flush=True)
    if n_dec >= 2 and shapes[2] >= 1:
        mp = pytest.MonkeyPatch()
        try:
            td.test_persistent_decoding_kernel_at_other_sequence_lengths(shapes, mp, vocab)
            print("DECODE ok  ", tag, flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("DECODE FAIL", tag, "::", str(e).splitlines()[0][:200] if str(e) else traceback.format_exc()[-300:], flush=True)
        finally:
            mp.undo()
print("fuzz: %d failures" % bad)
