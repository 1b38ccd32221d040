"""tests/test_decode_gpu.py's beam-search tests (whole search vs the fp32 oracle wherever the oracle's candidate gaps exceed the score error; shared vs
expanded beams; incremental vs full-recompute steps) at random sequence shapes and beam sizes:  python tools/fuzz_beam.py [count] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402

from tests import test_decode_gpu as td  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for _ in range(count):
    while True:
        shapes = (rnd.randint(5, 20), rnd.randint(1, 230), rnd.randint(1, 128), rnd.choice([2, 3, 5, 8, 12, 17, 30]))
        if sum(shapes) <= 384:
            break
    beam = rnd.choice([1, 2, 3, 5, 8])

    def models(layers=("n", "s", "s"), vocab=300, _s=shapes):
        from sam_textvqa_amd.params import prepare
        from tests.test_model_gpu import _small_full_model
        model, ref = _small_full_model(3, layers, _s, vocab=vocab)
        model.cuda().eval()
        prepare(model)
        return model, ref.eval(), _s
    tag = "shapes=%s N=%d beam=%d" % (shapes, sum(shapes), beam)
    cases = [("vs oracle", lambda mp: td.test_beam_search_whole_model_vs_oracle(beam, mp))]
    if beam > 1:
        cases += [("shared vs expanded", lambda mp: td.test_beams_sharing_one_copy_of_the_encoder_rows_decode_like_the_expanded_batch(beam, mp)),
                  ("incremental vs full", lambda mp: td.test_incremental_beam_steps_decode_like_the_full_recompute(rnd.random() < 0.5, mp))]
    for name, fn in cases:
        mp = pytest.MonkeyPatch()
        mp.setattr(td, "_models", models)
        try:
            fn(mp)
            print("ok  ", name, tag, flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("FAIL", name, tag, "::", (str(e).splitlines() or [repr(e)])[0][:240], flush=True)
        finally:
            mp.undo()
print("fuzz_beam: %d failures" % bad)
