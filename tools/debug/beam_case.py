"""one case of tools/fuzz_beam.py's incremental-vs-full check with the test's prints:  python tools/debug/beam_case.py T OBJ OCR DEC"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest  # noqa: E402

from tests import test_decode_gpu as td  # noqa: E402

shapes = tuple(int(x) for x in sys.argv[1:5])


def models(layers=("n", "s", "s"), vocab=300, _s=shapes):
    from sam_textvqa_amd.params import prepare
    from tests.test_model_gpu import _small_full_model
    model, ref = _small_full_model(3, layers, _s, vocab=vocab)
    model.cuda().eval()
    prepare(model)
    return model, ref.eval(), _s


for early in (False, True):
    mp = pytest.MonkeyPatch()
    mp.setattr(td, "_models", models)
    try:
        td.test_incremental_beam_steps_decode_like_the_full_recompute(early, mp)
        print("early=%s ok" % early)
    except Exception:      # noqa: BLE001
        print("early=%s FAILED" % early)
        traceback.print_exc()
    finally:
        mp.undo()
