"""tests/test_model_gpu.py::test_trainer_full_depth_batches_the_eight_wave_wgrad_declines (4 MMT + 3 TextBert layers, no reducer: the held layer pair merged with
TextBert's problems, against the unmerged path) over batch sizes the parametrisation does not list -- the partial last batch of an epoch can be anything:
python tools/fuzz_batch.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402

from tests import test_model_gpu as tm  # noqa: E402

bad = 0
for b in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 7, 13, 16, 31, 32, 33, 37, 50, 63, 64, 65, 96, 127, 128]:
    mp = pytest.MonkeyPatch()
    try:
        tm.test_trainer_full_depth_batches_the_eight_wave_wgrad_declines(b, mp)
        print("B=%3d ok" % b, flush=True)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("B=%3d FAIL :: %s" % (b, (str(e).splitlines() or [repr(e)])[0][:300]), flush=True)
    finally:
        mp.undo()
print("fuzz_batch: %d failures" % bad)
