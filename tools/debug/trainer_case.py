"""one case of tools/fuzz_shapes.py's trainer-trajectory check with the full traceback:  python tools/debug/trainer_case.py ctx layers T OBJ OCR DEC batch"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_model_gpu as tm  # noqa: E402

ctx, layers = int(sys.argv[1]), tuple(sys.argv[2])
shapes, batch = tuple(int(x) for x in sys.argv[3:7]), int(sys.argv[7])
try:
    tm.test_training_trajectory_matches_oracle_train_step(shapes, layers, ctx, batch, False)
    print("ok")
except Exception:      # noqa: BLE001
    traceback.print_exc()
