"""tests/test_attention_gpu.py::test_attention_sequence_length_edges over a list of lengths:  python tools/debug/attn_T_sweep.py 260 272 ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_attention_gpu as t  # noqa: E402

for T in [int(x) for x in sys.argv[1:]] or [257, 260, 272, 288, 300, 304, 320, 321, 336, 352, 353, 384]:
    try:
        t.test_attention_sequence_length_edges(T)
        print("T=%d ok" % T, flush=True)
    except AssertionError as e:
        print("T=%d FAILED: %s" % (T, str(e)[:200]), flush=True)
