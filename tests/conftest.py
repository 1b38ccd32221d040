import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "transport: spawns child processes that open an RCCL / gloo process group; collected LAST, so that a transport-level "
                                       "abort can never leave kernel-parity tests unreached under -x (round-4 driver record)")


def pytest_collection_modifyitems(config, items):
    # kernel parity first, process-group / subprocess tests last (stable within each class)
    items.sort(key=lambda it: 1 if ("transport" in it.keywords or "test_zz_" in it.nodeid) else 0)
    # GPU-marked tests never run without a device, whatever -m says.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
