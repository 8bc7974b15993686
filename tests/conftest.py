import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _settle_device_state(request):
    """A trainer and its captured hipGraphs form a reference cycle (trainer -> GraphedStep -> step
    closure -> trainer), so they die whenever the cyclic collector happens to run -- possibly in the
    middle of another test's capture or replay.  Collect them between tests instead, with the device
    idle, so that every graph is destroyed at a quiet point."""
    yield
    if "gpu" in request.keywords and torch.cuda.is_available():
        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def sub(d, prefix):
    """Entries of ``d`` under ``prefix`` with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
