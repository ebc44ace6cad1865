import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """HIP graphs captured by a test stay alive until the session ends.  Destroying a graph (and releasing its private
    memory pool) while later tests capture and replay their own has produced host faults inside hipGraphLaunch on this
    ROCm stack — intermittently with Python's own collector, deterministically when a fixture forced the collection
    after every test.  A training process never destroys its graphs mid-run; the test session now does not either."""
    if not _has_gpu():
        return
    import torch
    keep = []
    orig = torch.cuda.CUDAGraph.capture_end

    def capture_end(self):
        orig(self)
        keep.append(self)

    torch.cuda.CUDAGraph.capture_end = capture_end
    session.config._macjd_graphs_kept = keep
