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


@pytest.fixture(autouse=True)
def _quiet_point_for_graph_teardown(request):
    """GPU tests build HIP graphs (learner updates, rollouts) that die with the test's objects.  Left to Python's cyclic
    collector, such a graph — and its private memory pool — is destroyed at an arbitrary allocation point of a LATER test,
    possibly in the middle of that test's stream capture or graph launch; twice in this round a full run died with a host
    segmentation fault inside hipGraphLaunch that no single test reproduces.  So: no automatic collection while a GPU test
    runs, and an explicit one at a quiet point (device idle) after it."""
    if "gpu" not in request.keywords or not _has_gpu():
        yield
        return
    import gc
    import torch
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        torch.cuda.synchronize()
        gc.enable()
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()   # the finished test's graph pools and cached blocks go back to the driver now
