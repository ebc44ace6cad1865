import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


import macjd_amd  # noqa: E402,F401  (before the first torch.cuda call: reserves the graph-launch hardware queue, hipgraph.py)


def _reload_switches():
    """The package and the native library read their MACJD_* switches once: re-read after a test changed one."""
    from macjd_amd import options
    options.reload()
    if _has_gpu():
        from macjd_amd import _native
        _native.reload_options()


_MP = pytest.MonkeyPatch
_orig_setenv, _orig_delenv, _orig_undo = _MP.setenv, _MP.delenv, _MP.undo


def _setenv(self, name, value, prepend=None):
    _orig_setenv(self, name, value, prepend)
    if name.startswith("MACJD_"):
        _reload_switches()


def _delenv(self, name, raising=True):
    _orig_delenv(self, name, raising)
    if name.startswith("MACJD_"):
        _reload_switches()


def _undo(self):
    touched = any(isinstance(k, str) and k.startswith("MACJD_") for _, k, _ in getattr(self, "_setitem", []))
    _orig_undo(self)
    if touched:
        _reload_switches()


_MP.setenv, _MP.delenv, _MP.undo = _setenv, _delenv, _undo


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
    if _has_gpu() and os.environ.get("MACJD_SEGV_TRACE"):   # diagnostic: native backtrace of a host fault (scripts/segv_trace.c)
        import ctypes
        ctypes.CDLL(os.environ["MACJD_SEGV_TRACE"]).segv_trace_install()


@pytest.fixture(autouse=True)
def _teardown_graphs_after_every_gpu_test(request):
    """Every GPU test's HIP graphs are destroyed and their memory pools returned to the driver right after the test.
    This is the arrangement under which, in round 2, a later graph replay faulted deterministically inside
    hipGraphLaunch (the runtime's parallel-stream selection reads past its vector once destroyed graphs have left the
    hardware-queue reference counts uneven); with the replays launched from the high-priority stream
    (macjd_amd/hipgraph.py) the suite has to pass under exactly this stress.  MACJD_TEST_GRAPH_TEARDOWN=0 switches it off."""
    if os.environ.get("MACJD_TEST_GRAPH_TEARDOWN", "1") == "0" or "gpu" not in request.keywords or not _has_gpu():
        yield
        return
    import gc
    import torch
    from macjd_amd import _native
    _native.reload_options()   # (the previous test may have changed a MACJD_* switch through monkeypatch)
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        torch.cuda.synchronize()
        gc.enable()
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
