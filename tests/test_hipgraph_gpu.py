"""macjd_amd/hipgraph.py on the GPU box: the two launch arrangements.  Each case runs in its own process —
GPU_MAX_HW_QUEUES is read by the HIP runtime when it initialises."""
import os
import subprocess
import sys

import pytest

from _harness import REPO

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import contextlib, io, os, sys, warnings
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import macjd_amd
from macjd_amd import hipgraph
import numpy as np, torch
from test_nets_cpu import load, make_args, sd_from
from tests_golden_helpers import synthetic_batch
from macjd_amd.core.mac import BasicMAC
from macjd_amd.core.qmix import QMixLearner
from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
g, d = load("3j4r_h64")
T, N, B = 100, 40, 32
def build():
    args = make_args(d, device="cuda", use_cuda=True, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=5)
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(d["S"], args); mac.load_state(sd_from(g, "g5_agent0."))
        learner = QMixLearner(mac, args); buf = EpisodeReplayBuffer(args)
    full = synthetic_batch(np.random.default_rng(9), args, N, T)
    for k, v in buf.buffers.items():
        v.copy_(torch.as_tensor(full[k]).to(v.dtype))
    buf.current_size, buf.current_index = N, 0
    buf.episode_lengths[:] = T
    return learner, buf
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for rep in range(3):                      # learners come and go: graphs captured, replayed, released, re-captured
        learner, buf = build()
        learner.enable_graphs(buf, B, updates_per_graph=3)
        learner.train_from_buffer_many(7)
        learner.enable_graphs(buf, B, updates_per_graph=2)
        learner.train_from_buffer_many(4)
        learner.release_graphs()
        del learner, buf
    torch.cuda.synchronize()
print("MODE", hipgraph.launch_mode(), "QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "KEPT", len(hipgraph._KEPT_GRAPHS),
      "WARNED", int(any("kept alive" in str(x.message) for x in w)))
'''


def _run(**env):
    e = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "MACJD_GRAPH_REPLAY_STREAM")}
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _SCRIPT % {"repo": REPO}], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("MODE")][-1].split()
    return dict(zip(line[0::2], line[1::2]))


def test_reserved_queue_arrangement_destroys_graphs():
    r = _run()
    assert r == {"MODE": "high", "QUEUES": "3", "KEPT": "0", "WARNED": "0"}


def test_fallback_arrangement_keeps_graphs_alive_and_says_so():
    """A user setting that leaves no room for the launch queue: replays on the caller's stream, no captured graph is ever
    destroyed (the condition under which the runtime's parallel-stream selection stays in bounds), one warning."""
    r = _run(GPU_MAX_HW_QUEUES="4")
    assert r["MODE"] == "current" and r["QUEUES"] == "4" and r["WARNED"] == "1"
    assert int(r["KEPT"]) >= 6          # every capture of the three learners (graph A + grouped graph, twice each)
