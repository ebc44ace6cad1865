"""Is the replayed learner update GPU-bound or host-bound?  Times N train_from_buffer() calls three ways:
host time to ENQUEUE them (no sync inside the loop), total wall time with one sync at the end, and the GPU time
between two events around the loop.  GPU box: python scripts/probe_update_host.py"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402,F401
from macjd_amd import bench_rollout  # noqa: E402
from macjd_amd.scenario import Scenario, ring_scenario_dict  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402

cli = argparse.Namespace(hidden=64, no_gemm_tuning=True, no_graphs=False, warmup=0, steps=0)
dev = torch.device("cuda:0")
sc = Scenario.from_dict(ring_scenario_dict(3, 4))
env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=4096, device=dev, seed=42)
env.reset()
bench_rollout.make_step(cli, sc, env, dev, 0, 1, "train")
learner = [o for o in gc.get_objects() if type(o).__name__ == "QMixLearner"][0]
for _ in range(20):
    learner.train_from_buffer(sync_stats=False)
torch.cuda.synchronize()
N = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(N):
    learner.train_from_buffer(sync_stats=False)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"per update: host enqueue {1e6 * (t1 - t0) / N:.1f} us, wall incl. final sync {1e6 * (t2 - t0) / N:.1f} us, "
      f"GPU (events) {1e3 * e0.elapsed_time(e1) / N:.1f} us")
# split the host part: graph A replay, graph B replay, the rest (sampling, index upload)
import numpy as np
ta = tb = 0.0
for _ in range(N):
    a = time.perf_counter(); learner._graph_a.replay(); b = time.perf_counter()
    if not learner._g_single:   # one process: both halves are in graph A (set MACJD_SINGLE_UPDATE_GRAPH=0 for two)
        learner._graph_b.replay()
    c = time.perf_counter()
    ta += b - a; tb += c - b
torch.cuda.synchronize()
print(f"host time of graph A replay {1e6 * ta / N:.1f} us, graph B replay {1e6 * tb / N:.1f} us")
# one replay of graph A on an IDLE GPU: host time until replay() returns vs until the work has completed
hs, ws = [], []
for _ in range(50):
    torch.cuda.synchronize()
    a = time.perf_counter(); learner._graph_a.replay(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    hs.append(b - a); ws.append(c - a)
print(f"idle GPU: graph A replay() returns after {1e6 * sorted(hs)[len(hs) // 2]:.1f} us (median), work complete after {1e6 * sorted(ws)[len(ws) // 2]:.1f} us")
