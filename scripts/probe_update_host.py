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
    if not learner._g_single:   # one process: both halves are in graph A (enable_graphs(force_two_graphs=True) for two)
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

# where the host time of one train_from_buffer() goes (same statements, timed one by one while the GPU is busy)
import collections
acc = collections.defaultdict(float)
buf = learner._g_buffer
M = 300
for _ in range(M):
    t = [time.perf_counter()]
    idx = learner._sample_rng.choice(buf.current_size, learner._g_B, replace=False); t.append(time.perf_counter())
    idx = np.asarray(idx, dtype=np.int64); ok = int(buf.episode_lengths[idx].min()) == learner._g_T; t.append(time.perf_counter())
    learner.train_step += 1
    slot, ev = learner._g_idx_ring[learner.train_step % len(learner._g_idx_ring)]
    ev.synchronize(); t.append(time.perf_counter())
    slot.numpy()[:] = idx; t.append(time.perf_counter())
    learner._g_idx.copy_(slot, non_blocking=True); t.append(time.perf_counter())
    ev.record(); t.append(time.perf_counter())
    learner._graph_a.replay(); t.append(time.perf_counter())
    learner._after_step(); t.append(time.perf_counter())
    learner._pack_stats(*learner._g_out_a[:1], learner._g_out_b, *learner._g_out_a[1:], False); t.append(time.perf_counter())
    for name, a, b in zip(("sample", "check", "event wait", "fill slot", "copy_", "event record", "replay", "after_step", "pack_stats"), t, t[1:]):
        acc[name] += b - a
torch.cuda.synchronize()
print("host time per piece (us): " + ", ".join(f"{k} {1e6 * v / M:.1f}" for k, v in acc.items()))
