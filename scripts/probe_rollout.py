"""Where does a graphed rollout episode batch spend its time?  Times rollout_graphed() and end_episodes() separately
(device time, HIP events) for a dozen consecutive episode batches.  GPU box: python scripts/probe_rollout.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402,F401
from macjd_amd import bench_rollout  # noqa: E402
from macjd_amd.scenario import Scenario, ring_scenario_dict  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402

cli = argparse.Namespace(hidden=64, no_gemm_tuning=True, no_graphs=False, warmup=0, steps=10 ** 6)
dev = torch.device("cuda:0")
sc = Scenario.from_dict(ring_scenario_dict(3, 4))
env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=4096, device=dev, seed=42)
env.reset()
step_fn, _ = bench_rollout.make_step(cli, sc, env, dev, 0, 1, "rollout")
import gc
runner = [o for o in gc.get_objects() if type(o).__name__ == "BatchedEpisodeRunner"][0]
ev = lambda: torch.cuda.Event(enable_timing=True)
for k in range(12):
    a, b, c = ev(), ev(), ev()
    a.record()
    runner.rollout_graphed()
    b.record()
    runner.end_episodes()
    c.record()
    torch.cuda.synchronize()
    print(f"episode batch {k:2d}: rollout graph {a.elapsed_time(b):7.3f} ms   replay store {b.elapsed_time(c):7.3f} ms   buffer size {len(runner.buffer)}")
