#!/usr/bin/env python3
"""What an update costs in the multi-rank layouts, measured on ONE GPU with a one-rank RCCL group (the collective's launch
and stream hand-overs are real, the exchange over xGMI is not): two graphs around an eager all-reduce (today's default with
ranks) vs the all-reduce captured inside the update graph (MACJD_GRAPHED_ALLREDUCE=1), single updates and groups of 20,
next to the single-process update.  python scripts/probe_dist_update.py"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

from macjd_amd import bench_rollout  # noqa: E402
from macjd_amd.core.mac import BasicMAC  # noqa: E402
from macjd_amd.core.qmix import QMixLearner  # noqa: E402
from macjd_amd.runners.episode_runner import BatchedEpisodeRunner  # noqa: E402
from macjd_amd.scenario import Scenario, ring_scenario_dict  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402
from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer  # noqa: E402

dev = torch.device("cuda", 0)
sc = Scenario.from_dict(ring_scenario_dict(3, 4))


def build(ranks, **kw):
    env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=4096, device=dev, seed=1)
    args = bench_rollout.make_args(sc, 64, dev, batch_envs=4096)
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(args.obs_shape, args)
        buf = EpisodeReplayBuffer(args, device=dev)
        learner = QMixLearner(mac, args)
    if ranks:   # behave like one rank of several: separate LayerNorm-parameter launch, the all-reduce is issued
        learner._world_size = lambda: 2
    BatchedEpisodeRunner(env, mac, buf, args).run(sync_stats=False)
    learner.enable_graphs(buf, args.batch_size, **kw)
    return learner


def rate(learner, n=400):
    learner.train_from_buffer_many(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    learner.train_from_buffer_many(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, ranks, kw in (("single process, 20 updates per graph", False, dict(updates_per_graph=20)),
                        ("single process, 1 update per graph", False, dict(updates_per_graph=1)),
                        ("ranks: graph A -> eager all-reduce -> graph B", True, dict(updates_per_graph=1, graphed_allreduce=False)),
                        ("ranks: all-reduce inside the graph, 1 update per graph", True, dict(updates_per_graph=1, graphed_allreduce=True)),
                        ("ranks: all-reduce inside the graph, 20 updates per graph", True, dict(updates_per_graph=20, graphed_allreduce=True))):
    lr = build(ranks, **kw)
    print(f"{name}: {rate(lr):.1f} us / update  (graphed_ar={lr._g_graphed_ar}, single={lr._g_single}, group={lr._g_multi[0] if lr._g_multi else 1})", flush=True)
    lr.release_graphs()
dist.destroy_process_group()
