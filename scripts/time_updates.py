#!/usr/bin/env python3
"""Microseconds per learner update of the benchmark configuration WITHOUT a profiler attached (rocprofv3 inflates exactly
the cross-queue hand-overs one is usually trying to judge): groups of 20 updates replayed as one graph, wall clock over
n updates, several repetitions.  Environment switches (MACJD_*) select the variant:
    [UPG=updates per graph] [JAMMERS= RADARS=] python scripts/time_updates.py [n_updates] [repetitions]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402,F401
import torch  # noqa: E402

from macjd_amd import bench_rollout  # noqa: E402
from macjd_amd.core.mac import BasicMAC  # noqa: E402
from macjd_amd.core.qmix import QMixLearner  # noqa: E402
from macjd_amd.runners.episode_runner import BatchedEpisodeRunner  # noqa: E402
from macjd_amd.scenario import Scenario, ring_scenario_dict  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402
from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
J, R = int(os.environ.get("JAMMERS", 3)), int(os.environ.get("RADARS", 4))
dev = torch.device("cuda", 0)
sc = Scenario.from_dict(ring_scenario_dict(J, R))
E = int(os.environ.get("BATCH_ENVS", 4096))
env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=dev, seed=1)
args = bench_rollout.make_args(sc, 64, dev, batch_envs=E, mixer_dtype=os.environ.get("MIXER_DTYPE", "fp32"))
torch.manual_seed(42)
with contextlib.redirect_stdout(io.StringIO()):
    mac = BasicMAC(args.obs_shape, args)
    buf = EpisodeReplayBuffer(args, device=dev)
    learner = QMixLearner(mac, args)
BatchedEpisodeRunner(env, mac, buf, args).run(sync_stats=False)
learner.enable_graphs(buf, args.batch_size, updates_per_graph=int(os.environ.get("UPG", 20)))
learner.train_from_buffer_many(200)
torch.cuda.synchronize()
out = []
for _ in range(reps):
    t0 = time.perf_counter()
    learner.train_from_buffer_many(n)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / n * 1e6)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MACJD_") or k in ("UPG", "JAMMERS", "RADARS"))
print(f"us/update [{tag or 'defaults'}]: " + " ".join(f"{x:.1f}" for x in out) + f"   min {min(out):.1f}")
