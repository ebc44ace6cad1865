#!/usr/bin/env python3
"""The rollout's env launch ALONE, for a kernel trace whose AverageNs is that launch's duration: env_step_kernel<J,R> over
T x E work items (macjd_env_step_many), 50 launches replayed from one HIP graph, a few times.
    rocprofv3 --kernel-trace --stats -- python3 scripts/replay_many_step.py [--jammers 3 --radars 4 --batch-envs 4096]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402,F401
import torch  # noqa: E402
from macjd_amd.scenario import Scenario, ring_scenario_dict  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--jammers", type=int, default=3)
ap.add_argument("--radars", type=int, default=4)
ap.add_argument("--batch-envs", type=int, default=4096)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
J, R, E = a.jammers, a.radars, a.batch_envs
sc = Scenario.from_dict(ring_scenario_dict(J, R))
dev = torch.device("cuda", 0)
env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=dev, seed=1234)
env.reset()
T = sc.episode_limit
g = torch.Generator(device=dev).manual_seed(1234)
Tm = torch.randint(0, 2 * R + 1, (T, E, J, 1), dtype=torch.int32, device=dev, generator=g)
Pm = torch.rand((T, E, J, 1), device=dev, generator=g)
rew = torch.zeros((T, E, 1), device=dev)
ter = torch.zeros((T, E, 1), dtype=torch.bool, device=dev)
rd = torch.zeros((T, E, 3), device=dev)
ms = [env.time_step_many_kernel(Tm, Pm, rew, ter, rd, iters=50) for _ in range(a.reps)]
torch.cuda.synchronize()
print("us per launch (HIP events):", [round(m * 1e3, 3) for m in ms])
