#!/usr/bin/env python3
"""Times the env-step kernel variants (slot / lane) at a few batch sizes through macjd_env_step_timed.
MACJD_LIB selects another build of the library (kernel A/B runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as entry

if not os.environ.get("MACJD_LIB"):
    entry.build()
from macjd_amd import _native
from macjd_amd.scenario import Scenario, ring_scenario_dict
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment

J, R = 3, 4
sc = Scenario.from_dict(ring_scenario_dict(J, R))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for logE in [int(a) for a in (sys.argv[1:] or ["12", "16", "22"])]:
    E = 1 << logE
    env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=dev, seed=1)
    env.reset()
    T = torch.randint(0, 2 * R + 1, (J, E), generator=g, device=dev, dtype=torch.int32).t()
    P = torch.rand((J, E), generator=g, device=dev).t()
    out = []
    for label, flag in (("slot", _native.STEP_SLOT_KERNEL), ("lane", _native.STEP_LANE_KERNEL)):
        env.kernel_flags = flag
        env.time_step_kernel(T, P, iters=5)
        out.append(f"{label} {env.time_step_kernel(T, P, iters=100 if logE < 20 else 20) * 1e3:8.2f} us")
    print(f"E=2^{logE}: " + "   ".join(out), flush=True)
    env.close()

# per-env scenario tables (lane kernel streaming its env's table column), 4096 compiled scenarios cycled
if os.environ.get("MACJD_PROBE_PER_ENV", "1") != "0":
    from macjd_amd.scenario import ScenarioBatch
    batch = ScenarioBatch.randomized(ring_scenario_dict(J, R), 4096, seed=42)
    for logE in (12, 22):
        E = 1 << logE
        env = BatchedElectromagneticEnvironment(scenario_batch=batch.tile(E) if E > 4096 else batch, device=dev, seed=1)
        env.reset()
        T = torch.randint(0, 2 * R + 1, (J, E), generator=g, device=dev, dtype=torch.int32).t()
        P = torch.rand((J, E), generator=g, device=dev).t()
        env.time_step_kernel(T, P, iters=3)
        us = env.time_step_kernel(T, P, iters=100 if logE < 20 else 20) * 1e3
        print(f"per-env tables E=2^{logE}: {us:8.2f} us   {E * 461 / us / 1e3:7.1f} GB/s", flush=True)
        env.close()
