"""Attribute the launches of ONE learner update (the body the HIP graph captures) to Python call sites:
torch.profiler over an eager `_forward_backward_full` + `_clip_and_step`, kernels grouped by the innermost
macjd_amd frame.  Usage (GPU box): python scripts/profile_update.py > gpurun_out/update_ops.txt"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macjd_amd  # noqa: E402
from macjd_amd import bench_rollout  # noqa: E402
from macjd_amd.scenario import Scenario  # noqa: E402
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment  # noqa: E402


def main():
    cli = argparse.Namespace(hidden=64, no_gemm_tuning=True, no_graphs=False, warmup=0, steps=0)
    dev = torch.device("cuda:0")
    pkg = os.path.dirname(macjd_amd.scenario.__file__)
    sc = Scenario.from_yaml(os.path.join(pkg, "config", "scenario_3j4r.yaml"))
    args = bench_rollout.make_args(sc, 64, dev, batch_envs=4096)
    env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=4096, device=dev, seed=42)
    env.reset()
    step_fn, _ = bench_rollout.make_step(cli, sc, env, dev, 0, 1, "train")
    import gc
    learner = [o for o in gc.get_objects() if type(o).__name__ == "QMixLearner"][0]
    body = learner._graph_body_a
    for _ in range(2):
        body(); learner._clip_and_step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                with_stack=True, record_shapes=True) as prof:
        body(); learner._clip_and_step()
        torch.cuda.synchronize()
    # kernel events -> launching op -> innermost package frame
    by_site = collections.OrderedDict()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        site = "?"
        for fr in (ev.stack or []):
            if "marl_amd" in fr or "macjd" in fr:
                site = fr.split("marl_amd/")[-1]
                break
        for k in ev.kernels:
            shapes = str(ev.input_shapes)[:80] if ev.name in ("aten::copy_", "aten::cat", "aten::clamp_min", "aten::addmm", "aten::mm", "aten::fill_", "aten::mul") else ""
            key = (site + " " + shapes, ev.name, k.name[:50])
            by_site.setdefault(key, [0, 0.0])
            by_site[key][0] += 1
            by_site[key][1] += k.duration
    n = 0
    for (site, op, kern), (c, us) in by_site.items():
        n += c
        print(f"{c:3d} {us:8.1f} us  {op:38s} {kern:70s} @ {site}")
    print("total launches", n)


if __name__ == "__main__":
    main()
