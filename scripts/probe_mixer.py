#!/usr/bin/env python3
"""Times the fused mixer kernels (forward inference / forward training / backward) against the unfused path at the
learner's size (3232 rows, 3j/4r), launches replayed from a HIP graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as entry

if not os.environ.get("MACJD_LIB"):
    entry.build()
import bench
from macjd_amd import bench_rollout
from macjd_amd.core.networks import QMixer
from macjd_amd.scenario import Scenario, ring_scenario_dict

dev = torch.device("cuda", 0)
for J, R in (((3, 4), (6, 8)) if not os.environ.get("MX_ONLY34") else ((3, 4),)):
    sc = Scenario.from_dict(ring_scenario_dict(J, R))
    args = bench_rollout.make_args(sc, 64, dev)
    torch.manual_seed(0)
    mixer = QMixer(args).to(dev)
    mixer.enable_first_layer_cache()   # no torch.cat per call (the learner's mixers never concatenate either)
    M = 3232
    s = torch.randn(M, mixer.state_dim, device=dev)
    q = torch.randn(M, J, device=dev)
    gy = torch.randn(M, 1, device=dev)
    flops = 2.0 * M * (mixer.state_dim * 384 + 128 * J * 64 + 128 * 64 + 64 + J * 64 + 64)
    for fused in (True, False):
        QMixer.fused = fused
        with torch.no_grad():
            us_inf = bench.time_graph_replay(lambda: mixer(q, s), dev)
        qg = q.clone().requires_grad_(True)

        def fb():
            with torch.enable_grad():
                for p in mixer.parameters():
                    p.grad = None
                qg.grad = None
                mixer(qg, s).backward(gy)
        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fb()
        e1.record()
        torch.cuda.synchronize()
        print(f"{J}j/{R}r fused={fused}: forward (inference) {us_inf:7.2f} us = {flops / us_inf / 1e6:6.2f} TFLOP/s"
              f"   eager forward+backward {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
