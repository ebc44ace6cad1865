#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels against the library / eager form they replace, in ONE
process with interleaved rounds (median of rounds).  Usage: python scripts/bench_kernels.py [--rounds 7]"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def timeit(fn, iters=20, replays=10):
    """GPU-side time per call: `iters` calls captured into one HIP graph, replayed `replays` times (eager
    back-to-back launches of these small kernels are bound by the Python / launch path, not by the GPU)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * replays) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    entry.build()
    from macjd_amd import ops
    dev = "cuda:0"
    torch.manual_seed(0)
    cases = {}
    for name, dims, acts, N in (("actor 46-128-128-9 N=12288", (46, 128, 128, 9), (1, 1, 2), 12288),
                                ("fc1+gi 46-64-192 N=12288", (46, 64, 192), (1, 0), 12288),
                                ("actor N=9600", (46, 128, 128, 9), (1, 1, 2), 9600),
                                ("fc1+gi N=9600", (46, 64, 192), (1, 0), 9600)):
        x = torch.randn(N, dims[0], device=dev)
        layers = [(torch.randn(dims[l + 1], dims[l], device=dev) / dims[l] ** 0.5, torch.randn(dims[l + 1], device=dev),
                   acts[l]) for l in range(len(dims) - 1)]
        cases[name + " | fused MFMA"] = lambda x=x, layers=layers: ops.mlp_forward(x, layers)
        cases[name + " | torch GEMMs"] = lambda x=x, layers=layers: ops.mlp_reference(x, layers)
    # learner GRU scan: 2 networks x 96 sequences x 100 steps, H = 64
    B, T, J, H = 32, 100, 3, 64
    gis = [torch.randn(B, T, J, 3 * H, device=dev) for _ in range(2)]
    ws = [torch.randn(3 * H, H, device=dev) / 8 for _ in range(2)]
    bs = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
    cases["GRU scan 2x96 seq x 100 steps | fused kernel"] = lambda: ops.gru_sequence_multi(gis, ws, bs)
    with torch.no_grad():
        res = {k: [] for k in cases}
        for _ in range(args.rounds):
            for k, fn in cases.items():
                res[k].append(timeit(fn))
    for k, v in res.items():
        print(f"{k:55s} median {statistics.median(v):8.1f} us   min {min(v):8.1f} us")


if __name__ == "__main__":
    main()
