#!/usr/bin/env python3
"""GRU scan kernels side by side (H = 64): the unit-split scan (default) vs the K-split one (MACJD_GRU_SCAN=ksplit), for
the learner's shapes — 96 sequences x 101 steps (one shared body) and 2 x 96 (two controllers) — with a per-step input
transform and with one transform per sequence (static observation).  GPU time per launch by HIP-graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from bench_kernels import timeit  # noqa: E402


def main():
    entry.build()
    from macjd_amd import ops
    dev = torch.device("cuda:0")
    H, B, T, J = 64, 32, 101, 3
    torch.manual_seed(0)
    for nets in (1, 2):
        gis = [torch.randn(B, T, J, 3 * H, device=dev) for _ in range(nets)]
        ws = [torch.randn(3 * H, H, device=dev) / 8 for _ in range(nets)]
        bs = [0.1 * torch.randn(3 * H, device=dev) for _ in range(nets)]
        gst = [g[:, :1].contiguous() for g in gis]
        res = {}
        for scan in ("units", "ksplit"):
            if scan == "ksplit":
                os.environ["MACJD_GRU_SCAN"] = "ksplit"
            else:
                os.environ.pop("MACJD_GRU_SCAN", None)
            res[scan] = (timeit(lambda: ops.gru_sequence_multi(gis, ws, bs)),
                         timeit(lambda: ops.gru_sequence_multi(gst, ws, bs, n_steps=T)),
                         ops.gru_sequence_multi(gis, ws, bs)[0])
        err = float((res["units"][2] - res["ksplit"][2]).abs().max())
        for scan in ("units", "ksplit"):
            print(f"nets={nets} {scan:7s} per-step gi {res[scan][0]:7.2f} us   static gi {res[scan][1]:7.2f} us", flush=True)
        print(f"nets={nets} max |units - ksplit| = {err:.3e}", flush=True)
    os.environ.pop("MACJD_GRU_SCAN", None)


if __name__ == "__main__":
    main()
