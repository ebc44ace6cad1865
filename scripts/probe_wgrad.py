#!/usr/bin/env python3
"""The learner's grouped weight-gradient launch pair (csrc/macjd_wgrad.hip) on the update's own problem shapes
(3j/4r, 32 episodes x 101 steps): GPU time of the partial-products launch + reduce launch by HIP-graph replay.
MACJD_LIB selects another build (ablation builds: -DMACJD_WG_ABLATE=1 no MFMA loop, 2 no global loads, 4 no partial store)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from bench_kernels import timeit  # noqa: E402


def main():
    if not os.environ.get("MACJD_LIB"):
        entry.build()
    from macjd_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M, Na = 3232, 9696
    shapes = [(M, 384, 46, True), (M, 384, 46, False), (M, 192, 128, True), (M, 64, 128, True), (M, 1, 64, True),
              (Na, 64, 74, True), (Na, 1, 64, True)]
    probs = [(torch.randn(K, m, device=dev), torch.randn(K, n, device=dev), b) for K, m, n, b in shapes]

    def run(sel=None):
        with ops.deferred_wgrad():
            outs = [ops.linear_wgrad(g, x, want_bias=b) for i, (g, x, b) in enumerate(probs) if sel is None or i in sel]
        return outs

    print(f"all {len(probs)} problems: {timeit(run):7.2f} us per launch pair", flush=True)
    for i, (K, m, n, b) in enumerate(shapes):
        print(f"  problem {i} K={K} M={m} N={n}: {timeit(lambda i=i: run({i})):7.2f} us alone", flush=True)


if __name__ == "__main__":
    main()
