"""Synthetic learner batches in the replay buffer's schema (same generator as tests/golden/make_golden_nets.py)."""
import numpy as np


def synthetic_batch(rng, args, B, T, lengths=None):
    J, A, S, H = args.n_agents, args.n_actions, args.state_shape, args.rnn_hidden_dim
    lengths = lengths or [T] * B
    b = {
        "state": rng.standard_normal((B, T + 1, S)).astype(np.float32),
        "obs": rng.standard_normal((B, T + 1, J, S)).astype(np.float32),
        "actions_discrete": rng.integers(0, A, size=(B, T, J, 1)).astype(np.int32),
        "actions_continuous": rng.random((B, T, J, 1)).astype(np.float32),
        "avail_actions": np.ones((B, T + 1, J, A), dtype=np.int64),
        "reward": rng.standard_normal((B, T, 1)).astype(np.float32),
        "terminated": np.zeros((B, T, 1), dtype=np.bool_),
        "filled": np.zeros((B, T, 1), dtype=np.bool_),
        "hidden_state": (0.5 * rng.standard_normal((B, T + 1, J, H))).astype(np.float32),
    }
    for i, L in enumerate(lengths):
        b["filled"][i, :L] = True
        b["terminated"][i, L - 1:] = True
        b["reward"][i, L:] = 0
    b["max_seq_len"] = int(max(lengths))
    return b


def philox4x32_10(c, k):
    """Philox4x32-10 (Random123) on Python ints: counter c[4], key k[2] -> 4 words."""
    c, k = list(c), list(k)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def sample_episodes_mirror(n, N, counter, seed):
    """Host restatement of include/macjd_nets.h macjd_sampler_io: first n images of the keyed permutation of [0, N)."""
    def fmix(h):
        h ^= h >> 16; h = (h * 0x85ebca6b) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xc2b2ae35) & 0xFFFFFFFF; h ^= h >> 16
        return h
    ck = (counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF)
    sk = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    key = philox4x32_10((ck[0], ck[1], 0x53414d50, 0), sk) + philox4x32_10((ck[0], ck[1], 0x53414d50, 1), sk)
    k = 2
    while (1 << k) < N:
        k += 1
    rb, lb = k - k // 2, k // 2
    rmask, lmask = (1 << rb) - 1, (1 << lb) - 1
    out = []
    for t in range(n):
        x = t
        while True:
            L, R = x >> rb, x & rmask
            for r in range(0, 8, 2):
                L = (L ^ fmix(R ^ key[r])) & lmask
                R = (R ^ fmix(L ^ key[r + 1])) & rmask
            x = (L << rb) | R
            if x < N:
                break
        out.append(x)
    return out
