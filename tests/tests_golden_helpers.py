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
