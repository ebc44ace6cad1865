"""Nets leg of __graft_entry__.smoke(): one tiny MAC action selection + one learner update on cuda:0,
checked against the NumPy nets oracle.  Test infrastructure (it imports oracle/): lives under tests/, not in the
product package."""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch


def smoke_nets():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "oracle"))
    import nets_oracle
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    J, R, H = 3, 4, 64
    A, S = 2 * R + 1, 10 * R + 2 * J
    args = SimpleNamespace(n_agents=J, n_actions=A, state_shape=S, obs_shape=S, episode_limit=8, rnn_hidden_dim=H,
                           actor_hidden_dim=128, mixing_embed_dim=64, hyper_hidden_dim=128, lr=5e-4, gamma=0.99,
                           grad_norm_clip=1.0, target_update_interval=200, epsilon_start=1.0, epsilon_finish=0.05,
                           epsilon_anneal_time=100000, device="cuda", use_cuda=True, seed=1)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(S, args)
        learner = QMixLearner(mac, args)
    rng = np.random.default_rng(0)
    E = 16
    obs = rng.standard_normal((E, J, S)).astype(np.float32)
    mac.init_hidden(E)
    mac.keep_q_values = True
    T_, P_ = mac.select_actions(torch.from_numpy(obs).cuda(), torch.ones(E, J, A, dtype=torch.int32, device="cuda"),
                                t_env=0, test_mode=True)
    sd = {k: v.cpu().numpy() for k, v in mac.agent.state_dict().items()}
    h = nets_oracle.agent_forward(sd, obs.reshape(-1, S), np.zeros((E * J, H), np.float32))
    p = nets_oracle.actor_forward(sd, obs.reshape(-1, S))
    q = nets_oracle.q_all_actions(sd, h, p)
    np.testing.assert_allclose(mac.last_q_values.cpu().numpy().reshape(-1, A), q, atol=1e-5)
    np.testing.assert_allclose(mac.hidden_states.cpu().numpy(), h, atol=1e-5)
    B, T = 4, 8
    batch = {
        "state": rng.standard_normal((B, T + 1, S)).astype(np.float32),
        "obs": rng.standard_normal((B, T + 1, J, S)).astype(np.float32),
        "actions_discrete": rng.integers(0, A, (B, T, J, 1)).astype(np.int32),
        "actions_continuous": rng.random((B, T, J, 1)).astype(np.float32),
        "avail_actions": np.ones((B, T + 1, J, A), np.int64),
        "reward": rng.standard_normal((B, T, 1)).astype(np.float32),
        "terminated": np.zeros((B, T, 1), bool), "filled": np.ones((B, T, 1), bool),
        "hidden_state": (0.5 * rng.standard_normal((B, T + 1, J, H))).astype(np.float32), "max_seq_len": T,
    }
    stats = learner.train(batch, {})
    assert all(np.isfinite(v) for v in stats.values()), stats
    print(f"[smoke] nets ok: fused Q-head/GRU-scan/mixer kernels vs oracle, learner step loss={stats['loss']:.4f}")
