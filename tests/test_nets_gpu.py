"""GPU parity of the agent-side hot path: fused MP-DQN Q-head + selection kernel, MAC, mixer, learner,
batched runner + device replay — through the C-ABI library, against the NumPy nets oracle and the
reference-generated fixtures.  Tolerance on Q-values / Q_tot / hidden states: 1e-5 (BASELINE.json)."""
import contextlib
import io
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import yaml

from _harness import GOLDEN, REPO, load_scenario

sys.path.insert(0, os.path.join(REPO, "oracle"))
import nets_oracle  # noqa: E402

from test_nets_cpu import load, make_args, quiet, sd_from  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"


def _gpu_args(d, **kw):
    return make_args(d, device="cuda", use_cuda=True, **kw)


@pytest.mark.parametrize("tag", ["3j4r_h64", "2j2r_h128", "6j8r_h64", "12j16r_h64"])
@pytest.mark.parametrize("N_envs", [5, 4096])
def test_fused_qhead_vs_oracle(tag, N_envs):
    """All-action Q (one launch) == the reference's per-action loop (oracle), incl. at rollout size."""
    from macjd_amd import ops
    from macjd_amd.core.networks import RNNAgent
    g, d = load(tag)
    agent = RNNAgent(d["S"], _gpu_args(d)).to(DEV)
    agent.load_state_dict(sd_from(g, "agent."))
    sdn = {k: v.cpu().numpy() for k, v in agent.state_dict().items()}
    rng = np.random.default_rng(N_envs)
    N = N_envs * d["J"]
    h = (0.7 * rng.standard_normal((N, d["H"]))).astype(np.float32)
    P = rng.random((N, d["A"])).astype(np.float32)
    with torch.no_grad():
        q = agent.q_values_all_actions(torch.from_numpy(h).to(DEV), torch.from_numpy(P).to(DEV))
    q_ref = nets_oracle.q_all_actions(sdn, h, P)
    np.testing.assert_allclose(q.cpu().numpy(), q_ref, atol=TOL, rtol=0)
    # selection stage, greedy with a random availability mask: bit-exact actions wherever the top-2
    # margin exceeds the float tolerance; chosen power is the gathered actor output
    avail = (rng.random((N_envs, d["J"], d["A"])) < 0.7).astype(np.int64)
    avail[..., 0] |= (avail.sum(-1) == 0)
    l1, l2 = agent.fc2_q_head[0], agent.fc2_q_head[2]
    with torch.no_grad():
        base = torch.nn.functional.linear(torch.from_numpy(h).to(DEV), l1.weight[:, :d["H"]], l1.bias)
        T64, Psel, T32, Q = ops.qhead_select(base, torch.from_numpy(P).to(DEV), l1.weight, l2.weight, l2.bias,
                                             d["H"], d["A"], d["J"], torch.from_numpy(avail).to(DEV),
                                             epsilon=0.0, greedy_only=True, seed=1, counter=1, want_q=True)
    np.testing.assert_allclose(Q.cpu().numpy(), q_ref, atol=TOL, rtol=0)
    T_ref, P_ref = nets_oracle.select_greedy(q_ref, P, avail.reshape(N, d["A"]))
    qm = np.where(avail.reshape(N, -1) == 0, -np.inf, q_ref)
    srt = np.sort(qm, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4
    got_T = T64.cpu().numpy().reshape(N)
    np.testing.assert_array_equal(got_T[clear], T_ref[clear])
    assert clear.mean() > 0.9
    np.testing.assert_array_equal(T32.cpu().numpy().reshape(N), got_T)
    np.testing.assert_array_equal(Psel.cpu().numpy().reshape(N), P[np.arange(N), got_T])
    assert (avail.reshape(N, -1)[np.arange(N), got_T] == 1).all()


def test_double_q_helper_equals_argmax_gather():
    """Q_target(h', argmax_a Q_eval) from two Q-head launches == argmax + gather over the two full [N, A] Q tensors
    (reference core/qmix.py:138-147; no availability mask there)."""
    from macjd_amd import ops
    N, H, A = 9696, 64, 9
    gen = torch.Generator().manual_seed(11)
    mk = lambda *sh, sc=1.0: (sc * torch.randn(*sh, generator=gen)).to(DEV)
    be, bt = mk(N, H), mk(N, H)
    Pe, Pt = torch.rand(N, A, generator=gen).to(DEV), torch.rand(N, A, generator=gen).to(DEV)
    he = (mk(H, H + A + 1, sc=0.3), mk(1, H), mk(1, sc=0.1))
    ht = (mk(H, H + A + 1, sc=0.3), mk(1, H), mk(1, sc=0.1))
    got = ops.qhead_double_q(be, Pe, he, bt, Pt, ht, H, A)
    qe = ops.qhead_all_actions(be, Pe, he[0], he[1], he[2], H, A)
    qt = ops.qhead_all_actions(bt, Pt, ht[0], ht[1], ht[2], H, A)
    want = torch.gather(qt, 1, qe.argmax(dim=1, keepdim=True)).squeeze(1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("tag", ["3j4r_h64", "6j8r_h64", "12j16r_h64"])
@pytest.mark.parametrize("N", [9696, 37])
def test_double_q_from_hidden_states_kernel(tag, N):
    """One launch from the hidden states (base products on MFMA, both all-action Q-heads, arg-max, gather) == the
    NumPy oracle's per-action loop (pinned on G3) and the two-launch form on library-GEMM bases: arg-max identical except
    on near-ties, values within 1e-5."""
    from macjd_amd import ops
    from macjd_amd.core.networks import RNNAgent
    g, d = load(tag)
    args = _gpu_args(d)
    H, A = d["H"], d["A"]
    torch.manual_seed(1)
    with quiet():
        ae, at = RNNAgent(d["S"], args).to(DEV), RNNAgent(d["S"], args).to(DEV)
    ae.load_state_dict(sd_from(g, "agent."))
    rng = np.random.default_rng(N)
    h_e = torch.tensor(0.7 * rng.standard_normal((N, H)), dtype=torch.float32, device=DEV)
    h_t = torch.tensor(0.7 * rng.standard_normal((N, H)), dtype=torch.float32, device=DEV)
    P_e = torch.tensor(rng.random((N, A)), dtype=torch.float32, device=DEV)
    P_t = torch.tensor(rng.random((N, A)), dtype=torch.float32, device=DEV)
    heads = [(a.fc2_q_head[0].weight, a.fc2_q_head[0].bias, a.fc2_q_head[2].weight, a.fc2_q_head[2].bias) for a in (ae, at)]
    assert ops.qhead_double_q_fused_supported(h_e, H, A)
    with torch.no_grad():
        out, am = ops.qhead_double_q_from_h(h_e, P_e, heads[0], h_t, P_t, heads[1], H, A, want_argmax=True)
        same, am2 = ops.qhead_double_q_from_h(h_e, P_e, heads[0], h_e, P_e, heads[0], H, A, want_argmax=True)   # shared inputs
    sd_e = {k: v.cpu().numpy() for k, v in ae.state_dict().items()}
    sd_t = {k: v.cpu().numpy() for k, v in at.state_dict().items()}
    q_e = nets_oracle.q_all_actions(sd_e, h_e.cpu().numpy(), P_e.cpu().numpy())
    q_t = nets_oracle.q_all_actions(sd_t, h_t.cpu().numpy(), P_t.cpu().numpy())
    am_ref = q_e.argmax(axis=1)
    am_np = am.cpu().numpy()
    tie = np.take_along_axis(q_e, am_ref[:, None], 1)[:, 0] - np.take_along_axis(q_e, am_np[:, None], 1)[:, 0]
    assert (tie <= 1e-5).all() and (am_np == am_ref).mean() > 0.999
    np.testing.assert_allclose(out.cpu().numpy(), np.take_along_axis(q_t, am_np[:, None], 1)[:, 0], atol=TOL, rtol=0)
    np.testing.assert_allclose(same.cpu().numpy(), q_e.max(axis=1), atol=TOL, rtol=0)
    assert torch.equal(am2, am)


def test_fused_selection_writes_into_caller_rows():
    """out_T32 / out_P: the select kernel stores the chosen actions into caller-provided [E,J,1] rows (the runner's
    staging tensors) — same values as the default agent-major scratch outputs."""
    from macjd_amd import ops
    E, J, A, H = 513, 3, 9, 64
    gen = torch.Generator(device="cpu").manual_seed(5)
    base = torch.randn(E * J, H, generator=gen).to(DEV)
    P = torch.rand(E * J, A, generator=gen).to(DEV)
    W1 = (0.2 * torch.randn(H, H + A + 1, generator=gen)).to(DEV)
    w2, b2 = torch.randn(1, H, generator=gen).to(DEV), torch.zeros(1, device=DEV)
    avail = (torch.rand(E, J, A, generator=gen) < 0.8).to(torch.int32).to(DEV)
    avail[..., 0] = 1
    kw = dict(epsilon=0.25, greedy_only=False, seed=7, counter=3)
    T64, Psel, T32, _ = ops.qhead_select(base, P, W1, w2, b2, H, A, J, avail, **kw)
    stage_T = torch.full((2, E, J, 1), -7, dtype=torch.int32, device=DEV)
    stage_P = torch.full((2, E, J, 1), -7.0, device=DEV)
    T64b, Pb, T32b, _ = ops.qhead_select(base, P, W1, w2, b2, H, A, J, avail, out_T32=stage_T[1], out_P=stage_P[1], **kw)
    assert torch.equal(T64, T64b)
    assert torch.equal(stage_T[1], T64.to(torch.int32)) and torch.equal(stage_P[1], Psel.contiguous())
    assert Pb.data_ptr() == stage_P[1].data_ptr() and T32b.data_ptr() == stage_T[1].data_ptr()
    assert int((stage_T[0] != -7).sum()) == 0 and int((stage_P[0] != -7.0).sum()) == 0


def test_fused_selection_exploration_statistics():
    """epsilon-greedy in-kernel: P(explore) ~ eps, uniform over AVAILABLE actions, deterministic per
    (seed, counter), different across counters."""
    from macjd_amd import ops
    E, J, A, H = 20000, 3, 9, 64
    base = torch.zeros(E * J, H, device=DEV)
    P = torch.rand(E * J, A, device=DEV)
    W1 = torch.zeros(H, H + A + 1, device=DEV)
    W1[:, H + 4] = 1.0                       # action 4 is the greedy one
    w2, b2 = torch.ones(1, H, device=DEV), torch.zeros(1, device=DEV)
    avail = torch.ones(E, J, A, dtype=torch.int32, device=DEV)
    avail[:, :, 7:] = 0
    run = lambda eps, ctr: ops.qhead_select(base, P, W1, w2, b2, H, A, J, avail, epsilon=eps, greedy_only=False,
                                            seed=3, counter=ctr)[0].view(-1)
    t0 = run(0.0, 1)
    assert (t0 == 4).all()
    t1 = run(1.0, 1)
    cnt = torch.bincount(t1, minlength=A).float() / t1.numel()
    assert float(cnt[7:].sum()) == 0.0
    assert float((cnt[:7] - 1 / 7).abs().max()) < 0.01
    t3 = run(0.3, 2)
    frac_not_greedy = float((t3 != 4).float().mean())
    assert abs(frac_not_greedy - 0.3 * 6 / 7) < 0.01
    assert torch.equal(t3, run(0.3, 2)) and not torch.equal(t3, run(0.3, 3))


@pytest.mark.parametrize("tag", ["3j4r_h64", "2j2r_h128", "6j8r_h64", "12j16r_h64"])
def test_mac_on_gpu_matches_reference(tag):
    """BasicMAC.select_actions(test_mode=True) on the HIP path vs the reference's outputs (G3)."""
    from macjd_amd.core.mac import BasicMAC
    g, d = load(tag)
    mac = BasicMAC(d["S"], _gpu_args(d))
    mac.load_state(sd_from(g, "agent."))
    mac.cuda()
    mac.init_hidden(5)
    avail = torch.tensor(g["g3_sel_avail"]).to(DEV)
    obs0 = torch.tensor(g["g3_obs"]).view(5, d["J"], d["S"]).to(DEV)
    for t in range(3):
        T_, P_ = mac.select_actions(obs0 * (1.0 + 0.1 * t), avail, t_env=t, test_mode=True)
        assert T_.dtype == torch.int64 and T_.shape == (5, d["J"], 1) and P_.shape == (5, d["J"], 1)
        np.testing.assert_array_equal(T_.cpu().numpy(), g["g3_sel_T"][t])
        np.testing.assert_allclose(P_.cpu().numpy(), g["g3_sel_P"][t], atol=TOL, rtol=0)
        np.testing.assert_allclose(mac.hidden_states.cpu().numpy(), g["g3_sel_h"][t], atol=TOL, rtol=0)
        assert mac.last_actions_T32.dtype == torch.int32 and mac.last_actions_T32.stride() == (1, 5)


@pytest.mark.parametrize("tag", ["3j4r_h64", "2j2r_h128", "12j16r_h64"])
def test_learner_on_gpu_matches_reference(tag):
    """G5 on the device: stats within 1e-5 (relative to their scale), gradients, None-grad set."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    g, d = load(tag)
    args = _gpu_args(d)
    with quiet():
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "g5_agent0."))
        learner = QMixLearner(mac, args)
    learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
    learner._update_targets()
    assert next(mac.agent.parameters()).is_cuda
    for step in range(len(g["g5_stats"])):      # (12j/16r carries one step)
        pre = f"g5_b{step}_"
        batch = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        batch["max_seq_len"] = int(batch["max_seq_len"])
        stats = learner.train(batch, {})
        got = np.array([stats["loss"], stats["grad_norm"], stats["eval_qtot_avg"], stats["target_qtot_avg"]])
        np.testing.assert_allclose(got, g["g5_stats"][step], rtol=1e-4, atol=TOL)
        named = {"agent." + n: p for n, p in mac.agent.named_parameters()}
        named.update({"mixer." + n: p for n, p in learner.eval_qmix_net.named_parameters()})
        none_ref = set(json.loads(str(g[f"g5_s{step}_grad_none_json"])))
        assert {k for k, p in named.items() if p.grad is None} == none_ref
        for k, p in named.items():
            if k not in none_ref:
                ref = g[f"g5_s{step}_grad.{k}"]
                np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3,
                                           atol=2e-5 * max(1.0, float(np.abs(ref).max())))
        for prefix, module in ((f"g5_s{step}_agent.", mac.agent), (f"g5_s{step}_tmixer.", learner.target_qmix_net)):
            for k, v in sd_from(g, prefix).items():
                np.testing.assert_allclose(module.state_dict()[k].cpu().numpy(), v.numpy(), atol=5e-6, rtol=0)


def test_learner_unroll_on_gpu_vs_oracle_full_size():
    """Reference batch shape (B=32 episodes x T=100, 3j/4r, H=64): the time-parallel unroll + GRU scan +
    fused Q-head on the device == the per-step / per-action NumPy unroll."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    g, d = load("3j4r_h64")
    args = _gpu_args(d, episode_limit=100)
    with quiet():
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "agent."))
        learner = QMixLearner(mac, args)
    rng = np.random.default_rng(0)
    B, T = 32, 100
    obs = rng.standard_normal((B, T, d["J"], d["S"])).astype(np.float32)
    q, _ = learner._get_all_action_q_values_and_params(mac, {"obs": obs}, T)
    sdn = {k: v.cpu().numpy() for k, v in mac.agent.state_dict().items()}
    h_all = nets_oracle.gru_unroll(sdn, obs)
    p_ref = nets_oracle.actor_forward(sdn, obs.reshape(-1, d["S"]))
    q_ref = nets_oracle.q_all_actions(sdn, h_all.reshape(-1, d["H"]), p_ref).reshape(B, T, d["J"], d["A"])
    np.testing.assert_allclose(q.cpu().numpy(), q_ref, atol=TOL, rtol=0)
    np.testing.assert_allclose(mac.hidden_states.cpu().numpy(), h_all[:, -1].reshape(-1, d["H"]), atol=TOL, rtol=0)


def test_batched_runner_replay_learner_end_to_end():
    """E=4096 x 100-step rollout with no per-step host sync, stored into the device replay, sampled and
    trained on: shapes / dtypes / padding quirks of the reference's buffer, finite loss."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    sc, _ = load_scenario("3j4r")
    E = 4096
    env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=42)
    info = env.get_env_info()
    d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=64)
    args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=E + 100, batch_size=32, lr=5e-6,
                     target_update_interval=200)
    args.env_info = info
    torch.manual_seed(42)
    with quiet():
        mac = BasicMAC(info["obs_shape"], args)
        buf = EpisodeReplayBuffer(args)
        learner = QMixLearner(mac, args)
    runner = BatchedEpisodeRunner(env, mac, buf, args)
    ri = runner.run(test_mode=False)
    assert ri["episode_length"] == 100 and ri["n_episodes"] == E and np.isfinite(ri["episode_return"])
    assert abs(ri["action_distribution"].sum() - 1.0) < 1e-5
    assert ri["episode_return"] == pytest.approx(100 * (ri["avg_r_d"] + ri["avg_r_p"] + ri["avg_r_j"]), rel=1e-3)
    assert buf.current_size == E and runner.t_env == 100
    b = buf.buffers
    assert b["filled"][:E].all() and not b["filled"][E:].any()
    assert b["terminated"][:E, -1].all() and not b["terminated"][:E, :-1].any()
    assert float(b["state"][:E, 100].abs().sum()) == 0.0           # zero row at index T (reference quirk)
    assert float(b["hidden_state"][:E, 100].abs().sum()) == 0.0
    np.testing.assert_array_equal(b["state"][7, 0].cpu().numpy(), sc.state_vector())
    assert b["avail_actions"].dtype == torch.int64 and bool((b["avail_actions"][:E, :100] == 1).all())
    assert int(b["actions_discrete"][:E].max()) <= 2 * sc.num_radars and int(b["actions_discrete"][:E].min()) >= 0
    # the stored hidden state is the post-update h_t: row 0 equals one GRU step from zeros on the obs
    with torch.no_grad():
        h1 = mac.agent.forward(env.get_obs().reshape(-1, d["S"])[:3], torch.zeros(3, 64, device=DEV))
    np.testing.assert_allclose(b["hidden_state"][0, 0].cpu().numpy(), h1.cpu().numpy(), atol=TOL)
    # second rollout wraps the ring
    runner.run(test_mode=False)
    assert buf.current_size == E + 100 and buf.current_index == E - 100
    np.random.seed(0)
    for _ in range(3):
        stats = learner.train(buf.sample(32), {})
        assert all(np.isfinite(v) for v in stats.values())
    assert learner.train_step == 3


def test_reference_episode_end_to_end():
    """The reference's own EpisodeRunner episode (seeded exploration, 2 episodes x 100 steps) reproduced by
    the drop-in stack: facade env on the GPU + MAC on host tensors (same torch / NumPy RNG streams)."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import EpisodeRunner
    from macjd_amd.simulation.environment import ElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    g = np.load(os.path.join(GOLDEN, "episode_e2e.npz"))
    a = json.loads(str(g["args_json"]))
    path = os.path.join(tempfile.mkdtemp(prefix="macjd_e2e_"), "s.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(json.loads(str(g["scenario_json"])), f)
    args = SimpleNamespace(**a)
    np.random.seed(42)
    torch.manual_seed(42)
    with quiet():
        env = ElectromagneticEnvironment(args, path)
        args.env_info = env.get_env_info()
        mac = BasicMAC(args.obs_shape, args)
        buf = EpisodeReplayBuffer(args, device="cpu")
        runner = EpisodeRunner(env, mac, buf, args)
    for k, v in sd_from(g, "agent.").items():   # same seed -> same default init as the reference
        assert torch.equal(mac.agent.state_dict()[k], v), k
    with quiet():
        infos = [runner.run(test_mode=False) for _ in range(2)]
    for ep, ri in enumerate(infos):
        assert ri["episode_length"] == int(g[f"ep{ep}_episode_length"])
        for k in ("episode_return", "avg_step_reward", "avg_r_d", "avg_r_p", "avg_r_j", "avg_power_overall"):
            assert ri[k] == pytest.approx(float(g[f"ep{ep}_{k}"]), rel=1e-6, abs=1e-6), (ep, k)
        np.testing.assert_allclose(ri["action_distribution"], g[f"ep{ep}_action_distribution"], atol=1e-12)
    for k, v in buf.buffers.items():
        ref = g[f"buffer_{k}"]
        got = v[:2].numpy()
        if got.dtype.kind in "iub":
            np.testing.assert_array_equal(got, ref, err_msg=k)     # actions, masks, terminated, filled: bit-exact
        else:
            np.testing.assert_allclose(got, ref, atol=TOL, rtol=0, err_msg=k)
    assert mac.action_selector.epsilon == pytest.approx(float(g["epsilon_after"]))
    assert runner.t_env == int(g["t_env_after"])


def test_reference_greedy_episodes_with_mac_on_gpu():
    """BASELINE.json config C1 (reference plumbing, one env) with EVERYTHING on the HIP device: facade env + BasicMAC on
    cuda, so the reference's EpisodeRunner protocol drives fc1 / GRU gates / actor chain / the fused Q-head + selection
    kernel.  Fixture: two greedy (test_mode=True) episodes of the reference's own runner (tests/golden/episode_greedy.npz,
    per-step actions, powers, rewards, FSM states).  Greedy selection consumes no torch RNG, so the HIP path must give the
    same actions (bit-exact integers), rewards within 1e-5, and store nothing in the buffer."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import EpisodeRunner
    from macjd_amd.simulation.environment import ElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    g = np.load(os.path.join(GOLDEN, "episode_greedy.npz"))
    a = json.loads(str(g["args_json"]))
    a.update(device="cuda", use_cuda=True)
    path = os.path.join(tempfile.mkdtemp(prefix="macjd_greedy_"), "s.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(json.loads(str(g["scenario_json"])), f)
    args = SimpleNamespace(**a)
    np.random.seed(7)
    with quiet():
        env = ElectromagneticEnvironment(args, path)
        args.env_info = env.get_env_info()
        mac = BasicMAC(args.obs_shape, args)
        mac.load_state(sd_from(g, "agent."))
        mac.cuda()
        buf = EpisodeReplayBuffer(args, device="cuda")
        runner = EpisodeRunner(env, mac, buf, args)
    assert next(mac.agent.parameters()).is_cuda and runner.device.type == "cuda"
    log = []
    real_step = env.step
    def logged(actions):
        out = real_step(actions)
        log.append((np.array([x[0] for x in actions]), np.array([x[1] for x in actions], dtype=np.float64), out[1],
                    np.array([1 if s_["is_tracking"] else 0 for s_ in out[3]["radar_states"]], dtype=np.uint8)))
        return out
    env.step = logged
    with quiet():
        infos = [runner.run(test_mode=True) for _ in range(2)]
    np.testing.assert_array_equal(np.stack([l[0] for l in log]), g["step_T"])            # greedy actions: bit-exact
    np.testing.assert_array_equal(np.stack([l[3] for l in log]), g["step_track"])        # FSM states: bit-exact
    np.testing.assert_allclose(np.stack([l[1] for l in log]), g["step_P"], atol=TOL, rtol=0)
    np.testing.assert_allclose(np.array([l[2] for l in log]), g["step_reward"], atol=TOL, rtol=0)
    for ep, ri in enumerate(infos):
        assert ri["episode_length"] == int(g[f"ep{ep}_episode_length"])
        for k in ("episode_return", "avg_step_reward", "avg_r_d", "avg_r_p", "avg_r_j", "avg_power_overall"):
            assert ri[k] == pytest.approx(float(g[f"ep{ep}_{k}"]), rel=1e-6, abs=1e-5), (ep, k)
        np.testing.assert_allclose(ri["action_distribution"], g[f"ep{ep}_action_distribution"], atol=1e-12)
    assert buf.current_size == int(g["buffer_size_after"]) == 0 and runner.t_env == int(g["t_env_after"])


def test_fused_rollout_kernel_reproduces_reference_greedy_episodes():
    """The launches bench.py times, pinned on a reference output: two greedy episodes of the REFERENCE's own runner
    (tests/golden/episode_greedy.npz: 7 distinct greedy actions, 31 changes per episode) replayed by
    BatchedEpisodeRunner.rollout_fused(test_mode=True) — ops.agent_episode (GRU cell + all-action Q-head + arg-max for
    all 100 steps in one launch) + env.step_many — on E = 16 replicas of the one env.  The observation is static, so the
    greedy action sequence depends on the agent alone: actions bit-exact in every replica, chosen powers within 1e-5; the
    deterministic reward term (avg_r_p), the mean power and the action distribution equal the reference's run_info; the
    Monte-Carlo terms are compared with the step-by-step env.step() on the same actions (same Philox streams, same
    decisions: rewards within 1e-5, terminated flags identical)."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.scenario import Scenario
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    g = np.load(os.path.join(GOLDEN, "episode_greedy.npz"))
    a = json.loads(str(g["args_json"]))
    a.update(device="cuda", use_cuda=True)
    args = SimpleNamespace(**a)
    sc = Scenario.from_dict(json.loads(str(g["scenario_json"])))
    E, T, J = 16, args.episode_limit, args.n_agents
    assert len(np.unique(g["step_T"])) >= 5, "the fixture must exercise the arg-max"

    def build():
        env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=3)
        args.env_info = env.get_env_info()
        with quiet():
            mac = BasicMAC(args.obs_shape, args)
            mac.load_state(sd_from(g, "agent."))
            mac.cuda()
            buf = EpisodeReplayBuffer(args, device="cuda")
        return env, mac, BatchedEpisodeRunner(env, mac, buf, args)

    env, mac, runner = build()
    assert runner.fused_rollout_available()
    env2, _, _ = build()                      # step-by-step twin of the environment (same seed: same Philox streams)
    env2.reset()
    for ep in range(2):
        ri = runner.run(test_mode=True, store=False, sync_stats=True)
        st = runner.stage
        got_T = st["actions_discrete"][:T, :, :, 0].cpu().numpy()               # [T, E, J]
        got_P = st["actions_continuous"][:T, :, :, 0].cpu().numpy()
        ref_T, ref_P = g["step_T"][ep * T:(ep + 1) * T], g["step_P"][ep * T:(ep + 1) * T]     # [T, J]
        for e in range(E):
            np.testing.assert_array_equal(got_T[:, e], ref_T, err_msg=f"episode {ep}, replica {e}")
        np.testing.assert_allclose(got_P, np.broadcast_to(ref_P[:, None, :], got_P.shape), atol=TOL, rtol=0)
        for k in ("avg_r_p", "avg_power_overall"):
            assert ri[k] == pytest.approx(float(g[f"ep{ep}_{k}"]), rel=1e-6, abs=1e-5), (ep, k)
        np.testing.assert_allclose(ri["action_distribution"], g[f"ep{ep}_action_distribution"], atol=1e-6)
        # the env half of the fused rollout against single-step launches on the reference's actions
        if ep > 0:
            env2.reset()
        for t in range(T):
            rew, term, _ = env2.step(torch.from_numpy(np.broadcast_to(ref_T[t].astype(np.int32), (E, J)).copy()).to(DEV),
                                     torch.from_numpy(np.broadcast_to(ref_P[t].astype(np.float32), (E, J)).copy()).to(DEV))
            np.testing.assert_allclose(st["reward"][t].view(E).cpu().numpy(), rew.cpu().numpy(), atol=TOL, rtol=0)
            assert torch.equal(st["terminated"][t].view(E), term)


def test_replay_buffer_on_gpu_against_reference():
    """G7 with the buffer in HBM: 6 episodes (lengths 6,2,6,4,1,6) into a 4-slot ring — stored arrays incl. padding /
    filled / terminated, ring cursor, np.random-driven sampling (indices, max_seq_len, truncation) — and the learner's
    one-launch gather of sampled episodes (macjd_gather_rows) against index_select on the same device tensors."""
    from macjd_amd import ops
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    g = np.load(os.path.join(GOLDEN, "buffer_g7.npz"))
    d = json.loads(str(g["dims_json"]))
    args = _gpu_args(d, buffer_size=d["buffer_size"], episode_limit=d["episode_limit"])
    with quiet():
        buf = EpisodeReplayBuffer(args, device="cuda")
    keys = ("state", "obs", "actions_discrete", "actions_continuous", "avail_actions", "reward", "terminated",
            "hidden_state")
    for i in range(6):
        buf.store_episode({k: [g[f"g7_ep{i}_{k}"]] for k in keys})
        assert [buf.current_index, buf.current_size] == g[f"g7_after{i}_index_size"].tolist()
    for k, v in buf.buffers.items():
        ref = g[f"g7_final_{k}"]
        assert v.is_cuda and tuple(v.shape) == ref.shape and str(v.dtype).replace("torch.", "") == str(ref.dtype).replace("bool_", "bool"), k
        np.testing.assert_array_equal(v.cpu().numpy(), ref, err_msg=k)
    np.random.seed(5)
    for i, n in enumerate((2, 3, 4)):
        s = buf.sample(n)
        assert s["max_seq_len"] == int(g[f"g7_sample{i}_max_seq_len"])
        for k in ("state", "reward", "filled", "terminated", "actions_discrete", "hidden_state"):
            assert s[k].is_cuda
            np.testing.assert_array_equal(s[k].cpu().numpy(), g[f"g7_sample{i}_{k}"], err_msg=f"sample {i} {k}")
    # the graphed update's gather kernel on the same ring: whole rows of every key, destination rows padded to T + 1
    idx = torch.as_tensor(g["g7_sample2_idx"], dtype=torch.int64, device=DEV)
    # (keys whose episode row is a whole number of 32-bit words: at this fixture's T = 6 the bool keys are not, and
    # the learner then gathers with index_select — QMixLearner.enable_graphs checks gather_rows_supported)
    names = [k for k, v in buf.buffers.items() if k != "avail_actions" and (v[0].numel() * v.element_size()) % 4 == 0]
    srcs = [buf.buffers[k] for k in names]
    assert ops.gather_rows_supported(srcs) and len(names) >= 6
    T1 = d["episode_limit"] + 1
    dsts = [torch.zeros((idx.numel(), T1) + tuple(v.shape[2:]), dtype=v.dtype, device=DEV) for v in srcs]
    ops.gather_rows(idx, srcs, dsts)
    for k, v, o in zip(names, srcs, dsts):
        assert torch.equal(o[:, :v.shape[1]], v.index_select(0, idx)), k
        assert not o[:, v.shape[1]:].any(), k


@pytest.mark.parametrize("tag", ["3j4r_h64", "2j2r_h128", "6j8r_h64", "12j16r_h64"])
@pytest.mark.parametrize("M", [3232, 37])
def test_fused_mixer_chain_vs_reference_and_unfused_path(tag, M, monkeypatch):
    """The whole mixer as one MFMA launch per direction (csrc/macjd_mixer.hip): Q_tot against the reference's G4 values
    (incl. the clamp-saturating weight set), and — on random inputs at the learner's size and at a ragged size — Q_tot,
    dL/dq and every parameter gradient against the unfused path (LayerNorm + library GEMMs + tail kernel + autograd)."""
    import copy
    from macjd_amd import ops
    from macjd_amd.core.networks import QMixer
    g, d = load(tag)
    mixer = QMixer(_gpu_args(d)).to(DEV)
    sd = sd_from(g, "mixer.")
    mixer.load_state_dict(sd)
    assert mixer.fused_available(torch.zeros(1, device=DEV)), "every BASELINE.json size has the one-launch mixer"
    q4, s4 = torch.tensor(g["g4_q"]).to(DEV), torch.tensor(g["g4_s"]).to(DEV)
    with torch.no_grad():
        out = mixer(q4, s4)
        big = copy.deepcopy(mixer)
        for p in big.parameters():
            p.mul_(25.0)
        out_big = big(q4.view(4, 10, d["J"]), s4.view(4, 10, d["S"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["g4_qtot"], atol=TOL, rtol=1e-6)
    scale = float(np.abs(g["g4_qtot_big"]).max())
    np.testing.assert_allclose(out_big.cpu().numpy(), g["g4_qtot_big"], atol=TOL * scale, rtol=1e-5)
    # gradients: fused vs unfused on the same module
    rng = np.random.default_rng(M)
    q = torch.tensor(rng.standard_normal((M, d["J"])), dtype=torch.float32, device=DEV, requires_grad=True)
    s = torch.tensor(3.0 * rng.standard_normal((M, d["S"])), dtype=torch.float32, device=DEV)
    gy = torch.tensor(rng.standard_normal((M, 1)), dtype=torch.float32, device=DEV)
    with torch.no_grad():   # spread the hyper-network outputs so that every clamp has rows on both sides
        for p in mixer.parameters():
            p.mul_(3.0)
    def run(fused):
        monkeypatch.setattr(QMixer, "fused", fused)
        for p in list(mixer.parameters()) + [q]:
            p.grad = None
        y = mixer(q, s)
        y.backward(gy)
        return y.detach().cpu().numpy(), q.grad.cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in mixer.named_parameters()}
    y_f, gq_f, gp_f = run(True)
    y_u, gq_u, gp_u = run(False)
    ys = max(1.0, float(np.abs(y_u).max()))
    np.testing.assert_allclose(y_f, y_u, atol=TOL * ys, rtol=1e-5)
    np.testing.assert_allclose(gq_f, gq_u, atol=2e-5 * max(1.0, float(np.abs(gq_u).max())), rtol=1e-4)
    assert set(gp_f) == set(gp_u)
    for k in gp_u:
        tol = 3e-5 * max(1.0, float(np.abs(gp_u[k]).max()))
        np.testing.assert_allclose(gp_f[k], gp_u[k], atol=tol, rtol=1e-3, err_msg=k)
    # inference launch (no autograd) == training forward
    with torch.no_grad():
        monkeypatch.setattr(QMixer, "fused", True)
        assert torch.equal(mixer(q.detach(), s).cpu(), torch.from_numpy(y_f))


@pytest.mark.parametrize("tag", ["3j4r_h64", "6j8r_h64", "12j16r_h64"])
def test_bf16_mixer_error_bound_vs_reference(tag):
    """BASELINE.json config C5 ("bf16 mixer MFMA path"): hyper-network GEMMs with bf16 inputs / fp32 accumulation on the
    HIP device against the reference's fp32 Q_tot (G4 fixtures, incl. the weight set that saturates every clamp) and
    against the fp32 HIP path's parameter gradients.  bf16 keeps 8 significant bits: the stated bounds are
    |dQ_tot| <= 2^-6 max|Q_tot| (values; 2^-4 for the clamp-saturating weight set) and a relative L2 error <= 2^-3 per gradient tensor (measured: up to 0.07 on
    the first hyper-network layer, whose gradient passes through two bf16 products) — i.e. this option is
    OUTSIDE the 1e-5 parity bar by design (it is off by default), and the fp32 path on the same inputs holds 1e-5."""
    import copy
    from macjd_amd.core.networks import QMixer
    g, d = load(tag)
    mk = lambda dt: QMixer(_gpu_args(d, mixer_dtype=dt)).to(DEV)
    m32, m16 = mk("fp32"), mk("bf16")
    sd = sd_from(g, "mixer.")
    m32.load_state_dict(sd); m16.load_state_dict(sd)
    assert m16.bf16_hyper and not m32.bf16_hyper
    q, s = torch.tensor(g["g4_q"]).to(DEV), torch.tensor(g["g4_s"]).to(DEV)
    def run(m, scale=1.0):
        m = copy.deepcopy(m)
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(scale)
        for p in m.parameters():
            p.grad = None
        out = m(q.view(4, 10, d["J"]), s.view(4, 10, d["S"]))
        out.sum().backward()
        return out.detach().reshape(-1).cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()
                                                         if p.grad is not None}
    # (the x25 weight set drives the hyper-network outputs across their clamp thresholds, where a bf16 rounding flips a
    # whole term: measured 2.7 - 4.6 % of max|Q_tot| there, hence the looser value bound for that case)
    for scale, key, vbound in ((1.0, "g4_qtot", 2.0 ** -6), (25.0, "g4_qtot_big", 2.0 ** -4)):
        ref = g[key].reshape(-1)
        o32, g32 = run(m32, scale)
        o16, g16 = run(m16, scale)
        top = float(np.abs(ref).max())
        np.testing.assert_allclose(o32, ref, atol=TOL * max(1.0, top), rtol=1e-5)          # fp32 HIP path: the parity bar
        err = float(np.abs(o16 - ref).max())
        assert err <= vbound * top, (key, err, top)
        assert err > 0.0                                                                   # ... and bf16 really ran
        for k in g32:
            assert np.isfinite(g16[k]).all(), (key, k)
            den = float(np.linalg.norm(g32[k]))
            if den > 0 and scale == 1.0:   # (with every clamp saturated the gradients are near-zero noise: values only)
                rel = float(np.linalg.norm(g16[k] - g32[k])) / den
                assert rel <= 2.0 ** -3, (key, k, rel)


@pytest.mark.parametrize("scan", ["units", "ksplit"])
@pytest.mark.parametrize("H", [64, 128])
@pytest.mark.parametrize("B,T,J", [(1, 1, 1), (3, 7, 2), (32, 100, 3)])
def test_gru_sequence_kernel(H, B, T, J, scan, monkeypatch):
    """Fused GRU scan (one launch, eval + target weights together) == step-by-step torch.nn.GRUCell
    arithmetic on the host (the oracle formulation), with and without an initial state.  H = 64 has two kernels: the
    unit-split scan (default) and the K-split one (MACJD_GRU_SCAN=ksplit, also the H = 128 kernel)."""
    from macjd_amd import ops
    if scan == "ksplit":
        if H != 64:
            pytest.skip("H = 128 has the K-split kernel only")
        monkeypatch.setenv("MACJD_GRU_SCAN", "ksplit")
        from macjd_amd import _native
        _native.reload_options()
    rng = np.random.default_rng(H + T)
    gis = [torch.tensor(rng.standard_normal((B, T, J, 3 * H)), dtype=torch.float32) for _ in range(2)]
    ws = [torch.tensor(rng.standard_normal((3 * H, H)) / np.sqrt(H), dtype=torch.float32) for _ in range(2)]
    bs = [torch.tensor(0.1 * rng.standard_normal(3 * H), dtype=torch.float32) for _ in range(2)]
    h0 = torch.tensor(0.5 * rng.standard_normal((B * J, H)), dtype=torch.float32)
    ref = [ops.gru_sequence_reference(gis[0], ws[0], bs[0]), ops.gru_sequence_reference(gis[1], ws[1], bs[1], h0)]
    cell = torch.nn.GRUCell(H, H)   # cross-check the reference formulation against torch's own cell
    with torch.no_grad():
        cell.weight_hh.copy_(ws[0]); cell.bias_hh.copy_(bs[0])
        cell.weight_ih.copy_(torch.eye(3 * H)[:, :H]); cell.bias_ih.zero_()
    got = ops.gru_sequence_multi([g.to(DEV) for g in gis], [w.to(DEV) for w in ws], [b.to(DEV) for b in bs],
                                 [None, h0.to(DEV)])
    for g_, r_ in zip(got, ref):
        assert g_.shape == (B, T, J, H)
        np.testing.assert_allclose(g_.cpu().numpy(), r_.numpy(), atol=TOL, rtol=0)
    single = ops.gru_sequence(gis[0].to(DEV), ws[0].to(DEV), bs[0].to(DEV))
    assert torch.equal(single, got[0])
    # one input transform per sequence ([B, 1, J, 3H], static observation) == the same row at every step
    st = ops.gru_sequence_multi([g[:, :1].contiguous().to(DEV) for g in gis], [w.to(DEV) for w in ws], [b.to(DEV) for b in bs],
                                [None, h0.to(DEV)], n_steps=T)
    ref_st = [ops.gru_sequence_reference(gis[0][:, :1].expand(B, T, J, 3 * H), ws[0], bs[0]),
              ops.gru_sequence_reference(gis[1][:, :1].expand(B, T, J, 3 * H), ws[1], bs[1], h0)]
    for g_, r_ in zip(st, ref_st):
        np.testing.assert_allclose(g_.cpu().numpy(), r_.numpy(), atol=TOL, rtol=0)


@pytest.mark.parametrize("T,N,B,steps,single,shared", [(12, 40, 8, 7, True, "1"), (100, 48, 32, 4, True, "1"),
                                                       (100, 48, 32, 4, False, "1"), (100, 48, 32, 4, True, "0")])
def test_graphed_train_equals_eager_train(T, N, B, steps, single, shared, monkeypatch):
    """HIP-graph replay (one graph per update for a single process; two graphs around the gradient all-reduce otherwise,
    forced here with force_two_graphs so that the layout every rank of a multi-GPU job runs is covered too; with the
    frozen agent body evaluated once for both controllers, and — MACJD_SHARED_BODY=0 — once per controller)
    of the learner step == the eager step: same sampled episodes -> same losses and
    the same weights after several updates incl. a target sync.  The second size is the benchmark's (32 episodes x
    101 steps x 3 agents = 9696 rows): only there do the split-K / grouped weight gradients, the row-dot and the
    fused-ReLU paths engage (they need >= 1024 rows), and every step samples different episodes, so a gradient that
    arrived one replay late would show in grad_norm."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    monkeypatch.setenv("MACJD_SHARED_BODY", shared)
    g, d = load("3j4r_h64")
    def build():
        args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=3)
        with quiet():
            mac = BasicMAC(d["S"], args)
            mac.load_state(sd_from(g, "g5_agent0."))
            learner = QMixLearner(mac, args)
            buf = EpisodeReplayBuffer(args)
        learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
        learner._update_targets()
        full = synthetic_batch(np.random.default_rng(9), args, N, T)
        for k, v in buf.buffers.items():
            v.copy_(torch.as_tensor(full[k]).to(v.dtype))
        buf.current_size, buf.current_index = N, 0
        buf.episode_lengths[:] = T
        return mac, learner, buf
    mac_e, eager, buf_e = build()
    mac_g, graphed, buf_g = build()
    graphed.enable_graphs(buf_g, B, force_two_graphs=not single)   # warm-up updates are undone in place
    assert graphed._g_shared_body == (shared == "1")
    rng = np.random.default_rng(1)
    # the graphed update hands back views of ONE static output tensor; stats_row snapshots each update's four scalars
    # into its own row on the device (no host sync inside the loop), read back once at the end
    hist = torch.zeros((steps, 4), device=DEV)
    eager_stats = []
    for step in range(steps):
        idx = rng.choice(N, B, replace=False)
        eager_stats.append(eager.train(buf_e.sample(B, indices=idx), {}))
        sg = graphed.train_from_buffer(indices=idx, sync_stats=False, stats_row=hist[step])
        assert sg["loss"].data_ptr() == hist[step].data_ptr()
    rows = hist.cpu().numpy()
    for step, se in enumerate(eager_stats):
        for col, k in enumerate(("loss", "eval_qtot_avg", "target_qtot_avg", "grad_norm")):
            assert rows[step, col] == pytest.approx(se[k], rel=1e-4, abs=1e-6), (step, k)
    assert len({float(r[0]) for r in rows}) == steps      # every update logged its own loss, not the last one's
    assert graphed.train_step == eager.train_step == steps and graphed.last_target_update_step == 3 * (steps // 3)
    # full-size batches: every weight gradient was written straight into the flat gradient vector (no packing copy in
    # the captured graph); the small size runs on stock autograd gradients, which are packed
    assert (graphed.grad_pack_launches == 0) == (B * (T + 1) * 3 >= 1024), graphed.grad_pack_launches
    # weights after the Adam steps (lr 5e-4): Adam's m / sqrt(v) amplifies summation-order noise of near-zero gradient
    # entries (the graphed path reduces over full-length rows), hence 2e-5 rather than float epsilon
    for (k, a), b in zip(mac_e.agent.state_dict().items(), mac_g.agent.state_dict().values()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-5, rtol=0, err_msg=k)
    for (k, a), b in zip(eager.target_qmix_net.state_dict().items(), graphed.target_qmix_net.state_dict().values()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-5, rtol=0, err_msg=k)


@pytest.mark.parametrize("actor_in_scan", ["1", "0"])
def test_learner_static_observation_hoist_equals_per_step_evaluation(actor_in_scan, monkeypatch):
    """(actor_in_scan: the actor chain of the step-0 rows inside the scan launch's prologue — the default — or as its own
    launch on the origin stream.)  Episodes stored by the batched runner carry ``buffer.obs_static``: the graphed update then evaluates the input
    transform and the actor chain on the B * J step-0 rows only (the scan reads one input transform per sequence) instead
    of on all B * (T + 1) * J identical rows.  Same kernels on the same row values: statistics and weights after several
    updates equal the per-step evaluation (MACJD_LEARNER_STATIC_OBS=0); a single stored episode of unknown provenance
    switches the buffer's flag off and the captured update refuses to replay."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    sc, _ = load_scenario("3j4r")
    E, Bsz = 64, 32
    monkeypatch.setenv("MACJD_ACTOR_IN_SCAN", actor_in_scan)
    def build(hoist):
        monkeypatch.setenv("MACJD_LEARNER_STATIC_OBS", "1" if hoist else "0")
        env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=5)
        info = env.get_env_info()
        d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=64)
        args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=2 * E, batch_size=Bsz, lr=1e-3, epsilon_start=0.5,
                         target_update_interval=2)
        args.env_info = info
        torch.manual_seed(3)
        with quiet():
            mac = BasicMAC(info["obs_shape"], args)
            buf = EpisodeReplayBuffer(args)
            learner = QMixLearner(mac, args)
        runner = BatchedEpisodeRunner(env, mac, buf, args)
        runner.run(sync_stats=False)
        assert buf.obs_static is True
        learner.enable_graphs(buf, Bsz)
        assert learner._g_obs_static == hoist and learner._g_actor_in_scan == (actor_in_scan == "1")
        return learner, buf, mac, runner
    la, ba, ma, ra = build(True)
    lb, bb, mb, _ = build(False)
    for k in ba.buffers:
        assert torch.equal(ba.buffers[k], bb.buffers[k]), k
    rng = np.random.default_rng(0)
    for step in range(4):
        idx = rng.choice(E, Bsz, replace=False)
        sa, sb = la.train_from_buffer(indices=idx), lb.train_from_buffer(indices=idx)
        for k in sa:
            assert sa[k] == pytest.approx(sb[k], rel=1e-4, abs=1e-6), (step, k)
    # (the scan computes its input transform in-kernel on the VALU in this mode, the per-step path on the MFMA chain:
    # ~1e-7 on the hidden states; Adam's m / sqrt(v) amplifies that on near-zero gradient entries, hence 2e-5)
    for (k, a), b in zip(ma.agent.state_dict().items(), mb.agent.state_dict().values()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-5, rtol=0, err_msg=k)
    for (k, a), b in zip(la.eval_qmix_net.state_dict().items(), lb.eval_qmix_net.state_dict().values()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-5, rtol=0, err_msg=k)
    # the second rollout re-stores the same envs' static rows: those slots are recognised and not rewritten, a different
    # slot range is filled normally
    ra.run(sync_stats=False)
    assert ba.current_size == 2 * E and torch.equal(ba.buffers["obs"][:E], ba.buffers["obs"][E:])
    # an episode of unknown provenance: flag off, the captured static-observation update refuses to run
    ep = {k: [v[0].cpu().numpy()] for k, v in ba.buffers.items() if k != "filled"}
    with quiet():
        ba.store_episode(ep)
    assert ba.obs_static is False
    with pytest.raises(RuntimeError, match="enable_graphs"):
        la.train_from_buffer()


def test_shared_agent_body_equals_two_controller_unroll(monkeypatch):
    """Reference-faithful training never changes fc1 / GRU / actor (qmix.py:161-184), and the target controller is a
    copy (qmix.py:53): the learner evaluates that body ONCE for both networks.  Bitwise equal to the two-controller
    unroll (different Q-heads, as after training); the sharing stops as soon as either body is written to and comes back
    with the next hard target sync."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    g, d = load("3j4r_h64")
    args = _gpu_args(d, episode_limit=100)
    with quiet():
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "agent."))
        learner = QMixLearner(mac, args)
    with torch.no_grad():   # "trained" eval head: the heads differ, the bodies do not
        for p in mac.agent.fc2_q_head.parameters():
            p.add_(0.05 * torch.randn_like(p))
    assert learner._body_is_shared()
    obs = torch.randn(32, 101, d["J"], d["S"], device=DEV)
    with torch.no_grad():
        tq, eq = learner._all_action_q_multi([learner.target_mac, mac], obs)
        monkeypatch.setenv("MACJD_SHARED_BODY", "0")
        assert not learner._body_is_shared()
        tq2, eq2 = learner._all_action_q_multi([learner.target_mac, mac], obs)
        monkeypatch.delenv("MACJD_SHARED_BODY")
    assert torch.equal(tq, tq2) and torch.equal(eq, eq2) and not torch.equal(tq, eq)
    with torch.no_grad():
        mac.agent.fc1.weight.mul_(1.01)          # e.g. another optimiser training the body
    assert not learner._body_is_shared()
    with torch.no_grad():
        tq3, eq3 = learner._all_action_q_multi([learner.target_mac, mac], obs)
    assert torch.equal(tq3, tq) and not torch.equal(eq3, eq)   # the target body kept its weights, the eval one moved
    learner._update_targets()
    assert learner._body_is_shared()


@pytest.mark.parametrize("fused", [False, True])
def test_graphed_rollout_equals_eager_rollout(fused):
    """One HIP-graph launch per episode batch == eager launches — of the step-by-step rollout, and of the whole-episode
    (fused) rollout, whose reset / fills / three launches are captured as one graph too: identical actions, rewards,
    hidden states over two consecutive episode batches (epsilon annealing + Philox counters advance), and over a batch cut
    after 20 steps (its own, shorter graph)."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    sc, _ = load_scenario("3j4r")
    E = 512
    def build():
        env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=5)
        info = env.get_env_info()
        d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=64)
        args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=2 * E, epsilon_start=0.6,
                         epsilon_anneal_time=300)
        args.env_info = info
        torch.manual_seed(3)
        with quiet():
            mac = BasicMAC(info["obs_shape"], args)
            mac.cuda()
            buf = EpisodeReplayBuffer(args)
        r = BatchedEpisodeRunner(env, mac, buf, args)
        r.fused_rollout = fused
        assert r.fused_rollout_available() == fused
        return r, buf, mac
    r_e, b_e, m_e = build()
    r_g, b_g, m_g = build()
    r_g.enable_graph()
    if fused:
        r_g.enable_graph(n_steps=20)
        assert sorted(r_g._graphs) == [20, 100]
    for ep in range(2):
        ie = r_e.run(sync_stats=True)
        ig = r_g.run(sync_stats=True)
        assert ie["episode_return"] == pytest.approx(ig["episode_return"], rel=1e-6)
        np.testing.assert_allclose(ie["action_distribution"], ig["action_distribution"], atol=1e-7)
    assert r_e.t_env == r_g.t_env == 200
    assert m_e.action_selector.epsilon == pytest.approx(m_g.action_selector.epsilon)
    # the MAC's recurrent state after the batch is the captured launches' final h, not a stale eager tensor
    np.testing.assert_allclose(m_g.hidden_states.cpu().numpy(), m_e.hidden_states.cpu().numpy(), atol=1e-6, rtol=0)
    for k in b_e.buffers:
        a, b = b_e.buffers[k], b_g.buffers[k]
        if a.dtype.is_floating_point:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-6, rtol=0, err_msg=k)
        else:
            assert torch.equal(a, b), k
    if fused:   # a batch cut after 20 steps: eager launches vs the 20-step graph (same launches: bit for bit)
        r_e.rollout_fused(n_steps=20)
        r_g.rollout_fused(n_steps=20)
        assert r_e.t_env == r_g.t_env == 220
        for k in ("hidden_state", "actions_discrete", "actions_continuous", "reward", "terminated"):
            assert torch.equal(r_e.stage[k][:20], r_g.stage[k][:20]), k


@pytest.mark.parametrize("scenario,per_env", [("3j4r", False), ("3j4r", True), ("6j8r", False), ("2j2r_shipped", False), ("12j16r", False)])
def test_fused_episode_rollout_vs_step_by_step(scenario, per_env):
    """The whole episode batch in three launches (ops.agent_episode: GRU cell + all-action Q-head + epsilon-greedy for all
    100 steps; env.step_many: all env steps as independent work items) against the step-by-step rollout with the same
    weights, seeds and exploration counters.  Both paths use the same exploration draws and Monte-Carlo streams; the hidden
    states differ by the summation order of the recurrent product (library GEMM vs MFMA chain), so: hidden states within
    1e-5, the chosen actions identical except on near-ties (>= 99.5 % here), and wherever all agents of an env-step chose
    the same actions the rewards / terminated flags are bit-identical.  Every stored hidden state and action is also
    checked for self-consistency with stock torch ops (greedy pass: the chosen action's Q is the maximum within 1e-5)."""
    from macjd_amd import ops
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.scenario import ScenarioBatch
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    sc, g_ = load_scenario(scenario)
    E = 200   # not a multiple of 16: the last workgroup is ragged
    def build(fused):
        if per_env:
            batch = ScenarioBatch.randomized(json.loads(str(g_["scenario_json"])), E, seed=8)
            env = BatchedElectromagneticEnvironment(scenario_batch=batch, device=DEV, seed=5)
        else:
            env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=5)
        info = env.get_env_info()
        d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=64)
        args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=2 * E, epsilon_start=0.3, epsilon_anneal_time=500)
        args.env_info = info
        torch.manual_seed(3)
        with quiet():
            mac = BasicMAC(info["obs_shape"], args)
            with torch.no_grad():   # spread the Q-values: fewer near-ties between the two summation orders
                for p_ in mac.agent.fc2_q_head.parameters():
                    p_.mul_(3.0)
            mac.cuda()
            buf = EpisodeReplayBuffer(args)
        r = BatchedEpisodeRunner(env, mac, buf, args)
        r.fused_rollout = fused
        assert r.fused_rollout_available() == fused
        return r, buf, mac, args
    rf, bf, mf, args = build(True)
    rs, bs, ms, _ = build(False)
    for ep in range(2):
        i_f = rf.run(sync_stats=True)
        i_s = rs.run(sync_stats=True)
    assert rf.t_env == rs.t_env == 200 and mf.action_selector.epsilon == pytest.approx(ms.action_selector.epsilon)
    B = {k: (bf.buffers[k], bs.buffers[k]) for k in bf.buffers}
    np.testing.assert_allclose(B["hidden_state"][0].cpu().numpy(), B["hidden_state"][1].cpu().numpy(), atol=2e-5, rtol=0)
    same = (B["actions_discrete"][0] == B["actions_discrete"][1])                    # [N, T, J, 1]
    assert float(same.float().mean()) > 0.995
    agree = same.all(dim=2).squeeze(-1)                                              # [N, T]: all agents agree
    pa, pb = B["actions_continuous"]
    assert torch.equal(pa[same], pb[same])
    # (the many-step launch decides the Monte-Carlo compares through its float32 probability filter: same decisions, reward
    # terms within (R + J) 4e-7 of the all-float64 single-step launches)
    assert float((B["reward"][0].squeeze(-1)[agree] - B["reward"][1].squeeze(-1)[agree]).abs().max()) <= 1e-5
    assert torch.equal(B["terminated"][0], B["terminated"][1]) and torch.equal(B["filled"][0], B["filled"][1])
    for k in ("state", "obs", "avail_actions"):
        assert torch.equal(*B[k]), k
    assert torch.allclose(mf.hidden_states, ms.hidden_states, atol=2e-5)
    for k in ("episode_return", "avg_r_d", "avg_r_p", "avg_r_j", "avg_power_overall"):
        assert i_f[k] == pytest.approx(i_s[k], rel=2e-3, abs=2e-3), k
    # self-consistency of a GREEDY fused episode with stock torch ops on the stored rows
    rf.run(test_mode=True, store=False, sync_stats=False)
    st = rf.stage
    T, J, H = rf.episode_limit, rf.n_agents, 64
    a = mf.agent
    with torch.no_grad():
        obs = st["obs"][0].reshape(E * J, -1)
        P = a.actor(obs)
        x = torch.relu(a.fc1(obs))
        h = torch.zeros(E * J, H, device=DEV)
        worst = 0.0
        for t in range(T):
            h = a.rnn(x, h)
            np.testing.assert_allclose(st["hidden_state"][t].reshape(E * J, H).cpu().numpy(), h.cpu().numpy(), atol=2e-5, rtol=0)
            h = st["hidden_state"][t].reshape(E * J, H)      # follow the stored trajectory
            q = a.q_values_all_actions(h, P)
            ch = st["actions_discrete"][t].reshape(E * J).long()
            gap = q.max(dim=1).values - q.gather(1, ch.view(-1, 1)).squeeze(1)
            worst = max(worst, float(gap.max()))
            assert torch.allclose(st["actions_continuous"][t].reshape(E * J), P.gather(1, ch.view(-1, 1)).squeeze(1), atol=1e-6, rtol=1e-5)
        assert worst <= 1e-5 * max(1.0, float(q.abs().max())), worst


@pytest.mark.parametrize("per_env", [False, True])
def test_static_observation_hoist_gives_identical_buffers(per_env):
    """The observation never changes within an episode (env.observation_is_static), so the batched runner has the MAC
    evaluate the actor chain and the fc1 -> W_ih transform once per episode batch (one row when the observation is a
    broadcast, E rows with per-env scenarios) instead of at each of the 100 steps.  Same kernels on the same row values:
    every replay key is bit-identical to the per-step evaluation (hoist_static_obs = False), also from a HIP graph."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.scenario import ScenarioBatch, ring_scenario_dict
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    sc, _ = load_scenario("3j4r")
    E = 320
    def build(hoist, graph):
        if per_env:
            batch = ScenarioBatch.randomized(ring_scenario_dict(3, 4), E, seed=8)
            env = BatchedElectromagneticEnvironment(scenario_batch=batch, device=DEV, seed=5)
        else:
            env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=5)
        info = env.get_env_info()
        d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=64)
        args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=2 * E, epsilon_start=0.5, epsilon_anneal_time=300)
        args.env_info = info
        torch.manual_seed(3)
        with quiet():
            mac = BasicMAC(info["obs_shape"], args)
            mac.cuda()
            buf = EpisodeReplayBuffer(args)
        r = BatchedEpisodeRunner(env, mac, buf, args)
        assert r.hoist_static_obs
        r.fused_rollout = False      # this test is about the step-by-step path (eager and graph-replayed)
        r.hoist_static_obs = hoist
        if graph:
            r.enable_graph()
        return r, buf, mac
    runs = [build(True, False), build(False, False), build(True, True)]
    for ep in range(2):
        for r, _, _ in runs:
            r.run(sync_stats=False)
    assert runs[0][2].static_inputs is not None and runs[1][2].static_inputs is None
    if not per_env:   # one row, broadcast: nothing of size [E * J, ...] was computed or stored
        assert runs[0][2].static_inputs[0].stride(0) == 0 and runs[0][2].static_inputs[1].stride(0) == 0
    for k in runs[0][1].buffers:
        for other in (1, 2):
            assert torch.equal(runs[0][1].buffers[k], runs[other][1].buffers[k]), (k, other)
    assert float(runs[0][1].buffers["hidden_state"].abs().sum()) > 0


@pytest.mark.parametrize("N,H", [(1, 64), (63, 64), (12288, 64), (100, 128)])
def test_gru_gates_kernel_vs_torch_cell(N, H):
    """Gates + both stores in one launch == torch.nn.GRUCell on the host; strided gi / second destination."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(N + H)
    cell = torch.nn.GRUCell(H, H)
    with torch.no_grad():
        for p_ in cell.parameters():
            p_.copy_(0.3 * torch.randn(p_.shape, generator=g))
    x, h = torch.randn(N, H, generator=g), torch.randn(N, H, generator=g)
    with torch.no_grad():
        ref = cell(x, h)
        gi = torch.nn.functional.linear(x, cell.weight_ih, cell.bias_ih)
        gh = torch.nn.functional.linear(h, cell.weight_hh, cell.bias_hh)
    wide = torch.zeros(N, 3 * H + 8, device=DEV)
    wide[:, 4:4 + 3 * H] = gi.to(DEV)
    stage = torch.full((2, N, H), -7.0, device=DEV)
    out = ops.gru_gates(wide[:, 4:4 + 3 * H], gh.to(DEV), h.to(DEV), out2=stage[1])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=TOL, rtol=0)
    assert torch.equal(stage[1], out) and float((stage[0] != -7.0).sum()) == 0
    np.testing.assert_allclose(ops.gru_gates(gi, gh, h).numpy(), ref.numpy(), atol=TOL, rtol=0)   # host form


@pytest.mark.parametrize("scan", ["units", "ksplit"])
@pytest.mark.parametrize("tag", ["3j4r_h64", "2j2r_h128", "6j8r_h64"])
def test_gru_scan_with_in_kernel_input_transform_and_actor(tag, scan, monkeypatch):
    """Static observation: the scan launch computes each sequence's input transform (fc1 -> ReLU -> W_ih) and actor
    chain itself from the observation row it finds through an index into a [N, T+1, J, S] ring == torch layers on the
    gathered rows followed by the ordinary scan; the Double-DQN launch then reads one actor row per sequence."""
    from macjd_amd import ops
    from macjd_amd.core.networks import RNNAgent
    g, d = load(tag)
    args = _gpu_args(d)
    H, A, J, S = d["H"], d["A"], d["J"], d["S"]
    if scan == "ksplit":
        if H != 64:
            pytest.skip("H = 128 has the K-split kernel only")
        monkeypatch.setenv("MACJD_GRU_SCAN", "ksplit")
        from macjd_amd import _native
        _native.reload_options()
    torch.manual_seed(2)
    with quiet():
        agents = [RNNAgent(S, args).to(DEV), RNNAgent(S, args).to(DEV)]
    agents[0].load_state_dict(sd_from(g, "agent."))
    N, T1, B = 40, 13, 9
    ring = torch.randn(N, T1, J, S, device=DEV)
    idx = torch.tensor([3, 39, 0, 17, 17, 5, 22, 8, 31], dtype=torch.int64, device=DEV)
    with torch.no_grad():
        hs, ps = ops.gru_sequence_from_obs(ring, idx, agents, B, J, T1, with_actor=True)
        rows0 = ring[idx, 0].reshape(B * J, S)
        for a, h, p in zip(agents, hs, ps):
            gi = a.rnn.weight_ih.new_zeros(0)
            gi = torch.nn.functional.linear(torch.relu(a.fc1(rows0)), a.rnn.weight_ih, a.rnn.bias_ih).view(B, 1, J, 3 * H)
            ref = ops.gru_sequence_reference(gi.expand(B, T1, J, 3 * H).cpu(), a.rnn.weight_hh.cpu(), a.rnn.bias_hh.cpu())
            np.testing.assert_allclose(h.cpu().numpy(), ref.numpy(), atol=TOL, rtol=0)
            np.testing.assert_allclose(p.cpu().numpy(), a.actor(rows0).view(B, J, A).cpu().numpy(), atol=1e-6, rtol=1e-5)
        if H == 64 and ops.qhead_double_q_fused_supported(hs[0], H, A):
            heads = [(a.fc2_q_head[0].weight, a.fc2_q_head[0].bias, a.fc2_q_head[2].weight, a.fc2_q_head[2].bias) for a in agents]
            n = B * T1 * J
            ex = lambda p_: p_.view(B, 1, J, A).expand(B, T1, J, A).reshape(n, A)
            a_ = ops.qhead_double_q_from_h(hs[1].reshape(n, H), ps[1], heads[1], hs[0].reshape(n, H), ps[0], heads[0], H, A,
                                           p_row_map=(T1 * J, J))
            b_ = ops.qhead_double_q_from_h(hs[1].reshape(n, H), ex(ps[1]), heads[1], hs[0].reshape(n, H), ex(ps[0]), heads[0], H, A)
            assert torch.equal(a_, b_)


def test_gru_sequence_kernel_strided_initial_state():
    """h0 handed over as step 0 of a stored [B, T+1, J, H] tensor (batch stride (T+1) J H): no copy, same result."""
    from macjd_amd import ops
    B, T, J, H = 6, 9, 3, 64
    g = torch.Generator().manual_seed(3)
    gi = torch.randn(B, T, J, 3 * H, generator=g).to(DEV)
    w_hh = (0.2 * torch.randn(3 * H, H, generator=g)).to(DEV)
    b_hh = (0.1 * torch.randn(3 * H, generator=g)).to(DEV)
    stored = torch.randn(B, T + 1, J, H, generator=g).to(DEV)
    h0_view = stored[:, 0]
    assert not h0_view.is_contiguous()
    out = ops.gru_sequence(gi, w_hh, b_hh, h0_view)
    ref = ops.gru_sequence_reference(gi, w_hh, b_hh, h0_view.contiguous())
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=TOL, rtol=0)


def test_qhead_input_rows_kernel():
    """[h, onehot(a), P] rows in one launch == the reference's one_hot / cat sequence (networks.py:160-174),
    int32 and int64 indices, strided h, out-of-range index -> empty one-hot block."""
    from macjd_amd import ops
    n, H, A = 1003, 64, 9
    g = torch.Generator().manual_seed(4)
    wide = torch.randn(n, H + 7, generator=g).to(DEV)
    h = wide[:, 3:3 + H]
    P = torch.rand(n, 1, generator=g).to(DEV)
    for dt in (torch.int64, torch.int32):
        idx = torch.randint(0, A, (n, 1), generator=g).to(dt).to(DEV)
        got = ops.qhead_input(h, idx, P, A)
        want = torch.cat([h, torch.nn.functional.one_hot(idx.view(-1).long(), A).float(), P], dim=1)
        assert torch.equal(got, want)
    idx = torch.full((n,), A + 2, dtype=torch.int64, device=DEV)
    assert float(ops.qhead_input(h, idx, P, A)[:, H:H + A].abs().sum()) == 0.0


@pytest.mark.parametrize("M,S", [(3232, 46), (7, 5), (513, 130), (64, 1024)])
def test_layernorm_forward_kernel_and_backward(M, S):
    """One-launch LayerNorm forward (saving mean / rstd) + torch's native backward == F.layer_norm under autograd."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(M + S)
    x = (3.0 * torch.randn(M, S, generator=g) + 1.5).to(DEV).requires_grad_(True)
    w = (1.0 + 0.3 * torch.randn(S, generator=g)).to(DEV).requires_grad_(True)
    b = (0.2 * torch.randn(S, generator=g)).to(DEV).requires_grad_(True)
    up = torch.randn(M, S, generator=g).to(DEV)
    y = ops.layer_norm(x, w, b, 1e-5)
    gx, gw, gb = torch.autograd.grad((y * up).sum(), [x, w, b])
    y_ref = torch.nn.functional.layer_norm(x, (S,), w, b, 1e-5)
    rx, rw, rb = torch.autograd.grad((y_ref * up).sum(), [x, w, b])
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(gx.cpu().numpy(), rx.cpu().numpy(), atol=2e-5, rtol=1e-5)
    scale = max(1.0, float(rw.abs().max()))
    np.testing.assert_allclose(gw.cpu().numpy(), rw.cpu().numpy(), atol=1e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), rb.cpu().numpy(), atol=1e-5 * scale, rtol=1e-5)


def test_mixer_merged_first_layer_equals_concatenated_form():
    """QMixer on the flat-parameter views (no torch.cat, merged ReLU, strided b1 block, fused LayerNorm) ==
    the same module evaluated through the plain concatenation path: Q_tot and every parameter gradient."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    g, d = load("3j4r_h64")
    args = _gpu_args(d)
    with quiet():
        mac = BasicMAC(d["S"], args)
        learner = QMixLearner(mac, args)
    mx = learner.eval_qmix_net
    assert mx._merged_views is not None and learner.target_qmix_net._cat_cache is not None
    gen = torch.Generator().manual_seed(8)
    M = 777
    q = torch.randn(M, 1, d["J"], generator=gen).to(DEV)
    st = torch.randn(M, 1, mx.state_dim, generator=gen).to(DEV)
    up = torch.randn(M, 1, 1, generator=gen).to(DEV)
    params = list(mx.parameters())

    def run():
        for p in params:
            p.grad = None
        y = mx(q, st)
        (y * up).sum().backward()
        return y.detach().clone(), [p.grad.detach().clone() for p in params]

    y_m, g_m = run()
    views, mx._merged_views = mx._merged_views, None      # plain path: torch.cat of the four layers
    try:
        y_c, g_c = run()
    finally:
        mx._merged_views = views
    np.testing.assert_allclose(y_m.cpu().numpy(), y_c.cpu().numpy(), atol=TOL, rtol=0)
    for a, b in zip(g_m, g_c):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5 * max(1.0, float(b.abs().max())), rtol=1e-5)
    # the target copy answers from its cache, and the cache follows load_state_dict (the target sync)
    tm = learner.target_qmix_net
    with torch.no_grad():
        for p in mx.parameters():
            p.add_(0.01)
        tm.load_state_dict(mx.state_dict())
        np.testing.assert_allclose(tm(q, st).cpu().numpy(), mx(q, st).cpu().numpy(), atol=TOL, rtol=0)


@pytest.mark.parametrize("J,Em,M", [(3, 64, 3168), (6, 64, 100), (2, 32, 7), (3, 96, 33)])
def test_mixer_tail_kernel_forward_backward(J, Em, M):
    """Fused mixer tail (clamp / bmm / ELU / bmm) and its backward vs the stock-torch form, with inputs
    that land on both sides of every clamp bound and of the ELU knee."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(J * 1000 + Em)
    mk = lambda *s, scale=4.0: (torch.randn(*s, generator=g) * scale)
    q, w1, b1 = mk(M, J, scale=1.5), mk(M, J * Em), mk(M, Em)
    wf, v, gy = mk(M, Em), mk(M, 1), mk(M, 1, scale=1.0)
    ref_in = [t.clone().double().requires_grad_(True) for t in (q, w1, b1, wf, v)]
    y_ref = ops.mixer_tail_reference(*ref_in)
    y_ref.backward(gy.double())
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (q, w1, b1, wf, v)]
    y = ops.mixer_tail(*dev_in)
    y.backward(gy.to(DEV))
    scale = float(y_ref.abs().max())
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), atol=1e-5 * max(1.0, scale), rtol=1e-5)
    for a, b, name in zip(dev_in, ref_in, ("q", "w1", "b1", "wf", "v")):
        gs = float(b.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=1e-5 * max(1.0, gs), rtol=1e-4, err_msg=name)
    with torch.no_grad():  # target mixer path: forward only
        y2 = ops.mixer_tail(*[t.detach() for t in dev_in])
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("dims,acts,N", [
    ((46, 128, 128, 9), (1, 1, 2), 12288),     # actor, 3j/4r (weights resident in LDS)
    ((46, 64, 192), (1, 0), 9600),             # fc1 + GRU input transform, H=64
    ((24, 128, 192), (1, 0), 130),             # 12 output tiles
    ((184, 128, 128, 33), (1, 1, 2), 1000),    # actor, 12j/16r (staged per layer)
    ((92, 128, 128, 17), (1, 1, 2), 77),       # actor, 6j/8r
    ((64, 64), (0,), 63),                      # single layer, ragged row count
])
def test_fused_mlp_kernel(dims, acts, N):
    """Exact-f32 MFMA dense chain vs torch (float64 reference): <= 1e-5 abs on O(1) activations."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(sum(dims) + N)
    x = torch.randn(N, dims[0], generator=g) * 2.0
    layers = []
    for l in range(len(dims) - 1):
        w = torch.randn(dims[l + 1], dims[l], generator=g) / np.sqrt(dims[l])
        b = torch.randn(dims[l + 1], generator=g) * 0.3
        layers.append((w, b, acts[l]))
    assert ops.mlp_supported(list(dims))
    ref = ops.mlp_reference(x.double(), [(w.double(), b.double(), a) for w, b, a in layers])
    with torch.no_grad():
        y = ops.mlp_forward(x.to(DEV), [(w.to(DEV), b.to(DEV), a) for w, b, a in layers])
    assert y.shape == (N, dims[-1])
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=1e-5 * max(1.0, float(ref.abs().max())), rtol=1e-5)
    # strided input rows (a column slice of a wider tensor)
    xw = torch.randn(N, dims[0] + 5, generator=g).to(DEV)
    with torch.no_grad():
        y2 = ops.mlp_forward(xw[:, 2:2 + dims[0]], [(w.to(DEV), b.to(DEV), a) for w, b, a in layers])
    ref2 = ops.mlp_reference(xw[:, 2:2 + dims[0]].cpu().double(), [(w.double(), b.double(), a) for w, b, a in layers])
    np.testing.assert_allclose(y2.cpu().numpy(), ref2.numpy(), atol=1e-5 * max(1.0, float(ref2.abs().max())), rtol=1e-5)
    assert not ops.mlp_supported([300, 64]) and not ops.mlp_supported([46, 256, 9])
    assert not ops.mlp_supported([24, 128, 384])   # W_ih at H=128 (196 KB packed) exceeds LDS: library GEMMs


@pytest.mark.parametrize("scenario,E,H,mixer_dtype", [("6j8r", 4096, 64, "fp32"), ("12j16r", 2048, 64, "bf16"),
                                                      ("2j2r_shipped", 512, 128, "fp32")])
def test_other_baseline_configs_end_to_end(scenario, E, H, mixer_dtype):
    """BASELINE.json configs 3 and 5 at their per-GPU sizes (6j/8r E=4096; 12j/16r E=2048 with the bf16
    hyper-network option) and the shipped 2j/2r scenario with the reference's default H=128: graph-replayed
    rollout -> device replay -> graph-replayed updates; env outputs checked against the oracle on the
    actions the agents actually chose."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from _harness import OracleEnv
    sc, _ = load_scenario(scenario)
    env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=DEV, seed=11)
    info = env.get_env_info()
    d = dict(J=info["n_agents"], A=info["n_actions"], S=info["state_shape"], H=H)
    args = _gpu_args(d, episode_limit=info["episode_limit"], buffer_size=E, batch_size=32, lr=5e-6,
                     target_update_interval=200, mixer_dtype=mixer_dtype, epsilon_start=0.5)
    args.env_info = info
    torch.manual_seed(0)
    with quiet():
        mac = BasicMAC(info["obs_shape"], args)
        buf = EpisodeReplayBuffer(args)
        learner = QMixLearner(mac, args)
    assert learner.eval_qmix_net.bf16_hyper == (mixer_dtype == "bf16")
    runner = BatchedEpisodeRunner(env, mac, buf, args)
    runner.enable_graph()
    ri = runner.run()
    assert np.isfinite(ri["episode_return"]) and buf.current_size == E
    # oracle check of the stored rewards at three time steps, on the stored (chosen) actions
    b = buf.buffers
    ora = OracleEnv(sc, E, n_threads=8)
    ora.reset()   # first episode of every env, like the runner's (enable_graph leaves the episode indices untouched)
    assert env.episode_index.cpu().tolist() == ora.episode.tolist()
    for t in (0, 57, 99):
        ora.step_count[:] = t
        o = ora.step(b["actions_discrete"][:, t, :, 0].cpu().numpy(), b["actions_continuous"][:, t, :, 0].cpu().numpy(),
                     seed=11)
        np.testing.assert_allclose(b["reward"][:, t, 0].cpu().numpy(), o["reward"], atol=1e-5, rtol=0)
        np.testing.assert_array_equal(b["terminated"][:, t, 0].cpu().numpy(), o["terminated"].astype(bool))
    learner.enable_graphs(buf, 32)
    np.random.seed(0)
    losses = [learner.train_from_buffer()["loss"] for _ in range(4)]
    assert all(np.isfinite(l) for l in losses)
    # The graph-replayed update against the eager learner.train() on the SAME episodes from the SAME state: the four
    # statistics and the whole gradient vector at the G5 tolerances.  (The eager path itself is pinned on the reference's
    # gradients at every one of these sizes: test_learner_on_gpu_matches_reference incl. 12j16r_h64, and on the NumPy
    # oracle through tests/test_nets_cpu.py.)
    idx = np.random.default_rng(1).choice(E, 32, replace=False)
    snap = ([p_.detach().clone() for p_ in learner.params], learner._flat_exp_avg.clone(), learner._flat_exp_avg_sq.clone(),
            learner._adam_step.clone(), learner.train_step, learner.last_target_update_step)
    st_e = learner.train(buf.sample(32, indices=idx), {})
    g_e = learner._flat_grad.clone()
    with torch.no_grad():
        for p_, q_ in zip(learner.params, snap[0]):
            p_.copy_(q_)
        learner._flat_exp_avg.copy_(snap[1]); learner._flat_exp_avg_sq.copy_(snap[2]); learner._adam_step.copy_(snap[3])
    learner.train_step, learner.last_target_update_step = snap[4], snap[5]
    learner._mark_body_shared()      # (the restore wrote the unchanged body values back)
    st_g = learner.train_from_buffer(indices=idx)
    g_g = learner._flat_grad.clone()
    rt = 2e-2 if mixer_dtype == "bf16" else 1e-4     # (bf16 hyper-networks: 8 significant bits per product)
    for k in st_e:
        assert st_g[k] == pytest.approx(st_e[k], rel=rt, abs=TOL), k
    scale = float(g_e.abs().max())
    np.testing.assert_allclose(g_g.cpu().numpy(), g_e.cpu().numpy(), rtol=10 * rt, atol=(2e-5 if mixer_dtype != "bf16" else 2e-2) * max(scale, 1e-12))
    # ... and the grouped, pipelined form at this size (paired launches where the one-launch kernels apply; the bitwise
    # comparison with single updates is test_updates_grouped_into_one_graph_equal_single_updates at 3j/4r): re-captured in
    # groups of three, the allocator's cache emptied before the first replay (a baked address of a dead tensor faults)
    import gc
    learner.enable_graphs(buf, 32, updates_per_graph=3)
    assert learner._g_multi is not None and learner._g_multi[0] == 3
    gc.collect()
    torch.cuda.empty_cache()
    hist = torch.zeros(7, 4, device=DEV)
    learner.train_from_buffer_many(7, stats_out=hist)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(hist).all()) and float(hist[:, 3].min()) > 0.0


@pytest.mark.parametrize("B,T", [(32, 100), (4, 12), (3, 2)])
def test_td_loss_kernel(B, T):
    """Fused TD target + masked MSE (+ gradient, logged means) vs the stock-torch form, on the [:, :-1] slices
    of [B,T,1] tensors exactly as the learner passes them (no copies), with ragged `filled` / `terminated`."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(B * T)
    y = torch.randn(B, T - 1, 1, generator=g)
    tq = torch.randn(B, T - 1, 1, generator=g)
    reward = torch.randn(B, T, 1, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    steps = torch.arange(T).view(1, T, 1)
    filled = steps < lens.view(B, 1, 1)
    terminated = steps >= (lens.view(B, 1, 1) - 1)
    yr = y.clone().double().requires_grad_(True)
    loss_r, ev_r, tg_r = ops.td_loss_reference(yr, tq.double(), reward.double()[:, :-1], terminated[:, :-1], filled[:, :-1], 0.99)
    loss_r.backward()
    yd = y.clone().to(DEV).requires_grad_(True)
    loss, ev, tg = ops.td_loss(yd, tq.to(DEV), reward.to(DEV)[:, :-1], terminated.to(DEV)[:, :-1], filled.to(DEV)[:, :-1], 0.99)
    (loss * 1.0).backward()
    assert loss.item() == pytest.approx(loss_r.item(), rel=1e-5)
    assert ev.item() == pytest.approx(ev_r.item(), rel=1e-5, abs=1e-6)
    assert tg.item() == pytest.approx(tg_r.item(), rel=1e-5, abs=1e-6)
    np.testing.assert_allclose(yd.grad.cpu().numpy(), yr.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_logged_loss_sums_without_a_launch_of_their_own():
    """Updates whose loss gradient is formed inside the mixer's backward launch still log (loss, mean Q_tot, mean target):
    macjd_td_loss with gy = NULL writes stats[0..2] only, and macjd_mixer_fused_backward_td with td->stats computes the same
    three values in one extra workgroup of the backward launch — equal to the full loss launch's values to float rounding
    (the block sums are grouped differently), row[3] untouched, the backward's gradients bitwise unchanged."""
    from macjd_amd import ops
    from macjd_amd.core.networks import QMixer
    g_, d = load("3j4r_h64")
    B, T, J = 32, 101, d["J"]
    args = _gpu_args(d)
    torch.manual_seed(2)
    mixer = QMixer(args).to(DEV)
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, T, J, generator=g).to(DEV).requires_grad_(True)
    state = torch.randn(B, T, args.state_shape, generator=g).to(DEV)
    tq = torch.randn(B, T, 1, generator=g).to(DEV)
    reward = torch.randn(B, T, 1, generator=g).to(DEV)
    lens = torch.randint(2, T, (B,), generator=g)
    steps = torch.arange(T).view(1, T, 1)
    filled, terminated = (steps < lens.view(B, 1, 1)).to(DEV), (steps >= (lens.view(B, 1, 1) - 1)).to(DEV)
    params = [q] + list(mixer.parameters())

    def run(row):
        for p_ in params:
            p_.grad = None
        y = mixer(q, state)
        tot_m = ops.td_mask_sum(filled, T - 1)
        assert ops.fused_mixer_backward_will_run(y)
        gy = ops.td_grad_in_mixer_backward(y, tq, reward, terminated, filled, 0.99, T - 1, 1, tot_m, stats_row=row)
        y.backward(gy)
        return y.detach(), [p_.grad.clone() for p_ in params]

    y0, g0 = run(None)
    row = torch.full((4,), -7.0, device=DEV)
    y1, g1 = run(row)
    for a_, b_ in zip(g0, g1):
        assert torch.equal(a_, b_)
    full = ops.td_loss_and_grad(y0, tq, reward, terminated, filled, 0.99, T - 1, 1)[4]
    np.testing.assert_allclose(row[:3].cpu().numpy(), full[:3].cpu().numpy(), rtol=2e-6)
    assert float(row[3]) == -7.0
    alone = torch.full((4,), -7.0, device=DEV)
    ops.td_loss_sums_into(alone, y0, tq, reward, terminated, filled, 0.99, T - 1, 1)
    np.testing.assert_allclose(alone[:3].cpu().numpy(), full[:3].cpu().numpy(), rtol=2e-6)
    assert float(alone[3]) == -7.0


@pytest.mark.parametrize("max_norm", [1.0, 1e6])
def test_fused_clip_adam_vs_torch(max_norm):
    """Fused clip_grad_norm_ + Adam on a flat vector == torch.nn.utils.clip_grad_norm_ + torch.optim.Adam over
    the same parameters split into tensors, for several steps (bias corrections, clipping active / inactive)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 74), (64,), (1, 64), (1,), (192, 128), (46,)]
    n = sum(int(np.prod(s)) for s in shapes)
    flat0 = torch.randn(n, generator=g)
    ref_params = []
    off = 0
    for s in shapes:
        k = int(np.prod(s))
        ref_params.append(flat0[off:off + k].clone().view(s).double().requires_grad_(True))
        off += k
    opt = torch.optim.Adam(ref_params, lr=5e-4)
    p = flat0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step, gn, part = torch.zeros((), device=DEV), torch.zeros((), device=DEV), torch.zeros(256, device=DEV)
    for it in range(5):
        grad = torch.randn(n, generator=g) * (3.0 if it % 2 else 0.01)
        off = 0
        for rp in ref_params:
            rp.grad = grad[off:off + rp.numel()].view_as(rp).double().clone()
            off += rp.numel()
        total = torch.nn.utils.clip_grad_norm_(ref_params, max_norm)
        opt.step()
        gd = grad.to(DEV)
        ops.clip_adam_step(p, gd, m, v, step, gn, part, 5e-4, (0.9, 0.999), 1e-8, max_norm)
        assert float(gn) == pytest.approx(float(total), rel=1e-5)
        np.testing.assert_allclose(gd.cpu().numpy(), torch.cat([rp.grad.reshape(-1) for rp in ref_params]).numpy(),
                                   rtol=1e-5, atol=1e-9)   # gradients are left clipped in place
        assert float(step) == it + 1
        ref_flat = torch.cat([rp.detach().reshape(-1) for rp in ref_params])
        np.testing.assert_allclose(p.cpu().numpy(), ref_flat.numpy(), atol=2e-6, rtol=0)


def test_fused_gather_rows():
    from macjd_amd import ops
    g = torch.Generator().manual_seed(1)
    srcs = [torch.randn(50, 101, 46, generator=g), torch.randint(0, 9, (50, 100, 3, 1), generator=g, dtype=torch.int32),
            torch.rand(50, 100, 1, generator=g) > 0.5, torch.randn(50, 101, 3, 64, generator=g),
            torch.randn(50, 3, generator=g).double()]
    srcs = [s.to(DEV) for s in srcs]
    idx = torch.tensor([7, 0, 49, 7, 21], device=DEV)
    assert ops.gather_rows_supported(srcs)
    dsts = [torch.empty((5,) + tuple(s.shape[1:]), dtype=s.dtype, device=DEV) for s in srcs]
    ops.gather_rows(idx, srcs, dsts)
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s.index_select(0, idx))
    assert not ops.gather_rows_supported([torch.zeros(4, 5, dtype=torch.bool, device=DEV)])   # 5-byte rows


@pytest.mark.parametrize("K,M,N", [(3232, 384, 46), (9696, 64, 74), (3232, 1, 64), (1030, 192, 128), (3232, 64, 128),
                                   (9696, 1, 64), (1031, 1, 46), (777, 1, 100)])   # (one-row outputs: the row-vector work items)
def test_split_k_weight_gradient_kernel(K, M, N):
    """Split-K MFMA weight / bias gradient vs a float64 reference, incl. row-strided operands and the autograd
    Function it sits in (forward / input gradient stay library GEMMs)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(K + M + N)
    gout_w = torch.randn(K, M + 3, generator=g).to(DEV)
    inp_w = torch.randn(K, N + 5, generator=g).to(DEV)
    gout, inp = gout_w[:, 1:1 + M], inp_w[:, 2:2 + N]          # strided views (unit inner stride)
    dW, db = ops.linear_wgrad(gout, inp)
    ref_W = gout.double().t() @ inp.double()
    ref_b = gout.double().sum(0)
    sc = float(ref_W.abs().max())
    np.testing.assert_allclose(dW.cpu().numpy(), ref_W.cpu().numpy(), atol=2e-6 * sc * np.sqrt(K / 1000), rtol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), ref_b.cpu().numpy(), atol=2e-6 * float(ref_b.abs().max()) + 1e-5, rtol=1e-5)
    # through autograd: same gradients as torch.nn.functional.linear
    x = torch.randn(K, N, generator=g).to(DEV).requires_grad_(True)
    W = (torch.randn(M, N, generator=g) / np.sqrt(N)).to(DEV).requires_grad_(True)
    b = torch.randn(M, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(K, M, generator=g).to(DEV)
    y = ops.linear(x, W, b)
    y.backward(gy)
    x2, W2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, W, b))
    torch.nn.functional.linear(x2, W2, b2).backward(gy)
    for a, r in ((x.grad, x2.grad), (W.grad, W2.grad), (b.grad, b2.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), atol=1e-4 * float(r.abs().max()), rtol=1e-4)


@pytest.mark.parametrize("N,K", [(3232, 64), (9696, 64), (1024, 128), (5000, 4)])
def test_rowdot_kernel_single_output_linear(N, K):
    """Linear layers with one output feature on the row-dot kernel: forward and all gradients == F.linear, with a
    strided input block (the mixer's V head reads a column block of the merged ReLU output)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(N + K)
    wide = torch.randn(N, K + 12, generator=g).to(DEV)
    x = wide[:, 4:4 + K].detach().requires_grad_(True)
    W = (torch.randn(1, K, generator=g) / np.sqrt(K)).to(DEV).requires_grad_(True)
    b = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(N, 1, generator=g).to(DEV)
    y = ops.linear(x, W, b)
    assert y.shape == (N, 1)
    y.backward(gy)
    x2, W2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, W, b))
    y2 = torch.nn.functional.linear(x2, W2, b2)
    y2.backward(gy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y2.detach().cpu().numpy(), atol=2e-6 * np.sqrt(K), rtol=1e-5)
    for a, r in ((x.grad, x2.grad), (W.grad, W2.grad), (b.grad, b2.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), atol=1e-4 * max(1.0, float(r.abs().max())), rtol=1e-4)
    with torch.no_grad():
        assert torch.equal(ops.linear(x, W, b), y.detach())


def test_split_relu_fused_backward_equals_autograd():
    """ops.split_relu (one backward launch) == relu / split / slice under stock autograd, incl. an unused block."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(31)
    M, widths, Cp = 3232, [128, 128, 64], 64
    x = torch.randn(M, sum(widths) + Cp, generator=g).to(DEV).requires_grad_(True)
    ups = [torch.randn(M, w, generator=g).to(DEV) for w in widths + [Cp]]
    outs = ops.split_relu(x, widths, Cp)
    assert [o.shape[1] for o in outs] == widths + [Cp]
    (outs[0] * ups[0]).sum().add((outs[2] * ups[2]).sum()).add((outs[3] * ups[3]).sum()).backward()   # block 1 unused
    x2 = x.detach().clone().requires_grad_(True)
    Cr = sum(widths)
    r = torch.relu(x2[:, :Cr]).split(widths, dim=1)
    ((r[0] * ups[0]).sum() + (r[2] * ups[2]).sum() + (x2[:, Cr:] * ups[3]).sum()).backward()
    assert torch.equal(x.grad, x2.grad)
    for a, b in zip(outs[:3], r):
        assert torch.equal(a, b)


@pytest.mark.parametrize("deferred", [False, True])
def test_split_relu_with_folded_one_output_layer_equals_autograd(deferred):
    """ops.split_relu(..., dot=(k, w, b)): block k goes through a one-output Linear inside the node (row-dot forward,
    outer product folded into the one backward launch).  Values and every gradient == the unfused ops (bitwise)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(33)
    M, widths, Cp = 3232, [128, 128, 32], 32
    x = torch.randn(M, sum(widths) + Cp, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(1, 32, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    ups = [torch.randn(M, c, generator=g).to(DEV) for c in (128, 128, 1, Cp)]

    def run(fused):
        for t in (x, w, b):
            t.grad = None
        if fused:
            outs = ops.split_relu(x, widths, Cp, dot=(2, w, b))
        else:
            outs = list(ops.split_relu(x, widths, Cp))
            outs[2] = ops.linear(outs[2], w, b)
        loss = sum((o * u).sum() for o, u in zip(outs, ups))
        if deferred and fused:
            with ops.deferred_wgrad():
                loss.backward()
        else:
            loss.backward()
        return [o.detach().clone() for o in outs], [t.grad.clone() for t in (x, w, b)]

    o_ref, g_ref = run(False)
    o_got, g_got = run(True)
    assert o_got[2].shape == (M, 1)
    for a, c in zip(o_ref + g_ref, o_got + g_got):
        assert torch.equal(a, c)
    # against stock torch ops: same values up to the GEMM-vs-row-dot summation order of the one-output layer
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    r = list(torch.relu(x2[:, :sum(widths)]).split(widths, dim=1)) + [x2[:, sum(widths):]]
    r[2] = torch.nn.functional.linear(r[2], w2, b2)
    sum((o * u).sum() for o, u in zip(r, ups)).backward()
    assert torch.equal(g_got[0], x2.grad)
    torch.testing.assert_close(g_got[1], w2.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(o_got[2], r[2].detach(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("deferred", [False, True])
def test_linear_relu_dot_node_equals_unfused_layers(deferred):
    """ops.linear_relu_dot (the Q-head's two layers as one autograd node) == ops.linear(ops.linear_relu(...)) bitwise:
    output, input gradient and all four parameter gradients."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(34)
    N, K, Hd = 9696, 74, 64
    x = torch.randn(N, K, generator=g).to(DEV).requires_grad_(True)
    w1 = (torch.randn(Hd, K, generator=g) * 0.2).to(DEV).requires_grad_(True)
    b1 = torch.randn(Hd, generator=g).to(DEV).requires_grad_(True)
    w2 = torch.randn(1, Hd, generator=g).to(DEV).requires_grad_(True)
    b2 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    up = torch.randn(N, 1, generator=g).to(DEV)
    ps = (x, w1, b1, w2, b2)

    def run(fused):
        for t in ps:
            t.grad = None
        q = ops.linear_relu_dot(x, w1, b1, w2, b2) if fused else ops.linear(ops.linear_relu(x, w1, b1), w2, b2)
        if deferred and fused:
            with ops.deferred_wgrad():
                (q * up).sum().backward()
        else:
            (q * up).sum().backward()
        return [q.detach().clone()] + [t.grad.clone() for t in ps]

    ref, got = run(False), run(True)
    assert type(ops.linear_relu_dot(x, w1, b1, w2, b2).grad_fn).__name__.startswith("_LinearReluRowDot")
    for a, c in zip(ref, got):
        assert torch.equal(a, c)


def test_deferred_grouped_weight_gradients_equal_immediate():
    """Inside ops.deferred_wgrad() the split-K weight gradients of several layers run as ONE partial-products launch +
    ONE reduce launch at the end: bitwise the same results as the per-layer launches, for more problems than one
    batch holds, with strided operands and with / without bias."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(21)
    shapes = [(3232, 384, 46), (3232, 192, 128), (3232, 64, 128), (3232, 1, 64), (9696, 64, 74), (9696, 1, 64),
              (1030, 7, 9), (1500, 33, 65), (2000, 64, 64), (1234, 5, 5)]
    data = []
    for K, M, N in shapes:
        go = torch.randn(K, M + 4, generator=g).to(DEV)[:, 2:2 + M]
        x = torch.randn(K, N, generator=g).to(DEV)
        data.append((go, x))
    ref = [ops.linear_wgrad(go, x, want_bias=(i % 3 != 2)) for i, (go, x) in enumerate(data)]
    with ops.deferred_wgrad():
        got = [ops.linear_wgrad(go, x, want_bias=(i % 3 != 2)) for i, (go, x) in enumerate(data)]
    for (w1, b1), (w2, b2) in zip(ref, got):
        assert torch.equal(w1, w2)
        assert (b1 is None and b2 is None) or torch.equal(b1, b2)
    assert ops._DEFERRED_WGRAD is None


def test_deferred_weight_gradients_reach_dot_grad_exactly():
    """Through real autograd: parameters' .grad after a backward inside ops.deferred_wgrad() are bitwise those of an
    ordinary backward, also when the buffers held other values before (a replayed graph reuses them), and no copy of
    an unfilled buffer is taken (AccumulateGrad must adopt the tensors the flush fills)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(5)
    K = 3232
    x = torch.randn(K, 46, generator=g).to(DEV)
    W1 = (0.2 * torch.randn(128, 46, generator=g)).to(DEV).requires_grad_(True)
    b1 = torch.zeros(128, device=DEV, requires_grad=True)
    W2 = (0.2 * torch.randn(1, 128, generator=g)).to(DEV).requires_grad_(True)
    b2 = torch.zeros(1, device=DEV, requires_grad=True)
    params = [W1, b1, W2, b2]

    def run(deferred, scale):
        for p_ in params:
            p_.grad = None
        y = ops.linear(ops.linear_relu(x * scale, W1, b1), W2, b2)
        up = torch.ones_like(y)
        if deferred:
            with ops.deferred_wgrad():
                y.backward(up)
        else:
            y.backward(up)
        return [p_.grad.clone() for p_ in params]

    for scale in (1.0, -0.7, 2.5):      # different gradients each time: stale buffers would show
        ref = run(False, scale)
        got = run(True, scale)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


def test_weight_gradients_written_into_registered_destinations():
    """ops.deferred_wgrad(grad_dst=...): the gradients of the registered parameters land in the given tensors (slices
    of one flat vector) and .grad aliases them — bitwise the ordinary gradients; a parameter used TWICE in the graph
    still gets the sum (its second gradient forces the recorded ones out and is launched on the spot, autograd adds the
    two), and a gradient autograd drops (frozen layer) is kept alive until the deferred launch has written it."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(6)
    K = 3232
    x = torch.randn(K, 46, generator=g).to(DEV)
    W1 = (0.2 * torch.randn(64, 46, generator=g)).to(DEV).requires_grad_(True)
    b1 = torch.randn(64, generator=g).to(DEV).requires_grad_(True)
    W2 = (0.2 * torch.randn(1, 64, generator=g)).to(DEV).requires_grad_(True)
    b2 = torch.zeros(1, device=DEV, requires_grad=True)
    Ws = (0.2 * torch.randn(64, 64, generator=g)).to(DEV).requires_grad_(True)    # shared: used twice
    W3, b3 = torch.eye(64, device=DEV), torch.zeros(64, device=DEV)               # frozen layer: no destination
    params = [W1, b1, W2, b2, Ws]
    flat = torch.full((sum(p_.numel() for p_ in params),), 7.0, device=DEV)
    dst, off = {}, 0
    for p_ in params:
        dst[ops.grad_key(p_)] = flat[off:off + p_.numel()].view_as(p_)
        off += p_.numel()

    def run(use_dst, scale):
        for p_ in params:
            p_.grad = None
        h = ops.linear_relu(x * scale, W1, b1)
        h = ops.linear(ops.linear(h, Ws, None), Ws, None)
        y = ops.linear_relu_dot(h, W3, b3, W2, b2)
        if use_dst is None:
            y.backward(torch.ones_like(y))      # ordinary backward: every weight gradient launched on the spot
        else:
            with ops.deferred_wgrad(grad_dst=dst if use_dst else None):
                y.backward(torch.ones_like(y))
        return [p_.grad for p_ in params]

    for scale in (1.0, -0.6):
        ref = [t.clone() for t in run(None, scale)]
        for a, b in zip(ref, run(False, scale)):   # deferred, no destinations
            assert torch.equal(a, b)
        got = run(True, scale)
        off = 0
        for p_, a, b in zip(params, ref, got):
            assert torch.equal(a, b)
            if p_ is not Ws:    # (the sum for a shared parameter is a tensor of autograd's own)
                assert b.data_ptr() == flat.data_ptr() + 4 * off, "gradient was not written in place"
                assert torch.equal(flat[off:off + p_.numel()], a.reshape(-1))
            off += p_.numel()


def test_mlp_forward_pair_equals_two_single_launches():
    """Two independent dense chains in one launch == the two single launches, bitwise (actor + fc1/W_ih chains over
    the same rows; different row counts; one chain with weights staged per layer)."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(77)

    def chain(dims, acts):
        return [((torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i])).to(DEV),
                 (0.1 * torch.randn(dims[i + 1], generator=g)).to(DEV), acts[i]) for i in range(len(acts))]

    x = torch.randn(12288, 46, generator=g).to(DEV)
    x2 = torch.randn(1000, 184, generator=g).to(DEV)
    cases = [((x, chain((46, 128, 128, 9), (1, 1, 2))), (x, chain((46, 64, 192), (1, 0)))),
             ((x[:777], chain((46, 128, 128, 9), (1, 1, 2))), (x2, chain((184, 128, 128, 33), (1, 1, 2))))]
    for (xa, la), (xb, lb) in cases:
        ya, yb = ops.mlp_forward_pair(xa, la, xb, lb)
        assert torch.equal(ya, ops.mlp_forward(xa, la)) and torch.equal(yb, ops.mlp_forward(xb, lb))


@pytest.mark.parametrize("deferred", [False, True])
def test_norm_merged_linear_gradients_equal_autograd(deferred):
    """LayerNorm -> Linear as one autograd node whose backward derives the LayerNorm gamma / beta gradients from the
    Linear's weight-gradient products (dbeta = gb W, dgamma = sum_c W . (gout^T xhat)): same output and parameter
    gradients as F.layer_norm + F.linear under stock autograd, immediately and inside ops.deferred_wgrad()."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(13)
    M, K, sizes = 3232, 46, [128, 128, 64, 64]
    C = sum(sizes)
    x = (2.0 * torch.randn(M, K, generator=g) + 0.5).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(K, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(K, generator=g)).to(DEV).requires_grad_(True)
    w_cat = (torch.randn(C, K, generator=g) / np.sqrt(K)).to(DEV)
    b_cat = (0.1 * torch.randn(C, generator=g)).to(DEV)
    ws = [w.clone().requires_grad_(True) for w in w_cat.split(sizes, 0)]
    bs = [b.clone().requires_grad_(True) for b in b_cat.split(sizes, 0)]
    up = torch.randn(M, C, generator=g).to(DEV)
    out = ops.norm_merged_linear(x, gamma, beta, 1e-5, w_cat, b_cat, ws + bs)
    if deferred:
        with ops.deferred_wgrad():
            out.backward(up)
    else:
        out.backward(up)
    got = [gamma.grad, beta.grad] + [w.grad for w in ws] + [b.grad for b in bs]
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    w2, bb2 = w_cat.clone().requires_grad_(True), b_cat.clone().requires_grad_(True)
    ref_out = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (K,), g2, b2, 1e-5), w2, bb2)
    ref_out.backward(up)
    ref = [g2.grad, b2.grad] + list(w2.grad.split(sizes, 0)) + list(bb2.grad.split(sizes, 0))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.detach().cpu().numpy(), atol=2e-5, rtol=1e-5)
    for a, r in zip(got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), atol=2e-5 * max(1.0, float(r.abs().max())), rtol=1e-4)


from tests_golden_helpers import sample_episodes_mirror as _sample_episodes_mirror  # noqa: E402


@pytest.mark.parametrize("n,N", [(32, 8192), (32, 4096), (32, 33), (32, 32), (5, 7), (1, 1), (64, 100000)])
def test_device_episode_sampler(n, N):
    """The device-side draw of a batch's episodes: equals its host restatement word for word, indices distinct and in
    range, the counter advances by one per draw, and over many draws every stored episode is picked about equally often
    (uniform without replacement, like np.random.choice(N, n, replace=False) of the reference's buffer.sample)."""
    from macjd_amd import ops
    idx = torch.full((n,), -1, dtype=torch.int64, device=DEV)
    n_stored = torch.tensor([N], dtype=torch.int32, device=DEV)
    counter = torch.tensor([5], dtype=torch.int64, device=DEV)
    seed = 0x1234ABCD5678
    draws = []
    for d in range(3):
        ops.sample_episodes(idx, n_stored, counter, seed)
        got = idx.cpu().tolist()
        assert got == _sample_episodes_mirror(n, N, 5 + d, seed), (d, got)
        assert len(set(got)) == n and min(got) >= 0 and max(got) < N
        draws.append(got)
    assert int(counter.item()) == 8 and draws[0] != draws[1] or N == n == 1
    if N in (33, 4096):
        reps = 2000 if N == 33 else 3000
        counts = np.zeros(N)
        first = np.zeros(N)
        for _ in range(reps):
            ops.sample_episodes(idx, n_stored, counter, seed)
            g = idx.cpu().numpy()
            counts[g] += 1
            first[g[0]] += 1
        exp = reps * n / N                      # inclusion probability n / N per draw
        sd = np.sqrt(reps * (n / N) * (1 - n / N))
        assert np.abs(counts - exp).max() < 6 * sd + 1, (counts.min(), counts.max(), exp)
        assert first.max() < reps / N + 6 * np.sqrt(reps / N) + 1   # a given position of the batch is uniform too


def test_graphed_updates_on_device_drawn_batches(monkeypatch):
    """train_from_buffer() without indices: the batch was drawn by the previous replayed update's last launch (or by the
    stand-alone draw for the first update and after the population changed).  The indices found in the learner's index
    tensor before each update, handed to an eager learner, give the same statistics — so the draw is in place before the
    update that consumes it starts, and a store in between redraws from the new population."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    g, d = load("3j4r_h64")
    T, N, B = 100, 48, 32
    def build():
        args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=3)
        with quiet():
            mac = BasicMAC(d["S"], args)
            mac.load_state(sd_from(g, "g5_agent0."))
            learner = QMixLearner(mac, args)
            buf = EpisodeReplayBuffer(args)
        learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
        learner._update_targets()
        full = synthetic_batch(np.random.default_rng(9), args, N, T)
        for k, v in buf.buffers.items():
            v.copy_(torch.as_tensor(full[k]).to(v.dtype))
        buf.current_size, buf.current_index = 40, 40      # 40 of the 48 slots count as stored
        buf.episode_lengths[:] = T
        return learner, buf
    eager, buf_e = build()
    graphed, buf_g = build()
    graphed.enable_graphs(buf_g, B)
    assert graphed._g_dev_sampler
    seen = []
    for step in range(6):
        if step == 3:                                      # "a rollout stored 8 more episodes"
            for b_ in (buf_e, buf_g):
                b_.current_size, b_.current_index, b_.store_count = 48, 0, b_.store_count + 8
        # the draw this update consumes: number `step`, except that the draw made at the end of update 2 (from the old
        # population) is discarded for a fresh one when the population has changed
        draw_no = step if step < 3 else step + 1
        want = _sample_episodes_mirror(B, buf_g.current_size, draw_no, graphed._sampler_seed())
        if step in (0, 3):   # stand-alone draw inside train_from_buffer (first update / population changed)
            assert not graphed._g_idx_fresh or graphed._g_pop_seen != (buf_g.store_count, buf_g.current_size)
        else:                # drawn by the previous update's last launch: already in the index tensor
            assert graphed._g_idx_fresh and graphed._g_idx.cpu().tolist() == want
        sg = graphed.train_from_buffer()
        idx = np.array(want)
        seen.append(idx)
        assert len(set(want)) == B and idx.max() < buf_g.current_size
        se = eager.train(buf_e.sample(B, indices=idx), {})
        for k in se:
            assert sg[k] == pytest.approx(se[k], rel=1e-4, abs=1e-6), (step, k)
    assert int(graphed._g_draws.item()) == 8               # six consumed draws, one discarded, one waiting for the next update
    assert any(i.max() >= 40 for i in seen[3:])            # the enlarged population is being sampled
    # MACJD_DEVICE_SAMPLER=0 keeps the host draw + upload
    monkeypatch.setenv("MACJD_DEVICE_SAMPLER", "0")
    host, buf_h = build()
    host.enable_graphs(buf_h, B)
    assert not host._g_dev_sampler and np.isfinite(host.train_from_buffer()["loss"])


@pytest.mark.parametrize("paired", ["1", "0"])
def test_updates_grouped_into_one_graph_equal_single_updates(paired, monkeypatch):
    """train_from_buffer_many(n): groups of K updates replayed as ONE graph (each update's batch drawn on the device by
    the update before it) where no target sync falls inside the group, single replays otherwise == n calls of
    train_from_buffer(): same draws, same statistics per update, same weights, same target syncs.  The pipelined updates
    inside a group take their target branch as paired launches on the chain's stream (default) or on the side stream
    beside the eval head (MACJD_PAIRED_HEADS=0)."""
    monkeypatch.setenv("MACJD_PAIRED_HEADS", paired)
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    g, d = load("3j4r_h64")
    T, N, B, K, n = 100, 48, 32, 2, 7
    def build(k):
        args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=5, lr=1e-3)
        with quiet():
            mac = BasicMAC(d["S"], args)
            mac.load_state(sd_from(g, "g5_agent0."))
            learner = QMixLearner(mac, args)
            buf = EpisodeReplayBuffer(args)
        learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
        learner._update_targets()
        full = synthetic_batch(np.random.default_rng(9), args, N, T)
        for kk, v in buf.buffers.items():
            v.copy_(torch.as_tensor(full[kk]).to(v.dtype))
        buf.current_size, buf.current_index = N, 0
        buf.episode_lengths[:] = T
        learner.enable_graphs(buf, B, updates_per_graph=k)
        # every address baked into the captured launches must belong to a tensor that is still alive: cached blocks of
        # freed tensors go back to the driver here, as they do at the start of any later capture (round 3: the group's
        # second staging set was such a tensor — a GPU memory fault at 12j/16r once the rollout was captured afterwards)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return mac, learner
    mac_1, one = build(1)
    mac_k, many = build(K)
    assert one._g_multi is None and many._g_multi[0] == K
    ref = torch.zeros(n, 4, device=DEV)
    for i in range(n):
        one.train_from_buffer(sync_stats=False, stats_row=ref[i])
    got = torch.stack([r.clone() for r in many.train_from_buffer_many(n)]) if False else None
    # (the rows of a replayed group are static tensors that the next replay of the group overwrites: snapshot per call)
    many2_rows = []
    mac_k2, many2 = mac_k, many
    done = 0
    for chunk in (2, 2, 1, 2):     # the grouping train_from_buffer_many(7) itself chooses, spelled out to snapshot the rows
        many2_rows += [r.clone() for r in many2.train_from_buffer_many(chunk)]
        done += chunk
    got = torch.stack(many2_rows)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert many.train_step == one.train_step == n and many.last_target_update_step == one.last_target_update_step == 5
    assert int(many._g_draws.item()) == int(one._g_draws.item()) == n + 1
    for (k_, a), b in zip(mac_1.agent.state_dict().items(), mac_k.agent.state_dict().values()):
        assert torch.equal(a, b), k_
    for (k_, a), b in zip(one.target_qmix_net.state_dict().items(), many.target_qmix_net.state_dict().values()):
        assert torch.equal(a, b), k_
    # one call does the same split on its own, and snapshots every update's row when asked to
    mac_j, joint = build(K)
    snap = torch.zeros(n, 4, device=DEV)
    assert joint.train_from_buffer_many(n, stats_out=snap) is snap
    np.testing.assert_allclose(snap.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert joint.train_step == n and joint.last_target_update_step == 5
    for (k_, a), b in zip(mac_1.agent.state_dict().items(), mac_j.agent.state_dict().values()):
        assert torch.equal(a, b), k_


def test_body_written_through_data_is_noticed_when_graphs_are_captured():
    """The shared-body bookkeeping watches autograd version counters, which a write through ``p.data`` does not move
    (polyak-style target updates, hand-written broadcasts): enable_graphs() compares the VALUES of the two controllers'
    frozen bodies, so such a write ends the sharing at the next capture — and equal values restore it."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    g, d = load("3j4r_h64")
    T, N, B = 100, 40, 32
    args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B)
    with quiet():
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "g5_agent0."))
        learner = QMixLearner(mac, args)
        buf = EpisodeReplayBuffer(args)
    full = synthetic_batch(np.random.default_rng(3), args, N, T)
    for kk, v in buf.buffers.items():
        v.copy_(torch.as_tensor(full[kk]).to(v.dtype))
    buf.current_size, buf.current_index = N, 0
    buf.episode_lengths[:] = T
    assert learner._body_is_shared()
    learner.target_mac.agent.fc1.weight.data.mul_(1.5)          # invisible to the version counter
    assert learner._body_is_shared()                             # ... so the cheap check still says "shared"
    learner.enable_graphs(buf, B)
    assert not learner._g_shared_body and not learner._body_is_shared()
    assert np.isfinite(learner.train_from_buffer()["loss"])
    learner._update_targets()                                    # hard sync: identical again
    learner.enable_graphs(buf, B)
    assert learner._g_shared_body


def test_learner_graphs_recaptured_and_released_in_one_process():
    """enable_graphs() is meant to be called again mid-run (after the buffer lost its static-observation flag, the agent
    body changed, another grouping is wanted): every call destroys the previous graphs first (QMixLearner.release_graphs:
    device idle, grouped graph and graph B before graph A), and the re-captured learner continues the SAME sequence of
    updates — weights bitwise equal to a learner captured once.  Other learners' multi-branch graphs, created and destroyed
    in between, must not disturb it: this is the history (graph streams created and destroyed, their hardware-queue
    references returned) under which round 2's replay faulted inside hipGraphLaunch (macjd_amd/hipgraph.py)."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    g, d = load("3j4r_h64")
    T, N, B = 100, 48, 32

    def build(k):
        args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=5, lr=1e-3)
        with quiet():
            mac = BasicMAC(d["S"], args)
            mac.load_state(sd_from(g, "g5_agent0."))
            learner = QMixLearner(mac, args)
            buf = EpisodeReplayBuffer(args)
        learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
        learner._update_targets()
        full = synthetic_batch(np.random.default_rng(9), args, N, T)
        for kk, v in buf.buffers.items():
            v.copy_(torch.as_tensor(full[kk]).to(v.dtype))
        buf.current_size, buf.current_index = N, 0
        buf.episode_lengths[:] = T
        learner.enable_graphs(buf, B, updates_per_graph=k)
        return mac, learner, buf

    mac_r, ref, _ = build(2)
    mac_c, cut, buf_c = build(2)
    ref.train_from_buffer_many(9)
    cut.train_from_buffer_many(3)
    # other learners come and go in between (their graphs are destroyed explicitly or by the collector)
    for k in (1, 4):
        _, other, _ = build(k)
        other.train_from_buffer_many(4)
        other.release_graphs()
        with pytest.raises(RuntimeError, match="enable_graphs"):
            other.train_from_buffer()
        del other
    cut.enable_graphs(buf_c, B, updates_per_graph=4)      # re-capture, another grouping
    assert cut._g_multi[0] == 4
    cut.train_from_buffer_many(2)
    cut.enable_graphs(buf_c, B, updates_per_graph=1)      # ... and again, single updates
    assert cut._g_multi is None
    cut.train_from_buffer_many(4)
    assert cut.train_step == ref.train_step == 9 and int(cut._g_draws.item()) == int(ref._g_draws.item())
    for (k_, a), b in zip(mac_r.agent.state_dict().items(), mac_c.agent.state_dict().values()):
        assert torch.equal(a, b), k_
    for (k_, a), b in zip(ref.eval_qmix_net.state_dict().items(), cut.eval_qmix_net.state_dict().values()):
        assert torch.equal(a, b), k_
    # eager updates keep working after the graphs are gone
    cut.release_graphs()
    assert np.isfinite(cut.train(buf_c.sample(B), {})["loss"])


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("N,K,Hd", [(9696, 74, 64), (1500, 33, 20)])
def test_relu_backward_operand_formed_inside_the_weight_gradient_kernel(N, K, Hd, deferred, monkeypatch):
    """The Q-head node's backward when nobody wants the input gradient (the learner's case): the product
    gq w2 masked by the ReLU is formed by the weight-gradient kernel while it stages its operand
    (macjd_wgrad_io.outer_vec) == materialising it with the ReLU-backward launch first (MACJD_WGRAD_OUTER=0): all four
    parameter gradients bitwise, alone and inside a deferred group; ragged sizes too."""
    from macjd_amd import ops
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(N, K, generator=g).to(DEV)
    w1 = (torch.randn(Hd, K, generator=g) * 0.2).to(DEV).requires_grad_(True)
    b1 = torch.randn(Hd, generator=g).to(DEV).requires_grad_(True)
    w2 = torch.randn(1, Hd, generator=g).to(DEV).requires_grad_(True)
    b2 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    up = torch.randn(N, 1, generator=g).to(DEV)
    ps = (w1, b1, w2, b2)

    def run(outer):
        monkeypatch.setenv("MACJD_WGRAD_OUTER", "1" if outer else "0")
        for t in ps:
            t.grad = None
        q = ops.linear_relu_dot(x, w1, b1, w2, b2)
        assert type(q.grad_fn).__name__.startswith("_LinearReluRowDot")
        if deferred:
            with ops.deferred_wgrad():
                (q * up).sum().backward()
        else:
            (q * up).sum().backward()
        return [t.grad.clone() for t in ps]

    ref, got = run(False), run(True)
    for a, c in zip(ref, got):
        assert torch.equal(a, c)
    # and against stock autograd
    for t in ps:
        t.grad = None
    (torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, w1, b1)), w2, b2) * up).sum().backward()
    for t, c in zip(ps, got):
        np.testing.assert_allclose(c.cpu().numpy(), t.grad.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(t.grad.abs().max()))


@pytest.mark.parametrize("n,A,idt", [(9696, 9, torch.int32), (1037, 33, torch.int64), (2048, 5, torch.int64)])
def test_taken_action_qhead_in_one_launch(n, A, idt, monkeypatch):
    """macjd_qhead_taken (input rows + first layer on MFMA + ReLU + second layer's dot, one launch) == the three-launch
    form (macjd_qhead_input, library GEMM with bias / ReLU epilogue, macjd_rowdot): the input rows bitwise, Q-values and
    activations to 1e-5 (the first layer's summation order differs), all four parameter gradients to the weight-gradient
    tolerance; out-of-range action indices give an empty one-hot block; ragged row counts."""
    from macjd_amd import ops
    H = 64
    g = torch.Generator().manual_seed(n + A)
    h = torch.randn(n, H, generator=g).to(DEV)
    idx = torch.randint(-1, A + 1, (n,), generator=g).to(idt).to(DEV)     # incl. -1 and A: empty one-hot
    P = torch.rand(n, 1, generator=g).to(DEV)
    w1 = (torch.randn(H, H + A + 1, generator=g) * 0.2).to(DEV).requires_grad_(True)
    b1 = torch.randn(H, generator=g).to(DEV).requires_grad_(True)
    w2 = torch.randn(1, H, generator=g).to(DEV).requires_grad_(True)
    b2 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    up = torch.randn(n, 1, generator=g).to(DEV)
    ps = (w1, b1, w2, b2)
    assert ops.qhead_taken_supported(h, w1, w2, A)

    def run(fused):
        for t in ps:
            t.grad = None
        if fused:
            q = ops.qhead_taken(h, idx, P, w1, b1, w2, b2, A)
            assert type(q.grad_fn).__name__.startswith("_QheadTaken")
            x, act = q.grad_fn.saved_tensors[0], q.grad_fn.saved_tensors[2]
        else:
            x = ops.qhead_input(h, idx, P, A)
            q = ops.linear_relu_dot(x, w1, b1, w2, b2)
            act = q.grad_fn.saved_tensors[2]
        with ops.deferred_wgrad():
            (q * up).sum().backward()
        return q.detach().clone(), x.clone(), act.clone(), [t.grad.clone() for t in ps]

    q0, x0, a0, g0 = run(False)
    q1, x1, a1, g1 = run(True)
    assert torch.equal(x0, x1)
    scale = float(q0.abs().max())
    np.testing.assert_allclose(q1.cpu().numpy(), q0.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, scale))
    np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, float(a0.abs().max())))
    for a, c in zip(g0, g1):
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(a.abs().max()))
    # an activation within rounding of zero may sit on the other side of the ReLU in the two forms; everything else equal
    flipped = ((a0 > 0) != (a1 > 0)).float().mean().item()
    assert flipped < 1e-4
    # the module uses it exactly when the predicate says so
    monkeypatch.setenv("MACJD_QHEAD_TAKEN", "0")
    assert not ops.qhead_taken_supported(h, w1, w2, A)


@pytest.mark.parametrize("tag,n", [("3j4r_h64", 9696), ("6j8r_h64", 1206), ("12j16r_h64", 2412)])
def test_paired_launches_equal_the_single_launches(tag, n):
    """macjd_qheads_pair (taken-action Q-head + Double-DQN launch as one grid) and macjd_mixer_fused_forward_pair (saving +
    plain mixer forward as one grid) are the single launches' bodies behind a workgroup-index switch: every output —
    Q-values, the saved input rows / activations, both Q_tot, the gradients computed from the saved tensors — bitwise
    equal; a pair nobody picks up is reported."""
    from macjd_amd import ops
    from macjd_amd.core.networks import QMixer, RNNAgent
    g, d = load(tag)
    args = _gpu_args(d)
    H, A, J = d["H"], d["A"], d["J"]
    torch.manual_seed(3)
    with quiet():
        ae, at = RNNAgent(d["S"], args).to(DEV), RNNAgent(d["S"], args).to(DEV)
        me, mt = QMixer(args).to(DEV), QMixer(args).to(DEV)
    rng = np.random.default_rng(n)
    f = lambda *shape: torch.tensor(rng.standard_normal(shape), dtype=torch.float32, device=DEV)
    h_e, h_t, P_all = 0.7 * f(n, H), 0.7 * f(n, H), torch.rand(n, A, device=DEV)
    idx = torch.randint(0, A, (n, 1), device=DEV)
    P = torch.rand(n, 1, device=DEV)
    M = n // J
    state = 2.0 * f(M, args.state_shape)
    gy = f(M, 1)
    heads = [(a.fc2_q_head[0].weight, a.fc2_q_head[0].bias, a.fc2_q_head[2].weight, a.fc2_q_head[2].bias) for a in (ae, at)]
    params = list(ae.fc2_q_head.parameters()) + list(me.parameters())

    def run(paired):
        for p_ in params:
            p_.grad = None
        with torch.no_grad():
            launch = ops.pair_double_q_with_next_taken if paired else ops.qhead_double_q_from_h
            tq = launch(h_e, P_all, heads[0], h_t, P_all, heads[1], H, A)
        q = ae.get_q_value_for_action(h_e, idx, P, validate=False)
        assert type(q.grad_fn).__name__.startswith("_QheadTaken")
        with torch.no_grad():
            y_t = (mt.forward_paired_with_next_fused if paired else mt)(tq.view(M, J), state)
        y_e = me(q.view(M, J), state)
        ops.assert_pairs_launched()
        saved = [t.clone() for t in q.grad_fn.saved_tensors[:3:2]]
        y_e.backward(gy)
        return [tq.clone(), q.detach().clone(), y_t.clone(), y_e.detach().clone()] + saved + [p_.grad.clone() for p_ in params]

    single, pair = run(False), run(True)
    for i, (a, b) in enumerate(zip(single, pair)):
        assert torch.equal(a, b), i
    with torch.no_grad():
        ops.pair_double_q_with_next_taken(h_e, P_all, heads[0], h_t, P_all, heads[1], H, A)
    with pytest.raises(RuntimeError, match="paired launch not taken"):
        ops.assert_pairs_launched()
    ops.assert_pairs_launched()   # (cleared by the report)


def test_layernorm_param_grads_inside_the_squared_norm_launch(monkeypatch):
    """Single process: the LayerNorm-parameter launch behind the grouped weight gradients is held back and evaluated by
    the optimiser step's squared-norm launch (macjd_clip_adam_step_ln: K extra workgroups, their squares counted as
    extra partials, the generic workgroups skip the two ranges) == the separate launch (MACJD_LN_IN_SQNORM=0): same
    LayerNorm gradients bitwise, the gradient norm to float rounding (the partial sums are grouped differently), same
    weights after the steps to 1e-6."""
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    from tests_golden_helpers import synthetic_batch
    g, d = load("3j4r_h64")
    T, N, B = 100, 40, 32
    def build():
        args = _gpu_args(d, episode_limit=T, buffer_size=N, batch_size=B, target_update_interval=50, lr=1e-3)
        with quiet():
            mac = BasicMAC(d["S"], args)
            mac.load_state(sd_from(g, "g5_agent0."))
            learner = QMixLearner(mac, args)
            buf = EpisodeReplayBuffer(args)
        learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
        learner._update_targets()
        full = synthetic_batch(np.random.default_rng(9), args, N, T)
        for k, v in buf.buffers.items():
            v.copy_(torch.as_tensor(full[k]).to(v.dtype))
        buf.current_size, buf.current_index = N, 0
        buf.episode_lengths[:] = T
        return mac, learner, buf
    rng = np.random.default_rng(3)
    idxs = [rng.choice(N, B, replace=False) for _ in range(3)]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MACJD_LN_IN_SQNORM", mode)
        mac, learner, buf = build()
        stats, ln_grads = [], []
        for idx in idxs:
            stats.append(learner.train(buf.sample(B, indices=idx), {}))
            ln = learner.eval_qmix_net.state_norm
            ln_grads.append((ln.weight.grad.clone(), ln.bias.grad.clone(), float(stats[-1]["grad_norm"])))
        out[mode] = (stats, ln_grads, [p.detach().clone() for p in learner.params])
    for (s1, s0) in zip(out["1"][0], out["0"][0]):
        for k in s1:
            assert s1[k] == pytest.approx(s0[k], rel=1e-6, abs=1e-9), k
    # .grad holds the CLIPPED gradients after the step: compare through the clip coefficient of each run
    for (w1, b1, n1), (w0, b0, n0) in zip(out["1"][1], out["0"][1]):
        np.testing.assert_allclose(w1.cpu().numpy(), w0.cpu().numpy(), rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(b1.cpu().numpy(), b0.cpu().numpy(), rtol=2e-6, atol=1e-12)
        assert float(w1.abs().max()) > 0
    for a, c in zip(out["1"][2], out["0"][2]):
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=0, atol=1e-6)
