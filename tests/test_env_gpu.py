"""GPU parity tests of the HIP env-step kernel, called through the C-ABI (libmacjd_hip.so).

Bar (BASELINE.json north_star): integer outputs (FSM / terminated / masks) bit-exact; float rewards
within 1e-5 of the reference.  The float64 diagnostics are additionally held to 1e-9 relative, which
documents how close the device arithmetic actually is (device exp() may differ from the host's in the
last ulp)."""
import numpy as np
import pytest
import torch

from _harness import OracleEnv, load_scenario, random_actions

pytestmark = pytest.mark.gpu

SCENARIOS = ["2j2r_shipped", "3j4r", "6j8r", "12j16r", "3j3r_edge"]
TOL_REWARD = 1e-5  # north_star tolerance on float rewards


def _env(sc, E, **kw):
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    return BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device="cuda:0", **kw)


def _diag(E, R, J):
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda:0")
    return {"out64": z(E, 4), "pd64": z(E, R), "snr64": z(E, R), "prj64": z(E, J)}


def _cmp(o, rew, term, info, diag=None):
    np.testing.assert_array_equal(info["radar_tracking"].cpu().numpy(), o["track"])
    np.testing.assert_array_equal(term.cpu().numpy(), o["terminated"].astype(bool))
    np.testing.assert_array_equal(info["step_count"].cpu().numpy(), o["step"])
    np.testing.assert_allclose(rew.cpu().numpy(), o["reward"], rtol=0, atol=TOL_REWARD)
    for i, k in enumerate(("r_d", "r_p", "r_j")):
        np.testing.assert_allclose(info[k].cpu().numpy(), o["r_dpj"][:, i], rtol=0, atol=TOL_REWARD)
    np.testing.assert_allclose(info["radar_pds"].cpu().numpy(), o["pd"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(info["snr_with_jamming"].cpu().numpy(), o["snr_with"], rtol=1e-6, atol=0)
    if diag is not None:
        np.testing.assert_allclose(diag["out64"].cpu().numpy(), o["out64"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(diag["pd64"].cpu().numpy(), o["pd64"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(diag["snr64"].cpu().numpy(), o["snr64"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(diag["prj64"].cpu().numpy(), o["prj64"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", SCENARIOS)
@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_golden_trace_single_env(name, mode):
    """The reference's own traces (actions + logged MT19937 uniforms) through the HIP kernel, E=1."""
    sc, g = load_scenario(name)
    R, J = sc.num_radars, sc.num_jammers
    env = _env(sc, 1)
    diag = _diag(1, R, J)
    for seed in (42, 43, 44):
        pre = f"{mode}_s{seed}_"
        T, P, U = g[pre + "T"], g[pre + "P"], g[pre + "u"]
        Td = torch.from_numpy(T).cuda()
        Pd = torch.from_numpy(P.astype(np.float32) if mode == "f32" else P).cuda()
        Ud = torch.from_numpy(np.nan_to_num(U, nan=2.0)).cuda()
        got = {k: [] for k in ("out", "pd", "snr", "prj", "track", "term")}
        for t in range(T.shape[0]):
            if g[pre + "reset_before"][t]:
                env.reset()
            _, term, info = env.step(Td[t:t + 1], Pd[t:t + 1], Ud[t:t + 1], diag=diag)
            got["out"].append(diag["out64"].clone()); got["pd"].append(diag["pd64"].clone())
            got["snr"].append(diag["snr64"].clone()); got["prj"].append(diag["prj64"].clone())
            got["track"].append(info["radar_tracking"].clone()); got["term"].append(term.clone())
        cat = lambda k: torch.cat(got[k]).cpu().numpy()
        np.testing.assert_array_equal(cat("track"), g[pre + "track"])            # bit-exact FSM
        np.testing.assert_array_equal(cat("term"), g[pre + "terminated"])
        ref = np.stack([g[pre + "reward"], g[pre + "r_d"], g[pre + "r_p"], g[pre + "r_j"]], axis=1)
        np.testing.assert_allclose(cat("out"), ref, rtol=0, atol=TOL_REWARD)     # the stated bar
        np.testing.assert_allclose(cat("out"), ref, rtol=1e-9, atol=1e-12)       # what it actually achieves
        np.testing.assert_allclose(cat("pd"), g[pre + "pd"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(cat("snr"), g[pre + "snr_with"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(cat("prj"), g[pre + "prj"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", SCENARIOS)
@pytest.mark.parametrize("E", [1, 63, 64, 65, 257, 4096])
def test_batch_vs_oracle_supplied_uniforms(name, E):
    sc, _ = load_scenario(name)
    R, J = sc.num_radars, sc.num_jammers
    env, ora = _env(sc, E), OracleEnv(sc, E, n_threads=4)
    env.reset()
    rng = np.random.default_rng(E * 7 + R)
    diag = _diag(E, R, J)
    for t in range(4):
        T, P = random_actions(rng, E, J, R)
        u = rng.random((E, R + J))
        rew, term, info = env.step(torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda(),
                                   torch.from_numpy(u).cuda(), diag=diag)
        _cmp(ora.step(T, P, u=u), rew, term, info, diag)


@pytest.mark.parametrize("name", ["3j4r", "6j8r", "3j3r_edge"])
def test_philox_mode_and_layouts(name):
    """In-kernel Philox == oracle Philox; agent-major ([J,E]) action storage == env-major."""
    sc, _ = load_scenario(name)
    R, J, E = sc.num_radars, sc.num_jammers, 1000
    env = _env(sc, E, seed=1234, env_offset=5000)
    env2 = _env(sc, E, seed=1234, env_offset=5000)
    ora = OracleEnv(sc, E, n_threads=4)
    env.reset(); env2.reset(); ora.reset()
    rng = np.random.default_rng(3)
    diag = _diag(E, R, J)
    for t in range(6):
        T, P = random_actions(rng, E, J, R)
        o = ora.step(T, P, seed=1234, env_offset=5000)
        rew, term, info = env.step(torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda(), diag=diag)
        _cmp(o, rew, term, info, diag)
        # agent-major storage, passed as transposed views, int64 discrete actions with a trailing dim
        T_am = torch.from_numpy(np.ascontiguousarray(T.T)).cuda().to(torch.int64)
        P_am = torch.from_numpy(np.ascontiguousarray(P.T)).cuda()
        rew2, term2, info2 = env2.step(T_am.t().unsqueeze(-1), P_am.t().unsqueeze(-1))
        _cmp(o, rew2, term2, info2)


def test_arith_modes():
    """float64 power input, and float32 input with the ARITH_F64 flag, follow the oracle's f64 leg."""
    sc, _ = load_scenario("3j3r_edge")
    R, J, E = sc.num_radars, sc.num_jammers, 512
    rng = np.random.default_rng(9)
    T, P = random_actions(rng, E, J, R)
    u = rng.random((E, R + J))
    for variant in ("p64", "p32_flag"):
        env, ora = _env(sc, E), OracleEnv(sc, E)
        env.reset()
        diag = _diag(E, R, J)
        if variant == "p64":
            P_in = P.astype(np.float64) + 1e-9
            rew, term, info = env.step(torch.from_numpy(T).cuda(), torch.from_numpy(P_in).cuda(),
                                       torch.from_numpy(u).cuda(), diag=diag)
            o = ora.step(T, P_in, u=u)
        else:
            rew, term, info = env.step(torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda(),
                                       torch.from_numpy(u).cuda(), diag=diag, arith_f64=True)
            o = ora.step(T, P, u=u, arith_f64=True)
        _cmp(o, rew, term, info, diag)


def test_full_size_properties():
    """BASELINE.json sizes (E=4096 and 32768): size-independent properties + oracle spot check."""
    sc, _ = load_scenario("3j4r")
    R, J = sc.num_radars, sc.num_jammers
    for E in (4096, 32768):
        env = _env(sc, E, seed=42)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        ret = torch.zeros(E, device="cuda")
        for t in range(sc.episode_limit):
            T = torch.randint(0, 2 * R + 1, (J, E), generator=g, device="cuda", dtype=torch.int32).t()
            P = torch.rand((J, E), generator=g, device="cuda").t()
            rew, term, info = env.step(T, P)
            # reward decomposition and the closed form of r_p (environment.py:377): pmin=0 -> norm = P
            tot = info["r_d"] + info["r_p"] + info["r_j"]
            assert torch.allclose(rew, tot, atol=1e-6)
            rp = (sc.rp_max + (sc.rp_min - sc.rp_max) * P.double()).sum(1)
            assert torch.allclose(info["r_p"].double(), rp, atol=1e-6)
            # r_d is a sum of per-radar penalties over the tracking radars
            pen = torch.tensor(sc.tables["radar_rd_pen"], device="cuda")
            assert torch.allclose(info["r_d"].double(), (info["radar_tracking"].double() * pen).sum(1), atol=1e-6)
            assert bool(term.all()) == (t == sc.episode_limit - 1)
            assert bool(term.any()) == (t == sc.episode_limit - 1)
            ret += rew
        assert torch.isfinite(ret).all()
        # Pd floor: no suppression can push Pd below pd(0); tracking frequency ~ Pd
        assert float(info["radar_pds"].min()) >= 0.1029
        # idle jammers: r_j == 0 exactly, r_p == J * rp_max when P == 0
        env.reset()
        rew, term, info = env.step(torch.zeros((E, J), dtype=torch.int32, device="cuda"),
                                   torch.zeros((E, J), device="cuda"))
        assert float(info["r_j"].abs().max()) == 0.0
        assert torch.allclose(info["r_p"], torch.full((E,), J * sc.rp_max, device="cuda"), atol=1e-7)


def test_shard_invariance_and_determinism():
    """Sharding envs over ranks (env_offset) does not change any env's trajectory; same seed twice is
    bit-identical (multi-GPU row of SURVEY.md section 8e)."""
    sc, _ = load_scenario("3j4r")
    R, J, E = sc.num_radars, sc.num_jammers, 2048
    rng = np.random.default_rng(2)
    T, P = random_actions(rng, E, J, R)
    Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
    whole = _env(sc, E, seed=99)
    lo, hi = _env(sc, E // 2, seed=99, env_offset=0), _env(sc, E // 2, seed=99, env_offset=E // 2)
    again = _env(sc, E, seed=99)
    for e in (whole, lo, hi, again):
        e.reset()
    for t in range(5):
        r_w, _, i_w = whole.step(Td, Pd)
        r_a, _, i_a = again.step(Td, Pd)
        r_l, _, i_l = lo.step(Td[:E // 2], Pd[:E // 2])
        r_h, _, i_h = hi.step(Td[E // 2:], Pd[E // 2:])
        assert torch.equal(r_w, r_a) and torch.equal(i_w["radar_tracking"], i_a["radar_tracking"])
        assert torch.equal(r_w, torch.cat([r_l, r_h]))
        assert torch.equal(i_w["radar_tracking"], torch.cat([i_l["radar_tracking"], i_h["radar_tracking"]]))


def test_every_episode_draws_fresh_monte_carlo_values():
    """The reference consumes fresh np.random.rand() values in every episode (environment.py:341,430).  In Philox mode
    the per-env episode index (advanced by reset, also by a masked reset and by a reset replayed from a HIP graph) is
    part of the counter: the same env with the same actions sees different detection draws in consecutive episodes,
    every episode matches the oracle, and the values do not depend on how the envs are sharded."""
    sc, _ = load_scenario("3j4r")
    R, J, E = sc.num_radars, sc.num_jammers, 1536
    rng = np.random.default_rng(21)
    T, P = random_actions(rng, E, J, R, with_invalid=False)
    Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
    whole, ora = _env(sc, E, seed=31), OracleEnv(sc, E, n_threads=4)
    lo, hi = _env(sc, E // 2, seed=31), _env(sc, E // 2, seed=31, env_offset=E // 2)
    tracks = []
    for ep in range(3):
        for e_ in (whole, lo, hi):
            e_.reset()
        ora.reset()
        assert int(whole.episode_index.min()) == ep + 1 == int(whole.episode_index.max())
        ep_tracks = []
        for t in range(4):
            rew, term, info = whole.step(Td, Pd)
            o = ora.step(T, P, seed=31)
            _cmp(o, rew, term, info)
            r_l, _, i_l = lo.step(Td[:E // 2], Pd[:E // 2])
            r_h, _, i_h = hi.step(Td[E // 2:], Pd[E // 2:])
            assert torch.equal(rew, torch.cat([r_l, r_h]))
            assert torch.equal(info["radar_tracking"], torch.cat([i_l["radar_tracking"], i_h["radar_tracking"]]))
            ep_tracks.append(info["radar_tracking"].clone())
        tracks.append(torch.stack(ep_tracks))
    # identical actions, different episodes: the detection outcomes differ (pd ~ 0.1: ~18 % of the flags flip)
    for a, b in ((0, 1), (1, 2), (0, 2)):
        frac = (tracks[a] != tracks[b]).float().mean().item()
        assert 0.05 < frac < 0.4, frac
    # masked reset advances only the masked envs; a reset replayed from a HIP graph advances the counter as well
    mask = torch.zeros(E, dtype=torch.bool, device="cuda")
    mask[::2] = True
    whole.reset(mask)
    ora.reset(mask.cpu().numpy())
    assert whole.episode_index.cpu().tolist() == ora.episode.tolist()
    _cmp(ora.step(T, P, seed=31), *whole.step(Td, Pd))
    g = torch.cuda.CUDAGraph()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        whole.reset(); ora.reset()
    torch.cuda.current_stream().wait_stream(s_)
    with torch.cuda.graph(g):
        whole.reset()
        whole.step(Td, Pd)
    for _ in range(2):
        g.replay()
        ora.reset()
        o = ora.step(T, P, seed=31)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(whole.track.cpu().numpy(), o["track"])
        assert whole.episode_index.cpu().tolist() == ora.episode.tolist()


@pytest.mark.parametrize("pd32", [False, True])
@pytest.mark.parametrize("name,E,n", [("3j4r", 777, 23), ("6j8r", 300, 7), ("2j2r_shipped", 64, 100)])
def test_many_step_launch_equals_step_by_step(name, E, n, pd32, monkeypatch):
    """(pd32: the many-step launch's float32 detection-probability filter, the default — integer outputs and everything
    derived from them alone stay bit for bit, the reward's probability terms move by <= (R + J) 6e-7; off: all float64,
    every output bit for bit.)  macjd_env_step_many: the n steps of an episode batch as ONE launch over n x E independent work items (the FSM's
    next state does not depend on the previous one, core/radar.py:102-117, so given the actions of all steps the env-steps
    are independent) == n single-step launches, bit for bit: rewards, terminated, (r_d, r_p, r_j) per step and summed,
    final FSM state, step counters; also across an episode boundary of the counters and against the oracle."""
    from macjd_amd import _native
    monkeypatch.setenv("MACJD_ENV_PD32", "1" if pd32 else "0")
    _native.reload_options()
    sc, _ = load_scenario(name)
    R, J = sc.num_radars, sc.num_jammers
    rng = np.random.default_rng(E + n)
    Ts = np.stack([random_actions(rng, E, J, R, with_invalid=True)[0] for _ in range(n)])
    Ps = rng.random((n, E, J)).astype(np.float32)
    Td, Pd = torch.from_numpy(Ts).cuda(), torch.from_numpy(Ps).cuda()
    one, many, ora = _env(sc, E, seed=17, env_offset=40), _env(sc, E, seed=17, env_offset=40), OracleEnv(sc, E, n_threads=4)
    for rep in range(2):     # second repetition: another episode index
        one.reset(); many.reset(); ora.reset()
        rew1 = torch.zeros((n, E), device="cuda"); ter1 = torch.zeros((n, E), dtype=torch.uint8, device="cuda")
        rd1 = torch.zeros((n, E, 3), device="cuda"); sum1 = torch.zeros((E, 3), device="cuda")
        for t in range(n):
            _, _, info = one.step(Td[t], Pd[t], out_reward=rew1[t], out_terminated=ter1[t], want_info=False, rdpj_sum=sum1)
            rd1[t] = one._r_dpj
            if t in (0, n - 1):
                o = ora.step(Ts[t], Ps[t], seed=17, env_offset=40)
                np.testing.assert_allclose(rew1[t].cpu().numpy(), o["reward"], rtol=0, atol=TOL_REWARD)
                np.testing.assert_array_equal(one.track.cpu().numpy(), o["track"])
            else:
                ora.step_count += 1
        rewn = torch.zeros((n, E, 1), device="cuda"); tern = torch.zeros((n, E, 1), dtype=torch.bool, device="cuda")
        rdn = torch.zeros((n, E, 3), device="cuda"); sumn = torch.zeros((E, 3), device="cuda")
        many.step_many(Td.view(n, E, J, 1), Pd.view(n, E, J, 1), rewn, tern, rdn, rdpj_sum=sumn)
        assert torch.equal(tern.view(n, E).to(torch.uint8), ter1)
        if pd32:
            tol = (R + J) * 6e-7 + 1e-7
            assert torch.equal(rdn[..., :2], rd1[..., :2])                       # r_d (FSM bits), r_p: no probability value
            assert float((rdn[..., 2] - rd1[..., 2]).abs().max()) <= tol and float((rewn.view(n, E) - rew1).abs().max()) <= tol
            assert float((sumn - sum1).abs().max()) <= n * tol
        else:
            assert torch.equal(rewn.view(n, E), rew1) and torch.equal(rdn, rd1) and torch.equal(sumn, sum1)
        assert torch.equal(many.track, one.track) and torch.equal(many.step_count, one.step_count)
        assert int(many.step_count.min()) == n and torch.equal(many.episode_index, one.episode_index)
    with pytest.raises(ValueError):
        many.step_many(Td.to(torch.int64), Pd, rewn, tern, rdn)


def test_reset_mask_and_outputs_into_caller_buffers():
    sc, _ = load_scenario("3j4r")
    R, J, E = sc.num_radars, sc.num_jammers, 300
    env = _env(sc, E, seed=5)
    env.reset()
    T = torch.ones((E, J), dtype=torch.int32, device="cuda")
    P = torch.full((E, J), 0.5, device="cuda")
    rows = torch.zeros((4, E), device="cuda")
    terms = torch.zeros((4, E), dtype=torch.uint8, device="cuda")
    for t in range(4):
        rew, term, _ = env.step(T, P, out_reward=rows[t], out_terminated=terms[t])
        assert rew.data_ptr() == rows[t].data_ptr()
    assert (rows != 0).any() and int(env.step_count.min()) == 4
    mask = torch.zeros(E, dtype=torch.bool, device="cuda")
    mask[::3] = True
    env.reset(mask)
    sc_ = env.step_count.cpu().numpy()
    assert (sc_[::3] == 0).all() and (np.delete(sc_, np.arange(0, E, 3)) == 4).all()
    assert int(env.track[::3].sum()) == 0


def test_reward_component_sums_accumulate_in_kernel():
    """rdpj_sum += (r_d, r_p, r_j) inside the step launch (lane and slot kernels) == summing the per-step
    r_dpj output; the actions are read from a strided [E,J,1] view like the runner's staging rows."""
    sc, _ = load_scenario("3j4r")
    R, J, E = sc.num_radars, sc.num_jammers, 777
    rng = np.random.default_rng(9)
    from macjd_amd import _native
    for flag in (_native.STEP_LANE_KERNEL, _native.STEP_SLOT_KERNEL):
        env = _env(sc, E, seed=11)
        env.kernel_flags = flag
        env.reset()
        acc = torch.zeros((E, 3), device="cuda")
        ref = torch.zeros((E, 3), device="cuda")
        stage_T = torch.zeros((5, E, J, 1), dtype=torch.int32, device="cuda")
        stage_P = torch.zeros((5, E, J, 1), device="cuda")
        for t in range(5):
            T, P = random_actions(rng, E, J, R, with_invalid=False)
            stage_T[t].copy_(torch.from_numpy(T).view(E, J, 1))
            stage_P[t].copy_(torch.from_numpy(P).view(E, J, 1))
            env.step(stage_T[t], stage_P[t], rdpj_sum=acc)
            ref += env._r_dpj
        assert torch.equal(acc, ref) and float(acc.abs().sum()) > 0
    with pytest.raises(ValueError):
        env.step(stage_T[0], stage_P[0], rdpj_sum=torch.zeros((E, 2), device="cuda"))


RAND = ["3j4r_rand0", "3j4r_rand1", "3j4r_rand2"]


@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_per_env_scenarios_follow_the_reference_traces(mode):
    """Per-env scenario tables (SoA in HBM): env k of ONE batched environment runs fixture rand<k>'s scenario and
    replays that fixture's own action / uniform trace; every env reproduces the reference (FSM bit-exact)."""
    from macjd_amd.scenario import ScenarioBatch
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    loaded = [load_scenario(n) for n in RAND]
    batch = ScenarioBatch([sc for sc, _ in loaded])
    E, R, J = 3, 4, 3
    env = BatchedElectromagneticEnvironment(scenario_batch=batch, device="cuda:0")
    assert env.batch_envs == E
    np.testing.assert_array_equal(env.get_state().cpu().numpy(), np.stack([g["static_state"] for _, g in loaded]))
    np.testing.assert_array_equal(env.get_obs().cpu().numpy()[:, 1], np.stack([g["static_state"] for _, g in loaded]))
    pre = f"{mode}_s42_"
    n = loaded[0][1][pre + "T"].shape[0]
    diag = _diag(E, R, J)
    st = lambda key: np.stack([g[pre + key] for _, g in loaded], axis=1)     # [steps, E, ...]
    T, P, U = st("T"), st("P"), np.nan_to_num(st("u"), nan=2.0)
    Td = torch.from_numpy(T).cuda()
    Pd = torch.from_numpy(P.astype(np.float32) if mode == "f32" else P).cuda()
    Ud = torch.from_numpy(U).cuda()
    for t in range(n):
        if loaded[0][1][pre + "reset_before"][t]:
            env.reset()
        _, term, info = env.step(Td[t], Pd[t], Ud[t], diag=diag)
        np.testing.assert_array_equal(info["radar_tracking"].cpu().numpy(), st("track")[t], err_msg=f"step {t}")
        np.testing.assert_array_equal(term.cpu().numpy(), st("terminated")[t])
        ref = np.stack([st("reward")[t], st("r_d")[t], st("r_p")[t], st("r_j")[t]], axis=1)
        np.testing.assert_allclose(diag["out64"].cpu().numpy(), ref, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(diag["pd64"].cpu().numpy(), st("pd")[t], rtol=1e-12, atol=0)
        np.testing.assert_allclose(diag["prj64"].cpu().numpy(), st("prj")[t], rtol=1e-14, atol=0)
        np.testing.assert_allclose(info["snr_no_jamming"].cpu().numpy(), st("snr_no")[t], rtol=0, atol=0)


def test_per_env_batch_equals_single_scenario_envs_and_oracle():
    """E randomised scenarios in one launch == E single-scenario environments (bitwise, in-kernel Philox keyed by the
    global env index) == the oracle run per env on that env's own scenario description."""
    from macjd_amd.scenario import ScenarioBatch, ring_scenario_dict
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    E, R, J = 40, 4, 3
    batch = ScenarioBatch.randomized(ring_scenario_dict(3, 4), E, seed=3, env_offset=100)
    env = BatchedElectromagneticEnvironment(scenario_batch=batch, device="cuda:0", seed=77, env_offset=100)
    singles = [_env(sc, 1, seed=77, env_offset=100 + e) for e, sc in enumerate(batch.scenarios)]
    oracles = [OracleEnv(sc, 1) for sc in batch.scenarios]
    env.reset()
    for s_ in singles:
        s_.reset()
    for o_ in oracles:
        o_.reset()
    rng = np.random.default_rng(5)
    diag = _diag(E, R, J)
    for t in range(6):
        T, P = random_actions(rng, E, J, R)
        Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
        rew, term, info = env.step(Td, Pd, diag=diag)
        rews, tracks = [], []
        for e, s_ in enumerate(singles):
            r1, _, i1 = s_.step(Td[e:e + 1], Pd[e:e + 1])
            rews.append(r1.clone()); tracks.append(i1["radar_tracking"].clone())
            o = oracles[e].step(T[e:e + 1], P[e:e + 1], seed=77, env_offset=100 + e)
            np.testing.assert_array_equal(info["radar_tracking"][e].cpu().numpy(), o["track"][0])
            np.testing.assert_allclose(diag["out64"][e].cpu().numpy(), o["out64"][0], rtol=1e-9, atol=1e-12)
        assert torch.equal(rew, torch.cat(rews)) and torch.equal(info["radar_tracking"], torch.cat(tracks))
    assert float(rew.std()) > 0          # the scenarios really differ
    with pytest.raises(ValueError):
        BatchedElectromagneticEnvironment(scenario_batch=batch, batch_envs=E + 1, device="cuda:0")


@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("J,R", [(5, 7), (2, 2), (6, 8)])
def test_per_env_tables_other_sizes_vs_oracle(J, R, tiled, monkeypatch):
    """Per-env tables on the generic kernel variant (5j/7r: run-time sizes, table reads at their use sites) and on
    other compile-time sizes, against the oracle run per env on that env's own scenario; supplied uniforms."""
    from macjd_amd.scenario import ScenarioBatch, ring_scenario_dict
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    E = 33
    batch = ScenarioBatch.randomized(ring_scenario_dict(J, R), E, seed=J * 100 + R)
    # both device layouts of the tables: tiles of 8 envs (33 envs = 4 full tiles + a padded one) and the plain SoA
    monkeypatch.setattr(BatchedElectromagneticEnvironment, "pe_tiled", tiled)
    monkeypatch.setattr(BatchedElectromagneticEnvironment, "PE_TILE", 8)
    env = BatchedElectromagneticEnvironment(scenario_batch=batch, device="cuda:0")
    assert env._pe_tile == (8 if tiled else 0)
    oracles = [OracleEnv(sc, 1) for sc in batch.scenarios]
    env.reset()
    rng = np.random.default_rng(J + R)
    diag = _diag(E, R, J)
    for t in range(4):
        T, P = random_actions(rng, E, J, R)
        u = rng.random((E, R + J))
        rew, term, info = env.step(torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda(), torch.from_numpy(u).cuda(),
                                   diag=diag)
        for e in range(E):
            o = oracles[e].step(T[e:e + 1], P[e:e + 1], u=u[e:e + 1])
            np.testing.assert_array_equal(info["radar_tracking"][e].cpu().numpy(), o["track"][0])
            np.testing.assert_allclose(diag["out64"][e].cpu().numpy(), o["out64"][0], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(diag["pd64"][e].cpu().numpy(), o["pd64"][0], rtol=1e-12, atol=0)
            np.testing.assert_allclose(diag["prj64"][e].cpu().numpy(), o["prj64"][0], rtol=1e-14, atol=0)


@pytest.mark.parametrize("per_env", [False, True])
@pytest.mark.parametrize("J,R", [(3, 4), (6, 8), (12, 16), (2, 2)])
def test_fast_kernel_variant_equals_general_variant(J, R, per_env, monkeypatch):
    """The compile-time production variant of the lane kernel (Philox uniforms, float32 actions, no float64
    diagnostics; all-float64 probabilities: MACJD_ENV_PD32=0) == the general variant (selected here by asking for the
    diagnostics), bitwise, shared and per-env tables."""
    from macjd_amd import _native
    monkeypatch.setenv("MACJD_ENV_PD32", "0")
    _native.reload_options()
    from macjd_amd.scenario import Scenario, ScenarioBatch, ring_scenario_dict
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
    E = 300
    base = ring_scenario_dict(J, R)
    if per_env:
        batch = ScenarioBatch.randomized(base, 25, seed=9).tile(E)
        mk = lambda: BatchedElectromagneticEnvironment(scenario_batch=batch, device="cuda:0", seed=5, env_offset=64)
    else:
        sc = Scenario.from_dict(base)
        mk = lambda: _env(sc, E, seed=5, env_offset=64)
    fast, general = mk(), mk()
    fast.kernel_flags = general.kernel_flags = _native.STEP_LANE_KERNEL
    fast.reset(); general.reset()
    rng = np.random.default_rng(R)
    diag = _diag(E, R, J)
    for t in range(5):
        T, P = random_actions(rng, E, J, R)
        Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
        r1, t1, i1 = fast.step(Td, Pd)
        r2, t2, i2 = general.step(Td, Pd, diag=diag)
        assert torch.equal(r1, r2) and torch.equal(t1, t2)
        for k in ("r_d", "r_p", "r_j", "radar_tracking", "radar_pds", "snr_with_jamming"):
            assert torch.equal(i1[k], i2[k]), k


def test_bad_arguments_raise():
    sc, _ = load_scenario("3j4r")
    env = _env(sc, 8)
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 2), dtype=torch.int32, device="cuda"), torch.zeros((8, 2), device="cuda"))
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 3), dtype=torch.int32, device="cuda"), torch.zeros((8, 3), device="cuda"),
                 torch.zeros((8, 3), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("name", ["2j2r_shipped", "3j4r", "6j8r", "12j16r"])
@pytest.mark.parametrize("E", [1, 100, 4096])
def test_slot_kernel_equals_lane_kernel_bitwise(name, E):
    """The (env x slot) kernel and the one-lane-per-env kernel run the same arithmetic in the same order:
    float64 diagnostics are bit-identical, in supplied-uniform and in Philox mode."""
    from macjd_amd import _native
    sc, _ = load_scenario(name)
    R, J = sc.num_radars, sc.num_jammers
    rng = np.random.default_rng(E + J)
    a, b = _env(sc, E, seed=77, env_offset=123), _env(sc, E, seed=77, env_offset=123)
    a.kernel_flags, b.kernel_flags = _native.STEP_SLOT_KERNEL, _native.STEP_LANE_KERNEL
    a.reset(); b.reset()
    da, db = _diag(E, R, J), _diag(E, R, J)
    for t in range(4):
        T, P = random_actions(rng, E, J, R)
        u = torch.from_numpy(rng.random((E, R + J))).cuda() if t % 2 else None
        Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
        ra, ta, ia = a.step(Td, Pd, u, diag=da)
        rb, tb_, ib = b.step(Td, Pd, u, diag=db)
        for k in da:
            assert torch.equal(da[k], db[k]), (t, k)
        assert torch.equal(ra, rb) and torch.equal(ta, tb_)
        for k in ("r_d", "r_p", "r_j", "radar_tracking", "radar_pds", "snr_with_jamming", "step_count"):
            assert torch.equal(ia[k], ib[k]), (t, k)


@pytest.mark.parametrize("name", ["2j2r_shipped", "3j4r", "6j8r", "12j16r", "3j3r_edge"])
def test_regular_scenario_variant_equals_ieee_division_variant(name, monkeypatch):
    """Production lane kernel of a REGULAR scenario (divisions by table values through refined reciprocals from the
    tables, float32 quotients as twice-rounded float64 ones, guards a regular scenario cannot trigger dropped) ==
    the same kernel with IEEE divisions and every guard (MACJD_ENV_REGULAR=0) == the general variant, on every output
    the production configuration has, over 70 000 envs x 6 steps of random actions incl. invalid action types,
    out-of-range / NaN powers and, in the edge scenario, a jammer sitting on a radar; and vs the oracle."""
    from macjd_amd import _native
    from _harness import OracleEnv
    sc, _ = load_scenario(name)
    R, J = sc.num_radars, sc.num_jammers
    E = 70000 if name in ("3j4r", "3j3r_edge") else 20000
    reg32, reg, ieee = (_env(sc, E, seed=21, env_offset=5) for _ in range(3))
    assert reg.scenario_regular, "the shipped / benchmark / edge scenarios are regular"
    reg32.kernel_flags = reg.kernel_flags = ieee.kernel_flags = _native.STEP_LANE_KERNEL
    reg32.reset(); reg.reset(); ieee.reset()
    ora = OracleEnv(sc, 4096, n_threads=8)
    ora.reset()
    rng = np.random.default_rng(J * 100 + R)

    def options(**kv):     # the library reads its switches once: re-read after every change
        for k in ("MACJD_ENV_REGULAR", "MACJD_ENV_PD32"):
            monkeypatch.delenv(k, raising=False)
        for k, v in kv.items():
            monkeypatch.setenv(k, v)
        _native.reload_options()

    worst = 0.0
    for t in range(6):
        T, P = random_actions(rng, E, J, R)
        P = (P * 1.6 - 0.3).astype(np.float32)          # some below 0 and above 1 (np.clip)
        P[rng.random((E, J)) < 0.01] = np.nan           # NaN propagates through np.clip: the jammer is not recorded
        Td, Pd = torch.from_numpy(T).cuda(), torch.from_numpy(P).cuda()
        options()                                        # production: regular forms + float32 probability filter
        r0, t0, i0 = reg32.step(Td, Pd)
        options(MACJD_ENV_PD32="0")                      # regular forms, all-float64 probabilities
        r1, t1, i1 = reg.step(Td, Pd)
        options(MACJD_ENV_REGULAR="0")                   # IEEE divisions and every guard
        r2, t2, i2 = ieee.step(Td, Pd)
        options()
        same = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=-777.0), torch.nan_to_num(b.float(), nan=-777.0))
        assert torch.isnan(r1).any()                     # a NaN power makes r_p (and the reward) NaN, like the reference
        assert same(r1, r2) and torch.equal(t1, t2)
        for k in ("r_d", "r_p", "r_j", "radar_tracking", "radar_pds", "snr_with_jamming"):
            assert same(i1[k], i2[k]), (t, k)
        # the float32 filter: every integer output and everything that depends on them alone bit for bit; the reward's
        # probability terms within the stated bound
        assert torch.equal(t0, t1) and torch.equal(i0["radar_tracking"], i1["radar_tracking"]), t
        for k in ("r_d", "r_p"):
            assert same(i0[k], i1[k]), (t, k)
        # (the radars' SNR is a float32 quotient in the filter variant: 3.5e-7 relative)
        np.testing.assert_allclose(i0["snr_with_jamming"].cpu().numpy(), i1["snr_with_jamming"].cpu().numpy(), rtol=2e-6, atol=0)
        fin = torch.isfinite(r1)
        assert torch.equal(torch.isfinite(r0), fin)
        for a_, b_ in ((r0, r1), (i0["r_j"], i1["r_j"])):
            worst = max(worst, float((a_[fin].double() - b_[fin].double()).abs().max()))
        worst = max(worst, float((i0["radar_pds"].double() - i1["radar_pds"].double()).abs().max()))
        ora.step_count[:] = t
        o = ora.step(T[:4096], P[:4096], seed=21, env_offset=5)
        np.testing.assert_allclose(r1[:4096].cpu().numpy(), o["reward"], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(i1["radar_tracking"][:4096].cpu().numpy(), o["track"])
        np.testing.assert_allclose(r0[:4096].cpu().numpy(), o["reward"], rtol=0, atol=1e-5)
    assert worst <= (R + J) * 6e-7 + 1e-7, worst          # (float32 outputs: + half an ulp of the reward)
