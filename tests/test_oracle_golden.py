"""Pins the CPU oracle (oracle/macjd_oracle.c) and the scenario compiler on golden traces produced by
the REFERENCE itself (tests/golden/make_golden.py, generated in the build container).

Covers SURVEY.md section 8(c) fixtures G1 (env traces incl. edge-case actions and logged uniforms) and
G2 (static state / obs / avail-action mask / env_info)."""
import json

import numpy as np
import pytest

from _harness import OracleEnv, load_scenario, oracle_lib

SCENARIOS = ["2j2r_shipped", "3j4r", "6j8r", "12j16r", "3j3r_edge"]
SEEDS = [42, 43, 44]


@pytest.mark.parametrize("name", SCENARIOS)
def test_static_tables_and_masks(name):
    """G2: state vector, obs, integer avail-action mask (bit-exact) and env_info."""
    sc, g = load_scenario(name)
    state = sc.state_vector()
    assert state.dtype == np.float32
    np.testing.assert_array_equal(state, g["static_state"])
    np.testing.assert_array_equal(np.stack([state] * sc.num_jammers), g["static_obs"])
    avail = np.ones((sc.num_jammers, sc.n_actions), dtype=np.int32)
    assert g["static_avail_actions"].dtype == np.int32
    np.testing.assert_array_equal(avail, g["static_avail_actions"])
    assert sc.env_info() == json.loads(str(g["env_info_json"]))
    # static no-jamming SNR is what the reference reports every step
    np.testing.assert_array_equal(sc.tables["radar_snr_no"], g["f64_s42_snr_no"][0])


@pytest.mark.parametrize("name", SCENARIOS)
@pytest.mark.parametrize("mode", ["f32", "f64"])
@pytest.mark.parametrize("seed", SEEDS)
def test_env_trace(name, mode, seed):
    """G1: replay the reference's actions + logged uniforms through the oracle, one env, 200 steps."""
    sc, g = load_scenario(name)
    pre = f"{mode}_s{seed}_"
    T, P, U = g[pre + "T"], g[pre + "P"], g[pre + "u"]
    env = OracleEnv(sc, 1)
    n = T.shape[0]
    for t in range(n):
        if g[pre + "reset_before"][t]:
            env.reset()
        p = P[t].astype(np.float32) if mode == "f32" else P[t]
        u = np.nan_to_num(U[t], nan=2.0)  # slots the reference never drew must never be read as hits
        o = env.step(T[t][None], p[None], u=u[None])
        assert int(o["draws"][0]) == int(g[pre + "n_draws"][t]), f"step {t}: RNG draw count"
        np.testing.assert_array_equal(o["track"][0], g[pre + "track"][t], err_msg=f"step {t} FSM")
        assert bool(o["terminated"][0]) == bool(g[pre + "terminated"][t])
        ref = np.array([g[pre + "reward"][t], g[pre + "r_d"][t], g[pre + "r_p"][t], g[pre + "r_j"][t]])
        np.testing.assert_allclose(o["out64"][0], ref, rtol=1e-13, atol=1e-15, err_msg=f"step {t} rewards")
        np.testing.assert_allclose(o["pd64"][0], g[pre + "pd"][t], rtol=1e-13, atol=0)
        np.testing.assert_allclose(o["snr64"][0], g[pre + "snr_with"][t], rtol=1e-13, atol=0)
        np.testing.assert_allclose(o["prj64"][0], g[pre + "prj"][t], rtol=1e-14, atol=0)
        # float32 outputs are the rounded float64 ones
        np.testing.assert_array_equal(o["reward"][0], np.float32(o["out64"][0, 0]))


RAND = ["3j4r_rand0", "3j4r_rand1", "3j4r_rand2"]   # per-env randomised variations, light traces


@pytest.mark.parametrize("name", RAND)
@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_env_trace_randomised_scenarios(name, mode):
    """The oracle on the reference's traces of the randomised 3j/4r variations (100 steps, seed 42)."""
    test_env_trace(name, mode, 42)
    test_static_tables_and_masks(name)


def test_scenario_batch_is_the_fixture_scenarios():
    """ScenarioBatch.randomized(ring 3j/4r, seed 7) env k == the scenario the reference ran for fixture rand<k>:
    same dict, same compiled tables (SoA column k), same observation vector; shard-invariant by env_offset."""
    from macjd_amd.scenario import ScenarioBatch, ring_scenario_dict
    batch = ScenarioBatch.randomized(ring_scenario_dict(3, 4), 3, seed=7)
    R, J = 4, 3
    assert batch.tables.shape == (6 * R + 3 * J + J * R, 3) and batch.flags.shape == (J * R, 3)
    for k, name in enumerate(RAND):
        sc, g = load_scenario(name)
        col = np.concatenate([sc.tables[key].reshape(-1) for key in
                              ("radar_GaPs", "radar_Pn", "radar_D", "radar_pd_no", "radar_rd_pen", "radar_gr",
                               "jam_pmin", "jam_pmax", "jam_gj", "jr_denom")])
        np.testing.assert_array_equal(batch.tables[:, k], col)
        np.testing.assert_array_equal(batch.flags[:, k], sc.tables["jr_flags"].reshape(-1))
        np.testing.assert_array_equal(batch.state_vectors[k], g["static_state"])
        np.testing.assert_array_equal(batch.snr_no[k], g["f64_s42_snr_no"][0])
    shard = ScenarioBatch.randomized(ring_scenario_dict(3, 4), 2, seed=7, env_offset=1)
    np.testing.assert_array_equal(shard.tables, batch.tables[:, 1:])
    with pytest.raises(ValueError):
        ScenarioBatch([load_scenario("3j4r")[0], load_scenario("6j8r")[0]])


def test_f32_and_f64_modes_differ_in_reference():
    """Documents why both arithmetic modes exist: under NumPy 2 the reference's received power differs
    between np.float32 and python-float actions (float32 numerator, jammer.py:95)."""
    _, g = load_scenario("3j4r")
    prj32, prj64 = g["f32_s42_prj"], g["f64_s42_prj"]
    both = (prj32 > 0) & (prj64 > 0)
    assert both.any()
    rel = np.abs(prj32[both] - prj64[both]) / prj64[both]
    assert rel.max() > 1e-9 and rel.max() < 1e-6


def test_philox_known_answers():
    """Random123 Philox4x32-10 known-answer vectors (kat_vectors of the Random123 distribution)."""
    import ctypes
    lib = oracle_lib()
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kats:
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        lib.macjd_oracle_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_philox_uniform_range_and_determinism():
    lib = oracle_lib()
    vals = [lib.macjd_oracle_uniform(7, e, ep, s, k) for e in range(50) for ep in range(2) for s in range(3) for k in range(7)]
    a = np.array(vals)
    assert (a > 0).all() and (a < 1).all()
    assert len(np.unique(a)) == a.size          # every (env, episode, step, slot) has its own value
    assert lib.macjd_oracle_uniform(7, 3, 0, 1, 2) == lib.macjd_oracle_uniform(7, 3, 0, 1, 2)
    assert abs(a.mean() - 0.5) < 0.05
    # u = (word + 0.5) * 2^-32 of Philox block slot >> 2 (include/macjd.h): check against the raw block
    import ctypes
    c = (ctypes.c_uint32 * 4)(9, 4, 17, 1)      # env 9, episode 4, step 17, block 1 (slots 4..7)
    k = (ctypes.c_uint32 * 2)(7, 0)
    o = (ctypes.c_uint32 * 4)()
    lib.macjd_oracle_philox4x32_10(c, k, o)
    for w in range(4):
        assert lib.macjd_oracle_uniform(7, 9, 4, 17, 4 + w) == (o[w] + 0.5) / 4294967296.0


def test_oracle_episodes_draw_fresh_values_and_are_shard_invariant():
    """Philox mode: consecutive episodes of the same env with identical actions see different detection draws
    (the reference draws fresh np.random.rand() values every episode, environment.py:341,430); an env's values do
    not depend on which shard it lives in."""
    sc, _ = load_scenario("3j4r")
    rng = np.random.default_rng(4)
    E = 600
    T = rng.integers(0, 2 * sc.num_radars + 1, size=(E, sc.num_jammers)).astype(np.int32)
    P = rng.random((E, sc.num_jammers)).astype(np.float32)
    whole, lo, hi = OracleEnv(sc, E), OracleEnv(sc, E // 2), OracleEnv(sc, E // 2)
    per_ep = []
    for ep in range(3):
        for env in (whole, lo, hi):
            env.reset()
        tr = []
        for t in range(3):
            o = whole.step(T, P, seed=5)
            ol = lo.step(T[:E // 2], P[:E // 2], seed=5)
            oh = hi.step(T[E // 2:], P[E // 2:], seed=5, env_offset=E // 2)
            np.testing.assert_array_equal(o["track"], np.concatenate([ol["track"], oh["track"]]))
            np.testing.assert_array_equal(o["out64"], np.concatenate([ol["out64"], oh["out64"]]))
            tr.append(o["track"])
        per_ep.append(np.stack(tr))
    for a, b in ((0, 1), (1, 2)):
        frac = (per_ep[a] != per_ep[b]).mean()
        assert 0.05 < frac < 0.4, frac


def test_pd_floor_known_answer():
    """pd(0) = 0.10292481173711425 (reference import)."""
    sc, _ = load_scenario("3j4r")
    import ctypes
    desc, keep = sc.c_desc()
    v = oracle_lib().macjd_oracle_detection_probability(ctypes.addressof(desc), 0.0)
    assert v == pytest.approx(0.10292481173711425, rel=1e-15)
    assert oracle_lib().macjd_oracle_detection_probability(ctypes.addressof(desc), 1e9) == 1.0


def test_oracle_batch_matches_single_and_threads():
    """Batched / multi-threaded oracle == env-by-env oracle (envs are independent)."""
    sc, _ = load_scenario("6j8r")
    rng = np.random.default_rng(5)
    E = 257
    a, b = OracleEnv(sc, E), OracleEnv(sc, E, n_threads=4)
    for t in range(5):
        T = rng.integers(-1, 2 * sc.num_radars + 3, size=(E, sc.num_jammers)).astype(np.int32)
        P = rng.random((E, sc.num_jammers)).astype(np.float32)
        oa = a.step(T, P, seed=11, env_offset=100)
        ob = b.step(T, P, seed=11, env_offset=100)
        for k in ("out64", "pd64", "track", "terminated", "prj64"):
            np.testing.assert_array_equal(oa[k], ob[k])
    single = OracleEnv(sc, 1)
    single.track[:] = 0
    o1 = single.step(T[7:8], P[7:8], seed=11, env_offset=107)
    assert o1["out64"].shape == (1, 4)
