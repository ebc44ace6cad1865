"""Drop-in facade ``ElectromagneticEnvironment`` (E=1) against the reference's golden traces: same
constructor, list-of-tuples actions, NumPy outputs, and the same consumption of the global np.random
stream (so np.random.seed(s) reproduces the reference trajectory)."""
import contextlib
import io
import json
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import yaml

from _harness import load_scenario

pytestmark = pytest.mark.gpu


def _facade(name):
    from macjd_amd.simulation.environment import ElectromagneticEnvironment
    _, g = load_scenario(name)
    path = os.path.join(tempfile.mkdtemp(prefix="macjd_t_"), name + ".yaml")
    with open(path, "w") as f:
        yaml.safe_dump(json.loads(str(g["scenario_json"])), f)
    with contextlib.redirect_stdout(io.StringIO()) as out:
        env = ElectromagneticEnvironment(SimpleNamespace(), path)
    return env, g, out.getvalue()


@pytest.mark.parametrize("name", ["2j2r_shipped", "3j4r", "3j3r_edge", "12j16r"])
@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_facade_reproduces_reference_stream(name, mode):
    env, g, banner = _facade(name)
    assert "Environment Initialized" in banner
    seed = 43
    pre = f"{mode}_s{seed}_"
    T, P = g[pre + "T"], g[pre + "P"]
    np.random.seed(seed)  # the facade draws from the same global MT19937 stream as the reference
    with contextlib.redirect_stdout(io.StringIO()):
        for t in range(T.shape[0]):
            if g[pre + "reset_before"][t]:
                s0 = env.reset()
                assert s0.dtype == np.float32
                np.testing.assert_array_equal(s0, g["static_state"])
            if mode == "f32":
                acts = [(np.int64(T[t, i]), np.float32(P[t, i])) for i in range(T.shape[1])]
            else:
                acts = [(int(T[t, i]), float(P[t, i])) for i in range(T.shape[1])]
            obs, reward, term, info = env.step(acts)
            assert len(obs) == env.num_jammers and obs[0].dtype == np.float32
            assert abs(reward - g[pre + "reward"][t]) <= 1e-5
            assert reward == pytest.approx(g[pre + "reward"][t], rel=1e-9, abs=1e-12)
            assert term == bool(g[pre + "terminated"][t])
            np.testing.assert_array_equal([s["is_tracking"] for s in info["radar_states"]],
                                          g[pre + "track"][t].astype(bool))
            np.testing.assert_allclose(info["radar_pds"], g[pre + "pd"][t], rtol=1e-12)
            np.testing.assert_allclose(info["snr_with_jamming"], g[pre + "snr_with"][t], rtol=1e-13)
            np.testing.assert_array_equal(info["snr_no_jamming"], g[pre + "snr_no"][t])
            for k in ("r_d", "r_p", "r_j"):
                assert info[k] == pytest.approx(g[pre + k][t], rel=1e-9, abs=1e-12)
            prj = np.full(env.num_jammers, -1.0)
            for a in info["jammer_actions"]:
                prj[a["jammer_idx"]] = a["received_power"]
            np.testing.assert_allclose(prj, g[pre + "prj"][t], rtol=1e-14)
    # the facade consumed exactly as many uniforms as the reference did
    nxt = np.random.rand()
    np.random.seed(seed)
    for _ in range(int(g[pre + "n_draws"].sum())):
        np.random.rand()
    assert nxt == np.random.rand()


def test_facade_api_surface_and_errors():
    env, g, _ = _facade("2j2r_shipped")
    assert env.get_env_info() == {"state_shape": 24, "obs_shape": 24, "n_actions": 5, "n_agents": 2,
                                  "episode_limit": 100}
    av = env.get_avail_actions()
    assert len(av) == 2 and av[0].dtype == np.int32 and av[0].tolist() == [1] * 5
    np.testing.assert_array_equal(env.get_agent_obs(1), g["static_state"])
    with pytest.raises(ValueError, match="Invalid agent_id"):
        env.get_agent_obs(2)
    with pytest.raises(ValueError, match="Received 1 actions, but expected 2"):
        env.step([(0, 0.0)])
    # SURVEY.md section 8(c) known answers: seed 42, actions [(1,0.5),(4,0.25)]
    np.random.seed(42)
    env.reset()
    want = [0.9125037007160507, 0.9125037007160507, 0.11250370071605065]
    for w in want:
        _, r, term, info = env.step([(1, 0.5), (4, 0.25)])
        assert r == pytest.approx(w, rel=1e-12)
        assert not term
    assert info["r_d"] == pytest.approx(-0.8) and info["r_p"] == pytest.approx(-0.08750000000000001)
    # out-of-range T is not an error: warning + idle (environment.py:268)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        env.step([(99, 0.5), (0, 0.0)])
    assert "invalid discrete action T_i=99" in buf.getvalue()
    with contextlib.redirect_stdout(io.StringIO()):
        env.close()
