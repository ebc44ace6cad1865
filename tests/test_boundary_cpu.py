"""CPU-side checks of the drop-in boundary: the C-ABI library exports every symbol include/macjd.h
declares (no compute calls without a GPU), the scenario front-end validates like the reference
(simulation/environment.py:44-79,133-199), and the product path refuses to run without a HIP device
instead of silently falling back to a CPU implementation."""
import ctypes
import os
import re
import subprocess
import tempfile
from types import SimpleNamespace

import pytest
import yaml

import __graft_entry__ as entry
from _harness import REPO, StepIO as HarnessStepIO, load_scenario

from macjd_amd import _native
from macjd_amd.scenario import Scenario, _Desc, ring_scenario_dict


@pytest.fixture(scope="module")
def built():
    entry.build()
    return ctypes.CDLL(_native.LIB_PATH)


def _write(d):
    p = os.path.join(tempfile.mkdtemp(prefix="macjd_sc_"), "s.yaml")
    with open(p, "w") as f:
        yaml.safe_dump(d, f)
    return p


def test_library_exports_every_declared_symbol(built):
    inc = os.path.join(REPO, "include")
    hdr = "\n".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(macjd_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for sym in declared:
        assert hasattr(built, sym), sym
    built.macjd_abi_version.restype = ctypes.c_int
    assert built.macjd_abi_version() == _native.ABI_VERSION


def test_step_io_struct_layout_matches_header():
    """ctypes mirrors (product and test harness) agree with each other and with the C struct layout."""
    assert ctypes.sizeof(_native.StepIO) == ctypes.sizeof(HarnessStepIO)
    assert [f[0] for f in _native.StepIO._fields_] == [f[0] for f in HarnessStepIO._fields_]
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "macjd.h"\n'
           'int main(){printf("%zu %zu %zu %zu\\n", sizeof(macjd_step_io), sizeof(macjd_scenario_desc), '
           'offsetof(macjd_step_io, prj64), offsetof(macjd_step_io, track));return 0;}\n')
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "t.c"), "w") as f:
        f.write(src)
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), "-o", os.path.join(d, "t"),
                    os.path.join(d, "t.c")], check=True)
    out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_native.StepIO)
    assert int(out[1]) == ctypes.sizeof(_Desc)
    assert int(out[2]) == _native.StepIO.prj64.offset
    assert int(out[3]) == _native.StepIO.track.offset


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment, ElectromagneticEnvironment
    sc = Scenario.from_dict(ring_scenario_dict(3, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BatchedElectromagneticEnvironment(scenario=sc, batch_envs=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ElectromagneticEnvironment(SimpleNamespace(), _write(ring_scenario_dict(2, 2)))


def test_product_never_references_oracle():
    """Nothing in the product package may import, load, link or name anything under oracle/ (any module or file whose
    name contains "oracle", e.g. nets_oracle, macjd_oracle.c, libmacjd_oracle.so) nor the tests' harness."""
    pkg = os.path.join(REPO, "ma-cjd-cooperative-jamming-decision-making-via-marl_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            assert "oracle" not in f.lower(), f
            if f.endswith((".py", ".hip", ".h", ".cpp", ".yaml")):
                txt = open(os.path.join(root, f)).read()
                assert "libmacjd_oracle" not in txt and "macjd_oracle_" not in txt, f
                for m in re.finditer(r"^\s*(?:from\s+(\S+)\s+import\s+(.+)|import\s+(.+))$", txt, flags=re.M):
                    names = " ".join(x for x in m.groups() if x)
                    assert "oracle" not in names.lower() and "_harness" not in names, (f, m.group(0))
                assert not re.search(r"[\"']oracle[\"'/]", txt), f   # path pieces such as os.path.join(repo, "oracle")


def test_scenario_validation_matches_reference_errors(capsys):
    good = ring_scenario_dict(2, 2)
    with pytest.raises(FileNotFoundError):
        Scenario.from_yaml("/nonexistent/sim.yaml")
    bad = os.path.join(os.path.dirname(_write(good)), "bad.yaml")
    with open(bad, "w") as f:
        f.write("radars: [1, 2\n")
    with pytest.raises(yaml.YAMLError):
        Scenario.from_yaml(bad)
    with pytest.raises(ValueError, match="missing required 'radars' or 'jammers' keys"):
        Scenario.from_dict({"radars": []})
    d = ring_scenario_dict(2, 2); d.pop("protected_target")
    with pytest.raises(ValueError, match="missing required 'protected_target' key"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["protected_target"].pop("rcs")
    with pytest.raises(ValueError, match="must contain 'position' and 'rcs'"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["radars"][1].pop("pulse_compression_gain")
    with pytest.raises(KeyError, match="Radar config 1 missing required parameter: pulse_compression_gain"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["radars"][0]["type_id"] = 7
    with pytest.raises(ValueError, match="Radar config 0 invalid type_id"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["jammers"][0].pop("bj")
    with pytest.raises(KeyError, match="Jammer config 0 missing required parameter: bj"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["jammers"][1]["colour"] = "red"
    with pytest.raises(TypeError, match="unexpected keyword argument 'colour'"):
        Scenario.from_dict(d)
    d = ring_scenario_dict(2, 2); d["environment_params"]["rewards"]["rd_min"] = -0.5
    capsys.readouterr()
    Scenario.from_dict(d)
    assert "rd_min (-0.5) should be less than rd_max" in capsys.readouterr().out


def test_scenario_counts_follow_config_namespace():
    """num_jammers/num_radars/episode_limit come from getattr(config, ...) with the YAML as default
    (environment.py:82-84); main.py's namespace keeps them nested in env_args, so the YAML decides."""
    d = ring_scenario_dict(3, 4)
    sc = Scenario.from_dict(d, config=SimpleNamespace(env_args={"num_jammers": 2}))
    assert (sc.num_jammers, sc.num_radars, sc.episode_limit) == (3, 4, 100)
    sc = Scenario.from_dict(d, config=SimpleNamespace(num_jammers=2, num_radars=3, episode_limit=7))
    assert (sc.num_jammers, sc.num_radars, sc.episode_limit, sc.n_actions, sc.state_dim) == (2, 3, 7, 7, 34)
    with pytest.raises(ValueError):
        Scenario.from_dict(d, config=SimpleNamespace(num_radars=9))


def test_scenario_tables_known_answers():
    """SURVEY.md section 8(c) known answers for the shipped 2j/2r scenario."""
    sc, _ = load_scenario("2j2r_shipped")
    t = sc.tables
    assert t["radar_Ps"][0] == pytest.approx(3.3534683110189373e-10, rel=1e-15)
    assert t["radar_Ps"][1] == pytest.approx(2.0120809866113628e-11, rel=1e-15)
    assert t["radar_snr_no"][0] == pytest.approx(1.680715505856303e-05, rel=1e-15)
    assert t["radar_pd_no"][0] == pytest.approx(0.10292951362832024, rel=1e-15)
    assert t["radar_pd_no"][1] == pytest.approx(0.10292515026692867, rel=1e-15)
    assert t["radar_Pn"][0] == pytest.approx(0.001995262314968879, rel=1e-15)
    assert sc.pd_consts[0] == pytest.approx(13.337474757021274, rel=1e-15)
    assert sc.pd_consts[1] == pytest.approx(0.6191433636703569, rel=1e-15)
    assert sc.pd_consts[2] == pytest.approx(3.300496970842553, rel=1e-15)
    # Prj(j0->r0, 50 W) = 0.7981049259875517: (50 * gj * gr) / denom
    prj = (50.0 * t["jam_gj"][0] * t["radar_gr"][0]) / t["jr_denom"][0]
    assert prj == pytest.approx(0.7981049259875517, rel=1e-15)


def test_edge_scenario_flags():
    sc, _ = load_scenario("3j3r_edge")
    den = sc.tables["jr_denom"].reshape(3, 3)
    flg = sc.tables["jr_flags"].reshape(3, 3)
    assert den[0, 0] < 0            # jammer 0 sits on radar 0
    assert flg[2, 2] == 1           # Python-float (weak) denominator
    assert flg.sum() == 1 and (den[den >= 0] > 0).all()


def test_graph_launch_queue_reservation_rules():
    """macjd_amd/hipgraph.py: importing the package before the HIP runtime starts reserves one of the four hardware queues
    for graph launches (GPU_MAX_HW_QUEUES=3 -> replays go through the high-priority stream); a user setting that leaves no
    room selects the other arrangement (replays on the caller's stream, captured graphs never destroyed)."""
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import macjd_amd; from macjd_amd import hipgraph; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), hipgraph.launch_mode())" % REPO)
    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "MACJD_GRAPH_REPLAY_STREAM")}
        e.update(env)
        return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300).stdout.split()
    assert run() == ["3", "high"]
    assert run(GPU_MAX_HW_QUEUES="2") == ["2", "high"]
    assert run(GPU_MAX_HW_QUEUES="4") == ["4", "current"]
    assert run(GPU_MAX_HW_QUEUES="4", MACJD_GRAPH_REPLAY_STREAM="high") == ["4", "high"]
    assert run(MACJD_GRAPH_REPLAY_STREAM="current") == ["3", "current"]


def test_batch_exploration_schedule_equals_the_selectors_anneal():
    """The runner computes the exploration probabilities of a whole episode batch in one vector expression
    (BatchedEpisodeRunner._eps_schedule): bit-equal (as float32) to calling the selector's anneal() step by step
    (reference utils/action_selectors.py:30-32), the selector left at the same epsilon, test mode frozen."""
    import numpy as np
    from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
    from macjd_amd.utils.action_selectors import EpsilonGreedyActionSelector

    def selector():
        return EpsilonGreedyActionSelector(SimpleNamespace(epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=777))

    for t_env, n in ((0, 100), (700, 100), (123, 20), (5000, 7)):
        a, b = selector(), selector()
        want = np.zeros(100, dtype=np.float32)
        for t in range(n):
            want[t] = a.anneal(t_env + t)
        stub = SimpleNamespace(mac=SimpleNamespace(action_selector=b), t_env=t_env, episode_limit=100)
        got = BatchedEpisodeRunner._eps_schedule(stub, n, False)
        assert got.dtype == np.float32 and np.array_equal(got, want) and b.epsilon == a.epsilon
        frozen = BatchedEpisodeRunner._eps_schedule(stub, n, True)
        assert np.array_equal(frozen[:n], np.full(n, np.float32(b.epsilon))) and not frozen[n:].any() and b.epsilon == a.epsilon
