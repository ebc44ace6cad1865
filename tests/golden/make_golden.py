#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE implementation in the build container.

Run (build container only; /root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py [--only env|nets|all]

The reference (mfathulkr/MA-CJD-...) has no tests of its own, so these fixtures are what pins the
oracle (oracle/macjd_oracle.c, oracle/nets_oracle.py) and, through it, the HIP path.  Only DATA is
written: inputs and the reference's outputs (npz / json).  No reference source or bytecode is copied
(PYTHONDONTWRITEBYTECODE is set; the reference tree is read-only).

Environment used to generate the committed fixtures: Python 3.10.12, numpy 2.2.6,
torch 2.10.0+rocm7.0 (CPU), PyYAML 6.0.3.  NumPy >= 2 matters: with NEP-50 promotion the
runner-style np.float32 power scalars keep environment.py:271-277 / jammer.py:95 in float32;
both that mode ("f32") and python-float actions ("f64") are recorded.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
from types import SimpleNamespace

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MACJD_REFERENCE", "/root/reference")

sys.path.insert(0, REPO)
from macjd_amd.scenario import randomized_scenario_dict, ring_scenario_dict  # build-authored scenario definitions (data)


def edge_scenario_dict():
    """Build-authored 3-jammer / 3-radar scenario that exercises the reference's guard branches:
    jammer 0 sits exactly on radar 0 (distance <= 1e-6: actions ignored, environment.py:284);
    jammer 1 has power_min > 0 (reference __main__ example values, environment.py:595);
    jammer 2 has power_max == power_min (power_range <= 1e-6 -> norm 0, environment.py:276) and sits
    2e-5 m from radar 2 (d^2 <= 1e-9 -> Python-float denominator, jammer.py:83);
    threat levels outside [0.8, 1.2] so the r_d clip is active (environment.py:365)."""
    d = ring_scenario_dict(3, 3)
    d["radars"][0]["position"] = [400.0, 0.0]
    d["jammers"][0]["position"] = [400.0, 0.0]
    d["jammers"][1].update(power_max=120, power_min=10, gj=22, loss=4, bj=1.2e6)
    r2 = d["radars"][2]["position"]
    d["jammers"][2]["position"] = [r2[0] + 2e-5, r2[1]]
    d["jammers"][2].update(power_max=50.0, power_min=50.0)
    d["radars"][0]["threat_level"] = 0.5
    d["radars"][1]["threat_level"] = 1.5
    d["radars"][2]["threat_level"] = 1.0
    d["environment_params"]["rewards"] = dict(rd_min=-1.1, rd_max=-0.9, rp_min=-0.2, rp_max=-0.02)
    return d


def scenarios():
    with open(os.path.join(REF, "config", "simulation_config.yaml")) as f:
        shipped = yaml.safe_load(f)
    return {
        "2j2r_shipped": shipped,           # the reference's own scenario (config/simulation_config.yaml)
        "3j4r": ring_scenario_dict(3, 4),
        "6j8r": ring_scenario_dict(6, 8),
        "12j16r": ring_scenario_dict(12, 16),
        "3j3r_edge": edge_scenario_dict(),
        # per-env randomised scenarios: variations 0..2 of the 3j/4r ring, exactly the dicts
        # ScenarioBatch.randomized(ring_scenario_dict(3, 4), n, seed=7) compiles for envs 0..2
        **{f"3j4r_rand{k}": randomized_scenario_dict(ring_scenario_dict(3, 4), np.random.default_rng([7, k]))
           for k in range(3)},
    }


LIGHT = {"3j4r_rand0", "3j4r_rand1", "3j4r_rand2"}   # 100 steps, one seed per dtype mode


@contextlib.contextmanager
def reference_cwd():
    old = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        yield
    finally:
        os.chdir(old)
        sys.path.remove(REF)


def make_actions(rng, step, J, R):
    """Random actions with scripted edge cases."""
    T = rng.integers(-1, 2 * R + 3, size=J)            # includes T<0, T=0 and T>2R (idle + warning)
    P = rng.random(J)
    k = step % 25
    if k == 3:
        T[:] = 2 * (step % R) + 1                       # everyone suppresses the same radar
    elif k == 7:
        T[:] = 2 * (step % R) + 2                       # everyone deceives the same radar
    elif k == 11:
        P[:] = 0.0                                      # zero power (actual_power > 0 fails when pmin == 0)
    elif k == 13:
        P[0] = -0.25; P[-1] = 1.75                      # clipped
    elif k == 17:
        T[:] = 0                                        # all idle
    elif k == 19:
        T[:] = rng.integers(1, 2 * R + 1, size=J)      # all valid
        P[:] = 1.0
    return T.astype(np.int64), P


def gen_env(out_dir, names=None):
    with reference_cwd():
        from simulation.environment import ElectromagneticEnvironment
        scs = scenarios()
        tmpdir = tempfile.mkdtemp(prefix="macjd_golden_")
        for name, sc in scs.items():
            if names and name not in names:
                continue
            path = os.path.join(tmpdir, name + ".yaml")
            with open(path, "w") as f:
                yaml.safe_dump(sc, f)
            with contextlib.redirect_stdout(io.StringIO()):
                env = ElectromagneticEnvironment(SimpleNamespace(), path)
            J, R = env.num_jammers, env.num_radars
            info0 = env.get_env_info()
            static = {
                "state": env.reset(), "obs": np.array(env.get_obs()),
                "avail_actions": np.array(env.get_avail_actions()),
            }
            assert static["avail_actions"].dtype == np.int32
            rec = {"scenario_json": json.dumps(sc), "env_info_json": json.dumps(info0)}
            rec.update({"static_" + k: v for k, v in static.items()})
            n_steps = 100 if name in LIGHT else 200
            for mode in ("f32", "f64"):
                for seed in ((42,) if name in LIGHT else (42, 43, 44)):
                    rng = np.random.default_rng(1000 + seed)
                    np.random.seed(seed)
                    drawn = []
                    orig_rand = np.random.rand

                    def logged_rand(*a):
                        v = orig_rand(*a)
                        drawn.append(float(v))
                        return v
                    np.random.rand = logged_rand
                    try:
                        tr = {k: [] for k in ("T", "P", "u", "n_draws", "reward", "r_d", "r_p", "r_j", "pd",
                                              "snr_no", "snr_with", "track", "terminated", "prj", "reset_before")}
                        with contextlib.redirect_stdout(io.StringIO()):
                            for step in range(n_steps):
                                if step % 100 == 0:
                                    env.reset()
                                tr["reset_before"].append(step % 100 == 0)
                                T, P = make_actions(rng, step, J, R)
                                if mode == "f32":
                                    P_in = P.astype(np.float32)
                                    acts = [(T[i], P_in[i]) for i in range(J)]   # np.int64 / np.float32 scalars, as episode_runner.py:81
                                else:
                                    P_in = P.astype(np.float64)
                                    acts = [(int(T[i]), float(P_in[i])) for i in range(J)]
                                drawn.clear()
                                obs, reward, term, info = env.step(acts)
                                u = np.full(R + J, np.nan)
                                u[:len(drawn)] = drawn
                                prj = np.full(J, -1.0)
                                for a in info["jammer_actions"]:
                                    prj[a["jammer_idx"]] = a["received_power"]
                                tr["T"].append(T.astype(np.int32)); tr["P"].append(P_in.astype(np.float64))
                                tr["u"].append(u); tr["n_draws"].append(len(drawn))
                                tr["reward"].append(float(reward)); tr["r_d"].append(float(info["r_d"]))
                                tr["r_p"].append(float(info["r_p"])); tr["r_j"].append(float(info["r_j"]))
                                tr["pd"].append(np.array(info["radar_pds"], dtype=np.float64))
                                tr["snr_no"].append(np.array(info["snr_no_jamming"], dtype=np.float64))
                                tr["snr_with"].append(np.array(info["snr_with_jamming"], dtype=np.float64))
                                tr["track"].append(np.array([s["is_tracking"] for s in info["radar_states"]], dtype=np.uint8))
                                tr["terminated"].append(bool(term)); tr["prj"].append(prj)
                                assert all(np.array_equal(o, static["state"]) for o in obs)
                    finally:
                        np.random.rand = orig_rand
                    for k, v in tr.items():
                        rec[f"{mode}_s{seed}_{k}"] = np.array(v)
            np.savez_compressed(os.path.join(out_dir, f"env_{name}.npz"), **rec)
            print(f"env_{name}.npz  J={J} R={R} S={info0['state_shape']}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all", choices=["env", "nets", "all"])
    ap.add_argument("--names", default="", help="comma-separated env fixture names (default: all)")
    args = ap.parse_args()
    if args.only in ("env", "all"):
        gen_env(HERE, set(args.names.split(",")) if args.names else None)
    if args.only in ("nets", "all"):
        try:
            from make_golden_nets import gen_nets
        except ImportError:
            sys.path.insert(0, HERE)
            from make_golden_nets import gen_nets
        gen_nets(HERE, REF)
