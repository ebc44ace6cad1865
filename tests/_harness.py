"""Shared test helpers: ctypes views of include/macjd.h, the oracle loader and golden-trace readers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg touch oracle/ (it is the
checker, never the product path)."""
import ctypes
import json
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libmacjd_oracle.so")


class StepIO(ctypes.Structure):
    """ctypes mirror of ``macjd_step_io`` (include/macjd.h)."""
    _fields_ = [
        ("n_envs", ctypes.c_int64), ("env_offset", ctypes.c_int64), ("seed", ctypes.c_uint64),
        ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
        ("T", ctypes.c_void_p), ("T_se", ctypes.c_int64), ("T_sx", ctypes.c_int64),
        ("P32", ctypes.c_void_p), ("P64", ctypes.c_void_p), ("P_se", ctypes.c_int64), ("P_sx", ctypes.c_int64),
        ("u", ctypes.c_void_p), ("u_se", ctypes.c_int64), ("u_sx", ctypes.c_int64),
        ("episode", ctypes.c_void_p),
        ("track", ctypes.c_void_p), ("k_se", ctypes.c_int64), ("k_sx", ctypes.c_int64),
        ("step", ctypes.c_void_p),
        ("reward", ctypes.c_void_p), ("r_dpj", ctypes.c_void_p), ("terminated", ctypes.c_void_p),
        ("pd", ctypes.c_void_p), ("pd_se", ctypes.c_int64), ("pd_sx", ctypes.c_int64),
        ("snr_with", ctypes.c_void_p), ("sw_se", ctypes.c_int64), ("sw_sx", ctypes.c_int64),
        ("out64", ctypes.c_void_p), ("pd64", ctypes.c_void_p), ("snr64", ctypes.c_void_p),
        ("prj64", ctypes.c_void_p),
        ("pe_tables", ctypes.c_void_p), ("pe_flags", ctypes.c_void_p), ("pe_stride", ctypes.c_int64),
        ("pe_tile", ctypes.c_int32), ("reserved_pe", ctypes.c_int32),
        ("r_dpj_sum", ctypes.c_void_p),
    ]


_oracle = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle_lib():
    """Loads (building if needed) oracle/libmacjd_oracle.so."""
    global _oracle
    if _oracle is None:
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                    for f in ("macjd_oracle.c", "macjd_oracle_mt.c", "Makefile"))
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < src_m:
            build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.macjd_oracle_env_step.restype = ctypes.c_int
        lib.macjd_oracle_env_step.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_void_p]
        lib.macjd_oracle_env_step_mt.restype = ctypes.c_int
        lib.macjd_oracle_env_step_mt.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_int]
        lib.macjd_oracle_uniform.restype = ctypes.c_double
        lib.macjd_oracle_uniform.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        lib.macjd_oracle_detection_probability.restype = ctypes.c_double
        lib.macjd_oracle_detection_probability.argtypes = [ctypes.c_void_p, ctypes.c_double]
        lib.macjd_oracle_max_threads.restype = ctypes.c_int
        lib.macjd_oracle_bench.restype = ctypes.c_int
        lib.macjd_oracle_bench.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_double)]
        _oracle = lib
    return _oracle


def load_scenario(name):
    """Scenario object + raw golden npz for fixture ``env_<name>.npz``."""
    from macjd_amd.scenario import Scenario
    g = np.load(os.path.join(GOLDEN, f"env_{name}.npz"))
    sc = Scenario.from_dict(json.loads(str(g["scenario_json"])), source=f"golden:{name}")
    return sc, g


class OracleEnv:
    """Batched env state (host arrays) stepped by the C oracle.  Mirrors the argument block the HIP
    library takes so that the same inputs can be fed to both."""

    def __init__(self, scenario, n_envs, n_threads=1):
        self.sc = scenario
        self.E = int(n_envs)
        self.R, self.J = scenario.num_radars, scenario.num_jammers
        self.desc, self._keep = scenario.c_desc()
        self.track = np.zeros((self.E, self.R), dtype=np.uint8)
        self.step_count = np.zeros(self.E, dtype=np.int32)
        self.episode = np.zeros(self.E, dtype=np.int32)   # Philox episode index, advanced by reset() like the HIP env's
        self.n_threads = n_threads

    def reset(self, mask=None):
        sel = slice(None) if mask is None else np.asarray(mask, dtype=bool)
        self.track[sel] = 0
        self.step_count[sel] = 0
        self.episode[sel] += 1

    def step(self, T, P, u=None, seed=0, env_offset=0, arith_f64=False):
        """T int32[E,J]; P float32 or float64 [E,J]; u float64[E,R+J] or None (Philox)."""
        E, R, J = self.E, self.R, self.J
        T = np.ascontiguousarray(T, dtype=np.int32).reshape(E, J)
        out = {
            "reward": np.zeros(E, np.float32), "r_dpj": np.zeros((E, 3), np.float32),
            "terminated": np.zeros(E, np.uint8), "pd": np.zeros((E, R), np.float32),
            "snr_with": np.zeros((E, R), np.float32), "out64": np.zeros((E, 4), np.float64),
            "pd64": np.zeros((E, R), np.float64), "snr64": np.zeros((E, R), np.float64),
            "prj64": np.zeros((E, J), np.float64), "draws": np.zeros(E, np.int32),
        }
        io = StepIO()
        io.n_envs, io.env_offset, io.seed = E, env_offset, seed
        io.flags = 1 if arith_f64 else 0
        io.T, io.T_se, io.T_sx = T.ctypes.data, J, 1
        P = np.ascontiguousarray(P).reshape(E, J)
        if P.dtype == np.float32:
            io.P32, io.P64 = P.ctypes.data, None
        else:
            P = P.astype(np.float64)
            io.P32, io.P64 = None, P.ctypes.data
        io.P_se, io.P_sx = J, 1
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64).reshape(E, R + J)
            io.u, io.u_se, io.u_sx = u.ctypes.data, R + J, 1
        io.episode = self.episode.ctypes.data
        io.track, io.k_se, io.k_sx = self.track.ctypes.data, R, 1
        io.step = self.step_count.ctypes.data
        io.reward, io.r_dpj, io.terminated = out["reward"].ctypes.data, out["r_dpj"].ctypes.data, out["terminated"].ctypes.data
        io.pd, io.pd_se, io.pd_sx = out["pd"].ctypes.data, R, 1
        io.snr_with, io.sw_se, io.sw_sx = out["snr_with"].ctypes.data, R, 1
        io.out64, io.pd64, io.snr64, io.prj64 = (out["out64"].ctypes.data, out["pd64"].ctypes.data,
                                                 out["snr64"].ctypes.data, out["prj64"].ctypes.data)
        lib = oracle_lib()
        if self.n_threads > 1:
            rc = lib.macjd_oracle_env_step_mt(ctypes.addressof(self.desc), ctypes.byref(io), self.n_threads)
        else:
            rc = lib.macjd_oracle_env_step(ctypes.addressof(self.desc), ctypes.byref(io), out["draws"].ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle env_step failed: {rc}")
        out["track"] = self.track.copy()
        out["step"] = self.step_count.copy()
        return out


def oracle_bench(scenario, n_envs, n_threads, n_steps, seed=0):
    """Times ``n_steps`` env steps of ``n_envs`` environments inside C (macjd_oracle_bench): pre-allocated buffers,
    Philox uniforms, synthetic actions (T ~ U{0..2R}, P ~ U[0,1)), the same outputs the HIP kernel writes.
    Returns env-steps / second."""
    E, R, J = int(n_envs), scenario.num_radars, scenario.num_jammers
    rng = np.random.default_rng(seed)
    desc, keep = scenario.c_desc()
    T = rng.integers(0, 2 * R + 1, size=(E, J)).astype(np.int32)
    P = rng.random((E, J)).astype(np.float32)
    bufs = {"track": np.zeros((E, R), np.uint8), "step": np.zeros(E, np.int32), "episode": np.zeros(E, np.int32),
            "reward": np.zeros(E, np.float32), "r_dpj": np.zeros((E, 3), np.float32), "terminated": np.zeros(E, np.uint8),
            "pd": np.zeros((E, R), np.float32), "snr": np.zeros((E, R), np.float32)}
    io = StepIO()
    io.n_envs, io.env_offset, io.seed, io.flags = E, 0, seed, 0
    io.T, io.T_se, io.T_sx = T.ctypes.data, J, 1
    io.P32, io.P64, io.P_se, io.P_sx = P.ctypes.data, None, J, 1
    io.episode = bufs["episode"].ctypes.data
    io.track, io.k_se, io.k_sx = bufs["track"].ctypes.data, R, 1
    io.step = bufs["step"].ctypes.data
    io.reward, io.r_dpj, io.terminated = bufs["reward"].ctypes.data, bufs["r_dpj"].ctypes.data, bufs["terminated"].ctypes.data
    io.pd, io.pd_se, io.pd_sx = bufs["pd"].ctypes.data, R, 1
    io.snr_with, io.sw_se, io.sw_sx = bufs["snr"].ctypes.data, R, 1
    sec = ctypes.c_double(0.0)
    rc = oracle_lib().macjd_oracle_bench(ctypes.addressof(desc), ctypes.byref(io), int(n_threads), int(n_steps), ctypes.byref(sec))
    if rc != 0:
        raise RuntimeError(f"macjd_oracle_bench failed: {rc}")
    del keep
    return E * n_steps / sec.value, sec.value


def random_actions(rng, E, J, R, with_invalid=True):
    lo, hi = (-1, 2 * R + 3) if with_invalid else (0, 2 * R + 1)
    T = rng.integers(lo, hi, size=(E, J)).astype(np.int32)
    P = rng.random((E, J)).astype(np.float32)
    return T, P
