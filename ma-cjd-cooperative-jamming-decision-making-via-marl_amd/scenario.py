"""Scenario compiler: sim-config YAML -> static float64 tables for the HIP env-step kernel.

The reference rebuilds Radar/Jammer objects on every reset and recomputes echo power, distances and
the no-jamming SNR on every step (reference simulation/environment.py:123-206, 316-333;
core/radar.py:10-60; core/jammer.py:24-54; utils/math_utils.py:3-8,40-42).  Nothing in step() moves
an entity (environment.py:237-238 is a TODO), so all of it is a pure function of the YAML.  This
module evaluates those quantities ONCE, on the host, with the same Python/NumPy expression forms the
reference uses (so the table entries are bit-identical to what the reference computes per step) and
hands them to the native library as a ``macjd_scenario_desc`` (include/macjd.h).

Validation and error behaviour follow environment.py:44-79 and :133-199 (same exception types and
messages for the cases the reference checks).
"""
from __future__ import annotations

import ctypes
import warnings
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np
import yaml

DEFAULT_SIM_CONFIG_PATH = "config/simulation_config.yaml"  # environment.py:11

MAX_RADARS = 32   # MACJD_MAX_RADARS
MAX_JAMMERS = 32  # MACJD_MAX_JAMMERS

_REQUIRED_RADAR_PARAMS = [  # environment.py:136-140
    "pt", "gt", "gr", "wavelength", "rcs", "loss", "latm", "pn",
    "type_id", "position", "theta_m", "theta_a", "t_s",
    "pulse_compression_gain", "anti_jamming_factor",
]
_REQUIRED_JAMMER_PARAMS = ["gj", "loss", "latm", "bj", "position"]  # environment.py:179
_JAMMER_INIT_KEYS = {"power", "gj", "loss", "latm", "bj", "position"}  # jammer.py:24


def db_to_linear(db_value):
    """dB -> linear power ratio (utils/math_utils.py:3-8)."""
    return 10 ** (db_value / 10.0)


def detection_probability_constants(prfa=1e-6, m=10):
    """The three constants of Radar.detection_probability (core/radar.py:67-77), evaluated with the
    reference's own expression forms: A = ln(0.62/prfa); c1 = 5 log10(m) / (6.2 + 4.54/sqrt(m) + 0.44);
    denB = 1.7 + 0.12 A."""
    safe_prfa = max(prfa, 1e-18)
    A = np.log(0.62 / safe_prfa)
    safe_m = max(m, 1)
    log10_m = np.log10(safe_m) if safe_m > 0 else 0
    c1 = (5 * log10_m) / (6.2 + 4.54 / np.sqrt(safe_m) + 0.44)
    den_b = 1.7 + 0.12 * A
    return float(A), float(c1), float(den_b)


def detection_probability(snr, consts=None):
    """Host evaluation of core/radar.py:67-82 (used only for the static no-jamming Pd table)."""
    A, c1, den_b = consts if consts is not None else detection_probability_constants()
    snr_linear = max(snr, 0.0)
    Z = snr_linear + c1
    if abs(den_b) < 1e-9:
        return 0.0
    B = (10 * Z - A) / den_b
    if B > 700:
        return 1.0
    if B < -700:
        return 0.0
    return 1 / (1 + np.exp(-B))


class _Desc(ctypes.Structure):
    """ctypes mirror of ``macjd_scenario_desc`` (include/macjd.h)."""
    _fields_ = [
        ("n_radars", ctypes.c_int32), ("n_jammers", ctypes.c_int32),
        ("episode_limit", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("rp_min", ctypes.c_double), ("rp_max", ctypes.c_double),
        ("pd_A", ctypes.c_double), ("pd_c1", ctypes.c_double), ("pd_denB", ctypes.c_double),
        ("radar_GaPs", ctypes.c_void_p), ("radar_Pn", ctypes.c_void_p), ("radar_D", ctypes.c_void_p),
        ("radar_pd_no", ctypes.c_void_p), ("radar_snr_no", ctypes.c_void_p),
        ("radar_rd_pen", ctypes.c_void_p), ("radar_gr", ctypes.c_void_p),
        ("jam_pmin", ctypes.c_void_p), ("jam_pmax", ctypes.c_void_p), ("jam_gj", ctypes.c_void_p),
        ("jr_denom", ctypes.c_void_p), ("jr_flags", ctypes.c_void_p),
    ]


@dataclass
class Scenario:
    """Validated scenario + its static tables.  Build with :meth:`from_yaml` / :meth:`from_dict`."""
    num_radars: int
    num_jammers: int
    episode_limit: int
    max_radar_types: int
    rd_min: float
    rd_max: float
    rp_min: float
    rp_max: float
    radars: List[Dict[str, Any]]
    jammers: List[Dict[str, Any]]
    target_position: np.ndarray
    target_rcs: float
    tables: Dict[str, np.ndarray] = field(default_factory=dict)
    pd_consts: tuple = (0.0, 0.0, 0.0)
    source: str = "<dict>"

    # ---- derived dims (environment.py:91-101) ----
    @property
    def n_actions(self) -> int:
        return 2 * self.num_radars + 1

    @property
    def radar_feature_dim(self) -> int:
        return 1 + 1 + 1 + self.max_radar_types + 1 + 2

    @property
    def state_dim(self) -> int:
        return self.num_radars * self.radar_feature_dim + self.num_jammers * 2

    # ---- construction ----
    @classmethod
    def from_yaml(cls, sim_config_path: str = DEFAULT_SIM_CONFIG_PATH, config: Any = None) -> "Scenario":
        try:  # environment.py:45-55
            with open(sim_config_path, "r") as f:
                sim_config = yaml.safe_load(f)
        except FileNotFoundError:
            print(f"Error: Simulation configuration file not found at {sim_config_path}")
            raise
        except yaml.YAMLError as e:
            print(f"Error parsing simulation configuration file {sim_config_path}: {e}")
            raise
        return cls.from_dict(sim_config, config=config, source=sim_config_path)

    @classmethod
    def from_dict(cls, sim_config: Optional[dict], config: Any = None, source: str = "<dict>") -> "Scenario":
        if not sim_config or "radars" not in sim_config or "jammers" not in sim_config:  # :48-49
            raise ValueError(f"Simulation config file {source} is missing required 'radars' or 'jammers' keys.")
        if "protected_target" not in sim_config:  # :58-59
            raise ValueError(f"Simulation config file {source} is missing required 'protected_target' key.")
        target = sim_config["protected_target"]
        if "position" not in target or "rcs" not in target:  # :61-62
            raise ValueError("Protected target config must contain 'position' and 'rcs'.")
        target_position = np.array(target["position"])  # :64
        target_rcs = target["rcs"]

        env_params = sim_config.get("environment_params", {}) or {}  # :67-73
        max_radar_types = env_params.get("max_radar_types", 4)
        reward_params = env_params.get("rewards", {}) or {}
        rd_min = reward_params.get("rd_min", -1.2)
        rd_max = reward_params.get("rd_max", -0.8)
        rp_min = reward_params.get("rp_min", -0.1)
        rp_max = reward_params.get("rp_max", -0.01)
        if rd_min >= rd_max:  # :76-79 (warnings only)
            print(f"Warning: rd_min ({rd_min}) should be less than rd_max ({rd_max}) in config.")
        if rp_min >= rp_max:
            print(f"Warning: rp_min ({rp_min}) should be less than rp_max ({rp_max}) in config.")

        radar_cfgs_all = sim_config.get("radars", []) or []
        jammer_cfgs_all = sim_config.get("jammers", []) or []
        num_jammers = getattr(config, "num_jammers", len(jammer_cfgs_all))  # :82-84
        num_radars = getattr(config, "num_radars", len(radar_cfgs_all))
        episode_limit = getattr(config, "episode_limit", 100)

        radar_cfgs = radar_cfgs_all[:num_radars]  # :129-131
        jammer_cfgs = jammer_cfgs_all[:num_jammers]  # :172-174
        if len(radar_cfgs) < num_radars or len(jammer_cfgs) < num_jammers:
            # The reference only warns here and then indexes past the end of its entity lists as
            # soon as such an entity is addressed; the batched kernel needs rectangular tables.
            raise ValueError(
                f"Requested {num_radars} radars / {num_jammers} jammers, but the scenario defines only "
                f"{len(radar_cfgs)} / {len(jammer_cfgs)}.")
        if not (1 <= num_radars <= MAX_RADARS and 1 <= num_jammers <= MAX_JAMMERS):
            raise ValueError(f"Scenario size {num_jammers}j/{num_radars}r outside the supported range "
                             f"1..{MAX_JAMMERS} / 1..{MAX_RADARS}.")

        radars: List[Dict[str, Any]] = []
        for i, params in enumerate(radar_cfgs):  # :133-169
            try:
                for p_name in _REQUIRED_RADAR_PARAMS:
                    if p_name not in params:
                        raise KeyError(f"Radar config {i} missing required parameter: {p_name}")
                if not (0 <= params.get("type_id", -1) < max_radar_types):
                    raise ValueError(f"Radar config {i} invalid type_id")
            except KeyError as e:
                print(f"Error initializing radar {i}: Missing parameter {e} in simulation_config.yaml")
                raise
            except ValueError as e:
                print(f"Error initializing radar {i}: Invalid value - {e}")
                raise
            r = dict(params)
            r["threat_level"] = params.get("threat_level", 1.0)
            radars.append(r)

        jammers: List[Dict[str, Any]] = []
        for i, params in enumerate(jammer_cfgs):  # :176-199
            try:
                for p_name in _REQUIRED_JAMMER_PARAMS:
                    if p_name not in params:
                        raise KeyError(f"Jammer config {i} missing required parameter: {p_name}")
            except KeyError as e:
                print(f"Error initializing jammer {i}: Missing parameter {e} in simulation_config.yaml")
                raise
            extra = set(params) - _JAMMER_INIT_KEYS - {"power_max", "power_min"}
            if extra:  # Jammer(**init_params) in the reference, environment.py:186-190
                k = sorted(extra)[0]
                print(f"Error initializing jammer {i}: Jammer.__init__() got an unexpected keyword argument '{k}'")
                raise TypeError(f"Jammer.__init__() got an unexpected keyword argument '{k}'")
            j = dict(params)
            j["power_max"] = params.get("power_max", 100.0)
            j["power_min"] = params.get("power_min", 0.0)
            jammers.append(j)

        sc = cls(num_radars=num_radars, num_jammers=num_jammers, episode_limit=int(episode_limit),
                 max_radar_types=max_radar_types, rd_min=rd_min, rd_max=rd_max, rp_min=rp_min, rp_max=rp_max,
                 radars=radars, jammers=jammers, target_position=target_position, target_rcs=target_rcs,
                 source=source)
        sc._compile()
        return sc

    # ---- static tables ----
    def _compile(self) -> None:
        R, J = self.num_radars, self.num_jammers
        consts = detection_probability_constants()
        self.pd_consts = consts
        GaPs = np.zeros(R); Pn = np.zeros(R); D = np.zeros(R); pd_no = np.zeros(R)
        snr_no = np.zeros(R); rd_pen = np.zeros(R); gr_lin = np.zeros(R); Ps_tab = np.zeros(R)
        radar_pos = []
        for r, p in enumerate(self.radars):
            pt = float(p["pt"])                       # radar.py:11
            gt = db_to_linear(p["gt"])                # radar.py:12-13
            gr = db_to_linear(p["gr"])
            lam = p["wavelength"]
            loss = db_to_linear(p["loss"])            # radar.py:16-17
            latm = db_to_linear(p["latm"])
            pn_watts = 10 ** ((p["pn"] - 30) / 10)    # radar.py:19
            position = np.array(p["position"])
            radar_pos.append(position)
            # echo power of the PROTECTED TARGET at this radar, radar.py:35-60
            distance = np.linalg.norm(position - self.target_position)
            safe_distance = max(distance, 1e-6)
            numerator = pt * gt * gr * (lam ** 2) * self.target_rcs
            denominator = ((4 * np.pi) ** 3) * (safe_distance ** 4) * loss * latm
            Ps = 0.0 if denominator <= 1e-18 else numerator / denominator
            Ga = p["pulse_compression_gain"]
            ga_ps = Ga * Ps                           # environment.py:326
            s_no = ga_ps / pn_watts if pn_watts > 1e-18 else 0.0
            s_no = max(0.0, s_no)                     # environment.py:327
            Ps_tab[r] = Ps
            GaPs[r] = ga_ps
            Pn[r] = pn_watts
            D[r] = p["anti_jamming_factor"]
            snr_no[r] = s_no
            pd_no[r] = detection_probability(s_no, consts)           # environment.py:385
            rd_pen[r] = np.clip(-p["threat_level"], self.rd_min, self.rd_max)  # environment.py:362-365
            gr_lin[r] = gr                            # get_antenna_gain, radar.py:84-85

        pmin = np.zeros(J); pmax = np.zeros(J); gj_lin = np.zeros(J)
        denom = np.zeros((J, R)); flags = np.zeros((J, R), dtype=np.uint8)
        for j, p in enumerate(self.jammers):
            gj = db_to_linear(p["gj"])                # jammer.py:44-46
            loss = db_to_linear(p["loss"])
            latm = db_to_linear(p["latm"])
            bj = p["bj"]
            pmin[j] = p["power_min"]
            pmax[j] = p["power_max"]
            gj_lin[j] = gj
            jpos = np.array(p["position"])
            for r in range(R):
                distance = np.linalg.norm(np.array(jpos) - np.array(radar_pos[r]))  # math_utils.py:40-42
                if not (distance > 1e-6):             # environment.py:284
                    denom[j, r] = -1.0
                    continue
                distance_sq = max(1e-9, distance ** 2)    # jammer.py:83
                effective_bj = max(1e-9, bj)              # jammer.py:87
                den = distance_sq * loss * latm * effective_bj  # jammer.py:88
                denom[j, r] = den
                if not isinstance(den, np.floating):
                    # Python-float denominator: with a float32 numerator NumPy-2 keeps the division
                    # in float32 (weak scalar promotion)
                    flags[j, r] |= 1
        self.tables = {
            "radar_GaPs": GaPs, "radar_Pn": Pn, "radar_D": D, "radar_pd_no": pd_no,
            "radar_snr_no": snr_no, "radar_rd_pen": rd_pen, "radar_gr": gr_lin, "radar_Ps": Ps_tab,
            "jam_pmin": pmin, "jam_pmax": pmax, "jam_gj": gj_lin,
            "jr_denom": np.ascontiguousarray(denom.reshape(-1)),
            "jr_flags": np.ascontiguousarray(flags.reshape(-1)),
        }

    # ---- observation pieces (static; environment.py:479-565) ----
    def state_vector(self) -> np.ndarray:
        """f32[S]: per radar [pt, theta_m, t_s, onehot(type_id), theta_a, x, y], then per jammer [x, y]
        (environment.py:479-510; utils/state_utils.py:3-27)."""
        feats: List[Any] = []
        for p in self.radars:
            feats.append(float(p["pt"]))
            feats.append(p["theta_m"])
            feats.append(p["t_s"])
            type_id = p["type_id"]
            if not isinstance(type_id, int):
                raise TypeError(f"Index must be an integer, got {type(type_id)}")
            onehot = np.zeros(self.max_radar_types, dtype=np.float32)
            onehot[type_id] = 1.0
            feats.extend(onehot)
            feats.append(p["theta_a"])
            feats.extend(np.array(p["position"]).flatten())
        for p in self.jammers:
            feats.extend(np.array(p["position"]).flatten())
        return np.array(feats, dtype=np.float32)

    def env_info(self) -> Dict[str, int]:
        """environment.py:553-565."""
        return {"state_shape": self.state_dim, "obs_shape": self.state_dim, "n_actions": self.n_actions,
                "n_agents": self.num_jammers, "episode_limit": self.episode_limit}

    # ---- native description ----
    def c_desc(self):
        """Returns (``_Desc`` instance, keepalive list).  Pointers reference ``self.tables`` arrays."""
        t = self.tables
        d = _Desc()
        d.n_radars, d.n_jammers, d.episode_limit, d.reserved = self.num_radars, self.num_jammers, self.episode_limit, 0
        d.rp_min, d.rp_max = float(self.rp_min), float(self.rp_max)
        d.pd_A, d.pd_c1, d.pd_denB = self.pd_consts
        keep = []
        for name in ("radar_GaPs", "radar_Pn", "radar_D", "radar_pd_no", "radar_snr_no", "radar_rd_pen",
                     "radar_gr", "jam_pmin", "jam_pmax", "jam_gj", "jr_denom"):
            a = np.ascontiguousarray(t[name], dtype=np.float64)
            keep.append(a)
            setattr(d, name, a.ctypes.data)
        f = np.ascontiguousarray(t["jr_flags"], dtype=np.uint8)
        keep.append(f)
        d.jr_flags = f.ctypes.data
        return d, keep


# Row order of the per-env SoA table (include/macjd.h, macjd_step_io.pe_tables)
_PE_RADAR_ROWS = ("radar_GaPs", "radar_Pn", "radar_D", "radar_pd_no", "radar_rd_pen", "radar_gr")
_PE_JAMMER_ROWS = ("jam_pmin", "jam_pmax", "jam_gj")


def randomized_scenario_dict(base: dict, rng: np.random.Generator, position_jitter: float = 60.0,
                             threat_jitter: float = 0.15, power_jitter: float = 0.2) -> dict:
    """One random variation of a scenario dict in the reference's schema: radar and jammer
    positions moved by N(0, position_jitter) metres per axis, threat levels by N(0, threat_jitter), radar peak
    power ``pt`` and jammer ``power_max`` scaled by exp(N(0, power_jitter)).  Everything else is kept."""
    import copy
    d = copy.deepcopy(base)
    for r in d["radars"]:
        x, y = (float(v) for v in np.array(r["position"]).flatten()[:2])
        r["position"] = [x + float(rng.normal(0.0, position_jitter)), y + float(rng.normal(0.0, position_jitter))]
        r["threat_level"] = float(r.get("threat_level", 1.0) + rng.normal(0.0, threat_jitter))
        r["pt"] = float(r["pt"] * np.exp(rng.normal(0.0, power_jitter)))
    for j in d["jammers"]:
        x, y = (float(v) for v in np.array(j["position"]).flatten()[:2])
        j["position"] = [x + float(rng.normal(0.0, position_jitter)), y + float(rng.normal(0.0, position_jitter))]
        j["power_max"] = float(j.get("power_max", 100.0) * np.exp(rng.normal(0.0, power_jitter)))
    return d


class ScenarioBatch:
    """E scenarios of one shape (same R, J, episode limit, r_p bounds, Pd constants), compiled one by one with
    :class:`Scenario` (so each env's tables are exactly what a single-scenario environment would use) and stacked
    as the per-env SoA the step kernel streams: ``tables`` float64 [6R + 3J + JR, E], ``flags`` uint8 [JR, E],
    plus the per-env observation vectors ``state_vectors`` f32 [E, S] and ``snr_no`` f64 [E, R]."""

    def __init__(self, scenarios: List[Scenario]):
        if not scenarios:
            raise ValueError("ScenarioBatch needs at least one scenario")
        s0 = scenarios[0]
        for i, sc in enumerate(scenarios):
            same = (sc.num_radars == s0.num_radars and sc.num_jammers == s0.num_jammers
                    and sc.episode_limit == s0.episode_limit and sc.max_radar_types == s0.max_radar_types
                    and sc.rp_min == s0.rp_min and sc.rp_max == s0.rp_max and sc.pd_consts == s0.pd_consts)
            if not same:
                raise ValueError(f"ScenarioBatch: scenario {i} differs from scenario 0 in shape / episode limit / "
                                 f"r_p bounds / Pd constants (these are shared by the whole batch)")
        self.scenarios = list(scenarios)
        self.base = s0
        self.n_envs = len(scenarios)
        rows = []
        for key in _PE_RADAR_ROWS + _PE_JAMMER_ROWS + ("jr_denom",):
            rows.append(np.stack([np.asarray(sc.tables[key], dtype=np.float64).reshape(-1) for sc in scenarios], axis=1))
        self.tables = np.ascontiguousarray(np.concatenate(rows, axis=0))                     # [rows, E]
        self.flags = np.ascontiguousarray(np.stack([sc.tables["jr_flags"].reshape(-1) for sc in scenarios], axis=1)
                                          .astype(np.uint8))                                  # [J*R, E]
        self.state_vectors = np.stack([sc.state_vector() for sc in scenarios]).astype(np.float32)
        self.snr_no = np.stack([sc.tables["radar_snr_no"] for sc in scenarios]).astype(np.float64)
        R, J = s0.num_radars, s0.num_jammers
        assert self.tables.shape == (6 * R + 3 * J + J * R, self.n_envs)

    def tile(self, n_envs: int) -> "ScenarioBatch":
        """A batch of ``n_envs`` envs that cycles through this batch's scenarios (env e runs scenario e mod E) —
        for streaming-size measurements, where compiling millions of distinct scenarios on the host is pointless."""
        reps = (int(n_envs) + self.n_envs - 1) // self.n_envs
        out = object.__new__(ScenarioBatch)
        out.scenarios = None                       # too many to keep per-env objects; ``base`` carries the shape
        out.base, out.n_envs = self.base, int(n_envs)
        out.tables = np.ascontiguousarray(np.tile(self.tables, (1, reps))[:, :n_envs])
        out.flags = np.ascontiguousarray(np.tile(self.flags, (1, reps))[:, :n_envs])
        out.state_vectors = np.ascontiguousarray(np.tile(self.state_vectors, (reps, 1))[:n_envs])
        out.snr_no = np.ascontiguousarray(np.tile(self.snr_no, (reps, 1))[:n_envs])
        return out

    @classmethod
    def randomized(cls, base_dict: dict, n_envs: int, seed: int = 0, config: Any = None, env_offset: int = 0,
                   **jitter) -> "ScenarioBatch":
        """``n_envs`` random variations of ``base_dict``; env e of the batch is variation number ``env_offset + e``
        of the stream keyed by ``seed`` (shard-invariant: a rank's shard of a bigger batch is the same scenarios)."""
        scs = []
        for e in range(n_envs):
            rng = np.random.default_rng([int(seed), int(env_offset) + e])
            scs.append(Scenario.from_dict(randomized_scenario_dict(base_dict, rng, **jitter), config=config,
                                          source=f"<randomized {seed}:{env_offset + e}>"))
        return cls(scs)


def ring_scenario_dict(n_jammers: int, n_radars: int) -> dict:
    """Build-authored ring scenarios in the reference's YAML schema for sizes the reference does not
    ship (it ships 2 jammers / 2 radars only, config/simulation_config.yaml).  Definition from
    SURVEY.md section 8(d): radar r of R at 400*(cos, sin)(2 pi r / R) m, theta_a = deg(2 pi r / R),
    other radar parameters alternating between the two shipped radar parameter sets, type_id = r mod 4,
    threat_level = 0.8 + 0.4 r / (R-1); jammer j of J at 70*(cos, sin)(2 pi j / J + 0.3) m with the shipped
    jammer parameters (power_max defaults to 100 W, power_min to 0); protected target at the origin."""
    import math
    tmpl = [
        dict(pt=300.0, gt=30, gr=30, wavelength=0.03, rcs=1.0, loss=10, latm=2, pn=3, theta_m=2.0, t_s=5.0,
             pulse_compression_gain=100.0, anti_jamming_factor=10.0),
        dict(pt=180.0, gt=25, gr=25, wavelength=0.03, rcs=1.0, loss=10, latm=2, pn=3, theta_m=1.8, t_s=4.0,
             pulse_compression_gain=120.0, anti_jamming_factor=15.0),
    ]
    radars = []
    for r in range(n_radars):
        ang = 2.0 * math.pi * r / n_radars
        p = dict(tmpl[r % 2])
        p.update(type_id=r % 4, position=[400.0 * math.cos(ang), 400.0 * math.sin(ang)],
                 theta_a=math.degrees(ang),
                 threat_level=0.8 + 0.4 * r / (n_radars - 1) if n_radars > 1 else 1.0)
        radars.append(p)
    jammers = []
    for j in range(n_jammers):
        ang = 2.0 * math.pi * j / n_jammers + 0.3
        jammers.append(dict(power=1000, gj=20, loss=5, latm=2, bj=10,
                            position=[70.0 * math.cos(ang), 70.0 * math.sin(ang)]))
    return dict(radars=radars, jammers=jammers, protected_target=dict(position=[0, 0], rcs=1.0),
                environment_params=dict(max_radar_types=4,
                                        rewards=dict(rd_min=-1.2, rd_max=-0.8, rp_min=-0.1, rp_max=-0.01)))
