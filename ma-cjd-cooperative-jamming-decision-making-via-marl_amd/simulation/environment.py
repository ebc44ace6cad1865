"""Radar-jamming environment on MI355X.

Two classes over the same HIP kernel (csrc/macjd_env.hip, C-ABI include/macjd.h):

``BatchedElectromagneticEnvironment``
    E environments advanced per launch; actions, state and outputs stay in device tensors, no host
    synchronisation inside ``step``.  This is the hot-path object (BASELINE.json: batch_envs=4096).

``ElectromagneticEnvironment``
    Drop-in for the reference class of the same name (reference simulation/environment.py:29-573):
    same constructor ``(config, sim_config_path)``, ``reset() -> np.float32[S]``,
    ``step(list[(T_i, P_i)]) -> (obs_list, reward, terminated, info)``, ``get_state/get_obs/
    get_agent_obs/get_avail_actions/get_env_info/close`` and the same exceptions.  It runs ONE
    environment through the same kernel and, like the reference, consumes its Monte-Carlo uniforms
    from the global ``np.random`` stream (R draws for the radars, then one per valid deception action,
    environment.py:341,430), so ``np.random.seed(s)`` reproduces the reference's trajectory.

There is no CPU implementation of ``step`` in the product: both classes require a HIP device and
libmacjd_hip.so and raise otherwise.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _native
from ..scenario import DEFAULT_SIM_CONFIG_PATH, Scenario, ScenarioBatch

RADAR_STATE_SEARCH = "SEARCH"  # core/radar.py:5-7
RADAR_STATE_TRACK = "TRACK"


def _require_device(device) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "macjd_amd environment step needs a HIP device (torch.cuda.is_available() is False); "
            "there is no CPU fallback for the environment kernel.")
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError(f"macjd_amd environment tensors must live on a HIP device, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class BatchedElectromagneticEnvironment:
    """E independent copies of the reference environment, one HIP launch per step.

    Layout in HBM (all caller-visible tensors are views of these):
      track   uint8 [R, E]   radar-major so that radar r's flags of 64 consecutive envs are one line
      step    int32 [E]
      reward  f32 [E]; r_dpj f32 [E, 3]; terminated uint8 [E]
      pd, snr f32 [R, E]     radar-major (exposed as [E, R] transposed views)
    Actions may come in any strided layout; agent-major storage ([J, E], i.e. ``T.t()`` of a
    ``[J, E]`` tensor) gives fully coalesced loads.
    """

    # The observation is a pure function of static scenario parameters (environment.py:479-510; nothing
    # in step() moves an entity), so runners may fill their obs / state / avail rows once.
    observation_is_static = True
    PE_TILE = 256            # envs per tile of the per-env tables (= the lane kernel's workgroup at streaming sizes)
    pe_tiled = True          # A/B hook: False keeps the plain [rows, E] SoA on the device

    def __init__(self, config: Any = None, sim_config_path: str = DEFAULT_SIM_CONFIG_PATH,
                 batch_envs: Optional[int] = None, device=None, seed: Optional[int] = None,
                 env_offset: int = 0, scenario: Optional[Scenario] = None, verbose: bool = False,
                 scenario_batch: Optional[ScenarioBatch] = None):
        """``scenario_batch``: one scenario PER ENV (radar / jammer positions, threat levels, powers differing
        between envs — SURVEY.md 8f-3); its tables live in HBM as an SoA that every step streams, and the
        observation becomes a per-env [E, S] tensor (still constant in time)."""
        self.scenario_batch = scenario_batch
        if scenario_batch is not None:
            scenario = scenario_batch.base
            if batch_envs is None:
                batch_envs = scenario_batch.n_envs
            if int(batch_envs) != scenario_batch.n_envs:
                raise ValueError(f"batch_envs ({batch_envs}) != scenarios in the batch ({scenario_batch.n_envs})")
        self.scenario = scenario if scenario is not None else Scenario.from_yaml(sim_config_path, config)
        sc = self.scenario
        self.num_jammers, self.num_radars = sc.num_jammers, sc.num_radars
        self.episode_limit = sc.episode_limit
        self.state_dim = self.obs_dim = self.agent_obs_dim = sc.state_dim
        self.action_dim_discrete = sc.n_actions
        self.action_dim_continuous = 1
        if batch_envs is None:
            batch_envs = getattr(config, "batch_envs", 1)
        self.batch_envs = int(batch_envs)
        if self.batch_envs < 1:
            raise ValueError(f"batch_envs must be >= 1, got {batch_envs}")
        self.seed = int(seed if seed is not None else getattr(config, "seed", 0) or 0)
        self.env_offset = int(env_offset)
        self.device = _require_device(device)
        self._lib = _native.load()
        with torch.cuda.device(self.device):
            self._handle = _native.ScenarioHandle(sc)
        E, R, J = self.batch_envs, self.num_radars, self.num_jammers
        dev = self.device
        self._track = torch.zeros((R, E), dtype=torch.uint8, device=dev)
        self._step = torch.zeros(E, dtype=torch.int32, device=dev)
        # per-env episode index: advanced by every reset (in the reset kernel, so replayed HIP graphs advance it too)
        # and part of the in-kernel Philox counter — each episode draws fresh Monte-Carlo values like the reference's
        # np.random stream does (environment.py:341,430)
        self._episode = torch.zeros(E, dtype=torch.int32, device=dev)
        self._reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self._r_dpj = torch.zeros((E, 3), dtype=torch.float32, device=dev)
        self._terminated = torch.zeros(E, dtype=torch.uint8, device=dev)
        self._pd = torch.zeros((R, E), dtype=torch.float32, device=dev)
        self._snr = torch.zeros((R, E), dtype=torch.float32, device=dev)
        self._state_vec = torch.from_numpy(sc.state_vector()).to(dev)
        self._avail = torch.ones((1, 1, sc.n_actions), dtype=torch.int32, device=dev)
        self._snr_no = torch.from_numpy(sc.tables["radar_snr_no"]).to(dev)
        self._pe_tables = self._pe_flags = self._state_vecs = None
        if scenario_batch is not None:
            # tiled (AoSoA) device form: [n_tiles, rows, PE_TILE] — a workgroup's 256 consecutive envs read one
            # contiguous block per step (the plain [rows, E] SoA is 6R+3J+JR streams E*8 bytes apart)
            self._pe_tile = self.PE_TILE if getattr(self, "pe_tiled", True) else 0
            tb, fl = scenario_batch.tables, scenario_batch.flags
            if self._pe_tile:
                w = self._pe_tile
                n_tiles = (E + w - 1) // w
                pad = n_tiles * w - E
                tile = lambda a: np.ascontiguousarray(
                    np.pad(a, ((0, 0), (0, pad))).reshape(a.shape[0], n_tiles, w).transpose(1, 0, 2))
                tb, fl = tile(tb), tile(fl)
            self._pe_tables = torch.from_numpy(tb).to(dev)       # f64 [rows, E] or [n_tiles, rows, w]
            self._pe_flags = torch.from_numpy(fl).to(dev)        # u8  [J*R, E] or [n_tiles, J*R, w]
            self._state_vecs = torch.from_numpy(scenario_batch.state_vectors).to(dev)   # f32 [E, S]
            self._snr_no = torch.from_numpy(scenario_batch.snr_no).to(dev)          # f64 [E, R]
        self._io = _native.StepIO()
        self.kernel_flags = 0  # A/B hook: _native.STEP_LANE_KERNEL / STEP_SLOT_KERNEL force one kernel variant
        if verbose:
            print(f"Batched environment: {E} envs x {J} jammers / {R} radars on {dev} (from {sc.source})")

    # ---- reference-shaped getters, broadcast views (never materialised per env) ----
    def get_state(self) -> torch.Tensor:
        """f32 [E, S] expanded view of the static state vector (environment.py:479-510)."""
        if self._state_vecs is not None:
            return self._state_vecs
        return self._state_vec.unsqueeze(0).expand(self.batch_envs, -1)

    def get_obs(self) -> torch.Tensor:
        """f32 [E, J, S] expanded view: every agent observes the global state (environment.py:512-522)."""
        if self._state_vecs is not None:
            return self._state_vecs.unsqueeze(1).expand(-1, self.num_jammers, -1)
        return self._state_vec.view(1, 1, -1).expand(self.batch_envs, self.num_jammers, -1)

    def get_avail_actions(self) -> torch.Tensor:
        """int32 [E, J, A] expanded view of ones (environment.py:539-551)."""
        return self._avail.expand(self.batch_envs, self.num_jammers, -1)

    def get_env_info(self) -> Dict[str, int]:
        return self.scenario.env_info()

    @property
    def track(self) -> torch.Tensor:
        """uint8 [E, R] view, 1 = TRACK."""
        return self._track.t()

    @property
    def step_count(self) -> torch.Tensor:
        return self._step

    @property
    def episode_index(self) -> torch.Tensor:
        """int32 [E]: number of resets each env has seen (keys the in-kernel Monte-Carlo stream)."""
        return self._episode

    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All (or the masked) envs back to SEARCH / step 0 (environment.py:208-219).  Returns the
        [E, S] state view."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = mask.data_ptr()
        with torch.cuda.device(self.device):
            _native.check(self._lib.macjd_env_reset(self._handle.ptr, self.batch_envs, self._track.data_ptr(),
                                                    1, self.batch_envs, self._step.data_ptr(), mptr,
                                                    self._episode.data_ptr(), stream),
                          "macjd_env_reset")
        return self.get_state()

    # ---- the hot call ----
    def _fill_io(self, T: torch.Tensor, P: torch.Tensor, u: Optional[torch.Tensor], arith_f64: bool,
                 reward: Optional[torch.Tensor], terminated: Optional[torch.Tensor], want_info: bool,
                 diag: Optional[Dict[str, torch.Tensor]], rdpj_sum: Optional[torch.Tensor] = None) -> _native.StepIO:
        E, R, J = self.batch_envs, self.num_radars, self.num_jammers
        if T.dim() == 3:
            T = T.squeeze(-1)
        if P.dim() == 3:
            P = P.squeeze(-1)
        if tuple(T.shape) != (E, J) or tuple(P.shape) != (E, J):
            raise ValueError(f"Received actions of shape {tuple(T.shape)} / {tuple(P.shape)}, but expected ({E}, {J})")
        if T.dtype != torch.int32:
            T = T.to(torch.int32)
        if P.dtype not in (torch.float32, torch.float64):
            P = P.to(torch.float32)
        if T.device != self.device or P.device != self.device:
            raise ValueError("actions must live on the environment's device")
        io = self._io
        io.n_envs, io.env_offset, io.seed = E, self.env_offset, self.seed
        io.flags = (_native.STEP_ARITH_F64 if arith_f64 else 0) | self.kernel_flags
        io.T, io.T_se, io.T_sx = T.data_ptr(), T.stride(0), T.stride(1)
        if P.dtype == torch.float32:
            io.P32, io.P64 = P.data_ptr(), None
        else:
            io.P32, io.P64 = None, P.data_ptr()
        io.P_se, io.P_sx = P.stride(0), P.stride(1)
        if u is not None:
            if tuple(u.shape) != (E, R + J) or u.dtype != torch.float64 or u.device != self.device:
                raise ValueError(f"uniforms must be float64 [{E}, {R + J}] on {self.device}")
            io.u, io.u_se, io.u_sx = u.data_ptr(), u.stride(0), u.stride(1)
        else:
            io.u, io.u_se, io.u_sx = None, 0, 0
        io.episode = self._episode.data_ptr()
        io.track, io.k_se, io.k_sx = self._track.data_ptr(), 1, E
        io.step = self._step.data_ptr()
        rew = self._reward if reward is None else reward
        ter = self._terminated if terminated is None else terminated
        for name, t, dt in (("reward", rew, torch.float32), ("terminated", ter, torch.uint8)):
            if t.dtype != dt and not (dt == torch.uint8 and t.dtype == torch.bool):
                raise ValueError(f"{name} output must be {dt}")
            if t.numel() != E or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} output must be a contiguous [{E}] tensor on {self.device}")
        io.reward, io.terminated = rew.data_ptr(), ter.data_ptr()
        io.r_dpj = self._r_dpj.data_ptr()
        if want_info:
            io.pd, io.pd_se, io.pd_sx = self._pd.data_ptr(), 1, E
            io.snr_with, io.sw_se, io.sw_sx = self._snr.data_ptr(), 1, E
        else:
            io.pd, io.pd_se, io.pd_sx = None, 0, 0
            io.snr_with, io.sw_se, io.sw_sx = None, 0, 0
        for k in ("out64", "pd64", "snr64", "prj64"):
            setattr(io, k, diag[k].data_ptr() if diag is not None else None)
        io.r_dpj_sum = rdpj_sum.data_ptr() if rdpj_sum is not None else None
        if self._pe_tables is not None:
            io.pe_tables, io.pe_flags, io.pe_stride = self._pe_tables.data_ptr(), self._pe_flags.data_ptr(), E
            io.pe_tile = self._pe_tile
        # keep converted tensors alive until the launch has been enqueued (same-stream ordering
        # keeps their storage valid for the kernel: the caching allocator is stream-ordered)
        self._keep = (T, P, u, rew, ter, diag)
        return io

    def step(self, actions_T: torch.Tensor, actions_P: torch.Tensor, uniforms: Optional[torch.Tensor] = None,
             *, arith_f64: bool = False, out_reward: Optional[torch.Tensor] = None,
             out_terminated: Optional[torch.Tensor] = None, want_info: bool = True,
             diag: Optional[Dict[str, torch.Tensor]] = None, rdpj_sum: Optional[torch.Tensor] = None
             ) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
        """One step of all E envs (batched environment.py:221-477).

        actions_T  int   [E, J] or [E, J, 1]  discrete action index T_i (int32 avoids a cast kernel)
        actions_P  f32/f64 [E, J] or [E, J, 1] normalised power P_i
        uniforms   f64 [E, R+J] or None (None -> Philox in-kernel, keyed by seed / env / step)

        Returns ``(reward f32[E], terminated bool[E], info)``; the tensors are views of buffers that
        the next ``step`` overwrites (pass ``out_reward`` / ``out_terminated`` to write elsewhere,
        e.g. straight into a replay row).  No host synchronisation happens here.
        """
        if rdpj_sum is not None and not (rdpj_sum.dtype == torch.float32 and rdpj_sum.is_contiguous()
                                         and tuple(rdpj_sum.shape) == (self.batch_envs, 3) and rdpj_sum.device == self.device):
            raise ValueError(f"rdpj_sum must be a contiguous float32 [{self.batch_envs}, 3] tensor on {self.device}")
        io = self._fill_io(actions_T, actions_P, uniforms, arith_f64, out_reward, out_terminated, want_info, diag,
                           rdpj_sum)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(self._lib.macjd_env_step(self._handle.ptr, ctypes.byref(io), stream), "macjd_env_step")
        rew = self._reward if out_reward is None else out_reward
        ter = self._terminated if out_terminated is None else out_terminated
        info = {"r_d": self._r_dpj[:, 0], "r_p": self._r_dpj[:, 1], "r_j": self._r_dpj[:, 2],
                "radar_tracking": self._track.t(), "step_count": self._step}
        if want_info:
            info["radar_pds"] = self._pd.t()
            info["snr_with_jamming"] = self._snr.t()
            info["snr_no_jamming"] = self._snr_no
        return rew, (ter.view(torch.bool) if ter.dtype == torch.uint8 else ter), info

    def step_many(self, actions_T: torch.Tensor, actions_P: torch.Tensor, out_reward: torch.Tensor,
                  out_terminated: torch.Tensor, out_rdpj: torch.Tensor, rdpj_sum: Optional[torch.Tensor] = None) -> None:
        """All n steps of an episode batch in ONE launch (+ a tiny one that advances the counters), given the actions of
        all steps: ``actions_T`` int32 / ``actions_P`` float32 [n, E, J(,1)] contiguous (the runner's staging tensors);
        ``out_reward`` float32 [n, E(,1)], ``out_terminated`` bool / uint8 [n, E(,1)], ``out_rdpj`` float32 [n, E, 3] —
        all contiguous.  Legal because a step's outcome depends on the env's past only through the step counter (the
        FSM's next state equals `detected`, core/radar.py:102-117) — see include/macjd.h, macjd_env_step_many; meant for
        actions that do not depend on the env's outputs (static observation).  Philox uniforms, no pd / snr outputs."""
        E, J = self.batch_envs, self.num_jammers
        n = actions_T.shape[0]
        for name, t, dt in (("actions_T", actions_T, torch.int32), ("actions_P", actions_P, torch.float32),
                            ("out_reward", out_reward, torch.float32), ("out_rdpj", out_rdpj, torch.float32)):
            if t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} must be a contiguous {dt} tensor on {self.device}")
        if actions_T.numel() != n * E * J or actions_P.numel() != n * E * J or out_reward.numel() != n * E \
                or out_terminated.numel() != n * E or out_rdpj.numel() != n * E * 3 or not out_terminated.is_contiguous():
            raise ValueError("step_many: tensor sizes do not match [n, E, J] / [n, E] / [n, E, 3]")
        if out_terminated.dtype not in (torch.uint8, torch.bool):
            raise ValueError("out_terminated must be uint8 or bool")
        io = self._fill_io(actions_T[0].view(E, J), actions_P[0].view(E, J), None, False, out_reward.view(-1)[:E],
                           out_terminated.view(-1)[:E], False, None, rdpj_sum)
        io.r_dpj = out_rdpj.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(self._lib.macjd_env_step_many(self._handle.ptr, ctypes.byref(io), int(n), int(E * J), stream),
                          "macjd_env_step_many")
        io.r_dpj = self._r_dpj.data_ptr()
        self._keep = (actions_T, actions_P, out_reward, out_terminated, out_rdpj)

    def time_step_many_kernel(self, actions_T, actions_P, out_reward, out_terminated, out_rdpj, iters: int) -> float:
        """bench.py helper: average milliseconds per launch of the many-step kernel (n x E work items), ``iters``
        launches replayed from a HIP graph and bracketed by HIP events on the replay stream (macjd_env_step_many_timed)."""
        E, J = self.batch_envs, self.num_jammers
        n = actions_T.shape[0]
        io = self._fill_io(actions_T[0].view(E, J), actions_P[0].view(E, J), None, False, out_reward.view(-1)[:E],
                           out_terminated.view(-1)[:E], False, None, None)
        io.r_dpj = out_rdpj.data_ptr()
        ms = ctypes.c_float(0.0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(self._lib.macjd_env_step_many_timed(self._handle.ptr, ctypes.byref(io), int(n), int(E * J), int(iters),
                                                              stream, ctypes.byref(ms)), "macjd_env_step_many_timed")
        io.r_dpj = self._r_dpj.data_ptr()
        return float(ms.value)

    def time_step_kernel(self, actions_T: torch.Tensor, actions_P: torch.Tensor, iters: int,
                         uniforms: Optional[torch.Tensor] = None, want_info: bool = True) -> float:
        """bench.py helper: average milliseconds per env_step launch over ``iters`` back-to-back
        launches, measured with HIP events recorded on the launch stream (macjd_env_step_timed)."""
        io = self._fill_io(actions_T, actions_P, uniforms, False, None, None, want_info, None)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ms = ctypes.c_float(0.0)
        with torch.cuda.device(self.device):
            _native.check(self._lib.macjd_env_step_timed(self._handle.ptr, ctypes.byref(io), int(iters), stream,
                                                         ctypes.byref(ms)), "macjd_env_step_timed")
        return float(ms.value)

    @property
    def scenario_regular(self) -> bool:
        """The shared scenario's tables are regular (include/macjd.h, macjd_scenario_is_regular): the production lane
        kernel runs its short-division form.  Per-env scenario batches have no shared handle: False."""
        h = getattr(self, "_handle", None)
        return bool(h is not None and self._lib.macjd_scenario_is_regular(h.ptr) == 1)

    def close(self) -> None:
        self._handle.close()


class ElectromagneticEnvironment:
    """Single-environment drop-in for the reference class (simulation/environment.py:29-573)."""

    def __init__(self, config, sim_config_path: str = DEFAULT_SIM_CONFIG_PATH):
        self._batched = BatchedElectromagneticEnvironment(config, sim_config_path, batch_envs=1,
                                                          device=getattr(config, "env_device", None))
        sc = self.scenario = self._batched.scenario
        self.sim_config_path = sim_config_path
        self.num_jammers, self.num_radars = sc.num_jammers, sc.num_radars
        self.episode_limit = sc.episode_limit
        self.max_radar_types = sc.max_radar_types
        self.rd_min_penalty, self.rd_max_penalty = sc.rd_min, sc.rd_max
        self.rp_min_penalty, self.rp_max_penalty = sc.rp_min, sc.rp_max
        self.action_dim_discrete = sc.n_actions
        self.action_dim_continuous = 1
        self.state_dim = self.obs_dim = self.agent_obs_dim = sc.state_dim
        self.protected_target_config = {"position": sc.target_position, "rcs": sc.target_rcs}
        self._step_count = 0
        self._last_actions = np.zeros((self.num_jammers, 2))
        self._state = sc.state_vector()
        dev = self._batched.device
        R, J = self.num_radars, self.num_jammers
        self._diag = {"out64": torch.zeros((1, 4), dtype=torch.float64, device=dev),
                      "pd64": torch.zeros((1, R), dtype=torch.float64, device=dev),
                      "snr64": torch.zeros((1, R), dtype=torch.float64, device=dev),
                      "prj64": torch.zeros((1, J), dtype=torch.float64, device=dev)}
        # same one-time summary the reference prints (environment.py:115-121)
        rfd = sc.radar_feature_dim
        print(f"Environment Initialized: {J} Jammers, {R} Radars (from {sim_config_path})")
        print(f"State Dimension: {self.state_dim} (Radar features: {rfd}, Jammer features: 2, "
              f"Max Radar Types: {self.max_radar_types})")
        print(f"Observation Dimension (per agent): {self.agent_obs_dim}")
        print(f"Action Dimension (Discrete): {self.action_dim_discrete}")
        print(f"Protected Target: Pos={sc.target_position}, RCS={sc.target_rcs}")
        print(f"Episode Limit: {self.episode_limit}")
        print(f"Reward Params: rd=[{sc.rd_min}, {sc.rd_max}], rp=[{sc.rp_min}, {sc.rp_max}]")

    def reset(self) -> np.ndarray:
        self._batched.reset()
        self._step_count = 0
        self._last_actions.fill(0)
        return self.get_state()

    def _host_actions(self, actions: Sequence[Tuple[Any, Any]]):
        """Decode on the host only what decides HOW MANY uniforms the reference would draw, using the
        same NumPy expressions on the caller's own scalar types (environment.py:249-284)."""
        sc = self.scenario
        R, J = self.num_radars, self.num_jammers
        T = np.zeros(J, dtype=np.int32)
        all_f32 = all(isinstance(a[1], np.float32) for a in actions)
        P = np.zeros(J, dtype=np.float32 if all_f32 else np.float64)
        powers: List[Any] = []
        n_dec = 0
        denom = sc.tables["jr_denom"].reshape(J, R)
        for i, (t_i, p_i) in enumerate(actions):
            t_i = int(t_i)
            p_c = np.clip(p_i, 0.0, 1.0)
            pmin, pmax = sc.jammers[i]["power_min"], sc.jammers[i]["power_max"]
            actual = pmin + p_c * (pmax - pmin)
            powers.append(actual)
            T[i] = max(min(t_i, 2 ** 31 - 1), -(2 ** 31))
            P[i] = p_i
            if 1 <= t_i <= 2 * R:
                tgt = (t_i + 1) // 2 - 1
                if actual > 0 and denom[i, tgt] >= 0.0 and t_i % 2 == 0:
                    n_dec += 1
            elif t_i > 0:
                print(f"Warning: Jammer {i} chose invalid discrete action T_i={t_i}")  # environment.py:268
        return T, P, all_f32, powers, n_dec

    def step(self, actions):
        if len(actions) != self.num_jammers:  # environment.py:232-233
            raise ValueError(f"Received {len(actions)} actions, but expected {self.num_jammers}")
        self._step_count += 1
        sc, b = self.scenario, self._batched
        R, J = self.num_radars, self.num_jammers
        T, P, all_f32, powers, n_dec = self._host_actions(actions)
        u = np.full(R + J, 2.0)  # slots the reference would not draw can never compare as hits
        for k in range(R + n_dec):  # same consumption of the global stream as environment.py:341,430
            u[k] = np.random.rand()
        dev = b.device
        T_d = torch.from_numpy(T).to(dev).view(1, J)
        P_d = torch.from_numpy(P).to(dev).view(1, J)
        u_d = torch.from_numpy(u).to(dev).view(1, R + J)
        _, term, _ = b.step(T_d, P_d, u_d, diag=self._diag)
        out = self._diag["out64"].cpu().numpy()[0]
        pd = self._diag["pd64"].cpu().numpy()[0].copy()
        snr = self._diag["snr64"].cpu().numpy()[0].copy()
        prj = self._diag["prj64"].cpu().numpy()[0]
        tracking = b.track.cpu().numpy()[0].astype(bool)
        terminated = bool(self._step_count >= self.episode_limit)
        self._last_actions = np.array([[T[i], np.clip(actions[i][1], 0.0, 1.0)] for i in range(J)], dtype=np.float64)

        jammer_actions = []
        for i in range(J):
            if prj[i] >= 0.0:  # recorded in the reference's jammer_actions_details (environment.py:288-295)
                jammer_actions.append({"jammer_idx": i, "target_idx": (int(T[i]) + 1) // 2 - 1, "type": int(T[i]) % 2,
                                       "power": powers[i], "received_power": np.float64(prj[i])})
        radar_states = []
        for r in range(R):
            radar_states.append({  # core/radar.py:121-130
                "state": RADAR_STATE_TRACK if tracking[r] else RADAR_STATE_SEARCH,
                "threat": sc.radars[r]["threat_level"],
                "is_tracking": bool(tracking[r]),
                "locked_target": self.protected_target_config if tracking[r] else None,
            })
        info = {
            "radar_pds": pd,
            "radar_states": radar_states,
            "snr_no_jamming": sc.tables["radar_snr_no"].copy(),
            "snr_with_jamming": snr,
            "r_d": out[1], "r_p": out[2], "r_j": out[3],
            "jammer_actions": jammer_actions,
        }
        return self.get_obs(), np.float64(out[0]), terminated, info

    def get_state(self) -> np.ndarray:
        return self._state.copy()

    def get_obs(self) -> List[np.ndarray]:
        s = self.get_state()
        return [s for _ in range(self.num_jammers)]

    def get_agent_obs(self, agent_id: int) -> np.ndarray:
        if not (0 <= agent_id < self.num_jammers):  # environment.py:535-536
            raise ValueError(f"Invalid agent_id {agent_id} for {self.num_jammers} jammers.")
        return self.get_state()

    def get_avail_actions(self) -> List[np.ndarray]:
        return [np.ones(self.action_dim_discrete, dtype=np.int32) for _ in range(self.num_jammers)]

    def get_env_info(self) -> Dict[str, int]:
        return self.scenario.env_info()

    def close(self) -> None:
        print("Closing Electromagnetic Environment.")
        self._batched.close()
