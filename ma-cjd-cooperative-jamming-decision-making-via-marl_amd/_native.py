"""ctypes binding of libmacjd_hip.so (C-ABI in include/macjd.h).

There is deliberately NO fallback: if the HIP library is missing or no HIP device is present, every
entry point raises.  The CPU restatement under oracle/ is test infrastructure and is never imported
from here."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmacjd_hip.so"
# MACJD_LIB points at another build of the same sources (kernel A/B runs); the default is the in-tree library
LIB_PATH = os.environ.get("MACJD_LIB") or os.path.join(_PKG_DIR, LIB_NAME)

ABI_VERSION = 3
STEP_ARITH_F64 = 1
STEP_LANE_KERNEL = 2
STEP_SLOT_KERNEL = 4

EXPORTS = [
    "macjd_abi_version", "macjd_last_error", "macjd_reload_options", "macjd_device_count",
    "macjd_scenario_create", "macjd_scenario_destroy", "macjd_scenario_dims", "macjd_scenario_is_regular",
    "macjd_env_reset", "macjd_env_step", "macjd_env_step_timed",
    "macjd_qhead_select", "macjd_gru_sequence", "macjd_mixer_tail_forward", "macjd_mixer_tail_backward",
    "macjd_mlp_forward", "macjd_mlp_forward_pair", "macjd_td_loss", "macjd_clip_adam_step", "macjd_clip_adam_step_sample", "macjd_clip_adam_step_ln", "macjd_sample_episodes", "macjd_gather_rows",
    "macjd_linear_wgrad", "macjd_linear_wgrad_workspace_floats", "macjd_linear_wgrad_many", "macjd_qhead_input", "macjd_layernorm_forward", "macjd_layernorm_param_grad", "macjd_gru_gates", "macjd_rowdot", "macjd_splitrelu_backward",
    "macjd_mixer_fused_supported", "macjd_mixer_fused_forward", "macjd_mixer_fused_backward", "macjd_mixer_fused_backward_td", "macjd_td_mask_sum",
    "macjd_agent_episode_supported", "macjd_agent_episode", "macjd_env_step_many", "macjd_env_step_many_timed",
    "macjd_qhead_double_q_supported", "macjd_qhead_double_q", "macjd_qhead_taken_supported", "macjd_qhead_taken",
    "macjd_qheads_pair", "macjd_mixer_fused_forward_pair",
]


class NativeLibraryError(RuntimeError):
    pass


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


class QheadIO(ctypes.Structure):
    """ctypes mirror of ``macjd_qhead_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64),
        ("H", ctypes.c_int32), ("A", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("greedy_only", ctypes.c_int32),
        ("base", ctypes.c_void_p), ("base_ld", ctypes.c_int64),
        ("P_all", ctypes.c_void_p), ("p_ld", ctypes.c_int64),
        ("W1", ctypes.c_void_p), ("w1_ld", ctypes.c_int64),
        ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
        ("Q", ctypes.c_void_p), ("q_ld", ctypes.c_int64),
        ("avail", ctypes.c_void_p), ("avail_elem_size", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("av_se", ctypes.c_int64), ("av_sj", ctypes.c_int64), ("av_sa", ctypes.c_int64),
        ("epsilon", ctypes.c_float), ("reserved2", ctypes.c_float),
        ("seed", ctypes.c_uint64), ("counter", ctypes.c_uint64),
        ("eps_dev", ctypes.c_void_p), ("counter_dev", ctypes.c_void_p),
        ("T_out32", ctypes.c_void_p), ("T_out64", ctypes.c_void_p),
        ("t32_se", ctypes.c_int64), ("t32_sj", ctypes.c_int64), ("t64_se", ctypes.c_int64), ("t64_sj", ctypes.c_int64),
        ("P_out", ctypes.c_void_p), ("po_se", ctypes.c_int64), ("po_sj", ctypes.c_int64),
        ("argmax_out", ctypes.c_void_p), ("gather_idx", ctypes.c_void_p), ("q_gather_out", ctypes.c_void_p),
    ]


class GruIO(ctypes.Structure):
    """ctypes mirror of ``macjd_gru_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_nets", ctypes.c_int32), ("B", ctypes.c_int32), ("T", ctypes.c_int32), ("J", ctypes.c_int32),
        ("H", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("gi", ctypes.c_void_p * 2), ("w_hh", ctypes.c_void_p * 2), ("b_hh", ctypes.c_void_p * 2),
        ("h0", ctypes.c_void_p * 2), ("h_out", ctypes.c_void_p * 2), ("h0_sb", ctypes.c_int64 * 2),
        ("obs", ctypes.c_void_p), ("obs_sb", ctypes.c_int64), ("obs_sj", ctypes.c_int64), ("S", ctypes.c_int32), ("reserved2", ctypes.c_int32),
        ("obs_index", ctypes.c_void_p),
        ("fc1_w", ctypes.c_void_p * 2), ("fc1_b", ctypes.c_void_p * 2), ("w_ih", ctypes.c_void_p * 2), ("b_ih", ctypes.c_void_p * 2),
        ("p_out", ctypes.c_void_p * 2), ("act_w", (ctypes.c_void_p * 3) * 2), ("act_b", (ctypes.c_void_p * 3) * 2),
        ("Ah", ctypes.c_int32), ("A", ctypes.c_int32),
    ]


class MixerIO(ctypes.Structure):
    """ctypes mirror of ``macjd_mixer_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("M", ctypes.c_int64), ("J", ctypes.c_int32), ("Em", ctypes.c_int32),
        ("q", ctypes.c_void_p), ("w1_raw", ctypes.c_void_p), ("b1_raw", ctypes.c_void_p),
        ("wf_raw", ctypes.c_void_p), ("v_raw", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("gy", ctypes.c_void_p), ("gq", ctypes.c_void_p), ("gw1_raw", ctypes.c_void_p),
        ("gb1_raw", ctypes.c_void_p), ("gwf_raw", ctypes.c_void_p), ("gv_raw", ctypes.c_void_p),
        ("b1_ld", ctypes.c_int64),
    ]


class MixerFusedIO(ctypes.Structure):
    """ctypes mirror of ``macjd_mixerf_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("M", ctypes.c_int64), ("J", ctypes.c_int32), ("S", ctypes.c_int32), ("Hh", ctypes.c_int32), ("Em", ctypes.c_int32),
        ("save", ctypes.c_int32), ("reserved", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("reserved_f", ctypes.c_float),
        ("s", ctypes.c_void_p), ("s_ld", ctypes.c_int64), ("q", ctypes.c_void_p),
        ("ln_w", ctypes.c_void_p), ("ln_b", ctypes.c_void_p), ("W1", ctypes.c_void_p), ("b1", ctypes.c_void_p),
        ("W2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("Wf2", ctypes.c_void_p), ("bf2", ctypes.c_void_p),
        ("wV2", ctypes.c_void_p), ("bV2", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("sn", ctypes.c_void_p), ("xhat", ctypes.c_void_p), ("act", ctypes.c_void_p),
        ("gy", ctypes.c_void_p), ("gq", ctypes.c_void_p), ("gout1", ctypes.c_void_p), ("g_w1raw", ctypes.c_void_p),
        ("g_wfraw", ctypes.c_void_p), ("g_v", ctypes.c_void_p),
    ]


class AgentEpisodeIO(ctypes.Structure):
    """ctypes mirror of ``macjd_agent_episode_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_envs", ctypes.c_int64), ("T", ctypes.c_int32), ("J", ctypes.c_int32), ("H", ctypes.c_int32), ("A", ctypes.c_int32),
        ("greedy_only", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("gi", ctypes.c_void_p), ("gi_ld", ctypes.c_int64), ("P_all", ctypes.c_void_p), ("p_ld", ctypes.c_int64),
        ("h0", ctypes.c_void_p), ("w_hh", ctypes.c_void_p), ("b_hh", ctypes.c_void_p),
        ("W1", ctypes.c_void_p), ("w1_ld", ctypes.c_int64), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
        ("avail", ctypes.c_void_p), ("avail_elem_size", ctypes.c_int32), ("reserved2", ctypes.c_int32),
        ("av_se", ctypes.c_int64), ("av_sj", ctypes.c_int64), ("av_sa", ctypes.c_int64),
        ("eps", ctypes.c_void_p), ("seed", ctypes.c_uint64), ("counter_base", ctypes.c_void_p),
        ("hidden", ctypes.c_void_p), ("T_out", ctypes.c_void_p), ("P_out", ctypes.c_void_p), ("h_final", ctypes.c_void_p),
    ]


class DoubleQIO(ctypes.Structure):
    """ctypes mirror of ``macjd_doubleq_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("H", ctypes.c_int32), ("A", ctypes.c_int32),
        ("h_e", ctypes.c_void_p), ("he_ld", ctypes.c_int64), ("h_t", ctypes.c_void_p), ("ht_ld", ctypes.c_int64),
        ("P_e", ctypes.c_void_p), ("pe_ld", ctypes.c_int64), ("P_t", ctypes.c_void_p), ("pt_ld", ctypes.c_int64),
        ("W1_e", ctypes.c_void_p), ("w1e_ld", ctypes.c_int64), ("b1_e", ctypes.c_void_p), ("w2_e", ctypes.c_void_p), ("b2_e", ctypes.c_void_p),
        ("W1_t", ctypes.c_void_p), ("w1t_ld", ctypes.c_int64), ("b1_t", ctypes.c_void_p), ("w2_t", ctypes.c_void_p), ("b2_t", ctypes.c_void_p),
        ("out", ctypes.c_void_p), ("argmax_out", ctypes.c_void_p), ("p_group", ctypes.c_int64), ("p_inner", ctypes.c_int64),
    ]


class QinputIO(ctypes.Structure):
    """ctypes mirror of ``macjd_qinput_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("H", ctypes.c_int32), ("A", ctypes.c_int32),
        ("h", ctypes.c_void_p), ("h_ld", ctypes.c_int64),
        ("idx", ctypes.c_void_p), ("idx_elem_size", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("P", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_ld", ctypes.c_int64),
    ]


class GruGatesIO(ctypes.Structure):
    """ctypes mirror of ``macjd_grugates_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("H", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("gi", ctypes.c_void_p), ("gi_ld", ctypes.c_int64), ("gh", ctypes.c_void_p), ("gh_ld", ctypes.c_int64),
        ("h", ctypes.c_void_p), ("h_ld", ctypes.c_int64),
        ("h_out", ctypes.c_void_p), ("ho_ld", ctypes.c_int64), ("h_out2", ctypes.c_void_p), ("ho2_ld", ctypes.c_int64),
    ]


class RowdotIO(ctypes.Structure):
    """ctypes mirror of ``macjd_rowdot_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("K", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("x_ld", ctypes.c_int64), ("w", ctypes.c_void_p), ("b", ctypes.c_void_p),
        ("y", ctypes.c_void_p),
    ]


class SplitReluBwdIO(ctypes.Structure):
    """ctypes mirror of ``macjd_splitrelu_bwd_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("M", ctypes.c_int64), ("n_blocks", ctypes.c_int32), ("Cp", ctypes.c_int32), ("width", ctypes.c_int32 * 4),
        ("g0", ctypes.c_void_p), ("g1", ctypes.c_void_p), ("g2", ctypes.c_void_p), ("g3", ctypes.c_void_p),
        ("g_ld", ctypes.c_int64 * 4),
        ("g_pass", ctypes.c_void_p), ("gp_ld", ctypes.c_int64),
        ("act", ctypes.c_void_p), ("act_ld", ctypes.c_int64), ("gout", ctypes.c_void_p), ("gout_ld", ctypes.c_int64),
        ("ow0", ctypes.c_void_p), ("ow1", ctypes.c_void_p), ("ow2", ctypes.c_void_p), ("ow3", ctypes.c_void_p),
    ]


class LayerNormIO(ctypes.Structure):
    """ctypes mirror of ``macjd_layernorm_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("M", ctypes.c_int64), ("S", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("eps", ctypes.c_float), ("reserved2", ctypes.c_float),
        ("x", ctypes.c_void_p), ("x_ld", ctypes.c_int64), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("y_ld", ctypes.c_int64), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
        ("xhat", ctypes.c_void_p), ("xhat_ld", ctypes.c_int64),
    ]


class LnParamIO(ctypes.Structure):
    """ctypes mirror of ``macjd_lnparam_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("C", ctypes.c_int32), ("K", ctypes.c_int32),
        ("W", ctypes.c_void_p), ("w_ld", ctypes.c_int64), ("G", ctypes.c_void_p), ("g_ld", ctypes.c_int64),
        ("gb", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
    ]


class MlpIO(ctypes.Structure):
    """ctypes mirror of ``macjd_mlp_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("n_layers", ctypes.c_int32), ("dims", ctypes.c_int32 * 4),
        ("act", ctypes.c_int32 * 3), ("W", ctypes.c_void_p * 3), ("b", ctypes.c_void_p * 3),
        ("x", ctypes.c_void_p), ("x_ld", ctypes.c_int64), ("y", ctypes.c_void_p), ("y_ld", ctypes.c_int64),
    ]


class TdLossIO(ctypes.Structure):
    """ctypes mirror of ``macjd_tdloss_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("Tm1", ctypes.c_int32), ("gamma", ctypes.c_float), ("reserved", ctypes.c_float),
        ("y", ctypes.c_void_p), ("tq", ctypes.c_void_p),
        ("y_sb", ctypes.c_int64), ("tq_sb", ctypes.c_int64), ("gy_sb", ctypes.c_int64), ("gy_cols", ctypes.c_int64),
        ("reward", ctypes.c_void_p), ("r_sb", ctypes.c_int64), ("r_st", ctypes.c_int64),
        ("terminated", ctypes.c_void_p), ("t_sb", ctypes.c_int64), ("t_st", ctypes.c_int64),
        ("filled", ctypes.c_void_p), ("f_sb", ctypes.c_int64), ("f_st", ctypes.c_int64),
        ("stats", ctypes.c_void_p), ("gy", ctypes.c_void_p),
    ]


class AdamIO(ctypes.Structure):
    """ctypes mirror of ``macjd_adam_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n", ctypes.c_int64), ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
        ("eps", ctypes.c_float), ("max_norm", ctypes.c_float), ("reserved", ctypes.c_float),
        ("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
        ("exp_avg_sq", ctypes.c_void_p), ("step", ctypes.c_void_p), ("grad_norm", ctypes.c_void_p),
        ("partials", ctypes.c_void_p),
    ]


class QtakenIO(ctypes.Structure):
    """ctypes mirror of ``macjd_qtaken_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("H", ctypes.c_int32), ("A", ctypes.c_int32),
        ("h", ctypes.c_void_p), ("h_ld", ctypes.c_int64),
        ("idx", ctypes.c_void_p), ("idx_elem_size", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("P", ctypes.c_void_p),
        ("W1", ctypes.c_void_p), ("w1_ld", ctypes.c_int64),
        ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
        ("x", ctypes.c_void_p), ("x_ld", ctypes.c_int64),
        ("act", ctypes.c_void_p), ("act_ld", ctypes.c_int64),
        ("q", ctypes.c_void_p),
    ]


class SamplerIO(ctypes.Structure):
    """ctypes mirror of ``macjd_sampler_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("idx_out", ctypes.c_void_p), ("n", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("n_stored", ctypes.c_void_p), ("counter", ctypes.c_void_p), ("seed", ctypes.c_uint64),
    ]


class GatherIO(ctypes.Structure):
    """ctypes mirror of ``macjd_gather_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("n_tensors", ctypes.c_int32), ("n_rows", ctypes.c_int32), ("idx", ctypes.c_void_p),
        ("src", ctypes.c_void_p * 8), ("dst", ctypes.c_void_p * 8), ("row_bytes", ctypes.c_int64 * 8),
        ("dst_row_bytes", ctypes.c_int64 * 8),
    ]


class WgradIO(ctypes.Structure):
    """ctypes mirror of ``macjd_wgrad_io`` (include/macjd_nets.h)."""
    _fields_ = [
        ("K", ctypes.c_int64), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
        ("gout", ctypes.c_void_p), ("gout_ld", ctypes.c_int64), ("inp", ctypes.c_void_p), ("inp_ld", ctypes.c_int64),
        ("dW", ctypes.c_void_p), ("dw_ld", ctypes.c_int64), ("db", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("outer_vec", ctypes.c_void_p), ("outer_w", ctypes.c_void_p),
    ]


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load libmacjd_hip.so from the package directory; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the environment step.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    lib.macjd_abi_version.restype = ctypes.c_int
    lib.macjd_last_error.restype = ctypes.c_char_p
    lib.macjd_device_count.restype = ctypes.c_int
    lib.macjd_reload_options.restype = None
    lib.macjd_scenario_create.restype = ctypes.c_int
    lib.macjd_scenario_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.macjd_scenario_destroy.restype = None
    lib.macjd_scenario_destroy.argtypes = [ctypes.c_void_p]
    lib.macjd_scenario_is_regular.restype = ctypes.c_int
    lib.macjd_scenario_is_regular.argtypes = [ctypes.c_void_p]
    lib.macjd_scenario_dims.restype = ctypes.c_int
    lib.macjd_scenario_dims.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 3
    lib.macjd_env_reset.restype = ctypes.c_int
    lib.macjd_env_reset.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.macjd_env_step.restype = ctypes.c_int
    lib.macjd_env_step.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_void_p]
    lib.macjd_env_step_timed.restype = ctypes.c_int
    lib.macjd_env_step_timed.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_int, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_float)]
    lib.macjd_qhead_select.restype = ctypes.c_int
    lib.macjd_qhead_select.argtypes = [ctypes.POINTER(QheadIO), ctypes.c_void_p]
    lib.macjd_gru_sequence.restype = ctypes.c_int
    lib.macjd_gru_sequence.argtypes = [ctypes.POINTER(GruIO), ctypes.c_void_p]
    for name in ("macjd_mixer_tail_forward", "macjd_mixer_tail_backward"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [ctypes.POINTER(MixerIO), ctypes.c_void_p]
    lib.macjd_qhead_double_q_supported.restype = ctypes.c_int
    lib.macjd_qhead_double_q_supported.argtypes = [ctypes.c_int32] * 2
    lib.macjd_qhead_double_q.restype = ctypes.c_int
    lib.macjd_qhead_double_q.argtypes = [ctypes.POINTER(DoubleQIO), ctypes.c_void_p]
    lib.macjd_agent_episode_supported.restype = ctypes.c_int
    lib.macjd_agent_episode_supported.argtypes = [ctypes.c_int32] * 3
    lib.macjd_agent_episode.restype = ctypes.c_int
    lib.macjd_agent_episode.argtypes = [ctypes.POINTER(AgentEpisodeIO), ctypes.c_void_p]
    lib.macjd_env_step_many.restype = ctypes.c_int
    lib.macjd_env_step_many.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    lib.macjd_env_step_many_timed.restype = ctypes.c_int
    lib.macjd_env_step_many_timed.argtypes = [ctypes.c_void_p, ctypes.POINTER(StepIO), ctypes.c_int32, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.macjd_mixer_fused_supported.restype = ctypes.c_int
    lib.macjd_mixer_fused_supported.argtypes = [ctypes.c_int32] * 4
    for name in ("macjd_mixer_fused_forward", "macjd_mixer_fused_backward"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [ctypes.POINTER(MixerFusedIO), ctypes.c_void_p]
    lib.macjd_mixer_fused_backward_td.restype = ctypes.c_int
    lib.macjd_mixer_fused_backward_td.argtypes = [ctypes.POINTER(MixerFusedIO), ctypes.POINTER(TdLossIO), ctypes.c_void_p, ctypes.c_void_p]
    lib.macjd_td_mask_sum.restype = ctypes.c_int
    lib.macjd_td_mask_sum.argtypes = [ctypes.POINTER(TdLossIO), ctypes.c_void_p, ctypes.c_void_p]
    lib.macjd_qhead_taken_supported.restype = ctypes.c_int
    lib.macjd_qhead_taken_supported.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.macjd_qhead_taken.restype = ctypes.c_int
    lib.macjd_qhead_taken.argtypes = [ctypes.POINTER(QtakenIO), ctypes.c_void_p]
    lib.macjd_qheads_pair.restype = ctypes.c_int
    lib.macjd_qheads_pair.argtypes = [ctypes.POINTER(QtakenIO), ctypes.POINTER(DoubleQIO), ctypes.c_void_p]
    lib.macjd_mixer_fused_forward_pair.restype = ctypes.c_int
    lib.macjd_mixer_fused_forward_pair.argtypes = [ctypes.POINTER(MixerFusedIO), ctypes.POINTER(MixerFusedIO), ctypes.c_void_p]
    lib.macjd_qhead_input.restype = ctypes.c_int
    lib.macjd_qhead_input.argtypes = [ctypes.POINTER(QinputIO), ctypes.c_void_p]
    lib.macjd_layernorm_forward.restype = ctypes.c_int
    lib.macjd_layernorm_forward.argtypes = [ctypes.POINTER(LayerNormIO), ctypes.c_void_p]
    lib.macjd_gru_gates.restype = ctypes.c_int
    lib.macjd_gru_gates.argtypes = [ctypes.POINTER(GruGatesIO), ctypes.c_void_p]
    lib.macjd_splitrelu_backward.restype = ctypes.c_int
    lib.macjd_splitrelu_backward.argtypes = [ctypes.POINTER(SplitReluBwdIO), ctypes.c_void_p]
    lib.macjd_layernorm_param_grad.restype = ctypes.c_int
    lib.macjd_layernorm_param_grad.argtypes = [ctypes.POINTER(LnParamIO), ctypes.c_void_p]
    lib.macjd_rowdot.restype = ctypes.c_int
    lib.macjd_rowdot.argtypes = [ctypes.POINTER(RowdotIO), ctypes.c_void_p]
    lib.macjd_mlp_forward_pair.restype = ctypes.c_int
    lib.macjd_mlp_forward_pair.argtypes = [ctypes.POINTER(MlpIO), ctypes.POINTER(MlpIO), ctypes.c_void_p]
    lib.macjd_mlp_forward.restype = ctypes.c_int
    lib.macjd_mlp_forward.argtypes = [ctypes.POINTER(MlpIO), ctypes.c_void_p]
    lib.macjd_td_loss.restype = ctypes.c_int
    lib.macjd_td_loss.argtypes = [ctypes.POINTER(TdLossIO), ctypes.c_void_p]
    lib.macjd_clip_adam_step.restype = ctypes.c_int
    lib.macjd_clip_adam_step.argtypes = [ctypes.POINTER(AdamIO), ctypes.c_void_p]
    lib.macjd_clip_adam_step_sample.restype = ctypes.c_int
    lib.macjd_clip_adam_step_sample.argtypes = [ctypes.POINTER(AdamIO), ctypes.POINTER(SamplerIO), ctypes.c_void_p]
    lib.macjd_clip_adam_step_ln.restype = ctypes.c_int
    lib.macjd_clip_adam_step_ln.argtypes = [ctypes.POINTER(AdamIO), ctypes.POINTER(SamplerIO), ctypes.POINTER(LnParamIO),
                                            ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    lib.macjd_sample_episodes.restype = ctypes.c_int
    lib.macjd_sample_episodes.argtypes = [ctypes.POINTER(SamplerIO), ctypes.c_void_p]
    lib.macjd_gather_rows.restype = ctypes.c_int
    lib.macjd_gather_rows.argtypes = [ctypes.POINTER(GatherIO), ctypes.c_void_p]
    lib.macjd_linear_wgrad.restype = ctypes.c_int
    lib.macjd_linear_wgrad.argtypes = [ctypes.POINTER(WgradIO), ctypes.c_void_p]
    lib.macjd_linear_wgrad_many.restype = ctypes.c_int
    lib.macjd_linear_wgrad_many.argtypes = [ctypes.POINTER(WgradIO), ctypes.c_int32, ctypes.c_void_p]
    lib.macjd_linear_wgrad_workspace_floats.restype = ctypes.c_int64
    lib.macjd_linear_wgrad_workspace_floats.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    if lib.macjd_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_NAME}: ABI version {lib.macjd_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def reload_options() -> None:
    """Make the library re-read its MACJD_* environment switches (it reads them once, at the first launch)."""
    load().macjd_reload_options()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().macjd_last_error()
        raise NativeLibraryError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class ScenarioHandle:
    """Owns a ``macjd_scenario*`` (device copy of the scenario tables)."""

    def __init__(self, scenario):
        lib = load()
        desc, keep = scenario.c_desc()
        h = ctypes.c_void_p()
        check(lib.macjd_scenario_create(ctypes.addressof(desc), ctypes.byref(h)), "macjd_scenario_create")
        del keep
        self._h = h
        self._lib = lib

    @property
    def ptr(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.macjd_scenario_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
