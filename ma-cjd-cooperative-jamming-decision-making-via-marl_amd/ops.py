"""Tensor-level entry points of the fused HIP kernels (csrc/macjd_nets.hip, include/macjd_nets.h).

Dispatch rule: tensors on a HIP device ALWAYS go through libmacjd_hip.so — if the library is missing
the call raises (no silent eager substitute on the GPU, so a GPU run can never pass on a fallback).
Host (CPU) tensors are evaluated with stock torch ops; that branch exists only because the reference's
networks are device-agnostic (``main.py --device cpu``) and for the gloo multi-process tests — the
environment step itself has no CPU branch at all.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _native, options


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


def _qhead_fill(io: _native.QheadIO, base, P_all, W1, w2, b2, H: int, A: int, n_agents: int):
    io.n_rows, io.H, io.A, io.n_agents = base.shape[0], H, A, n_agents
    io.base, io.base_ld = base.data_ptr(), base.stride(0)
    io.P_all, io.p_ld = P_all.data_ptr(), P_all.stride(0)
    io.W1, io.w1_ld = W1.data_ptr(), W1.stride(0)
    io.w2, io.b2 = w2.data_ptr(), b2.data_ptr()


def qhead_all_actions_reference(base, P_all, W1, w2, b2, H: int, A: int) -> torch.Tensor:
    """Plain-torch evaluation of the decomposed Q-head, [N, A] (host tensors / numerics tests)."""
    w_a = W1[:, H:H + A]            # [H, A]
    w_p = W1[:, H + A]              # [H]
    pre = base.unsqueeze(1) + w_a.t().unsqueeze(0) + P_all.unsqueeze(2) * w_p.view(1, 1, -1)  # [N, A, H]
    return F.relu(pre) @ w2.reshape(-1) + b2.reshape(())


def qhead_all_actions(base, P_all, W1, w2, b2, H: int, A: int) -> torch.Tensor:
    """Q(h, a, P[:, a]) for all a, [N, A] float32.  ``base`` = W1[:, :H] h + b1 ([N, H])."""
    if not base.is_cuda:
        return qhead_all_actions_reference(base, P_all, W1, w2, b2, H, A)
    lib = _native.load()
    base, P_all, W1 = _f32c(base.detach()), _f32c(P_all.detach()), _f32c(W1.detach())
    w2, b2 = w2.detach().reshape(-1).contiguous(), b2.detach().reshape(-1).contiguous()
    Q = torch.empty((base.shape[0], A), dtype=torch.float32, device=base.device)
    io = _native.QheadIO()
    _qhead_fill(io, base, P_all, W1, w2, b2, H, A, 1)
    io.Q, io.q_ld = Q.data_ptr(), Q.stride(0)
    with torch.cuda.device(base.device):
        _native.check(lib.macjd_qhead_select(ctypes.byref(io), _stream(base)), "macjd_qhead_select")
    return Q


def qhead_double_q(base_eval, P_eval, head_eval, base_tgt, P_tgt, head_tgt, H: int, A: int) -> torch.Tensor:
    """Double-DQN target values without materialising either [N, A] Q tensor (reference core/qmix.py:138-147):
    a* = argmax_a Q_eval(h, a, P_a) (unmasked, as the reference), result [N] = Q_target(h', a*, P'_a*).
    ``head_* = (W1, w2, b2)`` of each network's fc2_q_head.  Two launches on a HIP device."""
    def q_all(base, P, head):
        return qhead_all_actions(base, P, head[0], head[1], head[2], H, A)
    if not base_eval.is_cuda:
        idx = q_all(base_eval, P_eval, head_eval).argmax(dim=1, keepdim=True)
        return torch.gather(q_all(base_tgt, P_tgt, head_tgt), 1, idx).squeeze(1)
    lib = _native.load()
    N = base_eval.shape[0]
    dev = base_eval.device
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    out = torch.empty(N, dtype=torch.float32, device=dev)
    keep = []
    for base, P, head, is_eval in ((base_eval, P_eval, head_eval, True), (base_tgt, P_tgt, head_tgt, False)):
        base, P, W1 = _f32c(base.detach()), _f32c(P.detach()), _f32c(head[0].detach())
        w2, b2 = head[1].detach().reshape(-1).contiguous(), head[2].detach().reshape(-1).contiguous()
        keep += [base, P, W1, w2, b2]
        io = _native.QheadIO()
        _qhead_fill(io, base, P, W1, w2, b2, H, A, 1)
        if is_eval:
            io.argmax_out = idx.data_ptr()
        else:
            io.gather_idx, io.q_gather_out = idx.data_ptr(), out.data_ptr()
        with torch.cuda.device(dev):
            _native.check(lib.macjd_qhead_select(ctypes.byref(io), _stream(base)), "macjd_qhead_select")
    return out


def qhead_double_q_fused_supported(h, H: int, A: int) -> bool:
    return h.is_cuda and bool(_native.load().macjd_qhead_double_q_supported(int(H), int(A)))


def qhead_double_q_from_h(h_eval, P_eval, head_eval, h_tgt, P_tgt, head_tgt, H: int, A: int, want_argmax: bool = False,
                          p_row_map=None):
    """Double-DQN target values [N] straight from the unrolled hidden states, ONE launch (csrc/macjd_episode.hip,
    qhead_double_q_kernel): both Q-head base products on the matrix cores, both all-action Q-heads, unmasked arg-max of
    the eval head, gather from the target head (reference core/qmix.py:138-147).  ``head_* = (W1 [H, H+A+1], b1 [H],
    w2, b2)`` of each network's fc2_q_head; h_* [N, H]; P_* [N, A]."""
    lib = _native.load()
    io, keep, out, am = _doubleq_io(h_eval, P_eval, head_eval, h_tgt, P_tgt, head_tgt, H, A, want_argmax, p_row_map)
    with torch.cuda.device(out.device):
        _native.check(lib.macjd_qhead_double_q(ctypes.byref(io), _stream(out)), "macjd_qhead_double_q")
    return (out, am) if want_argmax else out


def _doubleq_io(h_eval, P_eval, head_eval, h_tgt, P_tgt, head_tgt, H, A, want_argmax=False, p_row_map=None):
    """(macjd_doubleq_io, the tensors it points at, out [N], argmax [N] or None) of ``qhead_double_q_from_h``."""
    keep = []

    def c(t):
        t = t.detach()
        t = t if (t.dtype == torch.float32 and t.stride(-1) == 1) else t.float().contiguous()
        keep.append(t)
        return t

    h_e, h_t, P_e, P_t = c(h_eval), c(h_tgt), c(P_eval), c(P_tgt)
    N = h_e.shape[0]
    io = _native.DoubleQIO()
    io.n_rows, io.H, io.A = N, H, A
    io.h_e, io.he_ld, io.h_t, io.ht_ld = h_e.data_ptr(), h_e.stride(0), h_t.data_ptr(), h_t.stride(0)
    if P_e.dim() != 2:
        P_e, P_t = P_e.reshape(-1, A), P_t.reshape(-1, A)
    io.P_e, io.pe_ld, io.P_t, io.pt_ld = P_e.data_ptr(), P_e.stride(0), P_t.data_ptr(), P_t.stride(0)
    if p_row_map is not None:    # (rows per group, P rows per group): row n reads P row (n // group) * inner + n % inner
        io.p_group, io.p_inner = int(p_row_map[0]), int(p_row_map[1])
    for tag, head in (("e", head_eval), ("t", head_tgt)):
        W1, b1 = c(head[0]), c(head[1]).contiguous()
        w2, b2 = c(head[2]).reshape(-1).contiguous(), c(head[3]).reshape(-1).contiguous()
        keep += [b1, w2, b2]
        setattr(io, f"W1_{tag}", W1.data_ptr()); setattr(io, f"w1{tag}_ld", W1.stride(0))
        setattr(io, f"b1_{tag}", b1.data_ptr()); setattr(io, f"w2_{tag}", w2.data_ptr()); setattr(io, f"b2_{tag}", b2.data_ptr())
    out = torch.empty(N, dtype=torch.float32, device=h_e.device)
    io.out = out.data_ptr()
    am = None
    if want_argmax:
        am = torch.empty(N, dtype=torch.int64, device=h_e.device)
        io.argmax_out = am.data_ptr()
    keep += [P_e, P_t]
    return io, keep, out, am


# Paired launches of the learner update (csrc: macjd_qheads_pair, macjd_mixer_fused_forward_pair).  The update's target
# branch (Double-DQN Q-head launch -> target mixer) and its eval head (taken-action Q-head -> eval mixer, under autograd)
# are two independent chains of two launches each; as four launches they sat on two hardware queues and the serial
# chain paid a cross-queue hand-over where they meet.  A caller that knows both chains run ("pair_*" below, then the
# autograd call) gets them as two grids on ONE stream: the no-grad half is prepared here — outputs allocated, argument
# block built — and rides in the launch of the next autograd forward of the matching kind.
_PAIRED_DQ = None      # (macjd_doubleq_io, tensors kept alive, out) waiting for the next _QheadTaken.forward
_PAIRED_MIXER = None   # (macjd_mixerf_io, tensors kept alive, y) waiting for the next saving mixer_fused_forward


def pair_double_q_with_next_taken(h_eval, P_eval, head_eval, h_tgt, P_tgt, head_tgt, H: int, A: int, p_row_map=None):
    """``qhead_double_q_from_h`` whose launch happens inside the NEXT ``qhead_taken`` forward (one grid for both, on that
    call's stream).  Returns the [N] output tensor — valid once that forward has run (``assert_pairs_launched``)."""
    global _PAIRED_DQ
    assert _PAIRED_DQ is None, "a paired Double-DQN launch is already waiting"
    io, keep, out, _ = _doubleq_io(h_eval, P_eval, head_eval, h_tgt, P_tgt, head_tgt, H, A, False, p_row_map)
    _PAIRED_DQ = (io, keep, out)
    return out


def pair_mixer_forward_with_next_fused(q, s, params):
    """``mixer_fused_forward(q, s, params)`` (no activations saved) whose launch happens inside the NEXT differentiable
    ``mixer_fused`` forward (one grid for both).  Returns y [M, 1] — valid once that forward has run."""
    global _PAIRED_MIXER
    assert _PAIRED_MIXER is None, "a paired mixer launch is already waiting"
    q, s = q.detach().float().contiguous(), _f32c(s.detach())
    y = torch.empty((s.shape[0], 1), dtype=torch.float32, device=q.device)
    io = _mixerf_io(q, s, params)
    io.y = y.data_ptr()
    _PAIRED_MIXER = (io, (q, s, params), y)
    return y


def assert_pairs_launched():
    """Every launch handed to ``pair_*`` has gone out (it has not if the autograd call took another code path)."""
    global _PAIRED_DQ, _PAIRED_MIXER
    left = [n for n, v in (("Double-DQN", _PAIRED_DQ), ("mixer", _PAIRED_MIXER)) if v is not None]
    _PAIRED_DQ = _PAIRED_MIXER = None
    if left:
        raise RuntimeError("paired launch not taken by the autograd forward it was meant for: " + ", ".join(left))


def qhead_select(base, P_all, W1, w2, b2, H: int, A: int, n_agents: int, avail: Optional[torch.Tensor],
                 epsilon: float, greedy_only: bool, seed: int, counter: int, want_q: bool = False,
                 eps_dev: Optional[torch.Tensor] = None, counter_dev: Optional[torch.Tensor] = None,
                 out_T32: Optional[torch.Tensor] = None, out_P: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Fused all-action Q + mask + epsilon-greedy + gather on a HIP device.

    Returns ``(T64 [E, J, 1] int64, P [E, J, 1] float32, T32 [E, J] int32 view of agent-major
    storage, Q [N, A] or None)``.  ``T32``/``P`` are written agent-major ([J, E] storage) so the
    env-step kernel reads them fully coalesced; the returned tensors are transposed VIEWS with the
    reference's logical shapes."""
    if not base.is_cuda:
        raise RuntimeError("qhead_select is the HIP path; host tensors use the selector in utils.action_selectors")
    lib = _native.load()
    N = base.shape[0]
    E = N // n_agents
    dev = base.device
    base, P_all, W1 = _f32c(base.detach()), _f32c(P_all.detach()), _f32c(W1.detach())
    w2, b2 = w2.detach().reshape(-1).contiguous(), b2.detach().reshape(-1).contiguous()
    # chosen actions: agent-major scratch by default; or straight into caller-provided [E, J(,1)] tensors of any
    # strides (e.g. the runner's staging row, which the env-step kernel then reads in place)
    if out_T32 is not None:
        T32v = out_T32.view(E, n_agents) if out_T32.dim() == 3 else out_T32
        assert T32v.dtype == torch.int32 and tuple(T32v.shape) == (E, n_agents) and T32v.device == dev
    else:
        T32v = torch.empty((n_agents, E), dtype=torch.int32, device=dev).t()
    if out_P is not None:
        Pv = out_P.view(E, n_agents) if out_P.dim() == 3 else out_P
        assert Pv.dtype == torch.float32 and tuple(Pv.shape) == (E, n_agents) and Pv.device == dev
    else:
        Pv = torch.empty((n_agents, E), dtype=torch.float32, device=dev).t()
    T64 = torch.empty((E, n_agents, 1), dtype=torch.int64, device=dev)
    Q = torch.empty((N, A), dtype=torch.float32, device=dev) if want_q else None
    io = _native.QheadIO()
    _qhead_fill(io, base, P_all, W1, w2, b2, H, A, n_agents)
    io.greedy_only = 1 if greedy_only else 0
    if Q is not None:
        io.Q, io.q_ld = Q.data_ptr(), Q.stride(0)
    if avail is not None:
        if avail.dtype not in (torch.int32, torch.int64):
            avail = avail.to(torch.int32)
        if avail.device != dev:
            avail = avail.to(dev)
        io.avail, io.avail_elem_size = avail.data_ptr(), avail.element_size()
        io.av_se, io.av_sj, io.av_sa = avail.stride(0), avail.stride(1), avail.stride(2)
    io.epsilon, io.seed, io.counter = float(epsilon), int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1)
    if eps_dev is not None:      # float32 [1] on the device: exploration probability read at run time
        assert eps_dev.dtype == torch.float32 and eps_dev.device == dev
        io.eps_dev = eps_dev.data_ptr()
    if counter_dev is not None:  # int64 [1] on the device: added to ``counter`` at run time
        assert counter_dev.dtype == torch.int64 and counter_dev.device == dev
        io.counter_dev = counter_dev.data_ptr()
    io.T_out32, io.t32_se, io.t32_sj = T32v.data_ptr(), T32v.stride(0), T32v.stride(1)
    io.T_out64, io.t64_se, io.t64_sj = T64.data_ptr(), n_agents, 1
    io.P_out, io.po_se, io.po_sj = Pv.data_ptr(), Pv.stride(0), Pv.stride(1)
    with torch.cuda.device(dev):
        _native.check(lib.macjd_qhead_select(ctypes.byref(io), _stream(base)), "macjd_qhead_select")
    return T64, Pv.unsqueeze(-1), T32v, Q


def agent_episode_supported(J: int, H: int, A: int) -> bool:
    return bool(_native.load().macjd_agent_episode_supported(int(J), int(H), int(A)))


def agent_episode(gi, P_all, h0, w_hh, b_hh, W1, b1, w2, b2, n_envs: int, n_agents: int, T: int, avail, eps_sched,
                  greedy_only: bool, seed: int, counter_base, hidden_out, T_out, P_out, h_final=None):
    """All T agent steps of an episode batch in ONE launch (csrc/macjd_episode.hip; include/macjd_nets.h,
    macjd_agent_episode_io): GRU cell from the static input transform ``gi`` [E*J or 1, 3H], all-action MP-DQN Q-head on
    the static actor output ``P_all`` [E*J or 1, A], mask, epsilon-greedy (``eps_sched`` float32 [>= T] on the device,
    ``counter_base`` uint64 / int64 [1] on the device), gather.  Writes the staging rows ``hidden_out`` [>= T, E, J, H],
    ``T_out`` int32 [T, E, J(,1)], ``P_out`` float32 [T, E, J(,1)] (all contiguous) and ``h_final`` [E*J, H]."""
    lib = _native.load()
    H, A = w_hh.shape[1], P_all.shape[-1]
    io = _native.AgentEpisodeIO()
    io.n_envs, io.T, io.J, io.H, io.A = int(n_envs), int(T), int(n_agents), H, A
    io.greedy_only = 1 if greedy_only else 0
    keep = []

    def f32(t):
        t = t.detach()
        t = t if (t.dtype == torch.float32 and t.stride(-1) == 1) else t.float().contiguous()
        keep.append(t)
        return t

    gi, P_all = f32(gi), f32(P_all)
    io.gi, io.gi_ld = gi.data_ptr(), gi.stride(0)
    io.P_all, io.p_ld = P_all.data_ptr(), P_all.stride(0)
    if h0 is not None:
        h0 = f32(h0).contiguous()
        keep.append(h0)
        io.h0 = h0.data_ptr()
    w_hh, b_hh, W1 = f32(w_hh).contiguous(), f32(b_hh).contiguous(), f32(W1)
    keep += [w_hh, b_hh]
    io.w_hh, io.b_hh = w_hh.data_ptr(), b_hh.data_ptr()
    io.W1, io.w1_ld = W1.data_ptr(), W1.stride(0)
    b1, w2, b2 = f32(b1).contiguous(), f32(w2).reshape(-1).contiguous(), f32(b2).reshape(-1).contiguous()
    keep += [b1, w2, b2]
    io.b1, io.w2, io.b2 = b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    if avail is not None:
        if avail.dtype not in (torch.int32, torch.int64):
            avail = avail.to(torch.int32)
        keep.append(avail)
        io.avail, io.avail_elem_size = avail.data_ptr(), avail.element_size()
        io.av_se, io.av_sj, io.av_sa = avail.stride(0), avail.stride(1), avail.stride(2)
    if eps_sched is not None:
        assert eps_sched.dtype == torch.float32 and eps_sched.numel() >= T and eps_sched.is_contiguous()
        io.eps = eps_sched.data_ptr()
    io.seed = int(seed) & (2 ** 64 - 1)
    if counter_base is not None:
        assert counter_base.dtype in (torch.int64, torch.uint64) and counter_base.numel() >= 1
        io.counter_base = counter_base.data_ptr()
    for name, t_, dt in (("hidden_out", hidden_out, torch.float32), ("T_out", T_out, torch.int32), ("P_out", P_out, torch.float32)):
        assert t_.is_contiguous() and t_.dtype == dt and t_.device == gi.device, name
    assert hidden_out.numel() >= T * n_envs * n_agents * H and T_out.numel() >= T * n_envs * n_agents and P_out.numel() >= T_out.numel()
    io.hidden, io.T_out, io.P_out = hidden_out.data_ptr(), T_out.data_ptr(), P_out.data_ptr()
    if h_final is not None:
        assert h_final.is_contiguous() and h_final.dtype == torch.float32 and h_final.numel() == n_envs * n_agents * H
        io.h_final = h_final.data_ptr()
    with torch.cuda.device(gi.device):
        _native.check(lib.macjd_agent_episode(ctypes.byref(io), _stream(gi)), "macjd_agent_episode")


def gru_sequence_reference(gi: torch.Tensor, w_hh: torch.Tensor, b_hh: torch.Tensor,
                           h0: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Step-by-step GRU recurrence with stock torch ops (host tensors / numerics tests).

    gi = W_ih x_t + b_ih for every step, [B, T, J, 3H] (gate order r, z, n as torch.nn.GRUCell);
    returns every post-update hidden state, [B, T, J, H]:  r = s(gi_r + gh_r), z = s(gi_z + gh_z),
    n = tanh(gi_n + r * gh_n), h' = (h - n) * z + n, with gh = W_hh h + b_hh."""
    B, T, J, H3 = gi.shape
    H = H3 // 3
    h = gi.new_zeros(B * J, H) if h0 is None else h0.reshape(B * J, H)
    out = gi.new_empty(B, T, J, H)
    for t in range(T):
        g = gi[:, t].reshape(B * J, H3)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(g[:, :H] + gh[:, :H])
        z = torch.sigmoid(g[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(g[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (h - n) * z + n
        out[:, t] = h.view(B, J, H)
    return out


def gru_sequence(gi: torch.Tensor, w_hh: torch.Tensor, b_hh: torch.Tensor,
                 h0: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All T steps of the GRU recurrence for B*J independent sequences (inference path of the learner's
    unroll, reference core/qmix.py:241-253).  gi [B,T,J,3H] -> h_all [B,T,J,H]."""
    if not gi.is_cuda:
        return gru_sequence_reference(gi, w_hh, b_hh, h0)
    return gru_sequence_multi([gi], [w_hh], [b_hh], [h0])[0]


def gru_sequence_from_obs(obs, obs_index, agents, B: int, J: int, n_steps: int, with_actor: bool = False):
    """Scan of up to two agents whose observation is static within an episode, with the input transform computed INSIDE
    the scan launch: sequence (b, j) reads the observation row ``obs[obs_index[b], 0, j]`` (``obs`` [N, T+1, J, S]: the
    replay ring itself; ``obs_index`` int64 [B] on the device, or None for rows 0..B-1) and evaluates
    gi = W_ih ReLU(fc1 x + b) + b_ih once (reference core/networks.py:96-100).  No separate fc1 / W_ih launch and no
    dependence on a gather in front of the scan.  Returns [h_all [B, n_steps, J, H]] per agent (HIP device only);
    ``with_actor``: also the actor output of every sequence's observation row, ([h_all ...], [P [B, J, A] ...])."""
    lib = _native.load()
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 4 and obs.stride(3) == 1 and obs.shape[2] == J
    outs, pouts = [], []
    for start in range(0, len(agents), 2):
        part = agents[start:start + 2]
        H = part[0].rnn_hidden_dim
        if H not in (64, 128):
            raise _native.NativeLibraryError(f"macjd_gru_sequence supports rnn_hidden_dim 64 or 128, got {H}")
        io = _native.GruIO()
        io.n_nets, io.B, io.T, io.J, io.H = len(part), int(B), int(n_steps), int(J), H
        io.obs, io.obs_sb, io.obs_sj, io.S = obs.data_ptr(), obs.stride(0), obs.stride(2), obs.shape[3]
        if obs_index is not None:
            assert obs_index.dtype == torch.int64 and obs_index.is_cuda and obs_index.numel() >= B
            io.obs_index = obs_index.data_ptr()
        keep = []
        for k, a in enumerate(part):
            ts = [t.detach().float().contiguous() for t in (a.rnn.weight_hh, a.rnn.bias_hh, a.fc1.weight, a.fc1.bias,
                                                            a.rnn.weight_ih, a.rnn.bias_ih)]
            keep += ts
            io.w_hh[k], io.b_hh[k], io.fc1_w[k], io.fc1_b[k], io.w_ih[k], io.b_ih[k] = [t.data_ptr() for t in ts]
            o = torch.empty((B, n_steps, J, H), dtype=torch.float32, device=obs.device)
            io.h_out[k] = o.data_ptr()
            outs.append(o)
            if with_actor:
                layers = [(w.detach().float().contiguous(), b_.detach().float().contiguous()) for w, b_, _ in a.actor_layers()]
                keep += [t for pair in layers for t in pair]
                for l, (w, b_) in enumerate(layers):
                    io.act_w[k][l], io.act_b[k][l] = w.data_ptr(), b_.data_ptr()
                io.Ah, io.A = layers[0][0].shape[0], layers[2][0].shape[0]
                pk_ = torch.empty((B, J, io.A), dtype=torch.float32, device=obs.device)
                io.p_out[k] = pk_.data_ptr()
                pouts.append(pk_)
        with torch.cuda.device(obs.device):
            _native.check(lib.macjd_gru_sequence(ctypes.byref(io), _stream(obs)), "macjd_gru_sequence")
    return (outs, pouts) if with_actor else outs


def gru_sequence_multi(gis, w_hhs, b_hhs, h0s=None, n_steps=None):
    """Up to two networks (eval + target) scanned in ONE launch on a HIP device; host tensors loop.  ``n_steps``: the
    gi tensors are [B, 1, J, 3H] — one input transform per sequence, valid at every one of ``n_steps`` steps (static
    observation) — instead of [B, T, J, 3H]."""
    n = len(gis)
    h0s = h0s if h0s is not None else [None] * n
    static = n_steps is not None
    if static:
        assert all(g.shape[1] == 1 for g in gis)
    if not gis[0].is_cuda:
        if static:
            gis = [g.expand(-1, int(n_steps), -1, -1) for g in gis]
        return [gru_sequence_reference(g, w, b, h) for g, w, b, h in zip(gis, w_hhs, b_hhs, h0s)]
    lib = _native.load()
    B, T, J, H3 = gis[0].shape
    if static:
        T = int(n_steps)
    H = H3 // 3
    if H not in (64, 128):
        raise _native.NativeLibraryError(f"macjd_gru_sequence supports rnn_hidden_dim 64 or 128, got {H}")
    outs = []
    for start in range(0, n, 2):
        io = _native.GruIO()
        keep = []
        cnt = min(2, n - start)
        io.n_nets, io.B, io.T, io.J, io.H = cnt, B, T, J, H
        io.reserved = 1 if static else 0
        for k in range(cnt):
            g = gis[start + k].detach().float().contiguous()
            w = w_hhs[start + k].detach().float().contiguous()
            bb = b_hhs[start + k].detach().float().contiguous()
            h0 = h0s[start + k]
            if h0 is not None:
                # [B, J, H] with any batch stride (e.g. step 0 of a stored [B, T+1, J, H] tensor): no copy
                h0 = h0.detach().float().reshape(B, J, H) if h0.dim() != 3 else h0.detach().float()
                if h0.stride(2) != 1 or h0.stride(1) != H:
                    h0 = h0.contiguous()
                io.h0_sb[k] = h0.stride(0)
            o = torch.empty((B, T, J, H), dtype=torch.float32, device=g.device)
            keep += [g, w, bb, h0]
            io.gi[k], io.w_hh[k], io.b_hh[k] = g.data_ptr(), w.data_ptr(), bb.data_ptr()
            io.h0[k] = h0.data_ptr() if h0 is not None else None
            io.h_out[k] = o.data_ptr()
            outs.append(o)
        with torch.cuda.device(gis[0].device):
            _native.check(lib.macjd_gru_sequence(ctypes.byref(io), _stream(gis[0])), "macjd_gru_sequence")
    return outs


def gru_gates(gi, gh, h, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h' [N, H] of one GRUCell step from gi = W_ih x + b_ih and gh = W_hh h + b_hh (inference).  HIP device: one
    LDS-free elementwise launch that can also write h' to a second [N, H] destination (``out2``: the runner's staging
    row); host tensors: the same arithmetic with torch ops."""
    N, H = h.shape
    if not h.is_cuda or (H & 3):
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (h - n) * z + n
        if out2 is not None:
            out2.view(N, H).copy_(hn)
        return hn
    lib = _native.load()
    gi, gh, h = _f32c(gi.detach()), _f32c(gh.detach()), _f32c(h.detach())
    out = torch.empty((N, H), dtype=torch.float32, device=h.device)
    io = _native.GruGatesIO()
    io.n_rows, io.H = N, H
    io.gi, io.gi_ld, io.gh, io.gh_ld = gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0)
    io.h, io.h_ld, io.h_out, io.ho_ld = h.data_ptr(), h.stride(0), out.data_ptr(), out.stride(0)
    if out2 is not None:
        o2 = out2.view(N, H)
        assert o2.dtype == torch.float32 and o2.stride(1) == 1 and o2.device == h.device
        io.h_out2, io.ho2_ld = o2.data_ptr(), o2.stride(0)
    with torch.cuda.device(h.device):
        _native.check(lib.macjd_gru_gates(ctypes.byref(io), _stream(h)), "macjd_gru_gates")
    return out


# ---------------------------------------------------------------------------------------------
# QMix mixer tail (reference core/networks.py:283-315)
def mixer_tail_reference(q, w1_raw, b1_raw, wf_raw, v_raw):
    """Stock-torch form: q [M,J], w1_raw [M,J*Em], b1_raw [M,Em], wf_raw [M,Em], v_raw [M,1] -> y [M,1]."""
    M, J = q.shape
    Em = b1_raw.shape[1]
    w1 = torch.clamp(w1_raw, min=0.0, max=5.0).view(M, J, Em)
    b1 = torch.clamp(b1_raw, min=-5.0, max=5.0)
    wf = torch.clamp(wf_raw, min=0.0, max=5.0)
    v = torch.clamp(v_raw, min=-5.0, max=5.0)
    hidden = F.elu((q.unsqueeze(2) * w1).sum(dim=1) + b1)
    return (hidden * wf).sum(dim=1, keepdim=True) + v


def _mixer_io(q, w1_raw, b1_raw, wf_raw, v_raw):
    io = _native.MixerIO()
    io.M, io.J, io.Em = q.shape[0], q.shape[1], b1_raw.shape[1]
    io.q, io.w1_raw, io.b1_raw = q.data_ptr(), w1_raw.data_ptr(), b1_raw.data_ptr()
    io.wf_raw, io.v_raw = wf_raw.data_ptr(), v_raw.data_ptr()
    io.b1_ld = b1_raw.stride(0)      # may be a column block of the merged first-layer output
    return io


class _MixerTailHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, w1_raw, b1_raw, wf_raw, v_raw):
        lib = _native.load()
        args = [t.detach().float().contiguous() for t in (q, w1_raw)] + [_f32c(b1_raw.detach())] + \
               [t.detach().float().contiguous() for t in (wf_raw, v_raw)]
        y = torch.empty((q.shape[0], 1), dtype=torch.float32, device=q.device)
        io = _mixer_io(*args)
        io.y = y.data_ptr()
        with torch.cuda.device(q.device):
            _native.check(lib.macjd_mixer_tail_forward(ctypes.byref(io), _stream(q)), "macjd_mixer_tail_forward")
        ctx.save_for_backward(*args)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _native.load()
        q, w1_raw, b1_raw, wf_raw, v_raw = ctx.saved_tensors
        gy = gy.detach().float().contiguous()
        gq, gw1 = torch.empty_like(q), torch.empty_like(w1_raw)
        gb1 = torch.empty(b1_raw.shape, dtype=torch.float32, device=q.device)
        gwf, gv = torch.empty_like(wf_raw), torch.empty_like(v_raw)
        io = _mixer_io(q, w1_raw, b1_raw, wf_raw, v_raw)
        io.gy, io.gq, io.gw1_raw = gy.data_ptr(), gq.data_ptr(), gw1.data_ptr()
        io.gb1_raw, io.gwf_raw, io.gv_raw = gb1.data_ptr(), gwf.data_ptr(), gv.data_ptr()
        with torch.cuda.device(q.device):
            _native.check(lib.macjd_mixer_tail_backward(ctypes.byref(io), _stream(q)), "macjd_mixer_tail_backward")
        return gq, gw1, gb1, gwf, gv


def mixer_tail(q, w1_raw, b1_raw, wf_raw, v_raw):
    """y = ELU(q . clamp(w1) + clamp(b1)) . clamp(wf) + clamp(v), [M,1]; fused HIP forward/backward on a
    HIP device (differentiable), stock torch ops for host tensors."""
    if not q.is_cuda:
        return mixer_tail_reference(q, w1_raw, b1_raw, wf_raw, v_raw)
    return _MixerTailHip.apply(q, w1_raw, b1_raw, wf_raw, v_raw)


# ---------------------------------------------------------------------------------------------
# The whole mixer as one MFMA chain per direction (csrc/macjd_mixer.hip, include/macjd_nets.h: macjd_mixerf_io)
def mixer_fused_supported(J: int, S: int, Hh: int, Em: int) -> bool:
    return bool(_native.load().macjd_mixer_fused_supported(int(J), int(S), int(Hh), int(Em)))


def _mixerf_io(q, s, p):
    """p = dict(ln_w, ln_b, eps, W1, b1, W2, b2, Wf2, bf2, wV2, bV2) of contiguous float32 device tensors."""
    io = _native.MixerFusedIO()
    io.M, io.J, io.S, io.Hh, io.Em = q.shape[0], q.shape[1], p["W1"].shape[1], p["W2"].shape[1], p["Wf2"].shape[0]
    io.ln_eps = float(p["eps"])
    io.q = q.data_ptr()
    if s is not None:
        io.s, io.s_ld = s.data_ptr(), s.stride(0)
    for k in ("ln_w", "ln_b", "W1", "b1", "W2", "b2", "Wf2", "bf2", "wV2", "bV2"):
        setattr(io, k, p[k].data_ptr())
    return io


def _mixerf_params(ln_w, ln_b, eps, w_cat, b_cat, W2, b2, Wf2, bf2, wV2, bV2):
    c = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()
    return {"ln_w": c(ln_w), "ln_b": c(ln_b), "eps": eps, "W1": c(w_cat), "b1": c(b_cat), "W2": c(W2), "b2": c(b2),
            "Wf2": c(Wf2), "bf2": c(bf2), "wV2": c(wV2).reshape(-1), "bV2": c(bV2).reshape(-1)}


def mixer_fused_forward(q, s, params, save=False):
    """y [M,1] (+ (sn, xhat, act) when ``save``) of the fused mixer; q [M,J], s [M,S] float32 on a HIP device."""
    lib = _native.load()
    q, s = q.detach().float().contiguous(), _f32c(s.detach())
    M, S = s.shape
    y = torch.empty((M, 1), dtype=torch.float32, device=q.device)
    io = _mixerf_io(q, s, params)
    io.y = y.data_ptr()
    saved = None
    if save:
        width = 2 * io.Hh + 2 * io.Em
        sn = torch.empty((M, S), dtype=torch.float32, device=q.device)
        xhat = torch.empty((M, S), dtype=torch.float32, device=q.device)
        act = torch.empty((M, width), dtype=torch.float32, device=q.device)
        io.save, io.sn, io.xhat, io.act = 1, sn.data_ptr(), xhat.data_ptr(), act.data_ptr()
        saved = (sn, xhat, act)
    global _PAIRED_MIXER
    pair = None
    if save:
        pair, _PAIRED_MIXER = _PAIRED_MIXER, None
    with torch.cuda.device(q.device):
        if pair is not None:   # the target mixer rides in this grid (pair_mixer_forward_with_next_fused)
            _native.check(lib.macjd_mixer_fused_forward_pair(ctypes.byref(io), ctypes.byref(pair[0]), _stream(q)),
                          "macjd_mixer_fused_forward_pair")
        else:
            _native.check(lib.macjd_mixer_fused_forward(ctypes.byref(io), _stream(q)), "macjd_mixer_fused_forward")
    return y, q, saved


class _FusedMixer(torch.autograd.Function):
    """Q_tot = QMixer(q, s) as one forward and one backward launch (+ the grouped split-K weight gradients).  Inputs:
    q [M,J] (differentiable), s [M,S] (no gradient), the LayerNorm parameters, the merged first layer (w_cat / b_cat:
    views of the flat parameter vector or a concatenation; ``first`` = the eight underlying parameters, passed so that
    autograd routes their gradients: backward returns row blocks of ONE weight-gradient product), the three second
    layers."""

    @staticmethod
    def forward(ctx, q, s, ln_w, ln_b, eps, w_cat, b_cat, W2, b2, Wf2, bf2, wV2, bV2, *first):
        params = _mixerf_params(ln_w, ln_b, eps, w_cat, b_cat, W2, b2, Wf2, bf2, wV2, bV2)
        y, qc, (sn, xhat, act) = mixer_fused_forward(q, s, params, save=True)
        ctx.save_for_backward(qc, sn, xhat, act, w_cat, W2, Wf2, wV2, ln_w, ln_b, b2, bf2, bV2)
        ctx.params = {k: v for k, v in params.items() if k in ("eps",)}
        ctx.sizes = [p.shape[0] for p in first[:len(first) // 2]]
        ctx.keys = {"b_cat": grad_key(b_cat), "ln_w": grad_key(ln_w), "ln_b": grad_key(ln_b), "b2": grad_key(b2),
                    "bf2": grad_key(bf2), "bV2": grad_key(bV2)}
        ctx.n_first = len(first)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _native.load()
        q, sn, xhat, act, w_cat, W2, Wf2, wV2, ln_w, ln_b, b2, bf2, bV2 = ctx.saved_tensors
        M, J = q.shape
        Hh, Em = W2.shape[1], Wf2.shape[0]
        dev = q.device
        gy = gy.detach().float().contiguous()
        gq = torch.empty((M, J), dtype=torch.float32, device=dev)
        gout1 = torch.empty((M, 2 * Hh + 2 * Em), dtype=torch.float32, device=dev)
        g_w1 = torch.empty((M, J * Em), dtype=torch.float32, device=dev)
        g_wf = torch.empty((M, Em), dtype=torch.float32, device=dev)
        g_v = torch.empty((M, 1), dtype=torch.float32, device=dev)
        params = _mixerf_params(ln_w, ln_b, ctx.params["eps"], w_cat, gout1[0], W2, b2, Wf2, bf2, wV2, bV2)   # b1 unused here
        io = _mixerf_io(q, None, params)
        io.act, io.gy, io.gq, io.gout1 = act.data_ptr(), gy.data_ptr(), gq.data_ptr(), gout1.data_ptr()
        io.g_w1raw, io.g_wfraw, io.g_v = g_w1.data_ptr(), g_wf.data_ptr(), g_v.data_ptr()
        global _PENDING_TD
        pend, _PENDING_TD = _PENDING_TD, None
        with torch.cuda.device(dev):
            if pend is not None:   # the loss's gradient is formed in this launch: `gy` is a placeholder (td_grad_in_mixer_backward)
                assert pend[2] == M, "td_grad_in_mixer_backward: the loss rows are not this mixer's rows"
                _native.check(lib.macjd_mixer_fused_backward_td(ctypes.byref(io), ctypes.byref(pend[0]), pend[3].data_ptr(),
                                                                _stream(q)), "macjd_mixer_fused_backward_td")
            else:
                _native.check(lib.macjd_mixer_fused_backward(ctypes.byref(io), _stream(q)), "macjd_mixer_fused_backward")
        nd = ctx.needs_input_grad
        # weight / bias gradients: split-K products of the matrices the kernel wrote (recorded inside deferred_wgrad)
        gW1, gb1 = linear_wgrad(gout1, sn, want_bias=True, w_key=grad_key(w_cat), b_key=ctx.keys["b_cat"])
        G, _ = linear_wgrad(gout1, xhat, want_bias=False)
        gW2, gb2 = linear_wgrad(g_w1, act[:, :Hh], want_bias=True, w_key=grad_key(W2), b_key=ctx.keys["b2"], need=(nd[7], nd[8]))
        gWf, gbf = linear_wgrad(g_wf, act[:, Hh:2 * Hh], want_bias=True, w_key=grad_key(Wf2), b_key=ctx.keys["bf2"], need=(nd[9], nd[10]))
        gWv, gbv = linear_wgrad(g_v, act[:, 2 * Hh:2 * Hh + Em], want_bias=True, w_key=grad_key(wV2), b_key=ctx.keys["bV2"],
                                need=(nd[11], nd[12]))
        # LayerNorm parameter gradients from (W_cat, G, gb1), one small launch after the grouped products
        K = w_cat.shape[1]
        dgamma, dbeta = _grad_dst(ctx.keys["ln_w"], (K,)), _grad_dst(ctx.keys["ln_b"], (K,))
        if dgamma is None:
            dgamma = torch.empty(K, dtype=torch.float32, device=dev)
        if dbeta is None:
            dbeta = torch.empty(K, dtype=torch.float32, device=dev)
        lio = _native.LnParamIO()
        lio.C, lio.K = w_cat.shape[0], K
        lio.W, lio.w_ld, lio.G, lio.g_ld = w_cat.data_ptr(), w_cat.stride(0), G.data_ptr(), G.stride(0)
        lio.gb, lio.dgamma, lio.dbeta = gb1.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
        stream = _stream(q)

        def post(lio=lio, keep=(G, w_cat), dev=dev, stream=stream):
            with torch.cuda.device(dev):
                _native.check(_native.load().macjd_layernorm_param_grad(ctypes.byref(lio), stream), "macjd_layernorm_param_grad")

        post.lnparam = lio   # (a context that holds these launches back hands them to the optimiser step, see deferred_wgrad)
        if _DEFERRED_WGRAD is not None:
            _DEFERRED_POST.append(post)
        else:
            post()
        first = tuple(gW1.split(ctx.sizes, 0)) + tuple(gb1.split(ctx.sizes, 0))
        return (gq, None, dgamma, dbeta, None, None, None, gW2, gb2, gWf, gbf, gWv, gbv) + first


def mixer_fused(q, s, ln_w, ln_b, eps, w_cat, b_cat, W2, b2, Wf2, bf2, wV2, bV2, first_params):
    """Differentiable fused mixer (see _FusedMixer); y [M, 1]."""
    return _FusedMixer.apply(q, s, ln_w, ln_b, eps, w_cat, b_cat, W2, b2, Wf2, bf2, wV2, bV2, *first_params)


def enable_gemm_tuning(results_file: Optional[str] = None, max_tuning_ms: int = 30) -> bool:
    """Let PyTorch's TunableOp pick the rocBLAS / hipBLASLt solution for each library-GEMM shape on this
    GPU (first call per shape times the candidates).  The path's GEMMs are small and oddly shaped — e.g. the
    mixer's weight gradients are [192,128] outputs over K = 3168 rows, for which the default heuristic
    picks a few-workgroup kernel; tuning takes ~10 s once and cut the learner step by 15 % on MI355X.
    Must run BEFORE any HIP-graph capture.  Returns False when TunableOp is unavailable."""
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(int(max_tuning_ms))
        if results_file:
            tunable.set_filename(results_file)
        return True
    except Exception as e:  # pragma: no cover
        print(f"TunableOp unavailable: {e}")
        return False


# ---------------------------------------------------------------------------------------------
# Fused dense chain on the matrix cores (csrc/macjd_mlp.hip)
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
_MLP_LDS_FLOATS = 160 * 1024 // 4


def _fused_relu_ok(x, w, b):
    """hipBLASLt applies bias + ReLU in the GEMM epilogue through torch._addmm_activation: ONE kernel, bit-identical
    to relu(linear(x)) on this stack (checked on MI355X for the path's shapes)."""
    return (x.is_cuda and x.dim() == 2 and b is not None and x.dtype == torch.float32 and w.dtype == torch.float32
            and not torch.is_autocast_enabled() and hasattr(torch, "_addmm_activation"))


def mlp_reference(x, layers):
    """layers = [(weight [out,in], bias [out], act), ...] evaluated with library GEMMs / stock torch ops."""
    for w, b, act in layers:
        if act == ACT_RELU and not torch.is_grad_enabled() and _fused_relu_ok(x, w, b):
            x = torch._addmm_activation(b, x, w.t(), use_gelu=False)
            continue
        x = F.linear(x, w, b)
        x = F.relu(x) if act == ACT_RELU else (torch.sigmoid(x) if act == ACT_SIGMOID else x)
    return x


def _mlp_pitch(width: int) -> int:
    return ((width + 7) // 16) * 16 + 8


def mlp_supported(dims) -> bool:
    """Shape limits of macjd_mlp_forward (see include/macjd_nets.h; mirrors the host checks in csrc/macjd_mlp.hip)."""
    L = len(dims) - 1
    if not (1 <= L <= 3) or dims[0] > 256:
        return False
    biggest = 0
    for l in range(L):
        K, N, last = dims[l], dims[l + 1], l == L - 1
        if (l > 0 and K > 128) or (not last and N > 128) or (last and N > 384):
            return False
        if (N + 15) // 16 not in (1, 2, 3, 4, 8, 12, 24):
            return False
        n16, k16 = (N + 15) // 16 * 16, (K + 15) // 16 * 16
        biggest = max(biggest, n16 * _mlp_pitch(k16) + n16)
    lda = _mlp_pitch((max(dims[:-1]) + 15) // 16 * 16)
    return biggest <= _MLP_LDS_FLOATS - 4 * 16 * lda


def mlp_forward(x, layers):
    """Inference-only fused evaluation of up to three Linear(+activation) layers, [N, in] -> [N, out].
    HIP device + supported widths: ONE MFMA kernel launch; otherwise stock torch ops (library GEMMs)."""
    dims = [layers[0][0].shape[1]] + [w.shape[0] for w, _, _ in layers]
    if not x.is_cuda or not mlp_supported(dims):
        return mlp_reference(x, layers)
    lib = _native.load()
    x = _f32c(x.detach())
    if x.dim() != 2:
        x = x.reshape(-1, dims[0])
    y = torch.empty((x.shape[0], dims[-1]), dtype=torch.float32, device=x.device)
    io = _native.MlpIO()
    io.n_rows, io.n_layers = x.shape[0], len(layers)
    keep = []
    for l, (w, b, act) in enumerate(layers):
        w, b = w.detach().float().contiguous(), b.detach().float().contiguous()
        keep += [w, b]
        io.dims[l], io.W[l], io.b[l], io.act[l] = dims[l], w.data_ptr(), b.data_ptr(), int(act)
    io.dims[len(layers)] = dims[-1]
    io.x, io.x_ld, io.y, io.y_ld = x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0)
    with torch.cuda.device(x.device):
        _native.check(lib.macjd_mlp_forward(ctypes.byref(io), _stream(x)), "macjd_mlp_forward")
    return y


def _mlp_io(x, layers, keep):
    dims = [layers[0][0].shape[1]] + [w.shape[0] for w, _, _ in layers]
    x = _f32c(x.detach())
    if x.dim() != 2:
        x = x.reshape(-1, dims[0])
    y = torch.empty((x.shape[0], dims[-1]), dtype=torch.float32, device=x.device)
    io = _native.MlpIO()
    io.n_rows, io.n_layers = x.shape[0], len(layers)
    for l, (w, b, act) in enumerate(layers):
        w, b = w.detach().float().contiguous(), b.detach().float().contiguous()
        keep += [w, b]
        io.dims[l], io.W[l], io.b[l], io.act[l] = dims[l], w.data_ptr(), b.data_ptr(), int(act)
    io.dims[len(layers)] = dims[-1]
    io.x, io.x_ld, io.y, io.y_ld = x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0)
    keep.append(x)
    return io, y, dims


def mlp_forward_pair(x0, layers0, x1, layers1):
    """Two independent dense chains (see mlp_forward) in ONE launch on a HIP device: -> (y0, y1).  Falls back to two
    separate evaluations when either chain is outside the kernel's limits or the tensors live on the host."""
    d0 = [layers0[0][0].shape[1]] + [w.shape[0] for w, _, _ in layers0]
    d1 = [layers1[0][0].shape[1]] + [w.shape[0] for w, _, _ in layers1]
    if not (x0.is_cuda and x1.is_cuda and mlp_supported(d0) and mlp_supported(d1)) or x0.numel() == 0 or x1.numel() == 0:
        return mlp_forward(x0, layers0), mlp_forward(x1, layers1)
    lib = _native.load()
    keep = []
    io0, y0, _ = _mlp_io(x0, layers0, keep)
    io1, y1, _ = _mlp_io(x1, layers1, keep)
    with torch.cuda.device(x0.device):
        _native.check(lib.macjd_mlp_forward_pair(ctypes.byref(io0), ctypes.byref(io1), _stream(x0)), "macjd_mlp_forward_pair")
    return y0, y1


# ---------------------------------------------------------------------------------------------
# TD target + masked loss (reference core/qmix.py:155,190-194)
def td_loss_reference(y, tq, reward, terminated, filled, gamma):
    """Stock-torch form.  y, tq [B,Tm1,1]; reward / terminated / filled [B,Tm1,1] (views are fine).
    Returns (loss, mean(y), mean(target))."""
    targets = reward + gamma * (1 - terminated.float()) * tq
    m = filled.float()
    td = (y - targets.detach()) * m
    return (td ** 2).sum() / m.sum(), y.detach().mean(), targets.mean()


class _TdLossHip(torch.autograd.Function):
    """y_full [B, Ty, 1] (eval Q_tot, uses steps 0..Tm1-1), tq_full [B, Tq, 1] (target Q_tot, uses steps
    tq_off..tq_off+Tm1-1); reward / terminated / filled are [B, >=Tm1, 1] views.  No slicing happens in autograd:
    the gradient comes back as a full-length [B, Ty, 1] tensor whose unused steps are zero."""

    @staticmethod
    def forward(ctx, y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off):
        lib = _native.load()
        B, Ty = y_full.shape[0], y_full.shape[1]
        yc, tqc = y_full.detach().float().contiguous(), tq_full.detach().float().contiguous()
        stats = torch.empty(4, dtype=torch.float32, device=yc.device)
        gy = torch.empty((B, Ty, 1), dtype=torch.float32, device=yc.device)
        io = _native.TdLossIO()
        io.B, io.Tm1, io.gamma = B, int(Tm1), float(gamma)
        io.y, io.y_sb = yc.data_ptr(), Ty
        io.tq, io.tq_sb = tqc.data_ptr() + 4 * int(tq_off), tqc.shape[1]
        io.gy, io.gy_sb, io.gy_cols = gy.data_ptr(), Ty, Ty
        io.reward, io.r_sb, io.r_st = reward.data_ptr(), reward.stride(0), reward.stride(1)
        io.terminated, io.t_sb, io.t_st = terminated.data_ptr(), terminated.stride(0), terminated.stride(1)
        io.filled, io.f_sb, io.f_st = filled.data_ptr(), filled.stride(0), filled.stride(1)
        io.stats = stats.data_ptr()
        with torch.cuda.device(yc.device):
            _native.check(lib.macjd_td_loss(ctypes.byref(io), _stream(yc)), "macjd_td_loss")
        ctx.save_for_backward(gy)
        ctx.mark_non_differentiable(stats)
        stats.saved_gy = gy      # for td_loss_and_grad (plain attribute; autograd does not look at it)
        return stats[0], stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        (gy,) = ctx.saved_tensors
        return gy * g_loss, None, None, None, None, None, None, None


def td_loss_full(y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off):
    """Full-length form used by the graphed update: loss over eval steps 0..Tm1-1 against target steps
    tq_off..tq_off+Tm1-1, without slicing either tensor (HIP device only)."""
    loss, stats = _TdLossHip.apply(y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off)
    return loss, stats[1], stats[2]


def td_loss_and_grad(y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off):
    """(loss, mean(y), mean(target), dL/dy [B, Ty, 1], the kernel's [4] output tensor (loss, mean y, mean target, mask
    sum)) in one launch, nothing recorded for autograd: the caller seeds
    the backward pass with the gradient itself (``y_full.backward(gy)``) — ``loss.backward()`` would first fill a
    ones tensor for the scalar and multiply the saved gradient by it (three launch-bound kernels)."""
    with torch.no_grad():
        _, stats = _TdLossHip.apply(y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off)
    return stats[0], stats[1], stats[2], stats.saved_gy, stats


def td_loss_sums_into(row, y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off):
    """The logged sums of the TD loss — row[0:3] = (loss, mean(y), mean(target)), reference core/qmix.py:194, 212-213 — by
    one launch that writes no gradient and leaves row[3] alone: for updates whose loss gradient is formed inside the
    mixer's backward launch (``td_grad_in_mixer_backward``); this launch may then run any time later, off the chain."""
    lib = _native.load()
    io, keep = _tdloss_sums_io(row, y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off)
    with torch.cuda.device(row.device):
        _native.check(lib.macjd_td_loss(ctypes.byref(io), _stream(row)), "macjd_td_loss")
    return row


def _tdloss_sums_io(row, y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off):
    """(macjd_tdloss_io of the sums-only form, the tensors it points at)"""
    B, Ty = y_full.shape[0], y_full.shape[1]
    yc, tqc = y_full.detach(), tq_full.detach()
    assert yc.dtype == torch.float32 and tqc.dtype == torch.float32 and yc.is_contiguous() and tqc.is_contiguous()
    assert row.dtype == torch.float32 and row.numel() >= 3 and row.is_contiguous()
    io = _native.TdLossIO()
    io.B, io.Tm1, io.gamma = B, int(Tm1), float(gamma)
    io.y, io.y_sb = yc.data_ptr(), Ty
    io.tq, io.tq_sb = tqc.data_ptr() + 4 * int(tq_off), tqc.shape[1]
    io.reward, io.r_sb, io.r_st = reward.data_ptr(), reward.stride(0), reward.stride(1)
    io.terminated, io.t_sb, io.t_st = terminated.data_ptr(), terminated.stride(0), terminated.stride(1)
    io.filled, io.f_sb, io.f_st = filled.data_ptr(), filled.stride(0), filled.stride(1)
    io.stats, io.gy = row.data_ptr(), None
    return io, (row, yc, tqc, reward, terminated, filled)


def td_mask_sum(filled, Tm1):
    """float32 [1]: the number of loss-carrying steps of a batch, sum(filled[:, :Tm1]) — the one global quantity the TD
    loss's gradient needs (``td_grad_in_mixer_backward``).  ``filled`` [B, >= Tm1, 1] bool on a HIP device."""
    lib = _native.load()
    out = torch.empty(1, dtype=torch.float32, device=filled.device)
    io = _native.TdLossIO()
    io.B, io.Tm1 = filled.shape[0], int(Tm1)
    io.filled, io.f_sb, io.f_st = filled.data_ptr(), filled.stride(0), filled.stride(1)
    with torch.cuda.device(filled.device):
        _native.check(lib.macjd_td_mask_sum(ctypes.byref(io), out.data_ptr(), _stream(filled)), "macjd_td_mask_sum")
    return out


_PENDING_TD = None   # TD-loss inputs waiting for the fused mixer's backward launch (td_grad_in_mixer_backward)


def fused_mixer_backward_will_run(y_full) -> bool:
    """``y_full`` is the output of the fused mixer node (possibly behind views): its backward is ONE launch that can form
    the TD loss's gradient itself (macjd_mixer_fused_backward_td)."""
    fn = getattr(y_full, "grad_fn", None)
    for _ in range(4):
        if fn is None:
            return False
        if type(fn).__name__.startswith("_FusedMixer"):
            return True
        nxt = [f for f, _ in fn.next_functions if f is not None]
        if len(nxt) != 1:
            return False
        fn = nxt[0]
    return False


def td_grad_in_mixer_backward(y_full, tq_full, reward, terminated, filled, gamma, Tm1, tq_off, tot_m, stats_row=None):
    """Arrange for the fused mixer's backward launch (started by ``y_full.backward(placeholder)``, placeholder = the return
    value) to form dL/dy of the TD loss itself from the loss's inputs and ``tot_m`` (``td_mask_sum``): no loss launch in
    front of the backward pass.  The logged statistics: one extra workgroup of that launch writes (loss, mean y, mean
    target) into ``stats_row[0:3]`` when given; otherwise they are the caller's business (``td_loss_and_grad`` /
    ``td_loss_sums_into`` off the chain)."""
    global _PENDING_TD
    B, Ty = y_full.shape[0], y_full.shape[1]
    yc, tqc = y_full.detach().float().contiguous(), tq_full.detach().float().contiguous()
    io = _native.TdLossIO()
    io.B, io.Tm1, io.gamma = B, int(Tm1), float(gamma)
    io.y, io.y_sb = yc.data_ptr(), Ty
    io.tq, io.tq_sb = tqc.data_ptr() + 4 * int(tq_off), tqc.shape[1]
    io.gy, io.gy_sb, io.gy_cols = None, Ty, Ty
    io.reward, io.r_sb, io.r_st = reward.data_ptr(), reward.stride(0), reward.stride(1)
    io.terminated, io.t_sb, io.t_st = terminated.data_ptr(), terminated.stride(0), terminated.stride(1)
    io.filled, io.f_sb, io.f_st = filled.data_ptr(), filled.stride(0), filled.stride(1)
    if stats_row is not None:
        assert stats_row.dtype == torch.float32 and stats_row.numel() >= 3 and stats_row.is_contiguous()
        io.stats = stats_row.data_ptr()
    _PENDING_TD = (io, (yc, tqc, reward, terminated, filled, tot_m, stats_row), B * Ty, tot_m)
    return torch.empty((B, Ty, 1), dtype=torch.float32, device=yc.device)   # never read


def td_loss(y, tq, reward, terminated, filled, gamma):
    """(loss, mean(y), mean(target)) with loss differentiable in y.  One fused launch on a HIP device."""
    ok = (y.is_cuda and reward.dtype == torch.float32 and terminated.dtype == torch.bool and filled.dtype == torch.bool
          and y.shape[1] > 0)
    if not ok:
        return td_loss_reference(y, tq, reward, terminated, filled, gamma)
    loss, stats = _TdLossHip.apply(y, tq, reward, terminated, filled, gamma, y.shape[1], 0)
    return loss, stats[1], stats[2]


# ---------------------------------------------------------------------------------------------
# Fused clip_grad_norm_ + Adam on flat vectors (reference core/qmix.py:199-200)
def _sampler_io(idx_out, n_stored, counter, seed):
    assert idx_out.dtype == torch.int64 and idx_out.is_contiguous() and n_stored.dtype == torch.int32 and counter.dtype == torch.int64
    assert idx_out.is_cuda and n_stored.device == idx_out.device and counter.device == idx_out.device
    sp = _native.SamplerIO()
    sp.idx_out, sp.n, sp.n_stored, sp.counter = idx_out.data_ptr(), idx_out.numel(), n_stored.data_ptr(), counter.data_ptr()
    sp.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return sp


def sample_episodes(idx_out, n_stored, counter, seed):
    """Device-side draw of ``idx_out.numel()`` distinct episode indices, uniform over [0, n_stored) (the reference's
    ``np.random.choice(current_size, batch, replace=False)``, utils/replay_buffer.py:89, without the host): a keyed
    pseudo-random permutation, see include/macjd_nets.h ``macjd_sampler_io``.  ``n_stored`` int32 [1] and ``counter``
    int64 [1] live on the device; the counter advances by one."""
    lib = _native.load()
    sp = _sampler_io(idx_out, n_stored, counter, seed)
    with torch.cuda.device(idx_out.device):
        _native.check(lib.macjd_sample_episodes(ctypes.byref(sp), _stream(idx_out)), "macjd_sample_episodes")


def clip_adam_step(param, grad, exp_avg, exp_avg_sq, step, grad_norm, partials, lr, betas, eps, max_norm, sample_next=None,
                   lnparam=None):
    """In-place update of ``param`` / ``exp_avg`` / ``exp_avg_sq`` / ``step`` (all flat float32 on one HIP
    device); writes the pre-clip gradient norm into ``grad_norm``.  ``sample_next`` = (idx_out, n_stored, counter,
    seed): the update launch also draws the NEXT update's episodes (``sample_episodes``) when it is done.
    ``lnparam`` = LayerNorm-parameter launches held back by ``deferred_wgrad(hold_lnparam=True)``: a single one whose
    outputs are ranges of ``grad`` is evaluated inside the squared-norm launch; anything else is simply issued first."""
    lib = _native.load()
    io = _native.AdamIO()
    io.n, io.lr, io.beta1, io.beta2, io.eps, io.max_norm = param.numel(), lr, betas[0], betas[1], eps, max_norm
    io.param, io.grad, io.exp_avg, io.exp_avg_sq = param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
    io.step, io.grad_norm, io.partials = step.data_ptr(), grad_norm.data_ptr(), partials.data_ptr()
    sp = _sampler_io(*sample_next) if sample_next is not None else None
    spp = ctypes.byref(sp) if sp is not None else None
    held = list(lnparam or [])
    fuse = None
    if len(held) == 1:
        lio, esz, base = held[0].lnparam, grad.element_size(), grad.data_ptr()
        g_off, b_off = (lio.dgamma - base) // esz, (lio.dbeta - base) // esz
        if ((lio.dgamma - base) % esz == 0 and (lio.dbeta - base) % esz == 0 and 0 <= g_off <= grad.numel() - lio.K
                and 0 <= b_off <= grad.numel() - lio.K and partials.numel() >= 256 + lio.K):
            fuse = (lio, int(g_off), int(b_off))
    with torch.cuda.device(param.device):
        if fuse is not None:
            _native.check(lib.macjd_clip_adam_step_ln(ctypes.byref(io), spp, ctypes.byref(fuse[0]), fuse[1], fuse[2],
                                                      _stream(param)), "macjd_clip_adam_step_ln")
            return
        for fn in held:
            fn()
        _native.check(lib.macjd_clip_adam_step_sample(ctypes.byref(io), spp, _stream(param)), "macjd_clip_adam_step")


def gather_rows_supported(srcs) -> bool:
    return len(srcs) <= 8 and all(s.is_cuda and s.is_contiguous() and (s[0].numel() * s.element_size()) % 4 == 0
                                  for s in srcs)


def gather_rows(idx, srcs, dsts):
    """dst_k[i] = src_k[idx[i]] for up to 8 tensors in one launch (whole rows, raw bytes)."""
    lib = _native.load()
    io = _native.GatherIO()
    io.n_tensors, io.n_rows, io.idx = len(srcs), idx.numel(), idx.data_ptr()
    for k, (s_, d_) in enumerate(zip(srcs, dsts)):
        assert d_.is_contiguous() and d_.dtype == s_.dtype and d_.shape[0] == idx.numel() and d_[0].numel() >= s_[0].numel()
        io.src[k], io.dst[k], io.row_bytes[k] = s_.data_ptr(), d_.data_ptr(), s_[0].numel() * s_.element_size()
        io.dst_row_bytes[k] = d_.stride(0) * d_.element_size()   # destination rows may be longer (padded steps)
    with torch.cuda.device(idx.device):
        _native.check(lib.macjd_gather_rows(ctypes.byref(io), _stream(idx)), "macjd_gather_rows")


# ---------------------------------------------------------------------------------------------
# Linear with a split-K MFMA weight / bias gradient (csrc/macjd_wgrad.hip)
_DEFERRED_WGRAD = None   # list of (WgradIO, keep-alive tensors) while a ``deferred_wgrad`` context is active
_DEFERRED_POST = None    # callables run right after the grouped launches of that context
_GRAD_DST = None         # {grad_key: destination tensor} of the active context (see deferred_wgrad.__init__)
_DEFERRED_SEEN = None    # grad_keys with a recorded (or already flushed) gradient in the active context


def grad_key(t):
    """Identity of a parameter (or of a merged view over several) for the gradient-destination map."""
    return None if t is None else (t.data_ptr(), t.numel())


def _grad_dst(key, shape):
    """A FRESH view (autograd must be its only owner, see linear_wgrad) of the registered destination, or None."""
    if not _GRAD_DST or key is None:
        return None
    dst = _GRAD_DST.get(key)
    if dst is None or dst.numel() != int(np.prod(shape)) or not dst.is_contiguous():
        return None
    del _GRAD_DST[key]   # once per backward pass: a parameter used twice gets an ordinary second gradient, which
    return dst.view(shape)   # autograd then accumulates (in place) into the first


class deferred_wgrad:
    """``with ops.deferred_wgrad(): loss.backward()`` — inside the context a split-K weight gradient launches nothing:
    the problem is recorded (its operands kept alive) and ALL recorded problems run as one partial-products launch +
    one reduce launch when the context exits (same arithmetic and summation order per problem).  They only feed
    ``.grad`` and nothing in the backward pass waits for them.  The gradient tensors handed to autograd are filled by
    the flush: read them only after the context — and let autograd (or the caller) be their only owner until then.
    A parameter used twice in the graph is handled (its second gradient forces the recorded ones out first)."""

    def __init__(self, grad_dst=None, hold_lnparam=False):
        # {grad_key(parameter): preallocated gradient tensor of the parameter's shape}: inside the context the
        # weight / bias gradients of those parameters are written straight into these tensors (slices of the
        # learner's flat gradient vector) and autograd receives fresh views of them — no packing copy afterwards
        self._grad_dst = grad_dst
        # hold_lnparam: the LayerNorm-parameter launches that follow the grouped products are NOT issued at exit but
        # collected in ``held`` — the caller passes them to ``clip_adam_step(lnparam=...)``, whose squared-norm launch
        # evaluates them (one launch less), or calls them itself
        self._hold, self.held = bool(hold_lnparam), []

    def __enter__(self):
        global _DEFERRED_WGRAD, _DEFERRED_POST, _GRAD_DST, _DEFERRED_SEEN
        self._prev, _DEFERRED_WGRAD = _DEFERRED_WGRAD, []
        self._prev_post, _DEFERRED_POST = _DEFERRED_POST, []
        self._prev_dst, _GRAD_DST = _GRAD_DST, (dict(self._grad_dst) if self._grad_dst else None)
        self._prev_seen, _DEFERRED_SEEN = _DEFERRED_SEEN, set()
        return self

    def __exit__(self, *exc):
        global _DEFERRED_WGRAD, _DEFERRED_POST, _GRAD_DST, _DEFERRED_SEEN
        pending, _DEFERRED_WGRAD = _DEFERRED_WGRAD, self._prev
        post, _DEFERRED_POST = _DEFERRED_POST, self._prev_post
        _GRAD_DST, _DEFERRED_SEEN = self._prev_dst, self._prev_seen
        self._flush(exc, pending)
        if exc[0] is None:
            for fn in post:      # launches that consume the flushed products (e.g. LayerNorm parameter gradients)
                if self._hold and hasattr(fn, "lnparam"):
                    self.held.append(fn)
                else:
                    fn()
        return False

    @staticmethod
    def _flush(exc, pending):
        if exc[0] is None and pending:
            lib = _native.load()
            dev = pending[0][1][0].device
            for lo in range(0, len(pending), 8):
                part = pending[lo:lo + 8]
                arr = (_native.WgradIO * len(part))(*[io for io, _ in part])
                with torch.cuda.device(dev):
                    _native.check(lib.macjd_linear_wgrad_many(arr, len(part), _stream(part[0][1][0])),
                                  "macjd_linear_wgrad_many")


def linear_wgrad(gout, inp, want_bias=True, w_key=None, b_key=None, need=(True, True), outer=None):
    """dW [M,N] = gout[K,M]^T inp[K,N], db [M] = column sums of gout (HIP device, float32, row-strided inputs).
    ``w_key`` / ``b_key`` = grad_key of the parameters these are the gradients of: when the active ``deferred_wgrad``
    context maps them to destinations, the results are written there.  ``need`` = which of (dW, db) the caller hands
    to autograd for a tensor that requires grad; a result autograd will drop is kept alive until the deferred launch
    has written it (its memory must not be reused before)."""
    lib = _native.load()
    gout, inp = _f32c(gout), _f32c(inp)
    K, M = gout.shape
    N = inp.shape[1]
    if outer is not None:
        # ``gout`` is a ReLU's OUTPUT and the [K, M] operand is (gout > 0) ? vec[k] * w[m] : 0, formed while the kernel
        # stages it (include/macjd_nets.h, macjd_wgrad_io.outer_vec): no launch materialises the product
        o_vec, o_w = _f32c(outer[0]).reshape(-1).contiguous(), _f32c(outer[1]).reshape(-1).contiguous()
        assert o_vec.numel() == K and o_w.numel() == M
    dW = _grad_dst(w_key, (M, N))
    if dW is None:
        dW = torch.empty((M, N), dtype=torch.float32, device=gout.device)
    db = None
    if want_bias:
        db = _grad_dst(b_key, (M,))
        if db is None:
            db = torch.empty((M,), dtype=torch.float32, device=gout.device)
    ws = torch.empty(int(lib.macjd_linear_wgrad_workspace_floats(K, M, N)), dtype=torch.float32, device=gout.device)
    io = _native.WgradIO()
    io.K, io.M, io.N = K, M, N
    io.gout, io.gout_ld, io.inp, io.inp_ld = gout.data_ptr(), gout.stride(0), inp.data_ptr(), inp.stride(0)
    io.dW, io.dw_ld, io.db, io.workspace = dW.data_ptr(), dW.stride(0), (db.data_ptr() if want_bias else None), ws.data_ptr()
    if outer is not None:
        io.outer_vec, io.outer_w = o_vec.data_ptr(), o_w.data_ptr()
    if _DEFERRED_WGRAD is not None and any(k is not None and k in _DEFERRED_SEEN for k in (w_key, b_key)):
        # a parameter used twice in the graph: autograd ADDS this gradient to the first one as soon as it gets it, so
        # the recorded ones must be in memory by then — run them now, and this one immediately
        pending, posts = list(_DEFERRED_WGRAD), list(_DEFERRED_POST)
        del _DEFERRED_WGRAD[:], _DEFERRED_POST[:]
        deferred_wgrad._flush((None, None, None), pending)
        for fn in posts:
            fn()
    elif _DEFERRED_WGRAD is not None:
        _DEFERRED_SEEN.update(k for k in (w_key, b_key) if k is not None)
        # keep the OPERANDS and the workspace alive until the flush, but hold no reference to dW / db: autograd's
        # AccumulateGrad only adopts a gradient tensor it is the sole owner of — with a second reference it would
        # clone the still unfilled buffer into .grad right away (the flush would then fill a tensor nobody reads)
        unowned = tuple(t for t, n in ((dW, need[0]), (db, need[1])) if t is not None and not n)
        _DEFERRED_WGRAD.append((io, (ws, gout, inp) + unowned + ((o_vec, o_w) if outer is not None else ())))
        return dW, db
    with torch.cuda.device(gout.device):
        _native.check(lib.macjd_linear_wgrad(ctypes.byref(io), _stream(gout)), "macjd_linear_wgrad")
    return dW, db


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b.  Forward and the input gradient are library GEMMs; the weight / bias gradients, tiny outputs
    reduced over thousands of rows, run on the split-K MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.b_key = bias is not None, grad_key(bias)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy.matmul(weight) if ctx.needs_input_grad[0] else None
        gW = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gW, gb = linear_wgrad(gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1]), want_bias=ctx.has_bias,
                                  w_key=grad_key(weight), b_key=ctx.b_key,
                                  need=(ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]))
        return gx, gW, (gb if ctx.has_bias else None)


class _MergedLinear(torch.autograd.Function):
    """y = x W_cat^T + b_cat where W_cat / b_cat are VIEWS of a flat parameter vector covering several adjacent
    Linear layers (no torch.cat per call).  ``params`` = the layers' weights then biases, passed only so that autograd
    routes the gradients: backward returns row blocks of ONE split-K weight gradient."""

    @staticmethod
    def forward(ctx, x, w_cat, b_cat, *params):
        ctx.save_for_backward(x, w_cat)
        ctx.sizes, ctx.b_key = [p.shape[0] for p in params[:len(params) // 2]], grad_key(b_cat)
        return F.linear(x, w_cat, b_cat)

    @staticmethod
    def backward(ctx, gy):
        x, w_cat = ctx.saved_tensors
        gx = gy.matmul(w_cat) if ctx.needs_input_grad[0] else None
        gW, gb = linear_wgrad(gy, x, want_bias=True, w_key=grad_key(w_cat), b_key=ctx.b_key)
        return (gx, None, None) + tuple(gW.split(ctx.sizes, 0)) + tuple(gb.split(ctx.sizes, 0))


def merged_linear(x, w_cat, b_cat, params):
    return _MergedLinear.apply(x, w_cat, b_cat, *params)


def _layernorm_launch(x, weight, bias, eps, want_xhat):
    lib = _native.load()
    x = _f32c(x.detach())
    M, S = x.shape
    y = torch.empty((M, S), dtype=torch.float32, device=x.device)
    mean = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    xhat = torch.empty((M, S), dtype=torch.float32, device=x.device) if want_xhat else None
    io = _native.LayerNormIO()
    io.M, io.S, io.eps = M, S, float(eps)
    io.x, io.x_ld, io.y, io.y_ld = x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0)
    io.gamma = weight.data_ptr() if weight is not None else None
    io.beta = bias.data_ptr() if bias is not None else None
    io.mean, io.rstd = mean.data_ptr(), rstd.data_ptr()
    if want_xhat:
        io.xhat, io.xhat_ld = xhat.data_ptr(), xhat.stride(0)
    with torch.cuda.device(x.device):
        _native.check(lib.macjd_layernorm_forward(ctypes.byref(io), _stream(x)), "macjd_layernorm_forward")
    return x, y, mean, rstd, xhat


class _NormMergedLinear(torch.autograd.Function):
    """out = LayerNorm(x; gamma, beta) W_cat^T + b_cat with W_cat / b_cat views of the flat parameter vector (see
    _MergedLinear), for an input x that needs no gradient (the mixer's state): backward runs (inside
    ``deferred_wgrad``: records) two split-K problems — gout^T s (the layers' weight / bias gradients) and G = gout^T xhat — and, after the
    grouped launches, ONE small launch turns (W_cat, G, gb) into the LayerNorm's gamma / beta gradients
    (macjd_layernorm_param_grad).  No [M, K] input-gradient GEMM, no LayerNorm-backward launches."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w_cat, b_cat, *params):
        _, s_, _, _, xhat = _layernorm_launch(x, gamma, beta, eps, want_xhat=True)
        ctx.save_for_backward(s_, xhat, w_cat)
        ctx.sizes = [p.shape[0] for p in params[:len(params) // 2]]
        ctx.keys = (grad_key(b_cat), grad_key(gamma), grad_key(beta))
        return F.linear(s_, w_cat, b_cat)

    @staticmethod
    def backward(ctx, gy):
        s_, xhat, w_cat = ctx.saved_tensors
        assert not ctx.needs_input_grad[0], "norm_merged_linear is for inputs that need no gradient"
        gW, gb = linear_wgrad(gy, s_, want_bias=True, w_key=grad_key(w_cat), b_key=ctx.keys[0])
        G, _ = linear_wgrad(gy, xhat, want_bias=False)
        K = w_cat.shape[1]
        dgamma, dbeta = _grad_dst(ctx.keys[1], (K,)), _grad_dst(ctx.keys[2], (K,))
        if dgamma is None:
            dgamma = torch.empty(K, dtype=torch.float32, device=gy.device)
        if dbeta is None:
            dbeta = torch.empty(K, dtype=torch.float32, device=gy.device)
        io = _native.LnParamIO()
        io.C, io.K = w_cat.shape[0], K
        io.W, io.w_ld, io.G, io.g_ld = w_cat.data_ptr(), w_cat.stride(0), G.data_ptr(), G.stride(0)
        io.gb, io.dgamma, io.dbeta = gb.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
        dev, stream = gy.device, _stream(gy)

        def post(io=io, keep=(G, w_cat), dev=dev, stream=stream):   # gb / dgamma / dbeta live on as the parameters' .grad
            with torch.cuda.device(dev):
                _native.check(_native.load().macjd_layernorm_param_grad(ctypes.byref(io), stream), "macjd_layernorm_param_grad")

        post.lnparam = io
        if _DEFERRED_WGRAD is not None:
            _DEFERRED_POST.append(post)      # the products above are only recorded yet: run after the grouped launches
        else:
            post()
        return (None, dgamma, dbeta, None, None, None) + tuple(gW.split(ctx.sizes, 0)) + tuple(gb.split(ctx.sizes, 0))


def norm_merged_linear(x, gamma, beta, eps, w_cat, b_cat, params):
    return _NormMergedLinear.apply(x, gamma, beta, eps, w_cat, b_cat, *params)


def deferred_wgrad_active() -> bool:
    return _DEFERRED_WGRAD is not None


def qhead_input(h, idx, P, n_actions: int):
    """[h, onehot(idx), P] rows for the Q-head, [n, H + A + 1] float32 — one launch on a HIP device (the reference
    builds it with F.one_hot / torch.cat, core/networks.py:160-174).  Not differentiable (its inputs are data)."""
    n, H = h.shape
    if not h.is_cuda:
        onehot = (idx.reshape(n, 1) == torch.arange(n_actions, device=idx.device, dtype=idx.dtype)).to(h.dtype)
        return torch.cat([h, onehot, P.reshape(n, 1).to(h.dtype)], dim=1)
    lib = _native.load()
    h, P = _f32c(h.detach()), P.detach().float().reshape(n).contiguous()
    idx = idx.detach().reshape(n)
    if idx.dtype not in (torch.int32, torch.int64):
        idx = idx.to(torch.int64)
    idx = idx.contiguous()
    out = torch.empty((n, H + n_actions + 1), dtype=torch.float32, device=h.device)
    io = _native.QinputIO()
    io.n_rows, io.H, io.A = n, H, int(n_actions)
    io.h, io.h_ld, io.idx, io.idx_elem_size = h.data_ptr(), h.stride(0), idx.data_ptr(), idx.element_size()
    io.P, io.out, io.out_ld = P.data_ptr(), out.data_ptr(), out.stride(0)
    with torch.cuda.device(h.device):
        _native.check(lib.macjd_qhead_input(ctypes.byref(io), _stream(h)), "macjd_qhead_input")
    return out


class _LayerNormHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x, y, mean, rstd, _ = _layernorm_launch(x, weight, bias, eps, want_xhat=False)
        ctx.save_for_backward(x, mean, rstd, weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd, weight, bias = ctx.saved_tensors
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        gx, gw, gb = torch.ops.aten.native_layer_norm_backward(gy.contiguous(), x.contiguous(), [x.shape[1]], mean, rstd,
                                                               weight, bias, mask)
        return gx, gw, gb, None


def layer_norm(x, weight, bias, eps: float = 1e-5):
    """F.layer_norm over the last dim of a 2-D float32 tensor: one HIP launch forward (torch's is a moments kernel
    + a normalise kernel); backward is torch's native_layer_norm_backward on the saved mean / rstd."""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 1024 and weight is not None and bias is not None:
        return _LayerNormHip.apply(x, weight, bias, eps)
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


# The one-output Linear layers (Q-head second layer, mixer V head) ride inside the node that produces their input, so
# their backward outer product is folded into that node's one backward launch (measured: DESIGN.md 4.7)


def _splitrelu_backward_launch(act, widths, Cp, grads, g_pass, outer):
    """gout [M, sum(widths) + Cp] of macjd_splitrelu_backward; ``outer[k]`` (or None) is the one-row weight of a
    one-output Linear that consumed block k, ``grads[k]`` then being that layer's [M, 1] output gradient."""
    lib = _native.load()
    M = act.shape[0]
    gout = torch.empty((M, sum(widths) + Cp), dtype=torch.float32, device=act.device)
    io = _native.SplitReluBwdIO()
    io.M, io.n_blocks, io.Cp = M, len(widths), Cp
    keep = []
    for k, wk in enumerate(widths):
        io.width[k] = wk
        g = grads[k]
        if g is None:
            continue
        g = _f32c(g)
        keep.append(g)
        setattr(io, f"g{k}", g.data_ptr())
        io.g_ld[k] = g.stride(0)
        if outer[k] is not None:
            w = outer[k].detach().reshape(-1).float().contiguous()
            keep.append(w)
            setattr(io, f"ow{k}", w.data_ptr())
    if Cp and g_pass is not None:
        gp = _f32c(g_pass)
        keep.append(gp)
        io.g_pass, io.gp_ld = gp.data_ptr(), gp.stride(0)
    io.act, io.act_ld, io.gout, io.gout_ld = act.data_ptr(), act.stride(0), gout.data_ptr(), gout.stride(0)
    with torch.cuda.device(act.device):
        _native.check(lib.macjd_splitrelu_backward(ctypes.byref(io), _stream(act)), "macjd_splitrelu_backward")
    return gout


class _SplitRelu(torch.autograd.Function):
    """(ReLU(x[:, :w0]), ReLU(x[:, w0:w0+w1]), ..., x[:, Cr:]) for a [M, Cr + Cp] matrix: the ReLU blocks are column
    views of ONE activation buffer (one forward launch, as before); the backward writes the masked block gradients
    and the pass-through gradient into the [M, Cr + Cp] result with ONE launch (autograd: cat + threshold_backward +
    cat).  With ``dot_k >= 0`` block dot_k is not returned itself but fed through a one-output Linear (weight
    [1, w_k], bias [1]; the mixer's V head): forward = the row-dot launch, backward = the outer product folded into the
    same backward launch, the Linear's weight gradient through ``linear_wgrad`` like every other layer."""

    @staticmethod
    def forward(ctx, x, widths, Cp, dot_k, dot_w, dot_b):
        Cr = sum(widths)
        act = torch.relu(x[:, :Cr])                     # contiguous [M, Cr]
        ctx.widths, ctx.Cp, ctx.dot_k = list(widths), int(Cp), int(dot_k)
        outs = list(act.split(widths, dim=1))
        if dot_k >= 0:
            ctx.save_for_backward(act, dot_w)
            ctx.has_bias, ctx.b_key = dot_b is not None, grad_key(dot_b)
            outs[dot_k] = _rowdot_launch(outs[dot_k], dot_w, dot_b)
        else:
            ctx.save_for_backward(act)
        if Cp:
            outs.append(x[:, Cr:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        act = ctx.saved_tensors[0]
        nb, k = len(ctx.widths), ctx.dot_k
        outer = [None] * nb
        gW = gb = None
        if k >= 0:
            dot_w = ctx.saved_tensors[1]
            outer[k] = dot_w
            if grads[k] is not None and (ctx.needs_input_grad[4] or (ctx.has_bias and ctx.needs_input_grad[5])):
                h = act.split(ctx.widths, dim=1)[k]
                gW, gb = linear_wgrad(grads[k].reshape(-1, 1), h, want_bias=ctx.has_bias, w_key=grad_key(dot_w), b_key=ctx.b_key,
                                      need=(ctx.needs_input_grad[4], ctx.has_bias and ctx.needs_input_grad[5]))
        gout = _splitrelu_backward_launch(act, ctx.widths, ctx.Cp, grads, grads[nb] if ctx.Cp else None, outer)
        return gout, None, None, None, gW, (gb if k >= 0 and ctx.has_bias else None)


def split_relu(x, relu_widths, pass_width: int, dot=None):
    """Column blocks of a 2-D tensor: ReLU on the first ``sum(relu_widths)`` columns (returned as blocks of those
    widths), the last ``pass_width`` columns unchanged.  ``dot = (k, weight [1, w_k], bias)`` replaces block k in the
    result by ``F.linear(block_k, weight, bias)``.  HIP device + autograd: fused backward (see _SplitRelu)."""
    widths = [int(w) for w in relu_widths]
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad \
            and 1 <= len(widths) <= 4 and not torch.is_autocast_enabled():
        if dot is not None and _rowdot_ok(x[:, :widths[dot[0]]], dot[1]):
            return _SplitRelu.apply(x, widths, int(pass_width), int(dot[0]), dot[1], dot[2])
        outs = list(_SplitRelu.apply(x, widths, int(pass_width), -1, None, None))
    else:
        Cr = sum(widths)
        outs = list(torch.relu(x[:, :Cr]).split(widths, dim=1))
        if pass_width:
            outs.append(x[:, Cr:])
    if dot is not None:
        outs[dot[0]] = linear(outs[dot[0]], dot[1], dot[2])
    return tuple(outs)


def _rowdot_launch(x, w, b):
    lib = _native.load()
    x = _f32c(x.detach())
    w = w.detach().reshape(-1).float().contiguous()
    y = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
    io = _native.RowdotIO()
    io.n_rows, io.K = x.shape[0], x.shape[1]
    io.x, io.x_ld, io.w, io.y = x.data_ptr(), x.stride(0), w.data_ptr(), y.data_ptr()
    keep = b.detach().reshape(-1).float().contiguous() if b is not None else None
    io.b = keep.data_ptr() if keep is not None else None
    with torch.cuda.device(x.device):
        _native.check(lib.macjd_rowdot(ctypes.byref(io), _stream(x)), "macjd_rowdot")
    return y


class _RowDot(torch.autograd.Function):
    """y = x w^T + b for a weight with ONE output row ([1, K]): forward is one row-dot launch (the library path is
    a bias-broadcast copy + a 16 x 256-tile GEMM); backward: gx = gy w (broadcast product), gW / gb through the split-K
    weight-gradient kernel like every other Linear."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.b_key = bias is not None, grad_key(bias)
        return _rowdot_launch(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy * weight.reshape(1, -1) if ctx.needs_input_grad[0] else None
        gW = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gW, gb = linear_wgrad(gy.reshape(-1, 1), x.reshape(-1, x.shape[-1]), want_bias=ctx.has_bias,
                                  w_key=grad_key(weight), b_key=ctx.b_key,
                                  need=(ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]))
        return gx, gW, (gb if ctx.has_bias else None)


def _rowdot_ok(x, weight):
    return (x.is_cuda and x.dim() == 2 and weight.shape[0] == 1 and x.dtype == torch.float32 and x.stride(-1) == 1
            and weight.shape[1] % 4 == 0 and 4 <= weight.shape[1] <= 1024 and x.shape[0] >= 1024
            and not torch.is_autocast_enabled())


class _LinearReluSplitK(torch.autograd.Function):
    """y = relu(x W^T + b) with the bias + ReLU in the GEMM epilogue (one forward kernel instead of two); backward:
    mask by y > 0, library input gradient, split-K weight / bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        ctx.b_key = grad_key(bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(gy, y, 0.0)
        gx = g.matmul(weight) if ctx.needs_input_grad[0] else None
        gW, gb = linear_wgrad(g, x, want_bias=True, w_key=grad_key(weight), b_key=ctx.b_key,
                              need=(ctx.needs_input_grad[1], ctx.needs_input_grad[2]))
        return gx, gW, gb


def linear_relu(x, weight, bias):
    """relu(F.linear(x, weight, bias)); on a HIP device the ReLU rides in the GEMM epilogue (with autograd: the
    split-K weight gradient as in ``linear``)."""
    if _fused_relu_ok(x, weight, bias):
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad):
            if x.shape[0] >= 1024 and weight.shape[0] * weight.shape[1] <= 384 * 256 and x.stride(-1) == 1:
                return _LinearReluSplitK.apply(x, weight, bias)
            return F.relu(linear(x, weight, bias))
        return torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
    return F.relu(linear(x, weight, bias))


class _LinearReluRowDot(torch.autograd.Function):
    """q = relu(x W1^T + b1) w2^T + b2 with a one-row w2 (the MP-DQN Q-head, reference core/networks.py:75-79) as ONE
    autograd node: forward = GEMM with the bias + ReLU epilogue, row-dot launch; backward = ONE launch for
    gq w2 masked by the ReLU (autograd: broadcast product + threshold_backward), then both layers' weight gradients
    through ``linear_wgrad``."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        y = torch._addmm_activation(b1, x, w1.t(), use_gelu=False)
        ctx.save_for_backward(x, w1, y, w2)
        ctx.has_b2, ctx.b1_key, ctx.b2_key = b2 is not None, grad_key(b1), grad_key(b2)
        return _rowdot_launch(y, w2, b2)

    @staticmethod
    def backward(ctx, gq):
        x, w1, y, w2 = ctx.saved_tensors
        nd = ctx.needs_input_grad
        gW1 = gb1 = gW2 = gb2 = None
        if nd[0] or not options.on("WGRAD_OUTER"):
            g = _splitrelu_backward_launch(y, [y.shape[1]], 0, [gq], None, [w2])   # gq w2 masked by the ReLU, [n, H]
            gx = g.matmul(w1) if nd[0] else None
            if nd[1] or nd[2]:
                gW1, gb1 = linear_wgrad(g, x, want_bias=True, w_key=grad_key(w1), b_key=ctx.b1_key, need=(nd[1], nd[2]))
        else:
            # nobody wants the input gradient (the learner's case: the Q-head's input is data): that product is only
            # the first layer's weight-gradient operand, which the weight-gradient kernel forms while staging it
            gx = None
            if nd[1] or nd[2]:
                gW1, gb1 = linear_wgrad(y, x, want_bias=True, w_key=grad_key(w1), b_key=ctx.b1_key, need=(nd[1], nd[2]),
                                        outer=(gq, w2))
        if nd[3] or (ctx.has_b2 and nd[4]):
            gW2, gb2 = linear_wgrad(gq.reshape(-1, 1), y, want_bias=ctx.has_b2, w_key=grad_key(w2), b_key=ctx.b2_key,
                                    need=(nd[3], ctx.has_b2 and nd[4]))
        return gx, gW1, gb1, gW2, (gb2 if ctx.has_b2 else None)


class _QheadTaken(torch.autograd.Function):
    """q = Q-head([h, onehot(idx), P]) for the TAKEN action (reference core/networks.py:131-180) as ONE forward launch
    (macjd_qhead_taken: input rows, first layer on MFMA + ReLU, second layer's dot); the backward is
    ``_LinearReluRowDot``'s on the saved input rows and activations (h / idx / P are data: no input gradient)."""

    @staticmethod
    def forward(ctx, h, idx, P, w1, b1, w2, b2, n_actions):
        lib = _native.load()
        n, H = h.shape
        hc = h.detach()
        if hc.dtype != torch.float32 or hc.stride(1) != 1 or (hc.stride(0) & 3) or (hc.data_ptr() & 15):
            hc = hc.float().contiguous()
        Pc = P.detach().float().reshape(n).contiguous()
        ic = idx.detach().reshape(n)
        if ic.dtype not in (torch.int32, torch.int64):
            ic = ic.to(torch.int64)
        ic = ic.contiguous()
        w1c, b1c, w2c = _f32c(w1.detach()), _f32c(b1.detach()), _f32c(w2.detach()).reshape(-1)
        x = torch.empty((n, H + n_actions + 1), dtype=torch.float32, device=h.device)
        act = torch.empty((n, H), dtype=torch.float32, device=h.device)
        q = torch.empty((n, 1), dtype=torch.float32, device=h.device)
        io = _native.QtakenIO()
        io.n_rows, io.H, io.A = n, H, int(n_actions)
        io.h, io.h_ld, io.idx, io.idx_elem_size, io.P = hc.data_ptr(), hc.stride(0), ic.data_ptr(), ic.element_size(), Pc.data_ptr()
        io.W1, io.w1_ld, io.b1, io.w2 = w1c.data_ptr(), w1c.stride(0), b1c.data_ptr(), w2c.data_ptr()
        io.b2 = _f32c(b2.detach()).data_ptr() if b2 is not None else None
        io.x, io.x_ld, io.act, io.act_ld, io.q = x.data_ptr(), x.stride(0), act.data_ptr(), act.stride(0), q.data_ptr()
        global _PAIRED_DQ
        pair, _PAIRED_DQ = _PAIRED_DQ, None
        with torch.cuda.device(h.device):
            if pair is not None:   # the update's Double-DQN launch rides in this grid (pair_double_q_with_next_taken)
                _native.check(lib.macjd_qheads_pair(ctypes.byref(io), ctypes.byref(pair[0]), _stream(h)), "macjd_qheads_pair")
            else:
                _native.check(lib.macjd_qhead_taken(ctypes.byref(io), _stream(h)), "macjd_qhead_taken")
        ctx.save_for_backward(x, w1, act, w2)
        ctx.has_b2, ctx.b1_key, ctx.b2_key = b2 is not None, grad_key(b1), grad_key(b2)
        return q

    @staticmethod
    def backward(ctx, gq):
        x, w1, y, w2 = ctx.saved_tensors
        nd = ctx.needs_input_grad
        gW1 = gb1 = gW2 = gb2 = None
        if nd[3] or nd[4]:   # the first layer's operand gq w2 masked by the ReLU is formed inside the weight-gradient kernel
            gW1, gb1 = linear_wgrad(y, x, want_bias=True, w_key=grad_key(w1), b_key=ctx.b1_key, need=(nd[3], nd[4]),
                                    outer=(gq, w2))
        if nd[5] or (ctx.has_b2 and nd[6]):
            gW2, gb2 = linear_wgrad(gq.reshape(-1, 1), y, want_bias=ctx.has_b2, w_key=grad_key(w2), b_key=ctx.b2_key,
                                    need=(nd[5], ctx.has_b2 and nd[6]))
        return None, None, None, gW1, gb1, gW2, (gb2 if ctx.has_b2 else None), None


def qhead_taken_supported(h, w1, w2, n_actions: int) -> bool:
    """The one-launch form of the taken-action Q-head applies: HIP device, float32, H = 64 hidden units in both the GRU and
    the Q-head, enough rows for the split-K weight gradients, autograd on (it exists for the learner's forward)."""
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and w1.dtype == torch.float32 and not torch.is_autocast_enabled()
            and h.shape[0] >= 1024 and w1.shape[0] == h.shape[1] and w1.shape[1] == h.shape[1] + n_actions + 1
            and w2.shape[0] == 1 and torch.is_grad_enabled() and (w1.requires_grad or w2.requires_grad) and not h.requires_grad
            and options.on("QHEAD_TAKEN")
            and bool(_native.load().macjd_qhead_taken_supported(int(h.shape[1]), int(n_actions))))


def qhead_taken(h, idx, P, w1, b1, w2, b2, n_actions: int):
    """Q(h, idx, P) [n, 1] through ``_QheadTaken`` (callers check ``qhead_taken_supported``)."""
    return _QheadTaken.apply(h, idx, P, w1, b1, w2, b2, int(n_actions))


def linear_relu_dot(x, w1, b1, w2, b2):
    """F.linear(relu(F.linear(x, w1, b1)), w2, b2) for a one-output second layer; one autograd node on a HIP device."""
    if (_fused_relu_ok(x, w1, b1) and torch.is_grad_enabled() and (w1.requires_grad or w2.requires_grad or x.requires_grad)
            and x.dim() == 2 and x.shape[0] >= 1024 and w1.shape[0] * w1.shape[1] <= 384 * 256 and x.stride(-1) == 1
            and w2.shape[0] == 1 and w1.shape[0] % 4 == 0 and 4 <= w1.shape[0] <= 1024 and not torch.is_autocast_enabled()):
        return _LinearReluRowDot.apply(x, w1, b1, w2, b2)
    return linear(linear_relu(x, w1, b1), w2, b2)


def linear(x, weight, bias=None):
    """torch.nn.functional.linear with the split-K weight gradient when it pays: HIP device, autograd on, float32
    (not under the optional bf16 autocast of the mixer), and a reduction of >= 1024 rows into a small weight."""
    if _rowdot_ok(x, weight):   # one output feature: row-dot kernel (with or without autograd)
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
            return _RowDot.apply(x, weight, bias)
        return _rowdot_launch(x, weight, bias)
    if (x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad))
            and x.dtype == torch.float32 and x.numel() // x.shape[-1] >= 1024 and weight.shape[0] * weight.shape[1] <= 384 * 256
            and x.stride(-1) == 1):
        return _LinearSplitK.apply(x, weight, bias)
    return F.linear(x, weight, bias)
