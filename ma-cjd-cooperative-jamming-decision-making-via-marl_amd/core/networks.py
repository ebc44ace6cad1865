"""Agent and mixer networks for the MP-DQN / QMix hot path on MI355X.

Same module tree and ``state_dict`` keys as the reference (core/networks.py:16-315) so checkpoints
interchange (``agent.pth`` / ``qmix_net.pth``):

  RNNAgent   actor.{0,2,4}.{weight,bias}  fc1.*  rnn.{weight_ih,weight_hh,bias_ih,bias_hh}
             fc2_q_head.{0,2}.*
  QMixer     state_norm.*  hyper_w_1.{0,2}.*  hyper_w_final.{0,2}.*  hyper_b_1.*  V.{0,2}.*

What is different is HOW the hot calls are evaluated:

* ``RNNAgent.q_values_all_actions`` computes Q(h, a, P_a) for ALL discrete actions in one pass.  The
  reference runs the Q-head once per action on ``cat([h, onehot(a), P_a])`` (core/mac.py:115-135,
  core/qmix.py:260-274): A x (full, one_hot, cat, Linear, ReLU, Linear).  The first layer is linear in
  its input, so  W1 [h; onehot(a); P_a] + b1 = (W_h h + b1) + W_a[:, a] + w_p P_a : the H x H product is
  done once and each action only adds a column and a rank-1 term.  On a HIP device the add/ReLU/dot
  epilogue runs in one fused kernel (csrc/macjd_nets.hip) that never materialises the [N, A, H]
  tensor; results match the per-action loop to ~1e-6 (different summation order).
* ``QMixer.forward`` keeps the reference arithmetic (LayerNorm -> 4 hyper-networks -> clamp ->
  bmm/ELU/bmm, core/networks.py:250-315); everything after the hyper-network GEMMs (4 clamps, the two
  tiny batched products, ELU) and its backward is one fused kernel each way on a HIP device
  (``ops.mixer_tail``), instead of ~12 / ~25 elementwise + reduction launches.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class RNNAgent(nn.Module):
    """Actor MLP + (fc1 -> GRUCell) + MP-DQN Q-head (reference core/networks.py:16-180)."""

    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.n_actions = args.n_actions
        self.input_shape = input_shape
        self.rnn_hidden_dim = args.rnn_hidden_dim
        self.actor_input_dim = input_shape
        self.actor_hidden_dim = args.actor_hidden_dim
        self.actor_output_dim = self.n_actions
        self.actor = nn.Sequential(  # networks.py:54-61
            nn.Linear(self.actor_input_dim, self.actor_hidden_dim), nn.ReLU(),
            nn.Linear(self.actor_hidden_dim, self.actor_hidden_dim), nn.ReLU(),
            nn.Linear(self.actor_hidden_dim, self.actor_output_dim), nn.Sigmoid())
        self.fc1 = nn.Linear(self.input_shape, self.rnn_hidden_dim)  # networks.py:65-66
        self.rnn = nn.GRUCell(self.rnn_hidden_dim, self.rnn_hidden_dim)
        q_head_input_dim = self.rnn_hidden_dim + self.n_actions + 1  # networks.py:73-79
        self.fc2_q_head = nn.Sequential(
            nn.Linear(q_head_input_dim, self.rnn_hidden_dim), nn.ReLU(),
            nn.Linear(self.rnn_hidden_dim, 1))

    def init_hidden(self):
        """zeros [1, H] on the parameters' device (networks.py:81-86)."""
        return self.fc1.weight.new_zeros(1, self.rnn_hidden_dim)

    # Which chains go through the fused MFMA kernel (ops.mlp_forward) on a HIP device under no_grad.  Measured on MI355X
    # (bench.py, episode-aligned accounting): actor 46-128-128-9 at 12 288 rows: 15.5 us fused vs 39 us as 3 library
    # GEMMs + 3 activation launches; fc1 + GRU input transform 46-64-192: one 14 us launch vs 6.6 + 4.8 + 11.1 us in the
    # rollout step's serial chain (rollout 0.0793 -> 0.0744 ms / step); the learner's 9 696-row time-parallel transform
    # of both controllers as ONE pair launch instead of four GEMMs: no gain while the update was fed to the GPU just in
    # time (0.502 vs 0.505 ms), 0.3636 -> 0.356 ms once the host ran ahead (3 runs each, +-0.1 %).
    # class-level switches (tests / A/B scripts set the attribute): the dense-chain kernel for the actor, for the learner's
    # time-parallel fc1 -> W_ih transform and for the rollout step's (RNNAgent.forward)
    fused_actor = True
    fused_gi = True
    fused_gi_step = True

    def _fused_ok(self, t):
        """Inference on a HIP device: the fused MFMA chain (ops.mlp_forward) replaces Linear + activation
        launches.  With autograd on, the stock modules run (the fused kernels have no backward)."""
        return t.is_cuda and not torch.is_grad_enabled()

    def gru_input_transform(self, agent_inputs, fused=None):
        """W_ih ReLU(fc1 obs) + b_ih, [N, 3H]: the time-parallel half of the GRU step (networks.py:100)."""
        layers = [(self.fc1.weight, self.fc1.bias, ops.ACT_RELU), (self.rnn.weight_ih, self.rnn.bias_ih, ops.ACT_NONE)]
        if (self.fused_gi if fused is None else fused) and self._fused_ok(agent_inputs):
            return ops.mlp_forward(agent_inputs, layers)
        return ops.mlp_reference(agent_inputs, layers)

    def forward(self, agent_inputs, h_in, h_out2=None):
        """h' = GRUCell(ReLU(fc1 obs), h)  (networks.py:88-114).  ``h_out2``: optional second destination of h' on
        the HIP inference path (the batched runner's staging row)."""
        if self._fused_ok(agent_inputs) and agent_inputs.dim() == 2:
            gi = self.gru_input_transform(agent_inputs, fused=self.fused_gi_step)
            gh = F.linear(h_in.to(gi.device), self.rnn.weight_hh, self.rnn.bias_hh)
            return ops.gru_gates(gi, gh, h_in, out2=h_out2)                   # gates + both stores: one launch
        x = F.relu(self.fc1(agent_inputs))
        if h_in.device != x.device:
            h_in = h_in.to(x.device)
        return self.rnn(x.contiguous(), h_in.contiguous())

    def actor_layers(self):
        a = self.actor
        return [(a[0].weight, a[0].bias, ops.ACT_RELU), (a[2].weight, a[2].bias, ops.ACT_RELU),
                (a[4].weight, a[4].bias, ops.ACT_SIGMOID)]

    def gi_layers(self):
        return [(self.fc1.weight, self.fc1.bias, ops.ACT_RELU), (self.rnn.weight_ih, self.rnn.bias_ih, ops.ACT_NONE)]

    def step_forward(self, agent_inputs, h_in, h_out2=None):
        """One rollout step of the agent: (h' [N, H], continuous params for all actions [N, A]).  On the HIP inference
        path the actor chain and the fc1 -> W_ih chain — both read the same observation rows — are ONE launch."""
        x = agent_inputs
        if self._fused_ok(x) and x.dim() == 2 and self.fused_actor and self.fused_gi_step:
            params, gi = ops.mlp_forward_pair(x, self.actor_layers(), x, self.gi_layers())
            gh = F.linear(h_in.to(gi.device), self.rnn.weight_hh, self.rnn.bias_hh)
            return ops.gru_gates(gi, gh, h_in, out2=h_out2), params
        if h_out2 is not None and self._fused_ok(x):
            h = self.forward(x, h_in, h_out2=h_out2)
        else:
            h = self.forward(x, h_in)
            if h_out2 is not None:
                h_out2.view(h.shape).copy_(h.detach())
        return h, self.actor_forward(x)

    def static_step_inputs(self, rows):
        """(continuous params for all actions [n, A], gi = W_ih ReLU(fc1 obs) + b_ih [n, 3H]) for observation rows that
        stay the same for a whole episode: everything of the agent step that does not depend on the hidden state
        (networks.py:100,127).  One launch on the HIP inference path (the two chains read the same rows)."""
        if self._fused_ok(rows) and rows.dim() == 2 and self.fused_actor and self.fused_gi_step:
            return ops.mlp_forward_pair(rows, self.actor_layers(), rows, self.gi_layers())
        return self.actor_forward(rows), self.gru_input_transform(rows, fused=self.fused_gi_step)

    def step_from_gi(self, gi, h_in, h_out2=None):
        """h' of one GRUCell step given the precomputed input transform gi (see static_step_inputs): the recurrent
        GEMM + the gates launch."""
        gh = F.linear(h_in.to(gi.device), self.rnn.weight_hh, self.rnn.bias_hh)
        return ops.gru_gates(gi, gh, h_in, out2=h_out2)

    def actor_forward(self, inputs):
        """Continuous parameter for EVERY discrete action, [N, A] in (0,1)  (networks.py:116-129)."""
        if self.fused_actor and self._fused_ok(inputs):
            a = self.actor
            out = ops.mlp_forward(inputs.reshape(-1, inputs.shape[-1]),
                                  [(a[0].weight, a[0].bias, ops.ACT_RELU), (a[2].weight, a[2].bias, ops.ACT_RELU),
                                   (a[4].weight, a[4].bias, ops.ACT_SIGMOID)])
            return out.view(*inputs.shape[:-1], out.shape[-1])
        return self.actor(inputs)

    def get_q_value_for_action(self, hidden_state, discrete_action_index, continuous_param, validate=True):
        """Q(h, T, P) for one (action, parameter) per row (networks.py:131-180).  Differentiable; this
        is the only path through which the learner's loss reaches the agent (core/qmix.py:161-184).
        """
        n = hidden_state.shape[0]
        idx = discrete_action_index
        if idx.dim() > 1 and idx.shape[1] == 1:
            idx = idx.squeeze(1)
        # the bounds check reads the device back (host sync); ``validate=False`` is for trusted indices
        # inside a captured HIP graph (replay-buffer actions produced by select_actions)
        if validate and idx.numel() and (torch.any(idx < 0) or torch.any(idx >= self.n_actions)):  # networks.py:157-158
            raise IndexError(f"Action index out of bounds: {idx}, n_actions: {self.n_actions}")
        if continuous_param.dim() == 1:
            continuous_param = continuous_param.unsqueeze(1)
        elif continuous_param.dim() > 2 or (continuous_param.dim() == 2 and continuous_param.shape[1] != 1):
            try:
                continuous_param = continuous_param.view(n, 1)
            except RuntimeError:
                raise ValueError(f"Unexpected continuous_param shape: {continuous_param.shape}, expected ({n}, 1)")
        l1, l2 = self.fc2_q_head[0], self.fc2_q_head[2]
        # cat([h, onehot(a), P]) -> Linear -> ReLU -> Linear, exactly the reference's formulation (networks.py:171-176);
        # the one-hot is a compare + cast (any integer dtype), the Linears use the split-K weight gradient on a HIP device
        if ops.qhead_taken_supported(hidden_state, l1.weight, l2.weight, self.n_actions):
            # the learner's case on a HIP device: input rows, first layer + ReLU and the second layer's dot in ONE launch
            return ops.qhead_taken(hidden_state, idx, continuous_param, l1.weight, l1.bias, l2.weight, l2.bias, self.n_actions)
        q_head_input = ops.qhead_input(hidden_state, idx, continuous_param, self.n_actions)   # [h, onehot(a), P]
        return ops.linear_relu_dot(q_head_input, l1.weight, l1.bias, l2.weight, l2.bias)

    def q_values_all_actions(self, hidden_state, continuous_params_all):
        """Q(h, a, P[:, a]) for all a at once, [N, A].  Inference path (no autograd): replaces the
        reference's per-action loop (core/mac.py:115-135, core/qmix.py:260-274)."""
        H = self.rnn_hidden_dim
        l1, l2 = self.fc2_q_head[0], self.fc2_q_head[2]
        w_h = l1.weight[:, :H]
        base = F.linear(hidden_state, w_h, l1.bias)  # [N, H], the one H x H product
        return ops.qhead_all_actions(base, continuous_params_all, l1.weight, l2.weight, l2.bias, H, self.n_actions)


class QMixer(nn.Module):
    """QMix monotonic mixer with clamped hyper-network outputs (reference core/networks.py:182-315)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.n_agents = args.n_agents
        self.state_dim = int(np.prod(args.state_shape))
        self.embed_dim = args.mixing_embed_dim
        self.hyper_hidden_dim = args.hyper_hidden_dim
        self.state_norm = nn.LayerNorm(self.state_dim)  # networks.py:215
        self.hyper_w_1 = nn.Sequential(  # networks.py:223-248
            nn.Linear(self.state_dim, self.hyper_hidden_dim), nn.ReLU(),
            nn.Linear(self.hyper_hidden_dim, self.n_agents * self.embed_dim))
        self.hyper_w_final = nn.Sequential(
            nn.Linear(self.state_dim, self.hyper_hidden_dim), nn.ReLU(),
            nn.Linear(self.hyper_hidden_dim, self.embed_dim * 1))
        self.hyper_b_1 = nn.Linear(self.state_dim, self.embed_dim)
        self.V = nn.Sequential(
            nn.Linear(self.state_dim, self.embed_dim), nn.ReLU(),
            nn.Linear(self.embed_dim, 1))
        # BASELINE.json config 5 ("bf16 mixer MFMA path"): run the hyper-network GEMMs with bf16 inputs and
        # fp32 accumulation (bf16 MFMA).  Off by default: the reference is fp32 and the 1e-5 tolerance on
        # Q_tot only holds in fp32; LayerNorm and the clamp / ELU tail stay fp32 either way.
        self.bf16_hyper = str(getattr(args, "mixer_dtype", "fp32")).lower() in ("bf16", "bfloat16")
        self._merged_views = None   # (W_cat, b_cat) views of the learner's flat parameter vector (HIP device)
        self._cat_cache = None      # persistent concatenated first-layer weights of an inference-only copy
        self._cat_hooked = False

    def first_layer_params(self):
        """The four hyper-networks' FIRST Linear layers in merge order: weights, then biases."""
        mods = (self.hyper_w_1[0], self.hyper_w_final[0], self.V[0], self.hyper_b_1)
        return [m.weight for m in mods] + [m.bias for m in mods]

    def _refresh_first_layer_cache(self):
        if self._cat_cache is not None:
            with torch.no_grad():
                ps = self.first_layer_params()
                torch.cat([p.detach() for p in ps[:4]], dim=0, out=self._cat_cache[0])
                torch.cat([p.detach() for p in ps[4:]], out=self._cat_cache[1])

    def enable_first_layer_cache(self):
        """Inference-only networks (the target mixer): keep the concatenated first-layer weights in persistent
        buffers, refilled in place whenever ``load_state_dict`` runs (the target sync) — same addresses, so a
        captured HIP graph keeps seeing current values."""
        ps = self.first_layer_params()
        self._cat_cache = (torch.cat([p.detach() for p in ps[:4]], dim=0), torch.cat([p.detach() for p in ps[4:]]))
        if not self._cat_hooked:
            self.register_load_state_dict_post_hook(lambda module, incompatible: module._refresh_first_layer_cache())
            self._cat_hooked = True

    def _first_layer(self, s):
        """The four hyper-networks all read the same input, so their FIRST layers run as ONE GEMM over the
        concatenated weights [Hh + Hh + Em + Em, S] (forward, input-gradient, weight-gradient and bias-gradient
        each 4 -> 1 launches).  Same dot products as the four separate nn.Linear calls of the reference
        (networks.py:283-299).  The concatenation itself costs nothing when the learner has laid the parameters
        out adjacently in its flat vector (``_merged_views``) or, for an inference-only copy, cached it."""
        if self._merged_views is not None and s.is_cuda and not torch.is_autocast_enabled():
            w_cat, b_cat = self._merged_views
            if torch.is_grad_enabled():
                return ops.merged_linear(s, w_cat, b_cat, self.first_layer_params())
            return F.linear(s, w_cat, b_cat)
        if self._cat_cache is not None and not torch.is_grad_enabled() and s.device == self._cat_cache[0].device:
            return F.linear(s, self._cat_cache[0], self._cat_cache[1])
        ps = self.first_layer_params()
        return ops.linear(s, torch.cat(ps[:4], dim=0), torch.cat(ps[4:]))

    def _hyper_networks(self, s):
        """(w1_raw [M,J*Em], b1_raw [M,Em], wf_raw [M,Em], v_raw [M,1]) from the normalised state."""
        return self._hyper_tail(self._first_layer(s))           # [M, 2 Hh + 2 Em] -> the four heads

    def _hyper_tail(self, out):
        Hh, Em = self.hyper_hidden_dim, self.embed_dim
        # the three ReLUs are one launch over the first 2 Hh + Em columns, b1_raw is the remaining column block; the
        # backward of the whole split / ReLU / split is one launch too
        # (ops.split_relu); the V head's one-output second layer rides in the same node (its backward is an outer product
        # folded into that launch)
        h_w1, h_wf, v_raw, b1_raw = ops.split_relu(out, [Hh, Hh, Em], Em, dot=(2, self.V[2].weight, self.V[2].bias))
        w1_raw = ops.linear(h_w1, self.hyper_w_1[2].weight, self.hyper_w_1[2].bias)
        wf_raw = ops.linear(h_wf, self.hyper_w_final[2].weight, self.hyper_w_final[2].bias)
        return w1_raw, b1_raw, wf_raw, v_raw

    def hyper_outputs(self, states):
        """The state-only half of the mixer: LayerNorm + the four hyper-networks -> (w1_raw, b1_raw, wf_raw, v_raw).
        It does not depend on the agents' Q-values, so a caller may evaluate it early / on another stream and hand
        the result to :meth:`forward` through ``hyper=``."""
        ln = self.state_norm
        x = states.reshape(-1, self.state_dim)
        if (self._merged_views is not None and x.is_cuda and torch.is_grad_enabled() and not x.requires_grad
                and not self.bf16_hyper and ln.weight.requires_grad):
            # training path on a HIP device: LayerNorm + merged first layer as one autograd node whose backward gets
            # the LayerNorm's parameter gradients from the weight-gradient products (ops.norm_merged_linear)
            w_cat, b_cat = self._merged_views
            out = ops.norm_merged_linear(x, ln.weight, ln.bias, ln.eps, w_cat, b_cat, self.first_layer_params())
            return self._hyper_tail(out)
        s = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(self.bf16_hyper and s.is_cuda)):
            raw = self._hyper_networks(s)
        if self.bf16_hyper and s.is_cuda:
            raw = tuple(r.float() for r in raw)
        return raw

    # the whole mixer as one MFMA chain per direction (ops.mixer_fused, csrc/macjd_mixer.hip); off -> LayerNorm + library
    # GEMMs + the tail kernel (also the path of sizes the fused kernel does not cover and of the bf16 option)
    fused = True   # class-level switch (tests set it): the one-launch mixer where the size is covered

    def fused_available(self, t) -> bool:
        return (self.fused and t.is_cuda and not self.bf16_hyper and not torch.is_autocast_enabled()
                and ops.mixer_fused_supported(self.n_agents, self.state_dim, self.hyper_hidden_dim, self.embed_dim))

    def _first_layer_cat(self):
        """(W_cat [2Hh+2Em, S], b_cat) of the merged first layer: the flat-parameter views, the inference cache, or a
        fresh concatenation."""
        if self._merged_views is not None:
            return self._merged_views
        if self._cat_cache is not None and not torch.is_grad_enabled():
            return self._cat_cache
        ps = self.first_layer_params()
        return torch.cat([p.detach() for p in ps[:4]], dim=0), torch.cat([p.detach() for p in ps[4:]])

    def _forward_fused(self, agent_qs, states):
        q = agent_qs.reshape(-1, self.n_agents)
        s = states.reshape(-1, self.state_dim)
        ln, w_cat_b = self.state_norm, self._first_layer_cat()
        l2 = (self.hyper_w_1[2].weight, self.hyper_w_1[2].bias, self.hyper_w_final[2].weight, self.hyper_w_final[2].bias,
              self.V[2].weight, self.V[2].bias)
        if torch.is_grad_enabled() and (q.requires_grad or ln.weight.requires_grad or l2[0].requires_grad):
            return ops.mixer_fused(q, s, ln.weight, ln.bias, ln.eps, w_cat_b[0], w_cat_b[1], *l2, self.first_layer_params())
        params = ops._mixerf_params(ln.weight, ln.bias, ln.eps, w_cat_b[0], w_cat_b[1], *l2)
        return ops.mixer_fused_forward(q, s, params, save=False)[0]

    def forward_paired_with_next_fused(self, agent_qs, states):
        """No-grad forward whose launch rides in the NEXT differentiable fused-mixer forward (another QMixer of the same
        shape on the same number of rows — the learner's eval mixer): ``ops.pair_mixer_forward_with_next_fused``.  Callers
        check ``fused_available``.  The returned Q_tot is valid once that forward has run."""
        q = agent_qs.reshape(-1, self.n_agents)
        s = states.reshape(-1, self.state_dim)
        ln, w_cat_b = self.state_norm, self._first_layer_cat()
        params = ops._mixerf_params(ln.weight, ln.bias, ln.eps, w_cat_b[0], w_cat_b[1], self.hyper_w_1[2].weight,
                                    self.hyper_w_1[2].bias, self.hyper_w_final[2].weight, self.hyper_w_final[2].bias,
                                    self.V[2].weight, self.V[2].bias)
        y = ops.pair_mixer_forward_with_next_fused(q, s, params)
        q_tot = y.view(agent_qs.size(0), -1, 1)
        return q_tot.squeeze(1) if q_tot.shape[1] == 1 else q_tot

    def forward(self, agent_qs, states, hyper=None):
        """Q_tot = ELU(q . clamp(W1(s),0,5) + clamp(b1(s),-5,5)) . clamp(Wf(s),0,5) + clamp(V(s),-5,5)
        on the LayerNorm-ed state (networks.py:250-315; note clamp, not abs).  Output shape follows
        the reference: [B, T, 1] for [B, T, J] inputs, [B, 1] for [B, J]."""
        batch_size = agent_qs.size(0)
        if hyper is None and self.fused_available(agent_qs):
            y = self._forward_fused(agent_qs, states)
            q_tot = y.view(batch_size, -1, 1)
            return q_tot.squeeze(1) if q_tot.shape[1] == 1 else q_tot
        raw = hyper if hyper is not None else self.hyper_outputs(states)
        q = agent_qs.reshape(-1, self.n_agents)
        # clamp -> bmm -> ELU -> bmm (+ its backward) is one fused kernel on a HIP device (ops.mixer_tail)
        y = ops.mixer_tail(q, *raw)
        q_tot = y.view(batch_size, -1, 1)
        if q_tot.shape[1] == 1:
            q_tot = q_tot.squeeze(1)
        return q_tot
