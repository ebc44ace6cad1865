"""Shared-parameter multi-agent controller for MP-DQN (reference core/mac.py:18-254).

Same public surface: ``BasicMAC(input_shape, args)``, ``select_actions(obs[E,J,S], avail[E,J,A], t_env,
test_mode) -> (int64 [E,J,1], float32 [E,J,1])``, ``forward``, ``init_hidden``, ``parameters``,
``load_state``, ``state_dict``, ``cuda``, ``save_models`` (``agent.pth``), ``load_models``; attributes
``agent``, ``hidden_states``, ``action_selector``, ``n_agents``, ``args``; deep-copyable
(core/qmix.py:53).

On a HIP device ``select_actions`` is: fc1/GRU/actor GEMMs (rocBLAS via torch) + ONE fused kernel for
the MP-DQN multi-pass Q-head over all actions, the availability mask, epsilon-greedy and the gather of
the chosen power (ops.qhead_select).  No host synchronisation; the chosen actions are also kept in
agent-major int32/float32 storage (``last_actions_T32`` / ``last_actions_P``) for the env-step kernel.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..utils.action_selectors import EpsilonGreedyActionSelector
from .networks import RNNAgent


class BasicMAC:
    def __init__(self, input_shape, args):
        self.n_agents = args.n_agents
        self.args = args
        self.input_shape = int(np.prod(input_shape)) if isinstance(input_shape, tuple) else input_shape
        self._build_agents(self.input_shape, args)
        self.action_selector = EpsilonGreedyActionSelector(args)
        self.hidden_states = None
        self.select_seed = int(getattr(args, "seed", 0) or 0)
        self._select_calls = 0
        self.last_actions_T32 = None  # int32 [E, J] view of agent-major storage (HIP path)
        self.last_actions_P = None    # float32 [E, J, 1] view of agent-major storage (HIP path)
        self.last_q_values = None     # [E, J, A] unmasked Q-values of the last call when keep_q_values is set
        self.keep_q_values = False
        # (eps float32[1], counter int64[1]) device tensors; when set, the fused select kernel reads the
        # exploration probability / Philox call counter from them (HIP-graph replay, see BatchedEpisodeRunner)
        self.device_schedule = None
        # optional (int32 [E,J,1], float32 [E,J,1]) destinations for the chosen actions of the NEXT select_actions call
        # on a HIP device (the batched runner points them at its staging rows: no copy, the env kernel reads them there)
        self.action_out = None
        # optional float32 [E,J,H] destination that also receives the post-update hidden state of the NEXT
        # select_actions call (the batched runner's staging row: no separate copy launch)
        self.hidden_out = None
        # (params_all [N, A], gi [N, 3H]) of observations that do not change during an episode, set by
        # prepare_static_obs(); while set, select_actions skips the actor and the fc1 -> W_ih chains
        self.static_inputs = None

    def prepare_static_obs(self, obs_batch):
        """The environment's observation is constant within an episode (the batched runner asks the env:
        ``observation_is_static``; reference simulation/environment.py:479-522 builds it from static scenario parameters
        only, :237-238 is a TODO): the actor output and the GRU input transform are then the same tensors at every
        step, so they are computed here ONCE per episode batch — and for ONE row when every (env, agent) row is the
        same vector (shared scenario: the observation is a stride-0 broadcast) — instead of at each of the
        episode_limit steps.  ``obs_batch`` [E, J, S]; pass None to go back to per-step evaluation."""
        if obs_batch is None:
            self.static_inputs, self._static_key = None, None
            return
        device = next(self.agent.parameters()).device
        obs = obs_batch.to(device) if obs_batch.device != device else obs_batch
        E, J, S = obs.shape
        # the same observation tensor and an unchanged agent body (everything but the Q-head: frozen in reference-faithful
        # training) give the same tensors as last time: nothing to launch (version counters see every in-place write)
        key = (obs.data_ptr(), obs._version, tuple(obs.shape), tuple(obs.stride()),
               tuple((p.data_ptr(), p._version) for n_, p in self.agent.named_parameters() if not n_.startswith("fc2_q_head")))
        capturing = obs.is_cuda and torch.cuda.is_current_stream_capturing()   # a captured rollout recomputes at every replay
        if self.static_inputs is not None and getattr(self, "_static_key", None) == key and not capturing:
            return
        self._static_key = None if capturing else key
        with torch.no_grad():
            if obs.stride(0) == 0 and obs.stride(1) == 0:            # one vector broadcast over envs and agents
                p1, g1 = self.agent.static_step_inputs(obs[0, 0].reshape(1, S))
                params, gi = p1.expand(E * J, -1), g1.expand(E * J, -1)     # stride-0 rows: no memory, no copy
            elif obs.stride(1) == 0:                                 # one vector per env (per-env scenarios)
                pe, ge = self.agent.static_step_inputs(obs[:, 0].reshape(E, S))
                params, gi = pe.repeat_interleave(J, dim=0), ge.repeat_interleave(J, dim=0)
            else:
                params, gi = self.agent.static_step_inputs(obs.reshape(E * J, S))
        self.static_inputs = (params, gi)

    def select_actions(self, obs_batch, avail_actions_batch, t_env, test_mode=False):
        device = next(self.agent.parameters()).device
        if obs_batch.device != device:
            obs_batch = obs_batch.to(device)
        if self.hidden_states is None:
            self.init_hidden(batch_size=obs_batch.shape[0])
        if self.hidden_states.device != device:
            self.hidden_states = self.hidden_states.to(device)
        batch_size = obs_batch.shape[0]
        obs_reshaped = obs_batch.reshape(-1, self.input_shape)
        with torch.no_grad():
            if self.static_inputs is not None and self.static_inputs[0].shape[0] == obs_reshaped.shape[0]:
                params_all, gi = self.static_inputs
                h_new = self.agent.step_from_gi(gi, self.hidden_states, h_out2=self.hidden_out)
            else:
                h_new, params_all = self.forward(obs_reshaped, self.hidden_states, h_out2=self.hidden_out)
            self.hidden_out = None
            self.hidden_states = h_new.detach()  # mac.py:107
            agent = self.agent
            H, A = agent.rnn_hidden_dim, agent.n_actions
            l1, l2 = agent.fc2_q_head[0], agent.fc2_q_head[2]
            base = F.linear(h_new, l1.weight[:, :H], l1.bias)
            eps = self.action_selector.anneal(t_env, test_mode)
            if base.is_cuda:
                self._select_calls += 1
                eps_dev, ctr_dev, counter = None, None, self._select_calls
                if self.device_schedule is not None:
                    eps_dev, ctr_dev, counter = self.device_schedule
                T64, P_sel, T32, Q = ops.qhead_select(
                    base, params_all, l1.weight, l2.weight, l2.bias, H, A, self.n_agents, avail_actions_batch,
                    epsilon=eps, greedy_only=test_mode, seed=self.select_seed, counter=counter,
                    want_q=self.keep_q_values, eps_dev=eps_dev, counter_dev=ctr_dev,
                    out_T32=self.action_out[0] if self.action_out else None,
                    out_P=self.action_out[1] if self.action_out else None)
                self.action_out = None
                self.last_actions_T32, self.last_actions_P = T32, P_sel
                self.last_q_values = Q.view(batch_size, self.n_agents, A) if Q is not None else None
                return T64, P_sel
            # host tensors: same algebra with stock torch ops (device-agnostic nets, like the reference)
            if avail_actions_batch.device != device:
                avail_actions_batch = avail_actions_batch.to(device)
            q_all = ops.qhead_all_actions_reference(base, params_all, l1.weight, l2.weight, l2.bias, H, A)
            agent_qs = q_all.view(batch_size, self.n_agents, A)
            self.last_q_values = agent_qs.clone() if self.keep_q_values else None
            agent_qs = agent_qs.masked_fill(avail_actions_batch == 0, -float("inf"))  # mac.py:142
            chosen = self.action_selector.select_action(agent_qs, avail_actions_batch, t_env, test_mode=test_mode)
            chosen_p = torch.gather(params_all, 1, chosen.view(-1, 1).long()).view(batch_size, self.n_agents, 1)
            self.last_actions_T32 = chosen.squeeze(-1).to(torch.int32)
            self.last_actions_P = chosen_p
            return chosen, chosen_p

    def forward(self, agent_inputs_reshaped, hidden_states, h_out2=None):
        """-> (h' [N, H], continuous params for all actions [N, A])  (mac.py:168-187)."""
        if not torch.is_grad_enabled():
            return self.agent.step_forward(agent_inputs_reshaped, hidden_states, h_out2=h_out2)
        h_out = self.agent.forward(agent_inputs_reshaped, hidden_states)
        if h_out2 is not None:
            h_out2.view(h_out.shape).copy_(h_out.detach())
        return h_out, self.agent.actor_forward(agent_inputs_reshaped)

    def init_hidden(self, batch_size):
        """zeros [batch * n_agents, H] on the agent's device (mac.py:189-198)."""
        device = next(self.agent.parameters()).device
        self.hidden_states = torch.zeros(batch_size * self.n_agents, self.args.rnn_hidden_dim, device=device)

    def parameters(self):
        return self.agent.parameters()

    def load_state(self, other_mac_state_dict):
        self.agent.load_state_dict(other_mac_state_dict)

    def state_dict(self):
        return self.agent.state_dict()

    def cuda(self):
        self.agent.cuda()

    def to(self, device):
        self.agent.to(device)
        return self

    def save_models(self, path):
        os.makedirs(path, exist_ok=True)
        torch.save(self.agent.state_dict(), f"{path}/agent.pth")

    def load_models(self, path):
        device = next(self.agent.parameters()).device
        self.agent.load_state_dict(torch.load(f"{path}/agent.pth", map_location=device, weights_only=True))

    def _build_agents(self, input_shape, args):
        self.agent = RNNAgent(input_shape, args)
