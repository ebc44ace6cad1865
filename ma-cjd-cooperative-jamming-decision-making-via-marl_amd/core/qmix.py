"""QMix + Double-DQN learner over whole episodes, MP-DQN agents (reference core/qmix.py:25-333).

Same public surface: ``QMixLearner(mac, args)``; ``train(batch, train_info) -> {'loss', 'grad_norm',
'eval_qtot_avg', 'target_qtot_avg'}``; ``_update_targets``; ``cuda``; ``save_models`` / ``load_models``
(``agent.pth``, ``qmix_net.pth``, ``optimizer.pth``); attributes ``mac``, ``target_mac``,
``eval_qmix_net``, ``target_qmix_net``, ``optimizer``, ``params``, ``train_step``.

Reference behaviour that is reproduced on purpose:
  * only the Q-head and the mixer ever receive gradients: ``q_taken`` is evaluated from the BUFFERED
    hidden states and continuous actions (qmix.py:161-184); the unrolled eval network feeds an argmax
    only (qmix.py:134-143), so actor / fc1 / GRU parameters keep ``grad is None`` for ever;
  * the Double-DQN argmax does not apply the availability mask (qmix.py:141-142 is commented out);
  * the loss runs over steps 0..T-2 (``[:, :-1]`` slices, qmix.py:155,192).

What changes is the schedule of the work, not the algebra.  The reference unrolls
``for t: for a:`` in Python — 2 x T x (3 + 5A) small launches per call (qmix.py:241-274).  Everything
except the GRU recurrence is time-parallel, so here: fc1 / actor / GRU input transform / Q-head base are
ONE GEMM each over all B*T*J rows, the recurrence is one fused scan (ops.gru_sequence), and the
all-action Q-head is one fused kernel (ops.qhead_all_actions).  The unroll is inference-only, so it runs
under ``no_grad``.

Multi-GPU: when ``torch.distributed`` is initialised with world_size > 1 the
trainable gradients live in ONE flat buffer that is all-reduced (mean) once per step over RCCL before
clipping; every rank then takes the identical Adam step.
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from .. import ops
from .networks import QMixer


def _to(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(x), device=device).to(dtype)


class QMixLearner:
    def __init__(self, mac, args):
        self.args = args
        self.mac = mac
        self.n_agents = args.n_agents
        self.n_actions = args.n_actions
        self.state_shape = args.state_shape
        self.obs_shape = args.obs_shape
        self.device = torch.device(args.device if torch.cuda.is_available() and args.use_cuda else "cpu")

        self.eval_qmix_net = QMixer(args)
        self.target_mac = copy.deepcopy(mac)
        self.target_qmix_net = QMixer(args)
        self.target_qmix_net.load_state_dict(self.eval_qmix_net.state_dict())
        if args.use_cuda:
            self.cuda()
        print(f"QMix Learner Initialized on device: {self.device}")

        self.agent_params = list(self.mac.parameters())
        self.qmix_params = list(self.eval_qmix_net.parameters())
        self.params = self.agent_params + self.qmix_params
        self.optimizer = optim.Adam(params=self.params, lr=args.lr)
        self.last_target_update_step = 0
        self.train_step = 0
        self._flat_grad = None
        self._flat_params = None

    # ------------------------------------------------------------------ distributed gradients
    def _trainable(self):
        """Parameters the loss can reach: the Q-head and the mixer (see module docstring)."""
        return list(self.mac.agent.fc2_q_head.parameters()) + self.qmix_params

    def _bind_flat_grads(self):
        """Make every trainable ``.grad`` a view into one flat buffer (a single all-reduce, no packing)."""
        tr = self._trainable()
        n = sum(p.numel() for p in tr)
        flat = torch.zeros(n, dtype=tr[0].dtype, device=tr[0].device)
        off = 0
        for p in tr:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._flat_grad, self._flat_params = flat, tr

    def _allreduce_grads(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
        self._flat_grad.div_(dist.get_world_size())

    # ------------------------------------------------------------------ the training step
    def train(self, batch, train_info=None, sync_stats=True):
        self.train_step += 1
        dev = self.device
        max_seq_len = int(batch["max_seq_len"])
        T = max_seq_len
        states = _to(batch["state"], torch.float32, dev)[:, :T]
        obs = _to(batch["obs"], torch.float32, dev)[:, :T]
        actions_discrete = _to(batch["actions_discrete"], torch.long, dev)[:, :T]
        actions_continuous = _to(batch["actions_continuous"], torch.float32, dev)[:, :T]
        rewards = _to(batch["reward"], torch.float32, dev)[:, :T]
        terminated = _to(batch["terminated"], torch.bool, dev)[:, :T]
        mask = _to(batch["filled"], torch.float32, dev).squeeze(-1)[:, :T]
        hidden_states = _to(batch["hidden_state"], torch.float32, dev)[:, :T + 1]
        B = states.shape[0]

        # ---- all-action Q for target and eval networks (inference only), qmix.py:129-134 ----
        # The reference's unroll leaves the ROLLOUT controller's hidden state at the sampled batch's
        # h_T (qmix.py:241,253); it only trains between episodes so nothing reads it.  Here training
        # may interleave with a running batched rollout, so the rollout state is put back.
        rollout_hidden = self.mac.hidden_states
        with torch.no_grad():
            target_q_all, eval_q_all = self._all_action_q_multi([self.target_mac, self.mac], obs)  # [B, T, J, A]
            self.mac.hidden_states = rollout_hidden
            next_actions = eval_q_all[:, 1:].argmax(dim=3, keepdim=True)            # qmix.py:138-143 (no mask)
            target_q_taken = torch.gather(target_q_all[:, 1:], 3, next_actions).squeeze(3)  # qmix.py:147
            target_q_mixer = self.target_qmix_net(target_q_taken, states[:, 1:])    # qmix.py:151
            targets = rewards[:, :-1] + self.args.gamma * (1 - terminated[:, :-1].float()) * target_q_mixer  # :155

        # ---- Q(s_t, a_t) from buffered hidden states / actions, qmix.py:161-184 ----
        if T <= 1:
            q_taken = torch.empty((B, 0, self.n_agents), device=dev)
        else:
            n_eff = B * (T - 1) * self.n_agents
            q_taken = self.mac.agent.get_q_value_for_action(
                hidden_states[:, :T - 1].reshape(n_eff, self.args.rnn_hidden_dim),
                actions_discrete[:, :T - 1].reshape(n_eff, 1),
                actions_continuous[:, :T - 1].reshape(n_eff, 1)).view(B, T - 1, self.n_agents)
        eval_q_mixer = self.eval_qmix_net(q_taken, states[:, :-1])                  # qmix.py:187

        td_error = eval_q_mixer - targets.detach()                                   # qmix.py:190-194
        m = mask[:, :-1]
        loss = ((td_error * m.unsqueeze(-1)) ** 2).sum() / m.sum()

        if self._flat_grad is None:
            self._bind_flat_grads()
            for p in self.params:  # parameters the graph never reaches keep grad None, as in the reference
                if not any(p is q for q in self._flat_params):
                    p.grad = None
        self._flat_grad.zero_()
        loss.backward()
        self._allreduce_grads()
        grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.args.grad_norm_clip)  # qmix.py:199
        self.optimizer.step()

        if (self.train_step - self.last_target_update_step) >= self.args.target_update_interval:  # qmix.py:203-205
            self._update_targets()
            self.last_target_update_step = self.train_step

        stats = {"loss": loss.detach(), "grad_norm": grad_norm.detach() if torch.is_tensor(grad_norm) else grad_norm,
                 "eval_qtot_avg": eval_q_mixer.detach().mean(), "target_qtot_avg": targets.mean()}
        if sync_stats:  # the reference returns Python floats (4 x .item(), qmix.py:209-214)
            stacked = torch.stack([torch.as_tensor(v, device=dev, dtype=torch.float32).reshape(()) for v in stats.values()])
            vals = stacked.tolist()  # one host sync instead of four
            stats = dict(zip(stats.keys(), vals))
        return stats

    def _all_action_q_multi(self, macs, obs):
        """Q(s_t, a, P_a(s_t)) for every discrete action, [B, T, J, A] per controller; replaces the
        per-step / per-action unroll of qmix.py:217-280 (the discarded ``params`` tensor is not built).
        Everything but the recurrence is time-parallel: one GEMM each for fc1, the GRU input transform,
        the actor layers and the Q-head base over all B*T*J rows; the recurrences of all controllers
        (eval + target) run in ONE fused scan launch.  Leaves each controller's ``hidden_states`` at the
        final h_T like the reference's loop does."""
        B, T, J, S = obs.shape
        rows = obs.reshape(B * T * J, S)
        gis = []
        for m in macs:
            a = m.agent
            x = F.relu(a.fc1(rows))                                          # networks.py:100
            gis.append(F.linear(x, a.rnn.weight_ih, a.rnn.bias_ih).view(B, T, J, 3 * a.rnn_hidden_dim))
        h_alls = ops.gru_sequence_multi(gis, [m.agent.rnn.weight_hh for m in macs],
                                        [m.agent.rnn.bias_hh for m in macs])  # h_0 = 0 (qmix.py:241)
        out = []
        for m, h_all in zip(macs, h_alls):
            a = m.agent
            H, A = a.rnn_hidden_dim, a.n_actions
            m.hidden_states = h_all[:, T - 1].reshape(B * J, H) if T > 0 else None
            params_all = a.actor(rows)                                       # networks.py:127
            l1, l2 = a.fc2_q_head[0], a.fc2_q_head[2]
            base = F.linear(h_all.reshape(B * T * J, H), l1.weight[:, :H], l1.bias)
            out.append(ops.qhead_all_actions(base, params_all, l1.weight, l2.weight, l2.bias, H, A).view(B, T, J, A))
        return out

    def _all_action_q(self, mac_controller, obs):
        return self._all_action_q_multi([mac_controller], obs)[0]

    def _get_all_action_q_values_and_params(self, mac_controller, batch, max_seq_len):
        """Reference-named entry (qmix.py:217-280): returns (Q [B,T,J,A], params [B,T,J,A])."""
        obs = _to(batch["obs"], torch.float32, self.device)[:, :max_seq_len]
        with torch.no_grad():
            q = self._all_action_q(mac_controller, obs)
            B, T, J, S = obs.shape
            params = mac_controller.agent.actor(obs.reshape(-1, S)).view(B, T, J, -1)
        return q, params

    def _update_targets(self):
        self.target_mac.load_state(self.mac.state_dict())
        self.target_qmix_net.load_state_dict(self.eval_qmix_net.state_dict())

    def cuda(self):
        self.mac.cuda()
        self.target_mac.cuda()
        self.eval_qmix_net.cuda()
        self.target_qmix_net.cuda()
        self.device = torch.device("cuda", torch.cuda.current_device())

    def save_models(self, path):
        os.makedirs(path, exist_ok=True)
        self.mac.save_models(path)
        torch.save(self.eval_qmix_net.state_dict(), f"{path}/qmix_net.pth")
        torch.save(self.optimizer.state_dict(), f"{path}/optimizer.pth")

    def load_models(self, path):
        """Agent + mixer (not the optimizer, like qmix.py:317-333), then hard-sync the targets."""
        self.mac.load_models(path)
        self.eval_qmix_net.load_state_dict(torch.load(f"{path}/qmix_net.pth", map_location=self.device, weights_only=True))
        self._update_targets()
