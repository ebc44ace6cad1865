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

from .. import hipgraph, ops, options
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
        # capturable: the step counter lives on the device, so the update can sit inside a HIP graph
        self.optimizer = optim.Adam(params=self.params, lr=args.lr, capturable=(self.device.type == "cuda"))  # (HIP: the fused step below is used instead)
        self.last_target_update_step = 0
        self.train_step = 0
        self._flat_grad = None   # flat gradient vector (all-reduce buffer / input of the fused optimiser step)
        self._sample_rng = np.random.default_rng(int(getattr(args, "seed", 0) or 0))   # train_from_buffer's episode sampler
        self.grad_pack_launches = 0   # updates whose gradients had to be packed into the flat vector by a copy
        self._flat_param = None
        if self.device.type == "cuda":
            self._flatten_trainable()
        self._body_versions = None
        self._mark_body_shared()   # target_mac is a deep copy of mac (qmix.py:53)

    # ------------------------------------------------------------------ frozen agent body
    @staticmethod
    def _body_params(agent):
        """fc1 / GRU / actor: the parameters the loss never reaches (module docstring)."""
        return list(agent.fc1.parameters()) + list(agent.rnn.parameters()) + list(agent.actor.parameters())

    def _mark_body_shared(self):
        """Record that the two controllers' bodies hold identical values right now (after the deep copy in the
        constructor and after every hard target sync).  ``_body_is_shared`` stays true until any of those tensors is
        written again (their autograd version counters move; a write through ``p.data`` does NOT move them — enable_graphs
        therefore compares the values, _verify_body_shared): another optimiser training the body, a checkpoint loaded
        into one controller only, ...  The reference's own learner never changes them (qmix.py:161-184: the loss only
        reaches fc2_q_head and the mixer), so in reference-faithful training the eval and target unrolls share one
        input transform, one GRU scan and one actor chain for ever."""
        pe, pt = self._body_params(self.mac.agent), self._body_params(self.target_mac.agent)
        self._body_versions = [(p, p._version, q, q._version) for p, q in zip(pe, pt)]

    def _verify_body_shared(self):
        """The version counters miss writes made through ``p.data`` (or any other alias created before the mark).  Where it
        is cheap — enable_graphs, not the update — the VALUES are compared: returns whether the two bodies are identical
        now and brings the bookkeeping in line with that (marks them shared again, or forgets the mark)."""
        pe, pt = self._body_params(self.mac.agent), self._body_params(self.target_mac.agent)
        same = len(pe) == len(pt) and all(p.shape == q.shape and torch.equal(p.detach(), q.detach()) for p, q in zip(pe, pt))
        if same:
            self._mark_body_shared()
        else:
            self._body_versions = None
        return same and self._body_is_shared()

    def _body_is_shared(self):
        if not options.on("SHARED_BODY") or not self._body_versions:
            return False
        pe = self._body_params(self.mac.agent)
        if len(pe) != len(self._body_versions):
            return False
        return all(p is p0 and p._version == vp and q._version == vq
                   for p, (p0, vp, q, vq) in zip(pe, self._body_versions))

    # ------------------------------------------------------------------ distributed gradients
    def _trainable(self):
        """Parameters the loss can reach: the Q-head and the mixer (see module docstring), in the order of the
        flat vectors: the mixer's four first-layer weights, then their biases (adjacent, so that the merged
        first-layer GEMM reads them as ONE [2Hh+2Em, S] matrix without a torch.cat), then everything else."""
        first = self.eval_qmix_net.first_layer_params()
        ids = {id(p) for p in first}
        rest = [p for p in list(self.mac.agent.fc2_q_head.parameters()) + self.qmix_params if id(p) not in ids]
        return first + rest

    @staticmethod
    def _world_size():
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _flatten_trainable(self):
        """HIP device: make the trainable parameters (Q-head + mixer) views of ONE flat vector, with flat Adam
        moments, so clipping + Adam is a fused two-kernel step (ops.clip_adam_step) instead of ~20 foreach
        launches.  ``torch.optim.Adam`` keeps owning the state in its usual layout (per-parameter ``step`` /
        ``exp_avg`` / ``exp_avg_sq`` entries, here views of the flat vectors), so ``optimizer.pth`` and
        ``load_state_dict`` keep the reference's format (qmix.py:300-315)."""
        tr = self._trainable()
        # every parameter starts on a 16-byte boundary of the flat vector (float4 / MFMA-fragment loads of the kernels
        # that read weights in place); the padding elements are zero, get zero gradients and stay zero under Adam.
        # The merged first layer (4 weights, then 4 biases, all of 16-byte-multiple size at the supported shapes) stays
        # contiguous: it is read as ONE matrix.
        self._flat_offsets, off = [], 0
        for p in tr:
            self._flat_offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(off, dtype=tr[0].dtype, device=tr[0].device)
        for p, o in zip(tr, self._flat_offsets):
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
        self._flat_param = flat
        self._flat_exp_avg = torch.zeros_like(flat)
        self._flat_exp_avg_sq = torch.zeros_like(flat)
        self._adam_step = torch.zeros((), dtype=torch.float32, device=flat.device)
        self._grad_norm = torch.zeros((), dtype=torch.float32, device=flat.device)
        self._adam_partials = torch.zeros(256 + 1024, dtype=torch.float32, device=flat.device)   # + LayerNorm partials
        for p, off in zip(tr, self._flat_offsets):
            n = p.numel()
            p.data = flat[off:off + n].view_as(p)
            self.optimizer.state[p] = {"step": self._adam_step, "exp_avg": self._flat_exp_avg[off:off + n].view_as(p),
                                       "exp_avg_sq": self._flat_exp_avg_sq[off:off + n].view_as(p)}
        # gradients: ONE flat vector too (the all-reduce buffer); the weight-gradient kernels write their results
        # straight into its slices (ops.deferred_wgrad(grad_dst=...)), so nothing packs the gradients afterwards
        self._flat_grad = torch.zeros_like(flat)
        self._grad_dst = {}
        for p, off in zip(tr, self._flat_offsets):
            n = p.numel()
            self._grad_dst[ops.grad_key(p)] = self._flat_grad[off:off + n].view_as(p)
        # merged first layer of the eval mixer = the leading block of the flat vector; the target mixer (inference
        # only) keeps a cached concatenation that its load_state_dict refreshes in place
        mx = self.eval_qmix_net
        first = mx.first_layer_params()
        rows, S = sum(p.shape[0] for p in first[:4]), first[0].shape[1]
        assert self._flat_offsets[8] == rows * S + rows, "merged first layer must be contiguous in the flat vector"
        mx._merged_views = (flat[:rows * S].view(rows, S), flat[rows * S:rows * S + rows])
        # the merged views alias the first parameter of their block: same address, different size -> different key
        self._grad_dst[ops.grad_key(mx._merged_views[0])] = self._flat_grad[:rows * S].view(rows, S)
        self._grad_dst[ops.grad_key(mx._merged_views[1])] = self._flat_grad[rows * S:rows * S + rows]
        self.target_qmix_net.enable_first_layer_cache()

    def load_optimizer_state(self, state_dict):
        """``optimizer.load_state_dict`` + re-binding of the loaded moments into the flat vectors."""
        self.optimizer.load_state_dict(state_dict)
        if self._flat_param is None:
            return
        for p, off in zip(self._trainable(), self._flat_offsets):
            n = p.numel()
            st = self.optimizer.state.get(p, {})
            if "exp_avg" in st:
                self._flat_exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self._flat_exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self._adam_step.copy_(torch.as_tensor(st["step"], dtype=torch.float32).reshape(()))
            self.optimizer.state[p] = {"step": self._adam_step, "exp_avg": self._flat_exp_avg[off:off + n].view_as(p),
                                       "exp_avg_sq": self._flat_exp_avg_sq[off:off + n].view_as(p)}

    def grad_vector(self):
        """Current gradients of the trainable parameters as one flat vector (a copy unless already flat)."""
        if self._flat_grad is not None:
            return self._flat_grad
        return torch.cat([p.grad.reshape(-1) for p in self._trainable()])

    def _flatten_grads(self):
        """Every ``.grad`` becomes a view of ONE flat vector (the all-reduce buffer; clip + Adam read it).  On the
        graphed HIP path the weight-gradient kernels have already written into it (``_grad_dst``) and this is a
        host-side check only; gradients that stock autograd produced (small batches, CPU) are packed with one cat."""
        tr = self._trainable()
        if self._flat_grad is None:
            self._flat_grad = torch.empty(sum(p.numel() for p in tr), dtype=tr[0].dtype, device=tr[0].device)
        for p in tr:   # an empty loss (max_seq_len <= 1) reaches nothing: the reference then steps on NaN / None
            if p.grad is None:   # gradients (qmix.py:190-200); here the missing ones count as zero
                p.grad = torch.zeros_like(p)
        offs = getattr(self, "_flat_offsets", None)
        padded = offs is not None
        if not padded:
            offs, off = [], 0
            for p in tr:
                offs.append(off)
                off += p.numel()
        base, esz = self._flat_grad.data_ptr(), self._flat_grad.element_size()
        in_place = sum(int(p.grad.data_ptr() == base + off * esz and p.grad.is_contiguous()) for p, off in zip(tr, offs))
        if in_place == len(tr):
            return
        self.grad_pack_launches += 1
        if padded:   # 16-byte aligned slots: one small copy per parameter (fallback path only: small batches)
            packed = [p.grad.reshape(-1).clone() for p in tr]
            for g_, off in zip(packed, offs):
                self._flat_grad[off:off + g_.numel()].copy_(g_)
        elif in_place == 0:
            torch.cat([p.grad.reshape(-1) for p in tr], out=self._flat_grad)
        else:   # some already live in the buffer: pack through a temporary (cat must not read what it overwrites)
            self._flat_grad.copy_(torch.cat([p.grad.reshape(-1) for p in tr]))
        for p, off in zip(tr, offs):
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)

    def _ar_capturable(self):
        """The gradient all-reduce can be recorded into a HIP graph: an initialised RCCL process group."""
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"

    def _allreduce_grads(self, force=False):
        import torch.distributed as dist
        if self._world_size() <= 1 and not (force and self._ar_capturable()):   # (force: a one-rank group still issues it)
            return
        if dist.get_backend() == "nccl":   # RCCL averages in the collective: no separate divide launch
            dist.all_reduce(self._flat_grad, op=dist.ReduceOp.AVG)
        else:                              # gloo (CPU tests) has no AVG
            dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
            self._flat_grad.div_(dist.get_world_size())

    # ------------------------------------------------------------------ the training step
    def _forward_backward(self, batch, T, validate_actions=True):
        """Loss + gradients for one batch of device/host arrays (no optimiser step, no host sync when
        ``validate_actions`` is off).  Returns (loss, eval_q_mixer mean, targets mean) as 0-dim tensors."""
        dev = self.device
        states = _to(batch["state"], torch.float32, dev)[:, :T]
        obs = _to(batch["obs"], torch.float32, dev)[:, :T]
        actions_discrete = _to(batch["actions_discrete"], torch.long, dev)[:, :T]
        actions_continuous = _to(batch["actions_continuous"], torch.float32, dev)[:, :T]
        rewards = _to(batch["reward"], torch.float32, dev)[:, :T]
        terminated = _to(batch["terminated"], torch.bool, dev)[:, :T]
        filled = _to(batch["filled"], torch.bool, dev)[:, :T]
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

        # ---- Q(s_t, a_t) from buffered hidden states / actions, qmix.py:161-184 ----
        if T <= 1:
            q_taken = torch.empty((B, 0, self.n_agents), device=dev)
        else:
            n_eff = B * (T - 1) * self.n_agents
            q_taken = self.mac.agent.get_q_value_for_action(
                hidden_states[:, :T - 1].reshape(n_eff, self.args.rnn_hidden_dim),
                actions_discrete[:, :T - 1].reshape(n_eff, 1),
                actions_continuous[:, :T - 1].reshape(n_eff, 1), validate=validate_actions).view(B, T - 1, self.n_agents)
        eval_q_mixer = self.eval_qmix_net(q_taken, states[:, :-1])                  # qmix.py:187

        # targets = r + gamma (1 - terminated) Q_tot'(s', a')            qmix.py:155
        # loss = sum((filled (Q_tot - targets))^2) / sum(filled)          qmix.py:190-194   (one fused launch)
        loss, eval_mean, target_mean = ops.td_loss(eval_q_mixer, target_q_mixer, rewards[:, :-1], terminated[:, :-1],
                                                   filled[:, :-1], self.args.gamma)

        for p in self.params:   # autograd then ASSIGNS fresh gradients (no accumulate-add kernels, no memset);
            p.grad = None       # parameters the graph never reaches keep grad None, as in the reference
        loss.backward()
        if self._world_size() > 1 or self._flat_param is not None:
            self._flatten_grads()   # one cat: the all-reduce buffer and the fused optimiser's input
        return loss.detach(), eval_mean.detach(), target_mean.detach()

    def _scan_from_ring_early(self):
        """Static observations: launch the scan (with its in-kernel input transform) on the side stream BEFORE the
        gather — it reads the sampled episodes' step-0 observation rows straight from the replay ring through the index
        tensor, so the longest kernel of the update starts at time zero.  Returns the per-agent hidden states (side stream)."""
        dev = self.device
        origin = torch.cuda.current_stream(dev)
        if getattr(self, "_target_stream", None) is None:
            self._target_stream = torch.cuda.Stream(device=dev)
        ts = self._target_stream
        ts.wait_stream(origin)                                                                  # fork
        shared = self._body_is_shared()
        agents = [self.mac.agent] if shared else [self.target_mac.agent, self.mac.agent]
        with torch.cuda.stream(ts), torch.no_grad():
            # (the frozen actor chain of each sequence's observation row rides in the scan launch's prologue as well.
            # With the K-split scan (65 us) the prologue was on the update's critical path and the separate actor launch
            # on the origin stream was the better place; since the unit-split scan (27 us) the origin stream is the
            # longer branch: 0.220 -> 0.192 ms / step.  MACJD_ACTOR_IN_SCAN=0 keeps the separate launch.)
            res = ops.gru_sequence_from_obs(self._g_buffer.buffers["obs"], self._g_idx, agents, self._g_B, self.n_agents,
                                            self._g_T + 1, with_actor=self._g_actor_in_scan)
            h, ps = res if self._g_actor_in_scan else (res, None)
        if not self._g_actor_in_scan:
            return ([h[0], h[0]] if shared else h), None
        return ([h[0], h[0]] if shared else h), ([ps[0], ps[0]] if shared else ps)

    def _forward_backward_full(self, st, T, pre_scan=None, pre_actor=None, prefetched=None, after_join=None):
        """Same update as ``_forward_backward`` for a batch of FULL-LENGTH episodes held in contiguous staging
        tensors with T+1 steps on every key (actions padded with a zero row).  Every quantity is evaluated for
        all T+1 steps and the loss kernel picks the steps it needs through strides, so no slice of a [B,T+1,...]
        tensor is ever reshaped (= copied) and autograd sees no slicing (no zero-fill + copy in backward).
        Rows that the reference never evaluates (eval step T-1 and T, target step 0) cost ~2 % extra arithmetic,
        receive zero gradient and do not enter the loss: the result equals ``_forward_backward`` (tested).

        ``prefetched`` = (gather_done, scan_done) events: ``st`` was gathered and ``pre_scan`` / ``pre_actor`` were
        computed on the side stream during the PREVIOUS update (enable_graphs, pipelined group) — no fork here, the
        origin stream waits for the two events where it needs the data.  ``after_join()`` is called right behind the join:
        the place from which the next update's batch is prefetched."""
        B, T1 = st["state"].shape[0], T + 1
        J, H, A = self.n_agents, self.args.rnn_hidden_dim, self.n_actions
        n = B * T1 * J
        dev = st["state"].device
        macs = [self.target_mac, self.mac]
        rows = st["obs"].reshape(n, -1)
        heads = [(m.agent.fc2_q_head[0].weight, m.agent.fc2_q_head[2].weight, m.agent.fc2_q_head[2].bias) for m in macs]

        # Frozen agent body (fc1 / GRU / actor identical in both controllers, see _body_is_shared): ONE input transform,
        # ONE scan and ONE actor chain serve both networks; only the Q-heads differ.
        shared = self._body_is_shared()
        body = macs[1].agent   # the network whose body is evaluated when shared
        fused_dq = ops.qhead_double_q_fused_supported(rows, H, A)
        # Static observations (the replay buffer knows: every stored episode came from an env whose observation does not
        # change within an episode): everything of the agent that reads only the observation — the fc1 -> W_ih input
        # transform and the actor chain — is evaluated on the B * J step-0 rows instead of on all B * (T + 1) * J rows;
        # the scan takes the one input transform per sequence at every step.
        obs_static = bool(getattr(self, "_g_obs_static", False))
        Tg = 1 if obs_static else T1

        def rows_static():   # (a strided slice: the reshape is a copy launch — made on the stream and at the point where
            return st["obs"][:, 0].reshape(B * J, -1) if obs_static else rows   # it is needed, not up front)

        def actor_all(agent):
            p = agent.actor_forward(rows_static())                                             # networks.py:127
            return p.view(B, 1, J, A).expand(B, T1, J, A).reshape(n, A) if obs_static else p

        # Two streams inside the (captured) update (kernel timeline: scripts/timeline_update.py; the measured
        # alternatives are tabulated in DESIGN.md 4.8):
        #   side stream   (no grad) fc1 -> W_ih (dense-chain launch) -> the fused scan (~55 us latency chain) -> the
        #                 Q-head base GEMMs [-> the target controller's actor chain when the bodies differ]
        #   this stream   (no grad) the (eval) actor chain -> (autograd) Q-head on the STORED hidden states + eval
        #                 mixer -> (no grad) the target mixer's hyper-networks (they read only the state)
        #   ONE join, then on this stream the two Double-DQN Q-head launches, the target mixer tail and the loss.
        # Autograd only ever sees this stream.  Forks taken from a forked stream crash hipStreamEndCapture (ROCm 7.2):
        # every fork hangs off the capture's origin stream.  MACJD_UPDATE_STREAMS=1 runs everything on one stream.
        def scan_chain():
            ns = T1 if obs_static else None
            if pre_scan is not None:
                h_alls = pre_scan        # already running on the side stream (_scan_from_ring_early)
            elif obs_static and rows.is_cuda and getattr(self, "_g_scan_from_ring", False):
                # the scan launch computes each sequence's input transform itself, from the sampled episodes' step-0
                # observation rows in the replay ring: no fc1 / W_ih launch
                ring = self._g_buffer.buffers["obs"]
                h_alls = ops.gru_sequence_from_obs(ring, self._g_idx, [body] if shared else [m.agent for m in macs], B, J, T1)
                if shared:
                    h_alls = [h_alls[0], h_alls[0]]
            elif shared:
                gis = [body.gru_input_transform(rows_static()).view(B, Tg, J, 3 * H)]                 # networks.py:100
                h = ops.gru_sequence_multi(gis, [body.rnn.weight_hh], [body.rnn.bias_hh], n_steps=ns)[0]   # h_0 = 0, qmix.py:241
                h_alls = [h, h]
            else:
                a0, a1 = macs[0].agent, macs[1].agent
                if rows.is_cuda and a0.fused_gi and a1.fused_gi and not torch.is_grad_enabled():
                    # fc1 -> ReLU -> W_ih of both controllers: one launch of the dense-chain kernel
                    gis = [g.view(B, Tg, J, 3 * H) for g in ops.mlp_forward_pair(rows_static(), a0.gi_layers(), rows_static(), a1.gi_layers())]
                else:
                    gis = [m.agent.gru_input_transform(rows_static()).view(B, Tg, J, 3 * H) for m in macs]
                h_alls = ops.gru_sequence_multi(gis, [m.agent.rnn.weight_hh for m in macs],
                                                [m.agent.rnn.bias_hh for m in macs], n_steps=ns)
            if fused_dq:   # the Double-DQN launch takes the hidden states themselves (its base products run on MFMA)
                return [h.reshape(n, H) for h in h_alls]
            return [F.linear(h.reshape(n, H), hd[0][:, :H], m.agent.fc2_q_head[0].bias)
                    for m, h, hd in zip(macs, h_alls, heads)]

        def double_q(bases, params, paired=False):
            # a* = argmax_a Q_eval (no mask, qmix.py:138-143), Q_target(a*) (qmix.py:147), [B,T+1,J]: one launch from the
            # hidden states, or two Q-head launches on library-GEMM bases
            if fused_dq:
                hd4 = [(hd[0], m.agent.fc2_q_head[0].bias, hd[1], hd[2]) for m, hd in zip(macs, heads)]
                # P per sequence [B, J, A] (static observation): row n = (b, t, j) reads P[b, j]
                pmap = (T1 * J, J) if (pre_actor is not None and params[1].shape[0] != n) else None
                launch = ops.pair_double_q_with_next_taken if paired else ops.qhead_double_q_from_h
                return launch(bases[1], params[1], hd4[1], bases[0], params[0], hd4[0], H, A, p_row_map=pmap).view(B, T1, J)
            return ops.qhead_double_q(bases[1], params[1], heads[1], bases[0], params[0], heads[0], H, A).view(B, T1, J)

        def eval_forward():
            q_taken = self.mac.agent.get_q_value_for_action(
                st["hidden_state"].view(n, H), st["actions_discrete"].view(n, 1), st["actions_continuous"].view(n, 1),
                validate=False).view(B, T1, J)                                                  # qmix.py:161-184
            return self.eval_qmix_net(q_taken, st["state"])                                    # [B,T+1,1], qmix.py:187

        two_streams = dev.type == "cuda" and options.get("UPDATE_STREAMS") != "1"
        # A prefetched update (pipelined group) has the scan's outputs at its very start, so its whole TARGET branch —
        # Double-DQN launch and target mixer, which read the scan, the Q-heads and the target mixer but nothing of the eval
        # head — runs on the side stream BESIDE the eval head (taken-action Q-head + eval mixer on this stream) instead
        # of behind it; the two meet at the TD loss.  (-22 us of a 147 us update.)
        target_beside_head = (two_streams and prefetched is not None and fused_dq and pre_actor is not None
                              and self.target_qmix_net.fused_available(st["state"]))
        # ... or, better, not beside it but INSIDE its launches: both Q-head launches as one grid and both mixers as one
        # grid on this stream (ops.pair_*): the chain no longer crosses hardware queues to meet the target branch, and
        # stays on one queue from the previous Adam to this one.
        paired_heads = target_beside_head and self._paired_heads_ok(st, T)
        if paired_heads:
            origin = torch.cuda.current_stream(dev)
            # Side-stream work the caller wants beside this update (after_join: the next update's prefetch) forks from this
            # stream early, but its launches are captured behind this update's backward (_finish_update): the graph runtime
            # keeps a node on its predecessor's hardware queue only if it is that predecessor's FIRST captured dependent,
            # and every queue change on the chain costs ~10 us.
            # (Measured and dropped: every prefetch of a group up front, one staging set per update, so that the chain never
            # signals the side stream — the runtime then runs ALL side-stream nodes before the chain's first one.)
            side_work = None
            fork_at = torch.cuda.Event()
            if after_join is not None:
                side_work = lambda: after_join(fork_at)
            if prefetched[1] is not None:        # (None: a wait earlier on this stream already covers this update's prefetch)
                origin.wait_event(prefetched[1])  # the prefetched scan (hence the gather before it on that stream) is there
            with torch.no_grad():
                bases = scan_chain()
                p_eval = pre_actor[1]
                params = [p_eval if shared else pre_actor[0], p_eval]
                for t_ in list(bases) + [p for p in params if p is not None]:
                    t_.record_stream(origin)
                tq_agents = double_q(bases, params, paired=True)                                # launched with the next call
            q_taken = self.mac.agent.get_q_value_for_action(
                st["hidden_state"].view(n, H), st["actions_discrete"].view(n, 1), st["actions_continuous"].view(n, 1),
                validate=False).view(B, T1, J)                                                  # qmix.py:138-147, 161-184
            if after_join is not None:
                # (the fork: behind the Q-head grid rather than in front of it — the prefetch's scan then runs beside the
                # weight-gradient launches instead of the two mixer launches, which it slows more: 118.2 -> 116.7 us;
                # behind the mixer grid the prefetch is late for the next update: 120.7 us)
                fork_at.record(origin)
            with torch.no_grad():
                target_q_tot = self.target_qmix_net.forward_paired_with_next_fused(tq_agents, st["state"])
            eval_q_tot = self.eval_qmix_net(q_taken, st["state"])                               # qmix.py:151, 187
            ops.assert_pairs_launched()
            return self._finish_update(st, T, eval_q_tot, target_q_tot, tot_m=prefetched[2] if len(prefetched) > 2 else None,
                                       side_work=side_work, sums_in_backward=True)
        elif target_beside_head:
            origin = torch.cuda.current_stream(dev)
            ts = self._target_stream
            ts.wait_stream(origin)           # (fork) the previous update's Adam has written the Q-head this branch reads
            with torch.cuda.stream(ts), torch.no_grad():
                # scan outputs, actor rows and the gathered batch were produced on this very stream: stream order
                bases = scan_chain()
                p_eval = pre_actor[1]
                params = [p_eval if shared else pre_actor[0], p_eval]
                target_q_tot = self.target_qmix_net(double_q(bases, params), st["state"])       # qmix.py:138-151
                target_done = torch.cuda.Event()
                target_done.record(ts)
            if after_join is not None:
                # the NEXT update's prefetch goes straight behind the target branch on the side stream: the index tensor
                # and the other staging set are free already (their last readers ran earlier on that stream / before the
                # previous Adam), so it needs no wait for this stream and overlaps the eval head and the loss rather than
                # the weight-gradient launch (0.1461 -> 0.1440 ms)
                after_join(False)
            origin.wait_event(prefetched[0])                                                    # the gathered batch is there
            eval_q_tot = eval_forward()
            origin.wait_event(target_done)                                                      # join
            target_q_tot.record_stream(origin)
            return self._finish_update(st, T, eval_q_tot, target_q_tot, tot_m=prefetched[2] if len(prefetched) > 2 else None)
        elif two_streams:
            origin = torch.cuda.current_stream(dev)
            if getattr(self, "_target_stream", None) is None:
                self._target_stream = torch.cuda.Stream(device=dev)
            ts = self._target_stream
            if prefetched is None:
                ts.wait_stream(origin)                                                          # fork
                with torch.cuda.stream(ts), torch.no_grad():
                    bases = scan_chain()
                    p_target = None if (shared or pre_actor is not None) else actor_all(macs[0].agent)
            else:
                assert pre_scan is not None and pre_actor is not None
                origin.wait_event(prefetched[0])                                                # the gathered batch is there
                bases, p_target = scan_chain(), None                                            # (views of pre_scan: no launch)
            if pre_actor is not None:      # [B, J, A] per controller, from the scan launch's prologue: [target, eval]
                p_target, p_eval = (None if shared else pre_actor[0]), pre_actor[1]
            eval_q_tot = eval_forward()
            if pre_actor is None:          # only the Double-DQN launches behind the join read it: issued after the eval
                with torch.no_grad():      # forward, whose autograd chain is the longer part of this stream
                    p_eval = actor_all(body)
            with torch.no_grad():
                # unfused mixer: its state-only half (LayerNorm + hyper-networks) runs here, before the join; the fused
                # mixer is ONE launch that needs the target Q-values, i.e. it runs behind the join
                hyper = None if self.target_qmix_net.fused_available(st["state"]) else self.target_qmix_net.hyper_outputs(st["state"])
                if prefetched is None:
                    origin.wait_stream(ts)                                                      # join
                else:
                    origin.wait_event(prefetched[1])                                            # join (the scan ended long ago)
                if after_join is not None:
                    after_join()
                for t_ in list(bases) + ([p_target] if p_target is not None else []) + ([p_eval] if pre_actor is not None else []):
                    t_.record_stream(origin)
                if pre_actor is not None and not fused_dq:
                    # the two-launch Double-DQN form wants one row per (b, t, j): expanded HERE, behind the join — the
                    # rows come from the side stream's scan launch
                    ex = lambda p_: p_.view(B, 1, J, A).expand(B, T1, J, A).reshape(n, A)
                    p_target, p_eval = (None if p_target is None else ex(p_target)), ex(p_eval)
                params = [p_eval if shared else p_target, p_eval]
                target_q_tot = self.target_qmix_net(double_q(bases, params), st["state"], hyper=hyper)   # qmix.py:151
        else:
            with torch.no_grad():
                bases = scan_chain()
                p_eval = actor_all(body)
                params = [p_eval if shared else actor_all(macs[0].agent), p_eval]
                hyper = None if self.target_qmix_net.fused_available(st["state"]) else self.target_qmix_net.hyper_outputs(st["state"])
                target_q_tot = self.target_qmix_net(double_q(bases, params), st["state"], hyper=hyper)
            eval_q_tot = eval_forward()
        return self._finish_update(st, T, eval_q_tot, target_q_tot)

    def _paired_heads_ok(self, st, T):
        """A prefetched update can take its target branch as paired launches on the chain's stream (ops.pair_*): HIP device,
        two streams, the one-launch forms of both Q-heads and both mixers apply, MACJD_PAIRED_HEADS not 0.  (us / update,
        paired vs side-stream target branch: 3j/4r 111 vs 117, 6j/8r 179 vs 185, 12j/16r 318 vs 328 — at 12 agents only
        since the 33-action Double-DQN body stopped spilling: before, the pair was the slower form there, 388 vs 375.)"""
        B, T1 = st["state"].shape[0], T + 1
        J, H, A = self.n_agents, self.args.rnn_hidden_dim, self.n_actions
        n = B * T1 * J
        head = self.mac.agent.fc2_q_head
        return (st["state"].is_cuda and options.get("UPDATE_STREAMS") != "1" and options.on("PAIRED_HEADS")
                and ops.qhead_double_q_fused_supported(st["obs"].reshape(n, -1), H, A)
                and self.target_qmix_net.fused_available(st["state"]) and self.eval_qmix_net.fused_available(st["state"])
                and ops.qhead_taken_supported(st["hidden_state"].view(n, H), head[0].weight, head[2].weight, A))

    def _finish_update(self, st, T, eval_q_tot, target_q_tot, tot_m=None, side_work=None, sums_in_backward=False):
        # loss over eval steps 0..T-2 against targets built from target steps 1..T-1 (qmix.py:155,190-194)
        for p in self.params:
            p.grad = None
        if eval_q_tot.is_cuda:
            # the loss kernel also produces dL/dQ_tot: it seeds the backward pass directly (no ones-fill / multiply).
            # (Measured and dropped: the loss inside the eval mixer's backward launch — its loads and reductions in front of
            # the kernel's chain cost the 7 us the separate launch does: 26.2 vs 7.5 + 17.8 us.)
            stats_done = stats_branch = None
            if tot_m is not None and ops.fused_mixer_backward_will_run(eval_q_tot):
                # pipelined update: the batch's mask sum was computed behind its gather, so the eval mixer's backward
                # launch forms dL/dQ_tot itself (5 loads per row) and the loss launch leaves the serial chain: it still runs
                # — for the logged sums — on the side stream, ordered before the optimiser writes the gradient norm into
                # the same row
                origin = torch.cuda.current_stream(eval_q_tot.device)
                ts = self._target_stream
                head_done = torch.cuda.Event()
                if not sums_in_backward:
                    head_done.record(origin)

                def stats_branch():
                    with torch.cuda.stream(ts):
                        ts.wait_event(head_done)
                        out = ops.td_loss_and_grad(eval_q_tot, target_q_tot, st["reward"], st["terminated"], st["filled"],
                                                   self.args.gamma, T - 1, 1)
                        done = torch.cuda.Event()
                        done.record(ts)
                    return out, done

                if not sums_in_backward:
                    (loss, eval_mean, target_mean, _, self._last_stats4), stats_done = stats_branch()
                else:
                    # paired update: the logged sums feed nothing, so they get neither a launch nor a stream of their own —
                    # one extra workgroup of the mixer's backward launch computes them (the optimiser writes the gradient
                    # norm into the same row later)
                    row = self._last_stats4 = torch.empty(4, dtype=torch.float32, device=eval_q_tot.device)
                    loss, eval_mean, target_mean = row[0], row[1], row[2]
                gy = ops.td_grad_in_mixer_backward(eval_q_tot, target_q_tot, st["reward"], st["terminated"], st["filled"],
                                                   self.args.gamma, T - 1, 1, tot_m, stats_row=row if sums_in_backward else None)
            else:
                loss, eval_mean, target_mean, gy, self._last_stats4 = ops.td_loss_and_grad(
                    eval_q_tot, target_q_tot, st["reward"], st["terminated"], st["filled"], self.args.gamma, T - 1, 1)
            # the weight gradients: one grouped launch pair after the chain, written into the flat gradient vector
            # (single process: the LayerNorm-parameter launch behind the grouped products is held back and evaluated
            # inside the optimiser step's squared-norm launch — with more ranks the all-reduce needs it done first)
            hold = self._flat_param is not None and self._world_size() <= 1 and options.on("LN_IN_SQNORM")
            with ops.deferred_wgrad(grad_dst=getattr(self, "_grad_dst", None), hold_lnparam=hold) as dw:
                eval_q_tot.backward(gy)
            assert ops._PENDING_TD is None, "the fused mixer's backward did not take the TD loss's inputs"
            self._held_ln = dw.held
            if side_work is not None:
                # side-stream launches whose forks lie earlier (side_work: its own fork; the logged sums: head_done),
                # captured behind the chain's launches so that the chain's nodes stay first dependents of each other
                side_work()
            if stats_done is not None:
                torch.cuda.current_stream(eval_q_tot.device).wait_event(stats_done)
                for t_ in (self._last_stats4,):
                    t_.record_stream(torch.cuda.current_stream(eval_q_tot.device))
        else:
            loss, eval_mean, target_mean = ops.td_loss_full(eval_q_tot, target_q_tot, st["reward"], st["terminated"],
                                                            st["filled"], self.args.gamma, T - 1, 1)
            loss.backward()
        if self._world_size() > 1 or self._flat_param is not None:
            self._flatten_grads()
        return loss.detach(), eval_mean.detach(), target_mean.detach()

    def _clip_and_step(self, sample_next=None):
        if self._flat_param is not None:   # HIP device: fused clip_grad_norm_ + Adam on the flat vectors
            g = self.optimizer.param_groups[0]
            held, self._held_ln = getattr(self, "_held_ln", None), None
            ops.clip_adam_step(self._flat_param, self._flat_grad, self._flat_exp_avg, self._flat_exp_avg_sq,
                               self._adam_step, self._grad_norm, self._adam_partials, g["lr"], g["betas"], g["eps"],
                               self.args.grad_norm_clip, sample_next=sample_next, lnparam=held)
            return self._grad_norm
        grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.args.grad_norm_clip)  # qmix.py:199
        self.optimizer.step()
        return grad_norm.detach() if torch.is_tensor(grad_norm) else torch.as_tensor(grad_norm)

    def _after_step(self):
        if (self.train_step - self.last_target_update_step) >= self.args.target_update_interval:  # qmix.py:203-205
            self._update_targets()
            self.last_target_update_step = self.train_step

    def _pack_stats(self, loss, grad_norm, ev, tg, sync_stats):
        stats = {"loss": loss, "grad_norm": grad_norm, "eval_qtot_avg": ev, "target_qtot_avg": tg}
        if sync_stats:  # the reference returns Python floats (4 x .item(), qmix.py:209-214)
            stacked = torch.stack([torch.as_tensor(v, device=self.device, dtype=torch.float32).reshape(())
                                   for v in stats.values()])
            stats = dict(zip(stats.keys(), stacked.tolist()))  # one host sync instead of four
        return stats

    def train(self, batch, train_info=None, sync_stats=True):
        """One QMix update on a batch dict (reference signature, core/qmix.py:76-215)."""
        self.train_step += 1
        loss, ev, tg = self._forward_backward(batch, int(batch["max_seq_len"]))
        self._allreduce_grads()
        grad_norm = self._clip_and_step()
        self._after_step()
        if not sync_stats and torch.is_tensor(grad_norm) and grad_norm is self._grad_norm_tensor():
            grad_norm = grad_norm.clone()   # the fused optimiser step reuses ONE output tensor: hand out a snapshot
        return self._pack_stats(loss, grad_norm, ev, tg, sync_stats)

    def _grad_norm_tensor(self):
        return getattr(self, "_grad_norm", None)

    # ------------------------------------------------------------------ HIP-graph path
    def enable_graphs(self, buffer, batch_size, warmup_iters=3, force_two_graphs=False, updates_per_graph=None,
                      graphed_allreduce=None, prime=True):
        """Capture the update as two HIP graphs around the (eager) gradient all-reduce (ONE graph holding both halves
        when there is a single process, i.e. nothing to all-reduce):
          graph A  gather the sampled episodes from the device replay (static index tensor) + both
                   unrolls + mixers + loss + backward                      (~300 launches -> 1 graph launch)
          graph B  gradient clipping + Adam + the four logged scalars
        Valid while every stored episode has the full ``episode_limit`` length (true for this env: episodes
        only end at the limit, environment.py:460); ``train_from_buffer`` falls back to the eager path
        otherwise.  The replay tensors, parameters, gradients and optimiser state keep their addresses, so
        target syncs (in-place ``load_state_dict``) and new episodes are seen by the replayed graphs."""
        if self.device.type != "cuda":
            raise RuntimeError("enable_graphs needs the learner on a HIP device")
        self.release_graphs()   # a re-capture destroys the previous graphs first, explicitly and at a quiet point
        self._verify_body_shared()   # (values, not version counters: catches writes made through .data)
        same_buffer = getattr(self, "_g_buffer", None) is buffer
        self._g_buffer, self._g_B, self._g_T = buffer, int(batch_size), int(buffer.episode_limit)
        # static observations in every stored episode (see _forward_backward_full): baked into the captured launches
        self._g_obs_static = bool(getattr(buffer, "obs_static", False)) and options.on("LEARNER_STATIC_OBS")
        # the actor rows of each sequence's observation ride in the scan launch's prologue (MACJD_ACTOR_IN_SCAN=0: the
        # actor chain as its own launch on the origin stream)
        self._g_actor_in_scan = options.on("ACTOR_IN_SCAN")
        self._g_scan_from_ring = self._g_obs_static and int(buffer.buffers["obs"].shape[-1]) <= 3 * int(self.args.rnn_hidden_dim)
        # (a re-capture with the same batch size keeps the index tensor: the batch the last update drew for the next one is
        # still in it, so the sequence of updates continues as if nothing had been re-captured)
        prev_idx = getattr(self, "_g_idx", None)
        keep_idx = same_buffer and prev_idx is not None and prev_idx.numel() == self._g_B and prev_idx.device == self.device
        self._g_idx = prev_idx if keep_idx else torch.zeros(self._g_B, dtype=torch.int64, device=self.device)
        # The batch of an update whose caller passes no indices is drawn ON THE DEVICE by the previous update's last
        # launch (ops.sample_episodes: uniform without replacement over the stored episodes, like the reference's
        # np.random.choice in buffer.sample): no index upload between two replayed updates — that copy and its two
        # stream-order hops were ~10 us of a ~190 us step.  MACJD_DEVICE_SAMPLER=0: host draw + upload.
        self._g_dev_sampler = self._flat_param is not None and options.on("DEVICE_SAMPLER")
        if not keep_idx:
            self._g_n_stored = torch.zeros(1, dtype=torch.int32, device=self.device)
        if getattr(self, "_g_draws", None) is None:   # draws made so far (the sampler's counter; survives a re-capture)
            self._g_draws = torch.full((1,), int(getattr(self, "_resume_draws", 0)), dtype=torch.int64, device=self.device)
        if not keep_idx:
            self._g_pop_seen = None       # (store_count, current_size) the device-side population scalar / last draw refer to
            self._g_idx_fresh = False     # _g_idx holds a device draw from the current population that no update has used yet
        self._g_idx_ring = [(torch.zeros(self._g_B, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        if buffer.current_size < 1:
            raise RuntimeError("enable_graphs: the replay buffer is empty")

        keys = [k for k in buffer.buffers if k != "avail_actions"]   # the update never reads the mask (qmix.py:141-142)
        srcs = [buffer.buffers[k] for k in keys]
        fused = ops.gather_rows_supported(srcs) and self._g_T >= 2
        if fused:
            # persistent staging tensors, every key with T+1 steps (the action rows get a zero step T) so that the
            # full-length update never slices; ONE launch gathers all keys' episodes
            T1 = self._g_T + 1
            padded = ("actions_discrete", "actions_continuous")   # T-step keys that the update reshapes with T+1 rows
            stage = {k: torch.zeros((self._g_B, T1 if k in padded else v.shape[1]) + tuple(v.shape[2:]), dtype=v.dtype,
                                    device=v.device) for k, v in zip(keys, srcs)}

        def body_a():
            if fused:
                pre = (None, None)
                if self._g_scan_from_ring and options.get("UPDATE_STREAMS") != "1":
                    pre = self._scan_from_ring_early()
                ops.gather_rows(self._g_idx, srcs, [stage[k] for k in keys])
                return self._forward_backward_full(stage, self._g_T, pre_scan=pre[0], pre_actor=pre[1])
            b = {k: v.index_select(0, self._g_idx) for k, v in zip(keys, srcs)}
            return self._forward_backward(b, self._g_T, validate_actions=False)

        self._graph_body_a = body_a   # kept for scripts/profile_update.py (eager attribution of the captured launches)

        # Eager warm-up on a side stream: allocates the optimiser state BEFORE capture (state created during
        # capture would be re-initialised by every replay) and lets the libraries pick their kernels.  The
        # warm-up updates are then undone in place, so enabling graphs does not change the training state.
        was_shared = self._body_is_shared()
        snap_p = [p.detach().clone() for p in self.params]
        snap_o = {id(p): {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
                  for p, st in self.optimizer.state.items()}
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup_iters)):
                body_a()
                self._clip_and_step()   # (no draw: the warm-up leaves the sampler's counter alone)
        torch.cuda.current_stream(self.device).wait_stream(s)
        def restore_training_state():
            with torch.no_grad():
                for p, sp in zip(self.params, snap_p):
                    p.copy_(sp)
                for p, st in self.optimizer.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            old = snap_o.get(id(p), {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
            if was_shared:
                self._mark_body_shared()   # the restore wrote the (unchanged) body values back: still identical

        restore_training_state()
        self._graph_a, self._graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # a single process has no all-reduce between the two halves: one graph, one launch per update
        # (force_two_graphs: the layout every rank of a multi-GPU job runs, for single-process tests)
        # With ranks (RCCL): MACJD_GRAPHED_ALLREDUCE=1 captures the gradient all-reduce INSIDE the update graph —
        # torch.distributed records collectives issued on a capturing stream into the capture (the collective runs on the
        # process group's own stream, joined to the capture by events) — so ranks replay ONE graph per update, or per
        # group of K updates, like a single process does, instead of graph A -> eager all-reduce -> graph B.  Off by
        # default: no multi-GPU node was available to run it on; only its capture / replay
        # mechanics are tested, with a one-rank RCCL group (tests/test_dist_gpu.py).
        self._g_graphed_ar = bool(graphed_allreduce if graphed_allreduce is not None
                                  else options.get("GRAPHED_ALLREDUCE") == "1") and self._ar_capturable()
        self._g_single = (self._world_size() <= 1 and not force_two_graphs) or self._g_graphed_ar
        with hipgraph.capture(self._graph_a):
            self._g_out_a = body_a()
            # the four logged scalars of an update end up in ONE static [4] tensor: the loss kernel writes (loss,
            # mean Q_tot, mean target, mask sum) and the optimiser step then overwrites the unused mask sum with the
            # gradient norm — train_from_buffer(stats_row=...) snapshots it with a single 16-byte copy
            st4 = getattr(self, "_last_stats4", None) if fused else None
            if st4 is not None and st4.numel() == 4:
                self._grad_norm = st4[3]
            self._g_stats4 = st4 if (st4 is not None and st4.numel() == 4) else None
            nxt = (self._g_idx, self._g_n_stored, self._g_draws, self._sampler_seed()) if self._g_dev_sampler else None
            if self._g_single:
                if self._g_graphed_ar:
                    self._allreduce_grads(force=True)
                self._g_out_b = self._clip_and_step(sample_next=nxt)
        if not self._g_single:
            with hipgraph.capture(self._graph_b, pool=self._graph_a.pool()):
                self._g_out_b = self._clip_and_step(sample_next=nxt)
        # K consecutive updates as ONE graph (train_from_buffer_many): between two replayed graphs the stream pays a
        # launch-to-launch hand-over (~20 us here) that an edge inside a graph does not.  Needs the device-side draw (every
        # update's batch comes from the previous update's last launch) and a single process (no all-reduce in between).
        K = int(updates_per_graph if updates_per_graph is not None else options.get("UPDATES_PER_GRAPH"))
        self._g_multi = None
        if K > 1 and self._g_single and self._g_dev_sampler and fused and self._g_stats4 is not None:
            gm, rows, single_norm = torch.cuda.CUDAGraph(), [], self._grad_norm
            # Inside the group the updates are software-pipelined: everything of update k + 1 that depends on neither the
            # weights nor update k's results — the draw of its episodes, the gather into the OTHER staging set, the scan
            # launch (frozen body) — is issued on the side stream right behind update k's join and runs beside update k's
            # serial tail; update k + 1 then starts at its taken-action Q-head and finds the scan done at its join
            # (-17 us per pipelined update: the gather, its hand-over and the wait for the scan leave the chain).  Same
            # launches on the same data in an order that respects every dependence: same results (tested bitwise).
            pipelined = (self._g_scan_from_ring and self._g_actor_in_scan and options.get("UPDATE_STREAMS") != "1"
                         and options.on("PIPELINED_GROUP"))
            stage2 = {k: torch.zeros_like(v) for k, v in stage.items()} if pipelined else None
            stages = [stage, stage2]
            # (the captured launches write and read the second staging set through baked addresses: it has to live as long
            # as the graph.  Until round 3 nothing held it once this function returned — the freed blocks stayed cached, so
            # it went unnoticed until another capture's torch.cuda.empty_cache() unmapped them: a GPU memory fault in the
            # first replay of the group at 12j/16r.)
            self._g_stages = stages
            origin_dev = self.device

            def prefetch(dst, fork=True, draw=True):
                """draw + gather + scan of a LATER update on the side stream"""
                origin = torch.cuda.current_stream(origin_dev)
                ts = self._target_stream
                if isinstance(fork, torch.cuda.Event):
                    ts.wait_event(fork)     # (fork recorded earlier on the origin stream)
                elif fork:
                    ts.wait_stream(origin)  # (fork off the origin stream) the previous users of idx / dst are done
                shared = self._body_is_shared()
                agents = [self.mac.agent] if shared else [self.target_mac.agent, self.mac.agent]
                with torch.cuda.stream(ts), torch.no_grad():
                    if draw:
                        ops.sample_episodes(self._g_idx, self._g_n_stored, self._g_draws, self._sampler_seed())
                    ops.gather_rows(self._g_idx, srcs, [dst[k] for k in keys])
                    tot_m = ops.td_mask_sum(dst["filled"], self._g_T - 1)     # the loss's only global sum the gradient needs
                    ev_g = torch.cuda.Event()
                    ev_g.record(ts)
                    h, ps = ops.gru_sequence_from_obs(buffer.buffers["obs"], self._g_idx, agents, self._g_B, self.n_agents,
                                                      self._g_T + 1, with_actor=True)
                    ev_s = torch.cuda.Event()
                    ev_s.record(ts)
                return (([h[0], h[0]] if shared else h), ([ps[0], ps[0]] if shared else ps), (ev_g, ev_s, tot_m))

            with hipgraph.capture(gm, pool=self._graph_a.pool()):
                nxt_batch = None
                for k_upd in range(K):
                    last = k_upd == K - 1
                    box = {}
                    hook = (lambda fork=True: box.__setitem__("next", prefetch(stages[(k_upd + 1) % 2], fork))) if (pipelined and not last) else None
                    if pipelined and nxt_batch is not None:
                        h_pre, p_pre, evs = nxt_batch
                        self._forward_backward_full(stages[k_upd % 2], self._g_T, pre_scan=h_pre, pre_actor=p_pre,
                                                    prefetched=evs, after_join=hook)
                    elif pipelined:
                        pre = self._scan_from_ring_early()
                        ops.gather_rows(self._g_idx, srcs, [stage[k] for k in keys])
                        if self._paired_heads_ok(stage, self._g_T):
                            # the group's first update in the paired form too: its scan (side stream) and its gather + mask
                            # sum (this stream) run beside each other, then the same chain as every other update — and the
                            # second update's prefetch forks as early as everyone's (first two updates of a group:
                            # 169 + 134 -> 150 + 111 us)
                            tot_m0 = ops.td_mask_sum(stage["filled"], self._g_T - 1)
                            scan_done = torch.cuda.Event()
                            scan_done.record(self._target_stream)
                            self._forward_backward_full(stage, self._g_T, pre_scan=pre[0], pre_actor=pre[1],
                                                        prefetched=(None, scan_done, tot_m0), after_join=hook)
                        else:
                            self._forward_backward_full(stage, self._g_T, pre_scan=pre[0], pre_actor=pre[1], after_join=hook)
                    else:
                        body_a()
                    nxt_batch = box.get("next")
                    rows.append(self._last_stats4)          # this update's (loss, mean Q_tot, mean target, grad norm)
                    self._grad_norm = rows[-1][3]
                    if self._g_graphed_ar:
                        self._allreduce_grads(force=True)
                    # the update's last launch draws the next batch — unless the prefetch behind the join has done so
                    self._clip_and_step(sample_next=None if nxt_batch is not None else nxt)
                if pipelined:
                    torch.cuda.current_stream(self.device).wait_stream(self._target_stream)   # every fork rejoins
            self._grad_norm = single_norm
            self._g_multi = (K, gm, rows)
            self._g_pipelined = pipelined
        self._g_shared_body = self._body_is_shared()   # baked into the captured launches
        assert self._g_obs_static == (bool(getattr(buffer, "obs_static", False)) and options.on("LEARNER_STATIC_OBS"))
        # Prime the graphs: the FIRST launch of an instantiated graph pays one-off costs (its streams get their hardware
        # queues, kernel arguments are uploaded) that belong to setting up, not to the first update a caller times.  One
        # replay of each graph on valid indices, then the training state — weights, optimiser moments, the sampler's
        # counter and drawn batch — goes back to what it was.
        if prime:
            keep = (self._g_draws.clone(), self._g_idx.clone(), self._g_n_stored.clone())
            self._g_n_stored.fill_(int(buffer.current_size))
            self._g_idx.clamp_(0, max(0, int(buffer.current_size) - 1))
            hipgraph.replay(self._graph_a, self.device)
            if not self._g_single:
                self._allreduce_grads()
                hipgraph.replay(self._graph_b, self.device)
            if self._g_multi is not None:
                self._g_idx.clamp_(0, max(0, int(buffer.current_size) - 1))
                hipgraph.replay(self._g_multi[1], self.device)
            torch.cuda.synchronize(self.device)
            restore_training_state()
            self._g_draws.copy_(keep[0]); self._g_idx.copy_(keep[1]); self._g_n_stored.copy_(keep[2])
        self._graphs_ready = True

    def release_graphs(self):
        """Destroy the captured update graphs and everything that lives in their memory pool (staging tensors, static
        outputs, the side stream's events) — explicitly, with the device idle, the grouped graph and graph B (captured
        into graph A's pool) before graph A.  ``enable_graphs`` calls it before it captures again; call it yourself to
        return the pool's memory.  Afterwards ``train_from_buffer`` raises until ``enable_graphs`` ran again."""
        graphs = []
        m = getattr(self, "_g_multi", None)
        if m is not None:
            graphs.append(m[1])
        graphs += [getattr(self, "_graph_b", None), getattr(self, "_graph_a", None)]
        self._graphs_ready = False
        self._g_multi = self._g_stages = None
        self._graph_a = self._graph_b = None
        self._graph_body_a = None
        self._g_out_a = self._g_out_b = self._g_stats4 = self._last_stats4 = self._held_ln = None
        if self._flat_param is not None and any(g is not None for g in graphs):
            # the gradient norm was re-pointed at a static output row of the pool being released
            self._grad_norm = torch.zeros((), dtype=torch.float32, device=self._flat_param.device)
        hipgraph.destroy(graphs, self.device)

    def train_from_buffer(self, indices=None, sync_stats=True, stats_row=None):
        """Sample ``batch_size`` whole episodes (np.random.choice like the reference's buffer) and update.
        With ``sync_stats=False`` the returned scalars are views of the captured graph's static output tensors, which
        the NEXT update overwrites; pass ``stats_row`` (float32 [4] on the device) to receive this update's
        (loss, eval_qtot_avg, target_qtot_avg, grad_norm) as a snapshot instead (one 16-byte device copy, no sync)."""
        buf = self._g_buffer if getattr(self, "_graphs_ready", False) else None
        if buf is None:
            raise RuntimeError("call enable_graphs(buffer, batch_size) first")
        device_draw = False
        if indices is None:
            if buf.current_size < self._g_B:   # fewer stored episodes than a batch: sample what is there, like
                return self._snap(self.train(buf.sample(self._g_B), None, sync_stats=sync_stats), stats_row)   # buffer.sample does
            device_draw = self._g_dev_sampler and self._population_full_length(buf)
            if not device_draw:
                # uniform without replacement like the reference's buffer.sample (replay_buffer.py:89), but from the
                # learner's own numpy Generator: the legacy np.random.choice shuffles the whole population per call
                # (~90 us for 8192 stored episodes — more host time than the rest of the update's launch),
                # Generator.choice takes ~4 us.  EpisodeReplayBuffer.sample() keeps the reference's call.
                indices = self._sample_rng.choice(buf.current_size, self._g_B, replace=False)
        if not device_draw:
            indices = np.asarray(indices, dtype=np.int64)
            if len(indices) != self._g_B or int(buf.episode_lengths[indices].min()) != self._g_T:
                return self._snap(self.train(buf.sample(len(indices), indices=indices), None, sync_stats=sync_stats), stats_row)
        if self._g_obs_static and not buf.obs_static:
            raise RuntimeError("an episode with unknown / changing observations was stored after enable_graphs() captured "
                               "the static-observation update: call enable_graphs() again")
        if self._g_shared_body and not self._body_is_shared():
            raise RuntimeError("the agent body (fc1 / GRU / actor) of one controller changed after enable_graphs() captured "
                               "the shared-body update: call enable_graphs() again")
        self.train_step += 1
        if device_draw:
            self._device_draw_ready(buf)
        else:
            # index upload from a small ring of pinned buffers: a copy from pageable memory makes the host wait for the
            # stream (it could then never run ahead of the GPU and every node of the next replay would be issued just
            # in time); a slot is reused only after the copy that read it has completed
            k = self.train_step % len(self._g_idx_ring)
            slot, ev = self._g_idx_ring[k]
            ev.synchronize()
            slot.numpy()[:] = indices
            self._g_idx.copy_(slot, non_blocking=True)
            ev.record()
        hipgraph.replay(self._graph_a, self.device)
        # (the replayed update ends with the draw of the next batch from the population the device scalar names)
        self._g_idx_fresh = self._g_dev_sampler and (buf.store_count, buf.current_size) == self._g_pop_seen
        if not self._g_single:
            self._allreduce_grads()
            hipgraph.replay(self._graph_b, self.device)
        self._after_step()
        loss, ev, tg = self._g_out_a
        if stats_row is not None and not sync_stats and self._g_stats4 is not None:
            stats_row.copy_(self._g_stats4, non_blocking=True)
            return {"loss": stats_row[0], "grad_norm": stats_row[3], "eval_qtot_avg": stats_row[1], "target_qtot_avg": stats_row[2]}
        return self._snap(self._pack_stats(loss, self._g_out_b, ev, tg, sync_stats), stats_row)

    def _device_draw_ready(self, buf):
        """The previous update's last launch drew the next batch already — unless the population changed since (a rollout
        stored episodes) or that draw was overwritten by a caller's indices: then one small launch redraws."""
        pop = (buf.store_count, buf.current_size)
        if pop != self._g_pop_seen:
            self._g_n_stored.fill_(buf.current_size)
            self._g_pop_seen, self._g_idx_fresh = pop, False
        if not self._g_idx_fresh:
            ops.sample_episodes(self._g_idx, self._g_n_stored, self._g_draws, self._sampler_seed())

    def train_from_buffer_many(self, n, stats_out=None):
        """``n`` consecutive updates on device-drawn batches, the same sequence of updates as ``n`` calls of
        ``train_from_buffer()`` (same draws, same target syncs): groups of K = updates_per_graph updates replay ONE graph
        where nothing has to happen between them (no target sync due inside the group), the rest goes one by one.
        Returns the updates' statistics as float32 [4] device tensors (loss, eval_qtot_avg, target_qtot_avg, grad_norm);
        the tensors of a replayed group are static outputs that the next replay of the group overwrites — pass
        ``stats_out`` (float32 [n, 4] on the device) to get every update's row snapshotted (one small copy per group)."""
        buf = self._g_buffer if getattr(self, "_graphs_ready", False) else None
        if buf is None:
            raise RuntimeError("call enable_graphs(buffer, batch_size) first")
        out, done = [], 0
        while done < n:
            m = self._g_multi
            until_sync = self.args.target_update_interval - (self.train_step - self.last_target_update_step)
            if (m is not None and n - done >= m[0] and until_sync >= m[0] and buf.current_size >= self._g_B
                    and self._population_full_length(buf) and (not self._g_obs_static or buf.obs_static)
                    and (not self._g_shared_body or self._body_is_shared())):
                self._device_draw_ready(buf)
                self.train_step += m[0]
                hipgraph.replay(m[1], self.device)
                self._g_idx_fresh = (buf.store_count, buf.current_size) == self._g_pop_seen
                self._after_step()
                if stats_out is not None:
                    torch.stack(m[2], out=stats_out[done:done + m[0]])
                out += m[2]
                done += m[0]
            else:
                row = stats_out[done] if stats_out is not None else torch.empty(4, dtype=torch.float32, device=self.device)
                self.train_from_buffer(sync_stats=False, stats_row=row)
                out.append(row)
                done += 1
        return out if stats_out is None else stats_out

    def _sampler_seed(self):
        """Key of the device-side episode sampler: the learner's seed, set apart per rank like the host sampler's."""
        return getattr(self, "_sampler_seed_value", int(getattr(self.args, "seed", 0) or 0))

    def _population_full_length(self, buf):
        """Every stored episode has the full episode_limit length (what the captured update assumes): checked on the host
        copy of the lengths, once per change of the buffer's content."""
        key = (buf.store_count, buf.current_size)
        if getattr(self, "_g_full_key", None) != key:
            self._g_full_key = key
            self._g_full = bool((buf.episode_lengths[:buf.current_size] == self._g_T).all())
        return self._g_full

    @staticmethod
    def _snap(stats, stats_row):
        """Copy device-resident scalars of ``stats`` into ``stats_row`` (see train_from_buffer) and return views of it."""
        if stats_row is None or not torch.is_tensor(stats["loss"]):
            return stats
        order = ("loss", "eval_qtot_avg", "target_qtot_avg", "grad_norm")
        torch.stack([torch.as_tensor(stats[k], device=stats_row.device, dtype=torch.float32).reshape(()) for k in order],
                    out=stats_row)
        return {k: stats_row[i] for i, k in enumerate(order)}

    def _all_action_q_multi(self, macs, obs, keep_final_hidden=True):
        """Q(s_t, a, P_a(s_t)) for every discrete action, [B, T, J, A] per controller; replaces the
        per-step / per-action unroll of qmix.py:217-280 (the discarded ``params`` tensor is not built).
        Everything but the recurrence is time-parallel: one GEMM each for fc1, the GRU input transform,
        the actor layers and the Q-head base over all B*T*J rows; the recurrences of all controllers
        (eval + target) run in ONE fused scan launch.  The scan is a latency-bound chain that leaves most of
        the chip idle, so on a HIP device the actor chains — which do not depend on it — run concurrently on a
        forked side stream (also inside a captured graph: fork / join by stream waits).  Leaves each controller's
        ``hidden_states`` at the final h_T like the reference's loop does."""
        B, T, J, S = obs.shape
        rows = obs.reshape(B * T * J, S)
        # frozen, identical bodies (see _mark_body_shared): evaluate fc1 / GRU / actor once, the Q-heads per controller
        shared = len(macs) == 2 and self._body_is_shared() and {id(m) for m in macs} == {id(self.mac), id(self.target_mac)}
        body_macs = [self.mac] if shared else list(macs)
        side = None
        if rows.is_cuda:
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=rows.device)
            side, main = self._side_stream, torch.cuda.current_stream(rows.device)
            side.wait_stream(main)                                           # fork
            with torch.cuda.stream(side):
                params = [m.agent.actor_forward(rows) for m in body_macs]    # networks.py:127
        else:
            params = [m.agent.actor_forward(rows) for m in body_macs]
        gis = [m.agent.gru_input_transform(rows).view(B, T, J, 3 * m.agent.rnn_hidden_dim) for m in body_macs]  # networks.py:100
        h_alls = ops.gru_sequence_multi(gis, [m.agent.rnn.weight_hh for m in body_macs],
                                        [m.agent.rnn.bias_hh for m in body_macs])  # h_0 = 0 (qmix.py:241)
        if shared:
            params, h_alls = params * 2, h_alls * 2
        bases = []
        for m, h_all in zip(macs, h_alls):
            a = m.agent
            H = a.rnn_hidden_dim
            if keep_final_hidden:   # the reference's loop leaves h_T behind (a strided slice: this is a copy)
                m.hidden_states = h_all[:, T - 1].reshape(B * J, H) if T > 0 else None
            l1 = a.fc2_q_head[0]
            bases.append(F.linear(h_all.reshape(B * T * J, H), l1.weight[:, :H], l1.bias))
        if side is not None:
            torch.cuda.current_stream(rows.device).wait_stream(side)        # join
            for p in params:
                p.record_stream(torch.cuda.current_stream(rows.device))
        out = []
        for m, base, params_all in zip(macs, bases, params):
            a = m.agent
            l1, l2 = a.fc2_q_head[0], a.fc2_q_head[2]
            out.append(ops.qhead_all_actions(base, params_all, l1.weight, l2.weight, l2.bias, a.rnn_hidden_dim,
                                             a.n_actions).view(B, T, J, a.n_actions))
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
        self._mark_body_shared()

    def cuda(self):
        was_shared = getattr(self, "_body_versions", None) is not None and self._body_is_shared()
        self.mac.cuda()
        self.target_mac.cuda()
        self.eval_qmix_net.cuda()
        self.target_qmix_net.cuda()
        self.device = torch.device("cuda", torch.cuda.current_device())
        if was_shared:
            self._mark_body_shared()   # .cuda() re-homes the parameters; equal values stay equal

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
