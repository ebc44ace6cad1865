"""Rollout drivers (reference runners/episode_runner.py:9-277).

``EpisodeRunner``        the reference's one-env / one-episode protocol, same constructor, ``run(test_mode)``
                         return keys and buffer layout (including its quirks: the stored ``hidden_state`` is
                         the POST-update h_t, and row ``T`` of the T+1 arrays stays zero when the episode
                         fills ``episode_limit`` because ``EpisodeBatch.update_last`` is then a no-op,
                         episode_runner.py:246-255).  Works with the drop-in single-env facade.
``BatchedEpisodeRunner`` the hot path: E environments x episode_limit steps with every tensor resident in
                         HBM and NO host synchronisation inside the step loop; one transposing copy per
                         key moves the finished episodes into the replay buffer.
"""
from __future__ import annotations

import os
from functools import partial
from typing import Dict

import itertools

import numpy as np
import torch


_ENV_SERIAL = itertools.count(1)   # static-content tags of the replay slots (BatchedEpisodeRunner.end_episodes)

class EpisodeBatch:
    """Per-episode staging arrays (episode_runner.py:184-277)."""

    _T_PLUS_1 = ("state", "obs", "avail_actions", "hidden_state")

    def __init__(self, args, max_seq_length, n_agents, obs_shape):
        self.args = args
        self.max_seq_length = max_seq_length
        self.n_agents = n_agents
        self.obs_shape = obs_shape
        env_info = args.env_info
        self.scheme = {  # episode_runner.py:197-209
            "state": ((env_info["state_shape"],), None, np.float32),
            "obs": ((obs_shape,), "agents", np.float32),
            "actions_discrete": ((1,), "agents", np.int32),
            "actions_continuous": ((1,), "agents", np.float32),
            "avail_actions": ((env_info["n_actions"],), "agents", np.int64),
            "reward": ((1,), None, np.float32),
            "terminated": ((1,), None, np.bool_),
            "hidden_state": ((args.rnn_hidden_dim,), "agents", np.float32),
        }
        self.data: Dict[str, np.ndarray] = {}
        self.t = 0

    def _init_data(self):
        self.data = {}
        for key, (vshape, group, dtype) in self.scheme.items():
            seq = self.max_seq_length + 1 if key in self._T_PLUS_1 else self.max_seq_length
            shape = (seq, self.n_agents) + tuple(vshape) if group == "agents" else (seq,) + tuple(vshape)
            self.data[key] = np.zeros(shape, dtype=dtype)
        self.t = 0

    def push(self, transition_data):
        if self.t == 0:
            self._init_data()
        if self.t < self.max_seq_length:
            for key, value in transition_data.items():
                if key in self.data:
                    self.data[key][self.t] = value
            self.t += 1
        else:
            print("Warning: Episode length exceeded max_seq_length. Data not stored.")

    def update_last(self, last_data):
        if self.t < self.max_seq_length:  # no-op for a full-length episode (episode_runner.py:252)
            for key, value in last_data.items():
                if key in self.data:
                    self.data[key][self.t] = value

    def get_batch_data(self):
        out = {}
        for key in self.scheme:
            n = self.t + 1 if key in self._T_PLUS_1 else self.t
            out[key] = [self.data[key][:n]]
        return out


class EpisodeRunner:
    """One environment, one episode per ``run`` (episode_runner.py:9-181)."""

    def __init__(self, env, mac, buffer, args):
        self.env, self.mac, self.buffer, self.args = env, mac, buffer, args
        env_info = self.env.get_env_info()
        self.episode_limit = env_info["episode_limit"]
        self.n_agents = env_info["n_agents"]
        self.t = 0
        self.t_env = 0
        self.new_batch = partial(EpisodeBatch, self.args, self.episode_limit, self.n_agents, env_info["obs_shape"])
        self.device = torch.device(args.device if torch.cuda.is_available() and args.use_cuda else "cpu")

    def run(self, test_mode=False):
        batch = self.new_batch()
        self.mac.init_hidden(batch_size=1)
        terminated, episode_return, step = False, 0, 0
        rewards, r_d, r_p, r_j, ep_T, ep_P = [], [], [], [], [], []
        self.env.reset()
        state = self.env.get_state()
        obs_list = self.env.get_obs()
        info = {}
        while not terminated:
            obs_np = np.array(obs_list)
            avail_np = np.array(self.env.get_avail_actions())
            obs_t = torch.tensor(obs_np, dtype=torch.float32).unsqueeze(0).to(self.device)
            avail_t = torch.tensor(avail_np, dtype=torch.long).unsqueeze(0).to(self.device)
            T_t, P_t = self.mac.select_actions(obs_t, avail_t, self.t_env, test_mode=test_mode)
            hidden_np = self.mac.hidden_states.detach().cpu().reshape(self.n_agents, -1).numpy()  # post-update h_t
            T_np = T_t.detach().squeeze(0).cpu().numpy()
            P_np = P_t.detach().squeeze(0).cpu().numpy()
            ep_T.append(T_np)
            ep_P.append(P_np)
            next_obs_list, reward, terminated, info = self.env.step([(d[0], c[0]) for d, c in zip(T_np, P_np)])
            rewards.append(reward)
            r_d.append(info.get("r_d", 0)); r_p.append(info.get("r_p", 0)); r_j.append(info.get("r_j", 0))
            episode_return += reward
            batch.push({"state": np.array(state), "obs": obs_np, "actions_discrete": T_np,
                        "actions_continuous": P_np, "avail_actions": avail_np, "reward": np.array([reward]),
                        "terminated": np.array([terminated]), "hidden_state": hidden_np})
            state = self.env.get_state()
            obs_list = next_obs_list
            step += 1
            self.t_env += 1
            if terminated or step >= self.episode_limit:
                batch.update_last({"state": np.array(self.env.get_state()), "obs": np.array(next_obs_list),
                                   "avail_actions": np.array(self.env.get_avail_actions())})
                break
        if not test_mode:
            self.buffer.store_episode(batch.get_batch_data())

        all_T = np.array(ep_T) if ep_T else np.empty((0, self.n_agents, 1))
        all_P = np.array(ep_P) if ep_P else np.empty((0, self.n_agents, 1))
        avg_power_per_agent = np.mean(all_P, axis=0).flatten() if all_P.size > 0 else np.zeros(self.n_agents)
        counts = np.zeros(self.args.n_actions)
        if all_T.size > 0:
            counts = np.bincount(all_T.flatten().astype(int), minlength=self.args.n_actions)[:self.args.n_actions]
        run_info = {
            "episode_length": step,
            "episode_return": episode_return,
            "avg_step_reward": np.mean(rewards) if rewards else 0,
            "avg_r_d": np.mean(r_d) if r_d else 0,
            "avg_r_p": np.mean(r_p) if r_p else 0,
            "avg_r_j": np.mean(r_j) if r_j else 0,
            "avg_power_overall": np.mean(avg_power_per_agent),
            "action_distribution": counts / max(1, np.sum(counts)),
        }
        if "individual_rewards" in info:
            run_info["individual_rewards_final"] = info["individual_rewards"]
        return run_info

    def close_env(self):
        self.env.close()


class BatchedEpisodeRunner:
    """E environments per rollout, device-resident, no per-step host sync.

    Time-major staging tensors ``[T(+1), E, ...]`` make every per-step write a contiguous row, so the
    env-step kernel writes reward / terminated straight into them; the action kernel's agent-major
    outputs are copied in with one strided copy each.  The observation of this environment never
    changes (reference simulation/environment.py:479-510 is a pure function of static parameters), so
    when the env advertises ``observation_is_static`` the obs / state / avail rows are filled once.
    """

    def __init__(self, env, mac, buffer, args):
        self.env, self.mac, self.buffer, self.args = env, mac, buffer, args
        info = env.get_env_info()
        self.episode_limit = info["episode_limit"]
        self.n_agents = info["n_agents"]
        self.batch_envs = env.batch_envs
        self.device = env.device
        self.t_env = 0  # env steps per environment, drives the epsilon schedule like the reference's counter
        T, E, J = self.episode_limit, self.batch_envs, self.n_agents
        S, A, H = info["state_shape"], info["n_actions"], args.rnn_hidden_dim
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        self.stage = {
            "state": z((T + 1, E, S), torch.float32), "obs": z((T + 1, E, J, S), torch.float32),
            "avail_actions": z((T + 1, E, J, A), torch.int64), "hidden_state": z((T + 1, E, J, H), torch.float32),
            "actions_discrete": z((T, E, J, 1), torch.int32), "actions_continuous": z((T, E, J, 1), torch.float32),
            "reward": z((T, E, 1), torch.float32), "terminated": z((T, E, 1), torch.bool),
        }
        self._static = bool(getattr(env, "observation_is_static", False))
        # static observations: the agent's observation-only work (actor chain, fc1 -> W_ih) is done once per episode
        # batch instead of at every step (BasicMAC.prepare_static_obs); hoist_static_obs = False keeps it per step
        self.hoist_static_obs = self._static and hasattr(mac, "prepare_static_obs")
        self._static_filled = False
        self._rdpj_sum = z((E, 3), torch.float32)
        self._ep = -1
        self._graph = None

    def _fill_static(self):
        T = self.episode_limit
        self.stage["state"][:T] = self.env.get_state()
        self.stage["obs"][:T] = self.env.get_obs()
        self.stage["avail_actions"][:T] = self.env.get_avail_actions()
        self._static_filled = True

    def begin_episodes(self):
        """Reset the E envs and the recurrent state; (re)fill static rows once."""
        self._ep += 1  # episode-batch index; with the step index it forms the select kernel's Philox counter
        self.mac.init_hidden(batch_size=self.batch_envs)
        self.env.reset()
        if self._static and not self._static_filled:
            self._fill_static()
        self._rdpj_sum.zero_()
        if hasattr(self.mac, "prepare_static_obs"):
            self.mac.prepare_static_obs(self.env.get_obs() if self.hoist_static_obs else None)
        if self._static:
            # the filled staging row is a CONTIGUOUS [E, J, S] copy of the (broadcast) observation: the MAC can
            # view it as [E*J, S] without materialising 2 MB per step; the mask stays the env's broadcast view
            self._obs, self._avail = self.stage["obs"][0], self.env.get_avail_actions()
        else:
            self._obs, self._avail = self.env.get_obs(), self.env.get_avail_actions()

    def step(self, t, test_mode=False):
        """Batched step t of the current episodes: agent forward + MP-DQN Q + epsilon-greedy (one fused
        kernel), env-step kernel, staging writes.  Enqueues work only — no host synchronisation."""
        env, mac, st = self.env, self.mac, self.stage
        E, J = self.batch_envs, self.n_agents
        if not self._static:
            st["state"][t].copy_(env.get_state())
            st["obs"][t].copy_(self._obs)
            st["avail_actions"][t].copy_(self._avail)
        # exploration draws are keyed by (seed; row, counter) with counter = ep * (T + 1) + t + 1, the same in
        # eager and graph-replayed rollouts (select_actions pre-increments its call counter)
        mac._select_calls = self._ep * (self.episode_limit + 1) + t
        on_device = self._obs.is_cuda
        if on_device:   # the fused kernels write the chosen actions and the post-update h_t straight into this step's rows
            mac.action_out = (st["actions_discrete"][t], st["actions_continuous"][t])
            mac.hidden_out = st["hidden_state"][t]
        T64, P_sel = mac.select_actions(self._obs, self._avail, self.t_env, test_mode=test_mode)
        if not on_device:
            st["hidden_state"][t].copy_(mac.hidden_states.view(E, J, -1))  # post-update h_t
        if on_device:
            T32, P_sel = st["actions_discrete"][t], st["actions_continuous"][t]
        else:
            T32 = mac.last_actions_T32 if mac.last_actions_T32 is not None else T64.squeeze(-1).to(torch.int32)
            st["actions_discrete"][t].copy_(T32.view(E, J, 1))
            st["actions_continuous"][t].copy_(P_sel)
        # the env kernel reads the actions from the staging row, writes reward / terminated into their rows and adds
        # (r_d, r_p, r_j) to the running per-episode sums
        env.step(T32, P_sel, out_reward=st["reward"][t].view(E), out_terminated=st["terminated"][t].view(E),
                 want_info=False, rdpj_sum=self._rdpj_sum)
        self.t_env += 1
        if not self._static:
            self._obs, self._avail = env.get_obs(), env.get_avail_actions()

    def end_episodes(self, test_mode=False, store=True):
        """Move the E finished episodes into the replay buffer (one transposing copy per key).  Row T of
        the T+1 arrays stays zero for full-length episodes, as in the reference."""
        if store and not test_mode:
            tags = None
            if self._static:   # env e's static rows never change: identify them so that unchanged slots are not rewritten
                if getattr(self, "_static_tags", None) is None:
                    # a process-unique serial per environment object (an object's address can be handed to a later
                    # environment with another scenario, whose rows must not be mistaken for this one's)
                    base = next(_ENV_SERIAL) << 24
                    self._static_tags = base + np.arange(self.batch_envs, dtype=np.int64)
                tags = self._static_tags
            self.buffer.store_episodes_batched(self.stage, self.batch_envs, obs_static=self._static, static_tags=tags)

    # ---- HIP-graph replay of a whole episode batch ----
    def enable_graph(self, n_steps=None):
        """Capture reset + ``n_steps`` batched steps (default: a whole episode, ``episode_limit``) as ONE HIP graph.
        Kernel arguments are frozen at capture, so the two per-step scalars that change between episodes —
        the exploration probability and the Philox call counter of the select kernel — are read from device
        memory (``_eps_sched[t]``, ``_ctr_base``) that ``rollout_graphed`` refreshes before each replay.
        Shorter graphs (the first ``n_steps`` steps of an episode batch) can be captured alongside the full one;
        ``rollout_graphed(n_steps)`` picks the matching graph."""
        T = self.episode_limit
        n = T if n_steps is None else int(n_steps)
        if not 1 <= n <= T:
            raise ValueError(f"enable_graph: n_steps must be in 1..{T}, got {n_steps}")
        dev = self.device
        if getattr(self, "_graphs", None) is None:
            self._graphs = {}
            if getattr(self, "_eps_sched", None) is None:
                self._eps_sched = torch.zeros(T, dtype=torch.float32, device=dev)
                self._ctr_base = torch.zeros(1, dtype=torch.int64, device=dev)
        t_env0, ep0 = self.t_env, self._ep
        # the warm-up rollout resets the envs: put their episode indices (Monte-Carlo stream position) back afterwards,
        # so that enabling a graph does not change which values the following episodes draw
        env_ep = getattr(self.env, "episode_index", None)
        env_ep0 = env_ep.clone() if env_ep is not None else None
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):  # warm-up outside capture
            self._rollout_body(device_schedule=True, n_steps=n)
        torch.cuda.current_stream(dev).wait_stream(s)
        from .. import hipgraph
        if n in self._graphs:   # re-capture of the same length: the old graph goes first, explicitly
            hipgraph.destroy([self._graphs.pop(n)], dev)
        graph = torch.cuda.CUDAGraph()
        with hipgraph.capture(graph):
            self._rollout_body(device_schedule=True, n_steps=n)
        self._graphs[n] = graph
        # the tensors the captured launches were handed through attributes that later (eager) rollouts replace: held for
        # the graph's lifetime, and the MAC's recurrent state is pointed back at the captured one after each replay
        if getattr(self, "_graph_keep", None) is None:
            self._graph_keep = {}
        self._graph_keep[n] = (self.mac.hidden_states, getattr(self.mac, "static_inputs", None), self._avail, self._obs)
        if n == T:
            self._graph = graph
        if env_ep0 is not None:
            env_ep.copy_(env_ep0)
        self.t_env, self._ep = t_env0, ep0
        self.mac.device_schedule = None

    def release_graphs(self):
        """Destroy the captured rollout graphs explicitly, with the device idle (see hipgraph.py); ``run`` then falls back
        to the fused / step-by-step rollout until ``enable_graph`` is called again."""
        from .. import hipgraph
        graphs = list((getattr(self, "_graphs", None) or {}).values())
        self._graphs, self._graph, self._graph_keep = None, None, None
        hipgraph.destroy(list(reversed(graphs)), self.device)

    def _rollout_body(self, device_schedule=False, n_steps=None):
        T = self.episode_limit if n_steps is None else n_steps
        self.begin_episodes()
        if device_schedule and self.fused_rollout_available():
            self._fused_launches(T)     # whole-episode launches (static observation): the same rows as the loop below
            return
        for t in range(T):
            if device_schedule:
                self.mac.device_schedule = (self._eps_sched[t:t + 1], self._ctr_base, t + 1)
            self.step(t, test_mode=False)
        self.mac.device_schedule = None

    def rollout_graphed(self, n_steps=None):
        """Enqueue one episode batch (or its first ``n_steps`` steps) by replaying the captured graph (training mode)."""
        T = self.episode_limit
        n = T if n_steps is None else int(n_steps)
        self._upload_eps_schedule(self._eps_schedule(n, False))
        self._ep += 1
        self._ctr_base.fill_(self._ep * (T + 1))
        from .. import hipgraph
        hipgraph.replay(self._graphs[n], self.device)
        self.mac.hidden_states = self._graph_keep[n][0]   # h after the batch's last step, as the eager rollout leaves it
        self.t_env += n

    # ---- whole-episode launches: the agent's T steps as one kernel, the env's T steps as another ----
    fused_rollout = True   # class / instance switch (tests set it): whole-episode launches where available

    def fused_rollout_available(self) -> bool:
        """The observation is static (so the agent's steps do not depend on the env's outputs), the MAC is the stock
        RNNAgent on a HIP device at a size the episode kernel covers, and the env offers the many-step launch."""
        if not (self.fused_rollout and self.hoist_static_obs and self.device.type == "cuda" and hasattr(self.env, "step_many")):
            return False
        agent = getattr(self.mac, "agent", None)
        if agent is None or not hasattr(self.mac, "prepare_static_obs") or not next(agent.parameters()).is_cuda:
            return False
        from .. import ops
        return (getattr(self.env, "kernel_flags", 0) == 0
                and ops.agent_episode_supported(self.n_agents, agent.rnn_hidden_dim, agent.n_actions))

    def _eps_schedule(self, n, test_mode):
        """Exploration probability of the next ``n`` steps [episode_limit] float32 — the selector's ``anneal`` (reference
        utils/action_selectors.py:30-32) for t_env, t_env + 1, ... in one vector expression (same float64 arithmetic), the
        selector left at the last step's value as after n calls."""
        sel = self.mac.action_selector
        eps = np.zeros(self.episode_limit, dtype=np.float32)
        if test_mode:
            eps[:n] = sel.epsilon
        else:
            delta = (sel.epsilon_start - sel.epsilon_finish) / sel.epsilon_anneal_time
            sched = np.maximum(sel.epsilon_finish, sel.epsilon_start - delta * (self.t_env + np.arange(n, dtype=np.float64)))
            eps[:n] = sched
            sel.epsilon = float(sched[-1])
        return eps

    def _upload_eps_schedule(self, eps: np.ndarray):
        """The episode batch's exploration schedule -> device, through one of two pinned buffers (a copy from pageable memory
        makes the host wait for the stream: the launches of the batch would then start only after everything issued before
        has drained)."""
        ring = getattr(self, "_eps_ring", None)
        if ring is None:
            ring = self._eps_ring = [(torch.zeros(self.episode_limit, dtype=torch.float32).pin_memory(), torch.cuda.Event())
                                     for _ in range(2)]
            self._eps_slot = 0
        self._eps_slot ^= 1
        buf, ev = ring[self._eps_slot]
        ev.synchronize()                       # the copy that last read this buffer is done (two batches ago)
        buf.numpy()[:len(eps)] = eps
        self._eps_sched.copy_(buf, non_blocking=True)
        ev.record()

    def rollout_fused(self, test_mode=False, n_steps=None):
        """One episode batch (or its first ``n_steps`` steps) in THREE launches: ``ops.agent_episode`` (GRU cell + all-action
        Q-head + epsilon-greedy selection for all steps: nothing the agent computes depends on the env's outputs while the
        observation is static), ``env.step_many`` (all steps of all envs as independent work items + the counter
        update).  Same staging rows, exploration draws (row, episode * (T + 1) + t + 1) and Monte-Carlo streams as the
        step-by-step path; the hidden states / Q-values differ from it by the summation order of two matrix products
        (~1e-7), which can flip an arg-max only on a near-tie."""
        from .. import ops
        T = self.episode_limit
        n = T if n_steps is None else int(n_steps)
        if not 1 <= n <= T:
            raise ValueError(f"rollout_fused: n_steps must be in 1..{T}, got {n_steps}")
        st, mac, env = self.stage, self.mac, self.env
        E, J = self.batch_envs, self.n_agents
        if getattr(self, "_eps_sched", None) is None:
            self._eps_sched = torch.zeros(T, dtype=torch.float32, device=self.device)
            self._ctr_base = torch.zeros(1, dtype=torch.int64, device=self.device)
        if getattr(self, "_rdpj_steps", None) is None:
            self._rdpj_steps = torch.zeros((T, E, 3), dtype=torch.float32, device=self.device)
        if not test_mode and n in (getattr(self, "_graphs", None) or {}):
            # the batch's device work — reset, fills, the three launches — was captured (enable_graph): two small uploads
            # and ONE graph launch instead of ~10 eager launches (~200 us of host time, which a short run cannot hide)
            return self.rollout_graphed(n)
        self._upload_eps_schedule(self._eps_schedule(n, test_mode))
        self.begin_episodes()                       # episode index, zero hidden state, env reset, static rows + inputs
        self._ctr_base.fill_(self._ep * (T + 1))
        self._fused_launches(n, test_mode=test_mode)
        self.t_env += n

    def _fused_launches(self, n, test_mode=False):
        """The whole-episode launches of ``rollout_fused`` for the first ``n`` steps (behind ``begin_episodes``; exploration
        schedule and Philox counter base are read from device memory)."""
        from .. import ops
        st, mac, env = self.stage, self.mac, self.env
        E, J = self.batch_envs, self.n_agents
        if getattr(self, "_rdpj_steps", None) is None:
            self._rdpj_steps = torch.zeros((self.episode_limit, E, 3), dtype=torch.float32, device=self.device)
        params, gi = mac.static_inputs
        a = mac.agent
        l1, l2 = a.fc2_q_head[0], a.fc2_q_head[2]
        ops.agent_episode(gi, params, None, a.rnn.weight_hh, a.rnn.bias_hh, l1.weight, l1.bias, l2.weight, l2.bias,
                          E, J, n, self._avail, self._eps_sched, bool(test_mode), mac.select_seed, self._ctr_base,
                          st["hidden_state"], st["actions_discrete"], st["actions_continuous"], h_final=mac.hidden_states)
        env.step_many(st["actions_discrete"][:n], st["actions_continuous"][:n], st["reward"][:n], st["terminated"][:n],
                      self._rdpj_steps[:n], rdpj_sum=self._rdpj_sum)

    def run(self, test_mode=False, store=True, sync_stats=True):
        """One batch of E episodes.  Returns the reference's ``run_info`` keys as means over the E
        episodes (one host sync at the very end; pass ``sync_stats=False`` to get 0-dim tensors)."""
        st = self.stage
        T, E, J = self.episode_limit, self.batch_envs, self.n_agents
        if self.fused_rollout_available():
            self.rollout_fused(test_mode=test_mode)
        elif getattr(self, "_graph", None) is not None and not test_mode:
            self.rollout_graphed()
        else:
            self.begin_episodes()
            for t in range(T):
                self.step(t, test_mode=test_mode)
        self.end_episodes(test_mode=test_mode, store=store)
        ret = st["reward"].sum(dim=(0, 2))  # [E]
        stats = {
            "episode_length": T,
            "episode_return": ret.mean(),
            "avg_step_reward": ret.mean() / T,
            "avg_r_d": self._rdpj_sum[:, 0].mean() / T,
            "avg_r_p": self._rdpj_sum[:, 1].mean() / T,
            "avg_r_j": self._rdpj_sum[:, 2].mean() / T,
            "avg_power_overall": st["actions_continuous"].mean(),
            "action_distribution": torch.bincount(st["actions_discrete"].flatten().long(),
                                                  minlength=self.args.n_actions)[:self.args.n_actions].float()
            / float(T * E * J),
            "n_episodes": E,
        }
        if sync_stats:
            stats = {k: (v.item() if isinstance(v, torch.Tensor) and v.dim() == 0 else
                         (v.cpu().numpy() if isinstance(v, torch.Tensor) else v)) for k, v in stats.items()}
        return stats

    def close_env(self):
        self.env.close()
