"""bench.py glue for the rollout / train modes: builds MAC + replay + learner + batched runner for the
benchmark scenario and returns the per-step callable.

One benchmark "step" = one batched environment step of E envs per GPU:
  rollout  agent forward (fc1/GRU/actor GEMMs) + fused MP-DQN Q-head/epsilon-greedy kernel + env-step
           kernel + staging writes; every episode_limit steps the E finished episodes move into the
           device replay buffer (inside the timed region).
  train    rollout step + ONE QMixLearner.train on a sampled batch of `batch_size` whole episodes
           (reference cadence: train_interval = 1, i.e. one learner step per env step, main.py:212-216),
           including the RCCL gradient all-reduce when world_size > 1.  The replay buffer is pre-filled
           by one untimed rollout before warm-up so that sampling is valid from the first timed step.
"""
from __future__ import annotations

import contextlib
import io
import os
from types import SimpleNamespace

import numpy as np
import torch


def make_args(sc, hidden, device, batch_size=32, buffer_size=None, batch_envs=4096, seed=42, mixer_dtype="fp32"):
    info = sc.env_info()
    return SimpleNamespace(
        n_agents=info["n_agents"], n_actions=info["n_actions"], state_shape=info["state_shape"],
        obs_shape=info["obs_shape"], episode_limit=info["episode_limit"], env_info=info,
        rnn_hidden_dim=hidden, actor_hidden_dim=128, mixing_embed_dim=64, hyper_hidden_dim=128,
        lr=5e-6, gamma=0.99, grad_norm_clip=1.0, target_update_interval=200, batch_size=batch_size,
        buffer_size=buffer_size or 2 * batch_envs, epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=100000,
        device=str(device), use_cuda=True, seed=seed, mixer_dtype=mixer_dtype)


def _make_step_fn(cli, runner, learner, buf, args, T, mode, use_graphs, tails):
    """The per-step schedule of the benchmark (no device code here: tests drive it with a fake runner)."""
    state = {"graphed_episode": False}
    fused = use_graphs and getattr(runner, "fused_rollout_available", lambda: False)()
    group = int(getattr(learner, "_g_multi", None)[0]) if (learner is not None and getattr(learner, "_g_multi", None)) else 1

    def step_fn(i):
        # Episodes are aligned to the regions the caller times: the warm-up steps are their own run of episodes and
        # the timed steps start a fresh episode batch at step `warmup`.  A whole episode batch (reset + T batched
        # steps) is ONE graph launch; the partial episode at the end of a region replays its own shorter graph.
        # Without the alignment a graph launched during warm-up would do the work of later, timed, steps before
        # the clock starts.
        if i < cli.warmup:
            j, region = i, cli.warmup
        else:
            j, region = i - cli.warmup, cli.steps
        t = j % T
        if t == 0:
            left = region - j
            if fused:   # the whole (or the region's partial) episode batch: agent-episode launch + many-step env launch
                state["graphed_episode"] = True
                # (measured and dropped: the batch's launches on their own stream beside the updates of its first steps —
                # identical results with a Q-head snapshot, but the 0.75 ms agent kernel on every CU slows the updates
                # it overlaps by more than it hides: 0.1688 vs 0.1670 ms / step)
                runner.rollout_fused(n_steps=None if left >= T else left)
            else:
                state["graphed_episode"] = use_graphs and (left >= T or left in tails)
                if state["graphed_episode"]:
                    runner.rollout_graphed(None if left >= T else left)
                else:
                    runner.begin_episodes()
        if not state["graphed_episode"]:
            runner.step(t)
        if t == T - 1:
            runner.end_episodes()
        if mode == "train":
            if use_graphs and group > 1:
                # one update per env step, issued in groups: the K updates of steps j-K+1 .. j replay ONE graph after
                # step j (the reference trains after the episode, main.py:212-216; the env steps of the whole episode
                # batch were launched at t = 0 anyway); the region's remainder goes one by one
                state["owed"] = state.get("owed", 0) + 1
                if state["owed"] == group or j == region - 1:
                    learner.train_from_buffer_many(state["owed"])
                    state["owed"] = 0
            elif use_graphs:
                learner.train_from_buffer(sync_stats=False)
            else:
                learner.train(buf.sample(args.batch_size), None, sync_stats=False)

    return step_fn


def make_step(cli, sc, env, dev, rank, world, mode):
    from . import hipgraph
    from .core.mac import BasicMAC
    from .core.qmix import QMixLearner
    from .runners.episode_runner import BatchedEpisodeRunner
    from .utils.replay_buffer import EpisodeReplayBuffer

    args = make_args(sc, cli.hidden, dev, batch_envs=env.batch_envs, mixer_dtype=getattr(cli, "mixer_dtype", "fp32"))
    gemm_tuning = False
    if not getattr(cli, "no_gemm_tuning", False):
        import tempfile
        from . import ops
        gemm_tuning = ops.enable_gemm_tuning(os.path.join(tempfile.gettempdir(), f"macjd_tunableop_rank{rank}.csv"))
    torch.manual_seed(42)  # identical initial weights on every rank
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(args.obs_shape, args)
        mac.select_seed = 42 + 1000 * rank
        buf = EpisodeReplayBuffer(args, device=dev)
        learner = QMixLearner(mac, args)
    if world > 1:  # same seed already gives identical weights; the broadcast makes it independent of that
        from . import parallel
        parallel.broadcast_parameters([mac.agent, learner.eval_qmix_net])
        learner._update_targets()
    runner = BatchedEpisodeRunner(env, mac, buf, args)
    T = args.episode_limit
    np.random.seed(1234 + rank)  # replay sampling stream of buffer.sample (np.random.choice, like the reference)
    learner._sample_rng = np.random.default_rng(1234 + rank)   # ... and of the graphed update's own samplers
    learner._sampler_seed_value = 1234 + rank                  # (host draw for explicit / ragged batches, device draw)
    use_graphs = not getattr(cli, "no_graphs", False)
    if mode == "train":
        runner.run(test_mode=False, sync_stats=False)  # untimed pre-fill of the replay buffer
        if use_graphs:
            learner.enable_graphs(buf, args.batch_size, updates_per_graph=int(os.environ.get("MACJD_UPDATES_PER_GRAPH", "20")))
    tails = set()
    fused_rollout = use_graphs and runner.fused_rollout_available()
    if use_graphs:   # (fused rollout: reset + fills + the whole-episode launches as one graph; else the step-by-step loop)
        runner.enable_graph()
        # the partial episode at the end of the warm-up / timed region gets its own (shorter) graph, so every step of
        # the run is replayed from a graph whatever --steps / --warmup are
        tails = {r % T for r in (cli.warmup, cli.steps) if r % T}
        for n in sorted(tails):
            runner.enable_graph(n_steps=n)
    step_fn = _make_step_fn(cli, runner, learner if mode == "train" else None, buf, args, T, mode, use_graphs, tails)
    step_fn.runner, step_fn.learner, step_fn.buffer = runner, learner, buf   # (for tests and probes)

    st0 = (getattr(learner, "_g_stages", None) or [None])[0]   # the grouped update's first staging set (None: no group captured)
    extra = {"hidden": cli.hidden, "train_batch_episodes": args.batch_size if mode == "train" else 0,
             "train_calls_per_step": 1 if mode == "train" else 0,
             "updates_per_graph": (learner._g_multi[0] if (mode == "train" and use_graphs and learner._g_multi) else 1),
             "hip_graphs": bool(use_graphs), "gemm_tuning": bool(gemm_tuning),
             "graph_launch": {"stream": hipgraph.launch_mode(), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                              "allreduce_in_graph": bool(getattr(learner, "_g_graphed_ar", False))},
             "update": (("paired: Q-head launches as one grid, mixers as one grid, the chain on one queue"
                         if learner._paired_heads_ok(st0, learner._g_T) else "target branch on the side stream")
                        if (mode == "train" and use_graphs and st0 is not None) else None),
             "replay_capacity_episodes": args.buffer_size, "mixer_dtype": args.mixer_dtype,
             "mixer": ("one MFMA launch per direction (f32)" if (mode == "train" and learner.eval_qmix_net.fused_available(next(learner.eval_qmix_net.parameters())))
                       else ("library GEMMs (%s) + tail kernel" % args.mixer_dtype)) if mode == "train" else None,
             "rollout": ("fused: agent-episode launch + many-step env launch per episode batch" +
                         (", replayed as one graph" if use_graphs else "")) if fused_rollout else
                        ("HIP graph of the step-by-step rollout" if use_graphs else "eager step-by-step")}
    return step_fn, extra


def mixer_hyper_w1_roofline(sc, hidden, dev, M, timer, peak_tflops=157.3):
    """MFMA fraction of the mixer's ``hyper_w_1`` network (reference core/networks.py:223-248), the figure SURVEY.md 8(d)
    asks for: 2 M (S Hh + Hh J Em) flop over the learner's M = batch x (T + 1) rows, timed as the learner evaluates the
    state-only half of the mixer (``QMixer.hyper_outputs``: LayerNorm + all four hyper-networks, of which hyper_w_1 is
    ~85 % of the flops) and, separately, hyper_w_1 alone.  ``timer(fn, dev)`` -> microseconds per call."""
    from .core.networks import QMixer
    args = make_args(sc, hidden, dev)
    torch.manual_seed(0)
    mixer = QMixer(args).to(dev)
    S, J, Hh, Em = mixer.state_dim, mixer.n_agents, mixer.hyper_hidden_dim, mixer.embed_dim
    s = torch.randn(M, S, device=dev)
    flops_w1 = 2.0 * M * (S * Hh + Hh * J * Em)
    flops_all = flops_w1 + 2.0 * M * (S * Hh + Hh * Em) + 2.0 * M * S * Em + 2.0 * M * (S * Em + Em)
    mixer.enable_first_layer_cache()   # like the learner's mixers: the merged first layer is never concatenated per call
    q = torch.randn(M, J, device=dev)
    flops_fused = flops_all + 2.0 * M * (J * Em + Em)
    fused = mixer.fused_available(s)
    with torch.no_grad():
        us_fused = timer(lambda: mixer(q, s), dev) if fused else None
        QMixer.fused = False
        try:
            us_w1 = timer(lambda: mixer.hyper_w_1(s), dev)
            us_unfused = timer(lambda: mixer(q, s), dev)
        finally:
            QMixer.fused = True
    # hyper_w_1 is 60 % of the fused launch's flops and cannot be timed apart inside it: the fused launch is priced on
    # ALL of the mixer's flops (LayerNorm excluded), the library form of hyper_w_1 alone is reported next to it
    tf = flops_fused / (us_fused * 1e-6) / 1e12 if fused else flops_w1 / (us_w1 * 1e-6) / 1e12
    out = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(tf / peak_tflops, 4),
           "traffic": None,
           "kernel": ("mixer_fused_forward_kernel (whole QMixer.forward: %d-%d | %d-%d-%d | tail, %d rows)" % (S, 2 * Hh + 2 * Em, Hh, Hh, J * Em + Em, M))
                     if fused else "QMixer.hyper_w_1 on library GEMMs (%d-%d-%d, %d rows)" % (S, Hh, J * Em, M),
           "us_per_call": round(us_fused if fused else us_w1, 2), "flops_per_call": int(flops_fused if fused else flops_w1),
           "dtype": "f32 (exact-f32 MFMA)",
           "hyper_w_1_flops": int(flops_w1),
           "unfused": {"hyper_w_1_us": round(us_w1, 2), "hyper_w_1_frac": round(flops_w1 / (us_w1 * 1e-6) / 1e12 / peak_tflops, 4),
                       "forward_us": round(us_unfused, 2), "launches": 7,
                       "what": "LayerNorm + 3 library GEMMs + ReLU + row-dot + tail kernel (round 1's form of the same forward)"}}
    return out
