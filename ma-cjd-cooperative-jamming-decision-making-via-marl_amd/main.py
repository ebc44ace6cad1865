"""Training driver: the reference's ``main.py`` schedule on the MI355X hot path.

Same entry points and behaviour as reference ``main.py:50-328``:
  ``load_config(config_name, config_dir) -> SimpleNamespace``; ``run(args)``; CLI ``--config --env-config
  --device``; seeds (np / torch = ``args.seed``); env info copied into ``args``; the
  episodes-then-train cadence (``main.py:191-226``); the console block and the 13 + A scalar tags
  (``main.py:252-278``); checkpoints ``<save_model_dir>/<test_name>/step_<N>/{agent,qmix_net,optimizer}.pth``
  every ``save_interval`` episodes once ``total_steps > start_training_steps`` (``main.py:283-289``).

What is added (build flags, all optional):
  ``batch_envs``  (default 1)  E > 1 switches to the batched device-resident runner: one ``runner.run()`` is E
                  episodes.  The learner then does ``updates_per_rollout`` updates per ROLLOUT (default
                  ``episode_len // train_interval``: what the reference does per EPISODE, main.py:212-216; with
                  ``batch_envs = 1`` the two coincide exactly).  Update-to-data ratio: the reference samples
                  32 x 100 transitions per env step it collects; at E = 4096 and 100 updates per rollout this driver
                  samples 0.78 transitions per collected transition — config/default.yaml scales ``total_env_steps``
                  and ``epsilon_anneal_time`` so that a default run still makes ~49 k updates (reference: ~19 k) and
                  anneals epsilon over the first 5 % of the run like the reference.
  ``torchrun``    one process per GPU: envs shard over ranks (``env_offset = rank * batch_envs``), weights are
                  broadcast from rank 0, every update all-reduces the flat gradient (RCCL); rank 0 logs and saves.
  ``hip_graphs``  (default True on a HIP device with batch_envs > 1) replay rollouts / updates from HIP graphs.
  ``randomize_scenarios`` / ``--randomize-scenarios`` (batched runs): every env trains on its own random variation
                  of the scenario file (positions, threat levels, powers; ``scenario.ScenarioBatch``).
  ``--resume DIR`` restores agent, mixer, optimiser and ``trainer_state.json`` (env steps, episodes, epsilon
                  clock, learner step counters, and the positions of the three random streams: exploration counter,
                  replay sampler, per-env Monte-Carlo episode index) — the reference saves the optimiser but never
                  loads it and has no resume path (``core/qmix.py:317-333``).
  ``test_interval`` / ``test_nepisodes`` (present but unread in the reference's ``default.yaml:81-83``): greedy
                  evaluation episodes reporting return, radar lock fraction and mean power (the paper's
                  metrics, ``docs/impadd.md:60-64``).
  TensorBoard is optional (it is not installed in every image): scalars always go to ``scalars.jsonl`` under
  the run's log directory, and to a ``SummaryWriter`` as well when ``torch.utils.tensorboard`` imports.
"""
from __future__ import annotations

import argparse
import json
import os
import time
from collections import deque
from datetime import datetime
from types import SimpleNamespace

import numpy as np
import torch
import yaml

from .core.mac import BasicMAC
from .core.qmix import QMixLearner
from .runners.episode_runner import BatchedEpisodeRunner, EpisodeRunner
from .utils.replay_buffer import EpisodeReplayBuffer


def load_config(config_name="default", config_dir="config"):
    """YAML -> flat SimpleNamespace (main.py:50-79)."""
    path = os.path.join(config_dir, f"{config_name}.yaml")
    try:
        with open(path, "r") as f:
            config_dict = yaml.safe_load(f)
        print(f"Configuration loaded successfully from {path}")
        return SimpleNamespace(**config_dict)
    except FileNotFoundError:
        print(f"Error: Configuration file not found at {path}")
        raise
    except yaml.YAMLError as e:
        print(f"Error parsing configuration file {path}: {e}")
        raise


class ScalarLog:
    """``add_scalar(tag, value, step)`` sink: always ``scalars.jsonl``; TensorBoard too when available."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433 (optional dependency)
            self._tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            self._tb = None

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()


def _pick_device(args, local_rank=0):
    requested = str(getattr(args, "device_request", getattr(args, "device", "cuda"))).lower()
    if requested.startswith("cuda") and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda")
    else:
        if requested.startswith("cuda"):
            print("Warning: CUDA requested but not available. Falling back to CPU.")
        device = torch.device("cpu")
    args.device = str(device)
    args.use_cuda = device.type == "cuda"
    return device


def build_components(args, sim_config_path):
    """env, mac, buffer, learner, runner — single-env reference protocol or the batched hot path."""
    batch_envs = int(getattr(args, "batch_envs", 1) or 1)
    if batch_envs > 1:
        from .simulation.environment import BatchedElectromagneticEnvironment
        off = int(getattr(args, "env_offset", 0))
        batch = None
        if getattr(args, "randomize_scenarios", False):   # every env its own variation of the scenario file
            import yaml
            from .scenario import ScenarioBatch
            with open(sim_config_path) as f:
                base = yaml.safe_load(f)
            batch = ScenarioBatch.randomized(base, batch_envs, seed=args.seed, config=args, env_offset=off)
        env = BatchedElectromagneticEnvironment(args, sim_config_path, batch_envs=batch_envs, seed=args.seed,
                                                env_offset=off, verbose=True, scenario_batch=batch)
    else:
        from .simulation.environment import ElectromagneticEnvironment
        env = ElectromagneticEnvironment(config=args, sim_config_path=sim_config_path)
    env_info = env.get_env_info()
    args.n_agents = env_info["n_agents"]            # main.py:141-147
    args.n_actions = env_info["n_actions"]
    args.state_shape = env_info["state_shape"]
    args.obs_shape = env_info.get("obs_shape", args.state_shape)
    args.episode_limit = env_info["episode_limit"]
    args.env_info = env_info
    if batch_envs > 1 and args.buffer_size < batch_envs:
        raise ValueError(f"buffer_size ({args.buffer_size}) must hold at least one rollout of {batch_envs} episodes")
    mac = BasicMAC(input_shape=args.obs_shape, args=args)
    buffer = EpisodeReplayBuffer(args=args)
    learner = QMixLearner(mac, args=args)
    if args.use_cuda:
        mac.cuda()
    runner = (BatchedEpisodeRunner if batch_envs > 1 else EpisodeRunner)(env=env, mac=mac, buffer=buffer, args=args)
    return env, mac, buffer, learner, runner


def save_checkpoint(learner, runner, save_dir, episode, total_steps):
    """Reference files (qmix.py:300-315) + trainer_state.json for resume."""
    os.makedirs(save_dir, exist_ok=True)
    learner.save_models(save_dir)
    state = {"episode": episode, "total_steps": total_steps, "t_env": runner.t_env,
             "train_step": learner.train_step, "last_target_update_step": learner.last_target_update_step,
             "epsilon": learner.mac.action_selector.epsilon,
             # positions of the random streams (plain ints: JSON, nothing pickled)
             "runner_ep": int(getattr(runner, "_ep", -1)),
             "sample_rng": learner._sample_rng.bit_generator.state}
    draws = getattr(learner, "_g_draws", None)
    if draws is not None:   # the device-side episode sampler's counter (core/qmix.py, enable_graphs)
        state["sample_draws"] = int(draws.item())
    ep_idx = getattr(runner.env, "episode_index", None)
    if ep_idx is not None:   # every env's episode index (its Monte-Carlo stream position): envs reset under a mask differ
        state["env_episode"] = int(ep_idx.max().item())
        state["env_episode_index"] = [int(v) for v in ep_idx.cpu().tolist()]
    with open(os.path.join(save_dir, "trainer_state.json"), "w") as f:
        json.dump(state, f)


def load_checkpoint(learner, runner, load_dir):
    learner.load_models(load_dir)
    opt_path = os.path.join(load_dir, "optimizer.pth")
    if os.path.exists(opt_path):
        learner.load_optimizer_state(torch.load(opt_path, map_location=learner.device, weights_only=True))
    state = {"episode": 0, "total_steps": 0}
    st_path = os.path.join(load_dir, "trainer_state.json")
    if os.path.exists(st_path):
        with open(st_path) as f:
            state = json.load(f)
        runner.t_env = int(state.get("t_env", 0))
        learner.train_step = int(state.get("train_step", 0))
        learner.last_target_update_step = int(state.get("last_target_update_step", 0))
        learner.mac.action_selector.epsilon = float(state.get("epsilon", learner.mac.action_selector.epsilon))
        if "runner_ep" in state and hasattr(runner, "_ep"):
            runner._ep = int(state["runner_ep"])
        if "sample_rng" in state:
            learner._sample_rng.bit_generator.state = state["sample_rng"]
        learner._resume_draws = int(state.get("sample_draws", 0))   # applied when enable_graphs creates the counter ...
        if getattr(learner, "_g_draws", None) is not None:          # ... or right away when it exists already
            learner._g_draws.fill_(learner._resume_draws)
        ep_idx = getattr(runner.env, "episode_index", None)
        if ep_idx is not None and "env_episode_index" in state:
            vec = np.asarray(state["env_episode_index"], dtype=np.int64)
            if vec.shape == tuple(ep_idx.shape):
                ep_idx.copy_(torch.as_tensor(vec).to(ep_idx.dtype))
            else:   # another batch size than the checkpoint's: every env continues behind the furthest one
                ep_idx.fill_(int(vec.max()))
        elif ep_idx is not None and "env_episode" in state:
            ep_idx.fill_(int(state["env_episode"]))
    return state


def evaluate(runner, n_episodes):
    """Greedy (test_mode) episodes: mean return, radar lock fraction and mean normalised power."""
    rets, locks, powers = [], [], []
    n_radars = runner.env.num_radars
    rd_pen = np.asarray(runner.env.scenario.tables["radar_rd_pen"])
    runs = max(1, int(np.ceil(n_episodes / getattr(runner, "batch_envs", 1))))
    for _ in range(runs):
        info = runner.run(test_mode=True)
        rets.append(info["episode_return"])
        powers.append(info["avg_power_overall"])
        # r_d is the sum of per-radar penalties over tracking radars; with the mean penalty this gives the
        # mean fraction of radars locked per step
        locks.append(float(info["avg_r_d"]) / float(rd_pen.mean()) / n_radars if rd_pen.mean() != 0 else 0.0)
    return {"return": float(np.mean(rets)), "lock_fraction": float(np.mean(locks)), "power": float(np.mean(powers))}


def run(args):
    from . import parallel
    rank, world, local_rank = parallel.init_distributed()   # no-op unless launched by torchrun (WORLD_SIZE > 1)
    device = _pick_device(args, local_rank)
    print(f"Using device: {args.device}" + (f" (rank {rank} of {world}, local device {local_rank})" if world > 1 else ""))
    np.random.seed(args.seed + rank)   # replay sampling stream of this rank's shard
    torch.manual_seed(args.seed)       # identical initial weights on every rank (broadcast below as well)
    if args.use_cuda:
        torch.cuda.manual_seed(args.seed)
    if world > 1:
        args.env_offset = rank * int(getattr(args, "batch_envs", 1) or 1)
    test_name = getattr(args, "test_name", "ma_cjd_test")
    run_name = f"run_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
    log_dir = os.path.join(getattr(args, "results_path", "logs") or "logs", test_name, run_name)
    writer = ScalarLog(log_dir)
    print(f"Logs will be saved to: {log_dir}")

    sim_config_path = getattr(args, "sim_config_path", None) or os.path.join("config", f"{args.env_config}.yaml")
    if args.use_cuda and getattr(args, "gemm_tuning", True):
        from . import ops
        ops.enable_gemm_tuning(os.path.join(log_dir, "tunableop_results.csv"))
    env, mac, buffer, learner, runner = build_components(args, sim_config_path)
    if world > 1:
        mac.select_seed = int(args.seed) + 1000 * rank          # every rank explores with its own stream
        learner._sample_rng = np.random.default_rng(int(args.seed) + rank)
        learner._sampler_seed_value = int(args.seed) + rank
        parallel.broadcast_parameters([mac.agent, learner.eval_qmix_net])
        learner._update_targets()
    batch_envs = int(getattr(args, "batch_envs", 1) or 1)
    episodes_per_run = batch_envs
    use_graphs = bool(getattr(args, "hip_graphs", True)) and args.use_cuda and batch_envs > 1
    # config/default.yaml is scaled for batch_envs = 4096 (epsilon annealed over 2500 steps PER ENV, 200 M collected steps);
    # the reference's own protocol (one env) anneals over 100 000 of its 2 M steps (reference config/default.yaml:21-23,60)
    if batch_envs == 1 and (args.epsilon_anneal_time < 20000 or args.total_env_steps > 20_000_000):
        print(f"WARNING: batch_envs = 1 with epsilon_anneal_time = {args.epsilon_anneal_time} / total_env_steps = "
              f"{args.total_env_steps}: these look like the values scaled for thousands of envs; the reference protocol uses "
              f"epsilon_anneal_time: 100000 and total_env_steps: 2000000")

    episode, total_steps = 0, 0
    if getattr(args, "resume", None):
        st = load_checkpoint(learner, runner, args.resume)
        episode, total_steps = int(st.get("episode", 0)), int(st.get("total_steps", 0))
        print(f"Resumed from {args.resume}: episode {episode}, env steps {total_steps}")

    n_upd = args.log_interval * (args.episode_limit // args.train_interval)
    log_stats = {k: deque(maxlen=args.log_interval) for k in
                 ("episode_return", "episode_length", "avg_step_reward", "reward_r_d", "reward_r_p", "reward_r_j",
                  "avg_power", "action_dist")}
    log_stats.update({k: deque(maxlen=n_upd) for k in ("loss", "grad_norm", "eval_qtot_avg", "target_qtot_avg")})
    start_time = last_log_time = time.time()
    next_test = total_steps + getattr(args, "test_interval", 0) if getattr(args, "test_interval", 0) else None
    graphs_on = False
    print("Starting training...")
    while total_steps < args.total_env_steps:
        run_info = runner.run(test_mode=False)                                   # main.py:193
        episode += episodes_per_run
        current_episode_steps = run_info["episode_length"]
        total_steps += current_episode_steps * episodes_per_run
        log_stats["episode_return"].append(run_info["episode_return"])
        log_stats["episode_length"].append(current_episode_steps)
        log_stats["avg_step_reward"].append(run_info.get("avg_step_reward", 0))
        log_stats["reward_r_d"].append(run_info.get("avg_r_d", 0))
        log_stats["reward_r_p"].append(run_info.get("avg_r_p", 0))
        log_stats["reward_r_j"].append(run_info.get("avg_r_j", 0))
        log_stats["avg_power"].append(run_info.get("avg_power_overall", 0))
        if "action_distribution" in run_info:
            log_stats["action_dist"].append(np.asarray(run_info["action_distribution"]))

        if buffer.current_size >= args.batch_size and total_steps > args.start_training_steps:   # main.py:212
            if use_graphs and not graphs_on:
                learner.enable_graphs(buffer, args.batch_size, updates_per_graph=int(getattr(args, "updates_per_graph", 20) or 1))
                runner.enable_graph()
                graphs_on = True
            num_train_steps = int(getattr(args, "updates_per_rollout", 0) or current_episode_steps // args.train_interval)
            pending = []
            # every update's four scalars are snapshotted into their own row on the device (the graphed update reuses
            # ONE static output tensor): read back once after the block
            hist = torch.empty((num_train_steps, 4), dtype=torch.float32, device=device) if graphs_on else None
            if graphs_on and num_train_steps > 0:
                # groups of `updates_per_graph` updates replay one graph each (their batches are drawn on the device)
                learner.train_from_buffer_many(num_train_steps, stats_out=hist)
                pending = [{"loss": hist[i, 0], "grad_norm": hist[i, 3], "eval_qtot_avg": hist[i, 1], "target_qtot_avg": hist[i, 2]}
                           for i in range(num_train_steps)]
            for i_upd in range(0 if graphs_on else num_train_steps):
                if graphs_on:
                    pending.append(learner.train_from_buffer(sync_stats=False, stats_row=hist[i_upd]))
                else:
                    batch = buffer.sample(args.batch_size)
                    if batch is not None:
                        pending.append(learner.train(batch, {"total_steps": total_steps}, sync_stats=not args.use_cuda))
            if pending:  # one host sync for the whole block of updates
                if torch.is_tensor(pending[0]["loss"]):
                    keys = list(pending[0].keys())
                    vals = torch.stack([torch.stack([torch.as_tensor(p[k], device=device, dtype=torch.float32).reshape(())
                                                     for k in keys]) for p in pending]).tolist()
                    pending = [dict(zip(keys, v)) for v in vals]
                for st in pending:
                    for k in ("loss", "grad_norm", "eval_qtot_avg", "target_qtot_avg"):
                        log_stats[k].append(st[k])
                writer.add_scalar("Loss/train_episode_avg", float(np.mean([p["loss"] for p in pending])), total_steps)

        if next_test is not None and total_steps >= next_test:
            ev = evaluate(runner, getattr(args, "test_nepisodes", 20))
            writer.add_scalar("Test/Avg_Return", ev["return"], total_steps)
            writer.add_scalar("Test/Lock_Fraction", ev["lock_fraction"], total_steps)
            writer.add_scalar("Test/Avg_Power", ev["power"], total_steps)
            print(f"  [eval @ {total_steps}] return {ev['return']:.2f} | lock fraction {ev['lock_fraction']:.3f} | "
                  f"power {ev['power']:.3f}")
            next_test += args.test_interval

        now = time.time()
        if (now - last_log_time) >= args.log_interval_seconds or total_steps >= args.total_env_steps:   # main.py:231
            m = lambda k: float(np.mean(log_stats[k])) if log_stats[k] else 0.0
            dist = np.mean(np.array(log_stats["action_dist"]), axis=0) if log_stats["action_dist"] else np.zeros(args.n_actions)
            print(f"Steps: {total_steps}/{args.total_env_steps} | Episodes: {episode} | Time: {now - start_time:.2f}s")
            print(f"  Avg Return (last {len(log_stats['episode_return'])} eps): {m('episode_return'):.2f} | "
                  f"Avg Length: {m('episode_length'):.1f} | Avg Loss: {m('loss'):.4f}")
            print(f"  Avg Step Reward (last {len(log_stats['avg_step_reward'])} eps): {m('avg_step_reward'):.4f}")
            print(f"  Avg Rewards (r_d/r_p/r_j): {m('reward_r_d'):.4f} / {m('reward_r_p'):.4f} / {m('reward_r_j'):.4f}")
            print(f"  Avg QTot (Eval/Target): {m('eval_qtot_avg'):.4f} / {m('target_qtot_avg'):.4f} | "
                  f"Avg Grad Norm: {m('grad_norm'):.4f}")
            print(f"  Avg Power: {m('avg_power'):.3f} | Action Dist: [{' / '.join(f'{p:.2f}' for p in dist)}] "
                  f"(0=Idle, 1=S0, 2=D0, ...)")
            print(f"  Buffer Size: {len(buffer)}")
            print(f"  Epsilon: {mac.action_selector.epsilon:.3f}")
            for tag, val in (("Perf/Avg_Return", m("episode_return")), ("Perf/Avg_Length", m("episode_length")),
                             ("Perf/Avg_Step_Reward", m("avg_step_reward")), ("Loss/train_avg", m("loss")),
                             ("Params/Epsilon", mac.action_selector.epsilon), ("Params/Buffer_Size", len(buffer)),
                             ("Stats/grad_norm", m("grad_norm")), ("QValues/eval_qtot_avg", m("eval_qtot_avg")),
                             ("QValues/target_qtot_avg", m("target_qtot_avg")), ("Rewards/r_d_avg", m("reward_r_d")),
                             ("Rewards/r_p_avg", m("reward_r_p")), ("Rewards/r_j_avg", m("reward_r_j")),
                             ("Perf/Avg_Power", m("avg_power"))):
                writer.add_scalar(tag, val, total_steps)
            for act_idx, act_prob in enumerate(dist):
                writer.add_scalar(f"ActionDist/Action_{act_idx}", act_prob, total_steps)
            last_log_time = now

        boundary = (episode // args.save_interval) != ((episode - episodes_per_run) // args.save_interval)
        if args.save_model and rank == 0 and (boundary or total_steps >= args.total_env_steps):          # main.py:283-289
            if total_steps > args.start_training_steps:
                save_dir = os.path.join(args.save_model_dir, test_name, f"step_{total_steps}")
                print(f"Saving model to {save_dir}")
                save_checkpoint(learner, runner, save_dir, episode, total_steps)

    if hasattr(env, "close") and callable(env.close):
        env.close()
    writer.close()
    if world > 1:
        torch.distributed.barrier()
    print("Training finished.")
    return {"episodes": episode, "total_steps": total_steps, "log_dir": log_dir, "train_steps": learner.train_step}


def main(argv=None):
    parser = argparse.ArgumentParser(description="MA-CJD QMix Training (MI355X hot path)")
    parser.add_argument("--config", type=str, default="default")
    parser.add_argument("--env-config", type=str, default="simulation_config")
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--config-dir", type=str, default="config")
    parser.add_argument("--batch-envs", type=int, default=None)
    parser.add_argument("--resume", type=str, default=None)
    parser.add_argument("--randomize-scenarios", action="store_true",
                        help="batched runs: every env gets its own random variation of the scenario (per-env tables)")
    a = parser.parse_args(argv)
    config = load_config(config_name=a.config, config_dir=a.config_dir)
    config.config, config.env_config, config.device_request = a.config, a.env_config, a.device
    config.sim_config_path = os.path.join(a.config_dir, f"{a.env_config}.yaml")
    if a.batch_envs is not None:
        config.batch_envs = a.batch_envs
    config.resume = a.resume
    if a.randomize_scenarios:
        config.randomize_scenarios = True
    return run(config)


if __name__ == "__main__":
    main()
