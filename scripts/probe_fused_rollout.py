#!/usr/bin/env python3
"""Times the pieces of the fused episode rollout at the benchmark's size (E = 4096, 3j/4r, T = 100): the agent-episode
launch, the many-step env launch, the replay store, and the whole runner.run()."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as entry

if not os.environ.get("MACJD_LIB"):
    entry.build()
from macjd_amd import bench_rollout, ops
from macjd_amd.core.mac import BasicMAC
from macjd_amd.runners.episode_runner import BatchedEpisodeRunner
from macjd_amd.scenario import Scenario, ring_scenario_dict
from macjd_amd.simulation.environment import BatchedElectromagneticEnvironment
from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer

dev = torch.device("cuda", 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = Scenario.from_dict(ring_scenario_dict(3, 4))
args = bench_rollout.make_args(sc, 64, dev, batch_envs=E)
env = BatchedElectromagneticEnvironment(scenario=sc, batch_envs=E, device=dev, seed=42)
with contextlib.redirect_stdout(io.StringIO()):
    mac = BasicMAC(args.obs_shape, args)
    mac.cuda()
    buf = EpisodeReplayBuffer(args, device=dev)
runner = BatchedEpisodeRunner(env, mac, buf, args)
assert runner.fused_rollout_available()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


runner.rollout_fused()
st, T, J = runner.stage, runner.episode_limit, runner.n_agents
a = mac.agent
l1, l2 = a.fc2_q_head[0], a.fc2_q_head[2]
params, gi = mac.static_inputs
us_agent = timed(lambda: ops.agent_episode(gi, params, None, a.rnn.weight_hh, a.rnn.bias_hh, l1.weight, l1.bias, l2.weight,
                                           l2.bias, E, J, T, runner._avail, runner._eps_sched, False, 1, runner._ctr_base,
                                           st["hidden_state"], st["actions_discrete"], st["actions_continuous"],
                                           h_final=mac.hidden_states))
us_env = timed(lambda: env.step_many(st["actions_discrete"], st["actions_continuous"], st["reward"], st["terminated"],
                                     runner._rdpj_steps, rdpj_sum=runner._rdpj_sum))
us_store = timed(lambda: runner.end_episodes())
us_roll = timed(lambda: runner.rollout_fused())
us_run = timed(lambda: runner.run(sync_stats=False))
flops = 2.0 * E * J * T * (64 * 192 + 64 * 64)
print(f"E={E}: agent_episode {us_agent:8.1f} us ({us_agent / T:5.2f} us/step, {flops / us_agent / 1e6:6.2f} TFLOP/s in the two products)"
      f"   env.step_many {us_env:7.1f} us   replay store {us_store:7.1f} us   rollout_fused {us_roll:8.1f} us   run() {us_run:8.1f} us")
