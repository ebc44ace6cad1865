"""Training driver: schedule, scalar tags, checkpoint layout, resume and
the greedy evaluation loop, on the batched hot path and on the single-env reference protocol."""
import contextlib
import io
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from _harness import REPO

pytestmark = pytest.mark.gpu
PKG = os.path.join(REPO, "ma-cjd-cooperative-jamming-decision-making-via-marl_amd")

REFERENCE_TAGS = {"Perf/Avg_Return", "Perf/Avg_Length", "Perf/Avg_Step_Reward", "Loss/train_avg", "Loss/train_episode_avg",
                  "Params/Epsilon", "Params/Buffer_Size", "Stats/grad_norm", "QValues/eval_qtot_avg",
                  "QValues/target_qtot_avg", "Rewards/r_d_avg", "Rewards/r_p_avg", "Rewards/r_j_avg", "Perf/Avg_Power"}


def _cfg(tmp_path, **kw):
    from macjd_amd.main import load_config
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = load_config("default", os.path.join(PKG, "config"))
    cfg.device_request = "cuda"
    cfg.sim_config_path = os.path.join(PKG, "config", "scenario_3j4r.yaml")
    cfg.save_model_dir = str(tmp_path / "models")
    cfg.results_path = str(tmp_path / "logs")
    cfg.log_interval_seconds = 0
    cfg.gemm_tuning = False   # default library heuristics in tests (TunableOp is process-global)
    cfg.resume = None
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _scalars(log_dir):
    rows = [json.loads(l) for l in open(os.path.join(log_dir, "scalars.jsonl"))]
    return rows, {r["tag"] for r in rows}


def test_batched_training_run_checkpoint_resume_eval(tmp_path):
    from macjd_amd.main import run
    E = 64
    cfg = _cfg(tmp_path, batch_envs=E, buffer_size=4 * E, total_env_steps=3 * E * 100, start_training_steps=0,
               save_interval=2 * E, test_interval=2 * E * 100, test_nepisodes=E, batch_size=16, lr=1e-4)
    out_txt = io.StringIO()
    with contextlib.redirect_stdout(out_txt):
        res = run(cfg)
    assert res["total_steps"] == 3 * E * 100 and res["episodes"] == 3 * E
    assert res["train_steps"] == 3 * 100                # episode_len // train_interval updates per rollout
    rows, tags = _scalars(res["log_dir"])
    assert REFERENCE_TAGS <= tags and {f"ActionDist/Action_{a}" for a in range(9)} <= tags
    assert {"Test/Avg_Return", "Test/Lock_Fraction", "Test/Avg_Power"} <= tags
    lock = [r["value"] for r in rows if r["tag"] == "Test/Lock_Fraction"]
    assert all(0.0 <= v <= 1.0 for v in lock)
    txt = out_txt.getvalue()
    assert "Avg Rewards (r_d/r_p/r_j)" in txt and "Training finished." in txt
    ckpts = sorted(os.listdir(os.path.join(cfg.save_model_dir, cfg.test_name)))
    assert f"step_{2 * E * 100}" in ckpts and f"step_{3 * E * 100}" in ckpts
    last = os.path.join(cfg.save_model_dir, cfg.test_name, f"step_{3 * E * 100}")
    assert sorted(os.listdir(last)) == ["agent.pth", "optimizer.pth", "qmix_net.pth", "trainer_state.json"]
    st = json.load(open(os.path.join(last, "trainer_state.json")))
    # t_env also counts the greedy evaluation rollout: the reference's runner advances its epsilon clock
    # in test_mode too (episode_runner.py:119)
    assert st["total_steps"] == 3 * E * 100 and st["t_env"] == 400 and st["train_step"] == 300
    sd = torch.load(os.path.join(last, "agent.pth"), weights_only=True)
    assert "fc2_q_head.0.weight" in sd and "rnn.weight_hh" in sd
    # resume: continues the counters and the epsilon clock
    cfg2 = _cfg(tmp_path, batch_envs=E, buffer_size=4 * E, total_env_steps=4 * E * 100, start_training_steps=0,
                save_interval=10 ** 9, batch_size=16, lr=1e-4, resume=last, hip_graphs=False)
    with contextlib.redirect_stdout(io.StringIO()) as o2:
        res2 = run(cfg2)
    assert "Resumed from" in o2.getvalue()
    assert res2["total_steps"] == 4 * E * 100 and res2["episodes"] == 4 * E
    assert res2["train_steps"] == 300 + 100


def test_single_env_reference_protocol_run(tmp_path):
    """batch_envs = 1: the reference's own schedule (one episode, then episode_len // train_interval updates)."""
    from macjd_amd.main import run
    cfg = _cfg(tmp_path, batch_envs=1, buffer_size=8, total_env_steps=300, start_training_steps=150, batch_size=2,
               save_interval=1, train_interval=50, hip_graphs=False)
    with contextlib.redirect_stdout(io.StringIO()):
        res = run(cfg)
    assert res["episodes"] == 3 and res["total_steps"] == 300
    assert res["train_steps"] == 2 * 2                  # training starts once total_steps > 150 and 2 episodes stored
    _, tags = _scalars(res["log_dir"])
    assert REFERENCE_TAGS <= tags
    assert sorted(os.listdir(os.path.join(cfg.save_model_dir, cfg.test_name))) == ["step_200", "step_300"]


def test_batched_training_on_randomised_scenarios(tmp_path):
    """randomize_scenarios: every env trains on its own variation of the scenario file (per-env tables in HBM);
    the driver, graphed rollout and graphed update run unchanged and the per-env observations reach the replay."""
    from macjd_amd.main import build_components, run
    E = 64
    cfg = _cfg(tmp_path, batch_envs=E, buffer_size=2 * E, total_env_steps=2 * E * 100, start_training_steps=0,
               save_interval=10 ** 9, batch_size=16, lr=1e-4, randomize_scenarios=True)
    with contextlib.redirect_stdout(io.StringIO()):
        res = run(cfg)
    assert res["total_steps"] == 2 * E * 100 and res["train_steps"] == 2 * 100
    rows, tags = _scalars(res["log_dir"])
    assert REFERENCE_TAGS <= tags
    assert all(np.isfinite(r["value"]) for r in rows)
    with contextlib.redirect_stdout(io.StringIO()):
        env, mac, buffer, learner, runner = build_components(cfg, cfg.sim_config_path)[:5]
        runner.run(test_mode=False)
    obs = buffer.buffers["obs"][:E, 0, 0]                       # [E, S]: one row per env, all different
    assert torch.unique(obs, dim=0).shape[0] == E
    assert torch.equal(obs, env.get_state())
