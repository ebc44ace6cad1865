"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: env sharding arithmetic, the single
flat-gradient all-reduce of the learner, identical weights on all ranks after a step, and equivalence of
the averaged gradient with a single process training on the concatenated batch."""
import contextlib
import io
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_nets_cpu import load, make_args, sd_from
from tests_golden_helpers import synthetic_batch  # noqa: F401  (defined below via conftest path)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(g, d, args):
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "g5_agent0."))
        learner = QMixLearner(mac, args)
    learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
    learner._update_targets()
    return mac, learner


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from macjd_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    g, d = load("3j4r_h64")
    args = make_args(d)
    mac, learner = _build(g, d, args)
    if rank == 1:  # deliberately different start on rank 1: broadcast must fix it
        with torch.no_grad():
            for p in learner.eval_qmix_net.parameters():
                p.add_(0.5)
    parallel.broadcast_parameters([mac.agent, learner.eval_qmix_net, learner.target_mac.agent,
                                   learner.target_qmix_net])
    learner._update_targets()
    stats = []
    for step in range(2):
        batch = synthetic_batch(np.random.default_rng(500 + 10 * step + rank), args, 4, 12)
        stats.append(learner.train(batch, {}))
    flat = torch.cat([p.detach().reshape(-1) for p in learner._trainable()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged after all-reduced steps"
    assert learner._flat_grad is not None          # world_size 2: the flat all-reduce buffer is in use
    assert all(p.grad.data_ptr() >= learner._flat_grad.data_ptr() for p in learner._trainable())
    np.save(os.path.join(out_dir, f"grad_rank{rank}.npy"), learner.grad_vector().numpy())
    np.save(os.path.join(out_dir, f"w_rank{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


def test_shard_range():
    from macjd_amd.parallel import shard_range
    assert [shard_range(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    cover = [shard_range(10, r, 4) for r in range(4)]
    assert cover == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def test_two_rank_gradient_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "grad_rank0.npy"), np.load(tmp_path / "grad_rank1.npy")
    np.testing.assert_array_equal(g0, g1)                      # same averaged gradient on both ranks
    np.testing.assert_array_equal(np.load(tmp_path / "w_rank0.npy"), np.load(tmp_path / "w_rank1.npy"))
    # single process on the concatenated global batch (B = 8): same gradient, same weights
    g, d = load("3j4r_h64")
    args = make_args(d)
    mac, learner = _build(g, d, args)
    for step in range(2):
        parts = [synthetic_batch(np.random.default_rng(500 + 10 * step + r), args, 4, 12) for r in range(2)]
        batch = {k: (np.concatenate([p[k] for p in parts], axis=0) if isinstance(parts[0][k], np.ndarray) else parts[0][k])
                 for k in parts[0]}
        learner.train(batch, {})
    # (after clipping both took the same Adam step, so the last-step gradients agree as well)
    assert learner._flat_grad is None              # single process: gradients stay where autograd put them
    np.testing.assert_allclose(learner.grad_vector().numpy(), g0, rtol=2e-4, atol=1e-6)
    flat = torch.cat([p.detach().reshape(-1) for p in learner._trainable()]).numpy()
    np.testing.assert_allclose(flat, np.load(tmp_path / "w_rank0.npy"), rtol=0, atol=2e-6)


def test_bench_step_accounting_is_region_aligned():
    """bench_rollout.make_step's schedule: episodes restart at the warm-up / timed boundary, whole episodes replay the
    full graph, the partial episode at the end of a region replays its own shorter graph, so the steps a caller times
    do exactly the work of those steps (a whole-episode graph launched in warm-up used to pre-pay timed steps)."""
    from types import SimpleNamespace
    import macjd_amd.bench_rollout as br

    T = 100

    class FakeRunner:
        def __init__(self):
            self.log = []

        def rollout_graphed(self, n=None):
            self.log.append(("graph", T if n is None else n))

        def begin_episodes(self):
            self.log.append(("begin", 0))

        def step(self, t):
            self.log.append(("step", t))

        def end_episodes(self):
            self.log.append(("end", 0))

    def schedule(warmup, steps):
        runner = FakeRunner()
        cli = SimpleNamespace(warmup=warmup, steps=steps)
        step_fn = br._make_step_fn(cli, runner, None, None, None, T, "rollout", True,
                                   {r % T for r in (warmup, steps) if r % T})
        per_step = []
        for i in range(warmup + steps):
            n0 = len(runner.log)
            step_fn(i)
            per_step.append(runner.log[n0:])
        return per_step

    for warmup, steps in ((20, 200), (100, 1000), (2, 5), (30, 250), (0, 100)):
        per_step = schedule(warmup, steps)
        timed = per_step[warmup:]
        work = sum(n for ev in timed for kind, n in ev if kind == "graph") + sum(1 for ev in timed for kind, _ in ev if kind == "step")
        assert work == steps, (warmup, steps, work)                       # timed steps do exactly their own work
        pre = (sum(n for ev in per_step[:warmup] for kind, n in ev if kind == "graph")
               + sum(1 for ev in per_step[:warmup] for kind, _ in ev if kind == "step"))
        assert pre == warmup
        assert not any(kind == "step" for ev in per_step for kind, _ in ev)   # every step comes from a graph
        assert sum(1 for ev in timed for kind, _ in ev if kind == "end") == steps // T


def test_bench_train_schedule_issues_one_update_per_step_in_groups():
    """bench_rollout's train schedule with grouped updates: the updates of steps j-K+1 .. j are issued together after step j
    (one replayed graph), a region's remainder is flushed at its last step — every region does exactly one update per step,
    none is carried from the warm-up into the timed region — and the fused rollout launches whole (or region-cut) episodes."""
    from types import SimpleNamespace
    import macjd_amd.bench_rollout as br

    T, K = 100, 20

    class FakeRunner:
        def __init__(self):
            self.log = []

        def fused_rollout_available(self):
            return True

        def rollout_fused(self, n_steps=None):
            self.log.append(("rollout", T if n_steps is None else n_steps))

        def end_episodes(self):
            self.log.append(("end", 0))

    class FakeLearner:
        _g_multi = (K, None, None)

        def __init__(self, log):
            self.log = log

        def train_from_buffer_many(self, n):
            self.log.append(("updates", n))

    for warmup, steps in ((5, 20), (100, 1000), (30, 250), (0, 100), (7, 45)):
        runner = FakeRunner()
        learner = FakeLearner(runner.log)
        cli = SimpleNamespace(warmup=warmup, steps=steps)
        step_fn = br._make_step_fn(cli, runner, learner, None, SimpleNamespace(batch_size=32), T, "train", True, set())
        marks = []
        for i in range(warmup + steps):
            n0 = len(runner.log)
            step_fn(i)
            marks.append(runner.log[n0:])
        for lo, hi, n in ((0, warmup, warmup), (warmup, warmup + steps, steps)):
            evs = [e for m in marks[lo:hi] for e in m]
            assert sum(c for k, c in evs if k == "updates") == n, (warmup, steps)
            assert sum(c for k, c in evs if k == "rollout") == n, (warmup, steps)
            assert all(c <= K for k, c in evs if k == "updates")
        # inside the timed region full groups are the rule: at most one short group per episode boundary / region end
        timed = [c for m in marks[warmup:] for k, c in m if k == "updates"]
        assert sum(1 for c in timed if c < K) <= 1
