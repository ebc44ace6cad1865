"""Two ranks sharing the one GPU of the test box (gloo carries the HIP tensors): the graphed learner update of the
multi-GPU layout — graph A (forward / backward, weight gradients written into the flat gradient vector), the all-reduce
of that ONE vector, graph B (clip + Adam) — keeps the ranks' weights identical and equals a single process that
trains on the concatenated batch.  RCCL itself needs one device per rank, which a one-GPU box cannot give; what is
exercised here is everything around the collective call."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_nets_cpu import load, make_args, sd_from
from tests_golden_helpers import synthetic_batch

pytestmark = pytest.mark.gpu
T, N, B, STEPS = 100, 40, 16, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(seed):
    from macjd_amd.core.mac import BasicMAC
    from macjd_amd.core.qmix import QMixLearner
    from macjd_amd.utils.replay_buffer import EpisodeReplayBuffer
    g, d = load("3j4r_h64")
    args = make_args(d, device="cuda", use_cuda=True, episode_limit=T, buffer_size=N, batch_size=B,
                     target_update_interval=2)
    with contextlib.redirect_stdout(io.StringIO()):
        mac = BasicMAC(d["S"], args)
        mac.load_state(sd_from(g, "g5_agent0."))
        learner = QMixLearner(mac, args)
        buf = EpisodeReplayBuffer(args)
    learner.eval_qmix_net.load_state_dict(sd_from(g, "g5_mixer0."))
    learner._update_targets()
    full = synthetic_batch(np.random.default_rng(seed), args, N, T)
    for k, v in buf.buffers.items():
        v.copy_(torch.as_tensor(full[k]).to(v.dtype))
    buf.current_size, buf.current_index = N, 0
    buf.episode_lengths[:] = T
    return args, learner, buf


def _indices(step):
    return np.random.default_rng(100 + step).choice(N, B, replace=False)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    _, learner, buf = _build(seed=7 + rank)              # every rank owns its replay shard
    learner.enable_graphs(buf, B)
    assert not learner._g_single, "two ranks: the update must be two graphs around the all-reduce"
    for step in range(STEPS):
        learner.train_from_buffer(indices=_indices(step), sync_stats=False)
    torch.cuda.synchronize()
    assert learner.grad_pack_launches == 0              # the all-reduce buffer was filled by the kernels themselves
    flat = torch.cat([p.detach().reshape(-1) for p in learner._trainable()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
    np.save(os.path.join(out_dir, f"w_rank{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


def test_two_ranks_graphed_update_equals_single_process_on_the_global_batch(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    w0 = np.load(tmp_path / "w_rank0.npy")
    np.testing.assert_array_equal(w0, np.load(tmp_path / "w_rank1.npy"))
    # single process, eager, on the concatenation of both ranks' sampled episodes (global batch 2 B)
    args, learner, buf0 = _build(seed=7)
    _, _, buf1 = _build(seed=8)
    for step in range(STEPS):
        idx = _indices(step)
        parts = [b.sample(B, indices=idx) for b in (buf0, buf1)]
        batch = {k: (torch.cat([torch.as_tensor(p[k]) for p in parts], dim=0) if torch.is_tensor(parts[0][k]) or
                     isinstance(parts[0][k], np.ndarray) else parts[0][k]) for k in parts[0]}
        learner.train(batch, {})
    flat = torch.cat([p.detach().reshape(-1) for p in learner._trainable()]).cpu().numpy()
    # equal filled-step counts on both ranks => mean of the rank gradients == the global-batch gradient; Adam's
    # m / sqrt(v) amplifies summation-order noise of near-zero entries, hence 2e-5 (as in the graph-vs-eager test)
    np.testing.assert_allclose(flat, w0, rtol=0, atol=2e-5)


def _graphed_allreduce_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    _, plain, buf_p = _build(seed=7)
    _, graphed, buf_g = _build(seed=7)
    plain.enable_graphs(buf_p, B, updates_per_graph=2, graphed_allreduce=False)
    graphed.enable_graphs(buf_g, B, updates_per_graph=2, graphed_allreduce=True)
    assert graphed._g_graphed_ar and graphed._g_single and graphed._g_multi[0] == 2
    assert not plain._g_graphed_ar
    plain.train_from_buffer_many(5)
    graphed.train_from_buffer_many(5)
    torch.cuda.synchronize()
    a = torch.cat([p.detach().reshape(-1) for p in plain._trainable()]).cpu().numpy()
    b = torch.cat([p.detach().reshape(-1) for p in graphed._trainable()]).cpu().numpy()
    np.save(os.path.join(out_dir, "plain.npy"), a)
    np.save(os.path.join(out_dir, "graphed.npy"), b)
    graphed.release_graphs()
    dist.destroy_process_group()


def test_gradient_allreduce_captured_inside_the_update_graph(tmp_path):
    """MACJD_GRAPHED_ALLREDUCE / enable_graphs(graphed_allreduce=True): the RCCL all-reduce of the flat gradient vector is
    recorded into the update's HIP graph, so ranks keep the one-graph-per-update and K-updates-per-graph replay of a
    single process.  A one-GPU box can only host a ONE-rank RCCL group: this checks the mechanics — the collective is
    captured, grouped replays work, the (identity) all-reduce leaves the update sequence bit for bit — not the exchange
    between ranks (tests/test_dist_cpu.py and the two-rank gloo test above cover the arithmetic of that)."""
    mp.spawn(_graphed_allreduce_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "plain.npy"), np.load(tmp_path / "graphed.npy"))
