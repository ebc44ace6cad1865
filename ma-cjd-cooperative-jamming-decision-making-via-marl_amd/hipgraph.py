"""Capture / replay / teardown of the HIP graphs this package builds (learner update, step-by-step rollout).

Three rules, each of them the fix of an observed failure ("the hipGraphLaunch host fault"):

1. **Replays are launched from a high-priority stream** (``replay``).  A graph with parallel branches runs its extra
   branches on streams the instantiated graph owns.  The HIP runtime that ships inside the torch wheel (clr of ROCm
   7.0.2, ``torch/lib/libamdhip64.so``) creates ``max_streams`` of them per graph and, at every launch, hands the
   branches to those whose hardware queue differs from the launch stream's, skipping the others — with no bounds
   check on the skip (``Graph::UpdateStreams``, libamdhip64.so+0xaed90 <- ``GraphExec::Run`` +0xaf91f).  One spare
   stream covers ONE skip.  Hardware queues are handed out from a pool of ``GPU_MAX_HW_QUEUES`` (4) normal-priority
   queues by smallest reference count; while streams only ever get created the counts stay level and consecutive
   streams land on distinct queues, but every destroyed graph returns its streams' references, and once the counts
   are uneven two of a new graph's streams can get the launch stream's queue: the loop then reads one element past
   the vector (the next malloc chunk's size field, 0x81 in the captured trace) and dereferences it — a host SIGSEGV
   inside hipGraphLaunch, depending on the whole history of graph creations and destructions in the process.
   Queue pools are per priority and the graph's own streams are created with normal priority, so a launch stream of
   HIGH priority can never share a queue with any of them: no skip, no read past the end, whatever was destroyed
   before.  ``MACJD_GRAPH_REPLAY_STREAM=current`` launches from the caller's stream instead (A/B timing only).
2. **Nothing is collected during a capture** (``capture``): Python's cyclic collector may otherwise run the
   destructor of an unrelated, dead ``CUDAGraph`` inside the capture — ``hipGraphExecDestroy`` and a device
   synchronisation on the capturing thread.  Dead graphs are collected right before the capture starts instead.
3. **Graphs are destroyed explicitly, at a quiet point, newest first** (``destroy``), by the object that owns them
   (``QMixLearner.release_graphs``, ``BatchedEpisodeRunner.release_graphs``), not whenever the collector finds them.
"""
from __future__ import annotations

import contextlib
import gc
import os

import torch

_REPLAY_STREAMS = {}


def replay_stream(device) -> "torch.cuda.Stream":
    """The per-device high-priority stream graph replays are launched from (rule 1 of the module docstring)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _REPLAY_STREAMS.get(idx)
    if s is None:
        s = _REPLAY_STREAMS[idx] = torch.cuda.Stream(device=idx, priority=-1)
    return s


def replay(graph: "torch.cuda.CUDAGraph", device) -> None:
    """``graph.replay()`` ordered like a launch on the caller's current stream — everything enqueued there before runs
    first, everything enqueued there afterwards waits for the graph — but launched from ``replay_stream(device)``."""
    if os.environ.get("MACJD_GRAPH_REPLAY_STREAM", "high") == "current":
        graph.replay()
        return
    cur = torch.cuda.current_stream(device)
    ls = replay_stream(device)
    if cur == ls:
        graph.replay()
        return
    ls.wait_stream(cur)
    with torch.cuda.stream(ls):
        graph.replay()
    cur.wait_stream(ls)


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", pool=None):
    """``torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")`` with the cyclic collector run before
    and held off during the capture (rule 2).  thread_local: other threads (RCCL's watchdog polls events) may touch
    the runtime meanwhile."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
            yield graph
    finally:
        if was_enabled:
            gc.enable()


def destroy(graphs, device) -> None:
    """Destroy captured graphs NOW (rule 3): device idle before and after, in the order given (pass the newest first:
    graphs captured into another graph's pool go before the pool's owner)."""
    graphs = [g for g in graphs if g is not None]
    if not graphs:
        return
    torch.cuda.synchronize(device)
    for g in graphs:
        g.reset()
    torch.cuda.synchronize(device)
