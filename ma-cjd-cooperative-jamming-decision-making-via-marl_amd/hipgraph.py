"""Capture / replay / teardown of the HIP graphs this package builds (learner update, step-by-step rollout).

**The fault this module exists for** (round 2: host SIGSEGV inside ``hipGraphLaunch``, always in a learner's
``replay()``; native trace: profiles/r03_hipgraphlaunch_fault_native_trace.txt).  A graph with parallel
branches runs its extra branches on streams owned by the instantiated graph.  The HIP runtime inside the torch wheel
(clr of ROCm 7.0.2, ``torch/lib/libamdhip64.so``) creates ``max_streams`` such streams per graph
(``GraphExec::Init`` +0xafe20 -> ``CreateStreams`` +0xaeb30) and, at every launch, ``Graph::UpdateStreams`` (+0xaed90,
called from ``GraphExec::Run`` +0xaf91f) hands the branches to those whose HARDWARE QUEUE differs from the launch
stream's, skipping the others — the skip index is never checked against the vector's size.  ``max_streams`` streams for
``max_streams - 1`` branches leave room for ONE skip.  A stream gets its hardware queue on first use, from a pool of
``GPU_MAX_HW_QUEUES`` (4) normal-priority queues: the one with the fewest users, the first such in address order.  While
streams are only ever created, consecutive streams land on distinct queues (round robin) and at most one of a graph's
streams meets the launch stream's queue.  Every DESTROYED graph returns its streams' queue references; from the uneven
counts that leaves behind, two streams of a later graph can both be given the launch stream's queue — then the loop
reads ``parallel_streams[max_streams]``, one element past the heap buffer (the trace shows 0x81, the next malloc
chunk's size field), and calls through it.  Whether that happens depends on the whole history of stream / graph
creations and destructions of the process, which is why only full test sessions showed it, and deterministically so
once every test's graphs were torn down.

What the package does about it:

1. **Replays are launched from a high-priority stream** (``replay``).  Queue pools are per priority and a graph's own
   streams have normal priority, so that launch stream can never share a hardware queue with any of them: no skip, no
   read past the end, whatever was destroyed before.  One more hardware queue must not push the process over the four
   the GPU schedules without time-slicing (a fifth queue puts two inter-dependent branches on one hardware pipe and
   every hand-over between them costs ~50 us: train 0.14 -> 0.47 ms / step, measured), so the package reserves one:
   ``GPU_MAX_HW_QUEUES=3`` is put into the environment when the package is imported before the HIP runtime
   initialised (``reserve_launch_queue``; 3 pooled queues + the launch queue = 4; +2 % on the train step against
   the unprotected arrangement).
2. Where the reservation could not be made (runtime already up, or the user set ``GPU_MAX_HW_QUEUES`` > 3) replays stay
   on the caller's stream and **no captured graph is ever destroyed** (``capture`` keeps it, ``destroy`` parks it):
   with creations only, the queue counts stay in the round-robin state in which the runtime's one spare stream suffices.
   ``MACJD_GRAPH_REPLAY_STREAM=current|high`` forces either arrangement (A/B timing, tests).
3. **Nothing is collected during a capture** (``capture``): Python's cyclic collector may otherwise run the destructor
   of an unrelated dead ``CUDAGraph`` inside the capture — ``hipGraphExecDestroy`` plus a device synchronisation on
   the capturing thread.  Dead graphs are collected right before the capture starts.
4. **Graphs are destroyed explicitly, at a quiet point, newest first** (``destroy``), by the object that owns them
   (``QMixLearner.release_graphs``, ``BatchedEpisodeRunner.release_graphs``), not whenever the collector finds them.
"""
from __future__ import annotations

import contextlib
import gc
import os
import warnings

import torch

RESERVED_POOL_QUEUES = 3          # + the high-priority launch queue = the 4 hardware queues the runtime defaults to
_REPLAY_STREAMS = {}
_KEPT_GRAPHS = []                 # arrangement 2: graphs that must outlive their owners
_reserved = None                  # True: GPU_MAX_HW_QUEUES <= 3 is in effect for this process


def _hip_runtime_initialised() -> bool:
    """The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises, which opens the KFD device node."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink(f"/proc/self/fd/{fd}").endswith("/dev/kfd"):
                    return True
            except OSError:
                continue
    except OSError:
        pass
    return False


def reserve_launch_queue() -> bool:
    """Called when the package is imported: make room for the launch queue (see the module docstring, 1).  Returns
    whether the reservation holds; decided once per process."""
    global _reserved
    if _reserved is not None:
        return _reserved
    val = os.environ.get("GPU_MAX_HW_QUEUES")
    if val is not None:
        try:
            _reserved = int(val) <= RESERVED_POOL_QUEUES
        except ValueError:
            _reserved = False
    elif _hip_runtime_initialised():
        _reserved = False
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = str(RESERVED_POOL_QUEUES)
        _reserved = True
    return _reserved


def launch_mode() -> str:
    """"high": replays go through the high-priority stream; "current": through the caller's stream, graphs immortal."""
    forced = os.environ.get("MACJD_GRAPH_REPLAY_STREAM")
    if forced in ("high", "current"):
        return forced
    return "high" if reserve_launch_queue() else "current"


def replay_stream(device) -> "torch.cuda.Stream":
    """The per-device high-priority stream graph replays are launched from."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _REPLAY_STREAMS.get(idx)
    if s is None:
        s = _REPLAY_STREAMS[idx] = torch.cuda.Stream(device=idx, priority=-1)
    return s


def replay(graph: "torch.cuda.CUDAGraph", device) -> None:
    """``graph.replay()`` ordered like a launch on the caller's current stream — everything enqueued there before runs
    first, everything enqueued there afterwards waits for the graph — but launched from ``replay_stream(device)``."""
    if launch_mode() == "current":
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
    and held off during the capture.  thread_local: other threads (RCCL's watchdog polls events) may touch the runtime
    meanwhile."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
            yield graph
    finally:
        if was_enabled:
            gc.enable()
    if launch_mode() == "current":
        if not _KEPT_GRAPHS:
            warnings.warn("macjd_amd.hipgraph: no hardware queue could be reserved for graph launches (import the package "
                          "before the first torch.cuda call, GPU_MAX_HW_QUEUES unset or <= 3); captured HIP graphs are "
                          "kept alive until the process ends instead (see the module docstring)")
        _KEPT_GRAPHS.append(graph)


def destroy(graphs, device) -> None:
    """Destroy captured graphs NOW: device idle before and after, in the order given (pass the newest first: graphs
    captured into another graph's pool go before the pool's owner).  In the "current" arrangement the graphs are parked
    instead (they stay in ``_KEPT_GRAPHS``)."""
    graphs = [g for g in graphs if g is not None]
    if not graphs or launch_mode() == "current":
        return
    torch.cuda.synchronize(device)
    for g in graphs:
        g.reset()
    torch.cuda.synchronize(device)
