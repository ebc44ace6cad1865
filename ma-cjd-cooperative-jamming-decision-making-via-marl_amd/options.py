"""The package's MACJD_* switches: read from the environment ONCE (first use) into a table; ``reload()`` re-reads them
(tests and A/B scripts that change a switch inside a running process; the native library has its own:
``_native.reload_options()``).  Every switch selects a TESTED alternative of a default code path; nothing here is needed
for normal operation.

    MACJD_UPDATE_STREAMS      2 | 1      captured update on two streams (scan / prefetch beside the chain) or one
    MACJD_UPDATES_PER_GRAPH   K          updates captured per replayed graph when the caller does not say (bench / main: 20)
    MACJD_PIPELINED_GROUP     1 | 0      inside a group: next update's draw / gather / scan beside the current tail, target
                                         branch beside the head, loss gradient formed in the mixer's backward launch
    MACJD_SHARED_BODY         1 | 0      frozen agent body evaluated once for eval + target controller
    MACJD_LEARNER_STATIC_OBS  1 | 0      static observations: agent body once per sequence instead of per step
    MACJD_ACTOR_IN_SCAN       1 | 0      actor rows in the scan launch's prologue / as their own launch
    MACJD_DEVICE_SAMPLER      1 | 0      next batch drawn by the update's last launch / host draw + pinned upload
    MACJD_LN_IN_SQNORM        1 | 0      LayerNorm-parameter gradients inside the optimiser's first launch
    MACJD_WGRAD_OUTER         1 | 0      Q-head ReLU-backward operand formed inside the weight-gradient launch
    MACJD_QHEAD_TAKEN         1 | 0      taken-action Q-head as one launch / input rows + GEMM + row-dot
    MACJD_PAIRED_HEADS        1 | 0      pipelined update: both Q-head launches as one grid and both mixers as one grid on
                                         the chain's stream / target branch on the side stream beside the eval head
    MACJD_GRAPHED_ALLREDUCE   0 | 1      with ranks: RCCL all-reduce captured inside the update graph
"""
from __future__ import annotations

import os

_DEFAULTS = {
    "UPDATE_STREAMS": "2", "UPDATES_PER_GRAPH": "1", "PIPELINED_GROUP": "1", "SHARED_BODY": "1",
    "LEARNER_STATIC_OBS": "1", "ACTOR_IN_SCAN": "1", "DEVICE_SAMPLER": "1", "LN_IN_SQNORM": "1", "WGRAD_OUTER": "1",
    "QHEAD_TAKEN": "1", "PAIRED_HEADS": "1", "GRAPHED_ALLREDUCE": "0",
}
_values = None


def reload() -> None:
    global _values
    _values = {k: os.environ.get("MACJD_" + k, d) for k, d in _DEFAULTS.items()}


def get(name: str) -> str:
    if _values is None:
        reload()
    return _values[name]


def on(name: str) -> bool:
    """The switch is not "0"."""
    return get(name) != "0"
