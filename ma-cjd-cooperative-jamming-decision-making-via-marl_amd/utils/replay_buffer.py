"""Device-resident episode replay buffer (reference utils/replay_buffer.py:12-256).

Same schema, padding rules and API as the reference's NumPy ring (``store_episode``, ``sample``,
``current_size``, ``current_index``, ``len()``, ``.buffers``), but the arrays are torch tensors that
live in HBM next to the learner (3j/4r, H=64: ~176 KB per episode; 288 GB holds > 10^6 episodes), so
neither storing a rollout nor sampling a batch crosses PCIe:

  state f32 [N,T+1,S]   obs f32 [N,T+1,J,S]   actions_discrete i32 [N,T,J,1]
  actions_continuous f32 [N,T,J,1]   avail_actions i64 [N,T+1,J,A]   reward f32 [N,T,1]
  terminated bool [N,T,1]   filled bool [N,T,1]   hidden_state f32 [N,T+1,J,H]     (replay_buffer.py:49-68)

Host-side bookkeeping only: the ring cursor and each stored episode's length (so ``sample`` knows
``max_seq_len`` without reading ``filled`` back from the device).  Sampling draws its indices with
``np.random.choice(current_size, n, replace=False)`` exactly like the reference
(replay_buffer.py:178), which keeps the global NumPy stream in step with it.
"""
from __future__ import annotations

import threading
from typing import Dict, Optional

import numpy as np
import torch

_T_PLUS_1 = ("state", "obs", "avail_actions", "hidden_state")
_T_ONLY = ("actions_discrete", "actions_continuous", "reward", "terminated", "filled")


def _as_tensor(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(x), device=device).to(dtype)


class EpisodeReplayBuffer:
    def __init__(self, args, device=None):
        self.args = args
        self.buffer_size = args.buffer_size
        self.episode_limit = args.episode_limit
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.state_shape = int(np.prod(args.state_shape)) if isinstance(args.state_shape, tuple) else args.state_shape
        self.obs_shape = int(np.prod(args.obs_shape)) if isinstance(args.obs_shape, tuple) else args.obs_shape
        if device is None:
            want = getattr(args, "device", "cpu")
            use_cuda = getattr(args, "use_cuda", str(want).startswith("cuda")) and torch.cuda.is_available()
            device = torch.device(want if use_cuda else "cpu")
        self.device = torch.device(device)
        N, T, J = self.buffer_size, self.episode_limit, self.n_agents
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        self.buffers: Dict[str, torch.Tensor] = {
            "state": z((N, T + 1, self.state_shape), torch.float32),
            "obs": z((N, T + 1, J, self.obs_shape), torch.float32),
            "actions_discrete": z((N, T, J, 1), torch.int32),
            "actions_continuous": z((N, T, J, 1), torch.float32),
            "avail_actions": z((N, T + 1, J, self.n_actions), torch.int64),
            "reward": z((N, T, 1), torch.float32),
            "terminated": z((N, T, 1), torch.bool),
            "filled": z((N, T, 1), torch.bool),
            "hidden_state": z((N, T + 1, J, args.rnn_hidden_dim), torch.float32),
        }
        self.episode_lengths = np.zeros(N, dtype=np.int64)  # host copy of sum(filled) per slot
        # True while every stored episode came from an environment whose observation does not change within an episode
        # (the batched runner says so when it stores; None = nothing stored yet).  The learner may then evaluate the
        # observation-only parts of the agent once per sampled episode instead of once per step.
        self.obs_static: Optional[bool] = None
        # which static (state / obs / avail_actions) content each slot already holds: a caller-chosen integer per
        # episode; a batched store skips re-copying those keys into slots that hold the very same content already
        self._static_tag = np.full(N, -1, dtype=np.int64)
        self.current_index = 0
        self.current_size = 0
        self.store_count = 0     # episodes ever stored: lets a consumer notice that the ring's content changed
        self.lock = threading.Lock()
        print(f"Replay Buffer Initialized: Size={self.buffer_size}, Episode Limit={self.episode_limit} ({self.device})")

    # ---- storing ----
    def store_episode(self, episode_batch) -> None:
        """One episode: dict key -> [array of length T+1 / T]  (replay_buffer.py:78-151).  Arrays may
        be NumPy (reference runner) or tensors.  Rows past the episode's end are padded exactly as the
        reference pads them: zeros, ``terminated=True``, ``filled=False``."""
        batch_size = len(episode_batch["state"])
        if batch_size != 1:
            print("Warning: EpisodeReplayBuffer expects batch_size=1 from runner")
        with self.lock:
            idx = int(self._get_storage_idx(inc=batch_size)[0])
            self.obs_static = False          # nothing is known about a single stored episode
            self._static_tag[idx] = -1
            ep = {k: v[0] for k, v in episode_batch.items()}
            L = int(ep["reward"].shape[0])
            b = self.buffers
            for k in _T_PLUS_1:
                if k == "hidden_state" and k not in ep:
                    print("Warning: 'hidden_state' key not found in episode_batch data during buffer storage.")
                    b[k][idx, :L + 1] = 0
                else:
                    b[k][idx, :L + 1] = _as_tensor(ep[k], b[k].dtype, self.device)
                b[k][idx, L + 1:] = 0
            for k in ("actions_discrete", "actions_continuous", "reward", "terminated"):
                b[k][idx, :L] = _as_tensor(ep[k], b[k].dtype, self.device)
            b["filled"][idx] = False
            b["filled"][idx, :L] = True
            b["actions_discrete"][idx, L:] = 0
            b["actions_continuous"][idx, L:] = 0.0
            b["reward"][idx, L:] = 0
            b["terminated"][idx, L:] = True  # padded steps count as terminated (replay_buffer.py:149)
            self.episode_lengths[idx] = L

    def store_episodes_batched(self, stage: Dict[str, torch.Tensor], n_episodes: int, length: Optional[int] = None,
                               obs_static: bool = False, static_tags: Optional[np.ndarray] = None) -> None:
        """E full-length episodes at once from the batched runner's TIME-MAJOR staging tensors
        (``stage[key]`` is [T(+1), E, ...] on this device): one transposing copy per key into a
        contiguous slot range (split in two when the ring wraps).  ``length`` < episode_limit pads like
        ``store_episode``.  ``obs_static``: these episodes' observations are constant in time (see ``self.obs_static``).
        ``static_tags`` (int64 [E], >= 0): identity of each episode's state / obs / avail_actions content — slots that
        already hold the same content (the same env's static rows, stored by an earlier rollout) are not rewritten,
        which saves the larger half of the store's traffic (obs + state + mask: 393 of 720 MB at 3j/4r, E = 4096)."""
        T = self.episode_limit
        L = T if length is None else int(length)
        with self.lock:
            idx = self._get_storage_idx(inc=n_episodes)
            self.obs_static = bool(obs_static) if self.obs_static is None else (self.obs_static and bool(obs_static))
            runs = []  # contiguous (slot_lo, slot_hi, src_lo) runs
            start = 0
            for i in range(1, len(idx) + 1):
                if i == len(idx) or idx[i] != idx[i - 1] + 1:
                    runs.append((int(idx[start]), int(idx[i - 1]) + 1, start))
                    start = i
            b = self.buffers
            for lo, hi, s in runs:
                n = hi - lo
                tags = None if (static_tags is None or L < T) else np.asarray(static_tags[s:s + n], dtype=np.int64)
                same_static = tags is not None and bool((tags >= 0).all()) and bool((self._static_tag[lo:hi] == tags).all())
                for k in _T_PLUS_1:
                    if same_static and k != "hidden_state":
                        continue        # these slots already hold exactly this content
                    b[k][lo:hi, :L + 1].copy_(stage[k][:L + 1, s:s + n].transpose(0, 1))
                    if L < T:
                        b[k][lo:hi, L + 1:] = 0
                self._static_tag[lo:hi] = tags if tags is not None else -1
                for k in ("actions_discrete", "actions_continuous", "reward", "terminated"):
                    b[k][lo:hi, :L].copy_(stage[k][:L, s:s + n].transpose(0, 1))
                b["filled"][lo:hi, :L] = True
                if L < T:
                    b["filled"][lo:hi, L:] = False
                    b["actions_discrete"][lo:hi, L:] = 0
                    b["actions_continuous"][lo:hi, L:] = 0.0
                    b["reward"][lo:hi, L:] = 0
                    b["terminated"][lo:hi, L:] = True
                self.episode_lengths[lo:hi] = L

    # ---- sampling ----
    def sample(self, batch_size: int, indices: Optional[np.ndarray] = None):
        """Uniform sample of whole episodes without replacement, truncated to the longest one
        (replay_buffer.py:153-214).  Returns device tensors + ``'max_seq_len'``."""
        if self.current_size < batch_size:
            print(f"Warning: Sampling {batch_size} but buffer only contains {self.current_size} episodes. "
                  f"Sampling {self.current_size}.")
            actual = self.current_size
        else:
            actual = batch_size
        if actual <= 0:
            print("Error: Cannot sample 0 or negative episodes.")
            return None
        if indices is None:
            indices = np.random.choice(self.current_size, actual, replace=False)
        indices = np.asarray(indices, dtype=np.int64)
        max_seq_len = int(self.episode_lengths[indices].max()) if len(indices) else 0
        idx_t = torch.from_numpy(indices).to(self.device, non_blocking=True)
        out = {}
        for k, data in self.buffers.items():
            sel = data.index_select(0, idx_t)
            out[k] = sel[:, :max_seq_len + 1] if k in _T_PLUS_1 else sel[:, :max_seq_len]
        out["max_seq_len"] = max_seq_len
        return out

    def _get_storage_idx(self, inc=None) -> np.ndarray:
        """Ring cursor (replay_buffer.py:216-251)."""
        inc = inc or 1
        if self.current_index + inc <= self.buffer_size:
            idx = np.arange(self.current_index, self.current_index + inc)
            self.current_index += inc
        elif inc <= self.buffer_size:
            overflow = inc - (self.buffer_size - self.current_index)
            idx = np.concatenate((np.arange(self.current_index, self.buffer_size), np.arange(0, overflow)))
            self.current_index = overflow
        else:
            raise ValueError("Attempting to store more episodes than the buffer capacity in a single call.")
        self.current_size = min(self.current_size + inc, self.buffer_size)
        self.store_count += inc
        return idx

    def __len__(self):
        with self.lock:
            return self.current_size
