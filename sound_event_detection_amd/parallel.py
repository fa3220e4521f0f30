"""Clip sharding across the GPUs of one node (one process per GPU,
``torch.distributed`` with the ``nccl`` backend = RCCL over xGMI).

The reference has no parallelism of any kind (single device, training.py:21);
its clips are independent of each other (feature_extraction.py:35-68,
training.py:345-348), so the path shards by clip with replicated weights and
has exactly two optional exchange steps:

* ``all_reduce_mel_stats`` — sums the float64 per-mel-bin statistics
  (2 * n_mels + 1 numbers) behind the optional dataset-level normalisation;
* ``all_gather_probs``     — gathers frame probabilities (10 KB per clip) when
  the caller wants every prediction on every rank, e.g. to decode recordings
  whose clips straddle a rank boundary.

Both take whatever tensors the active backend supports, so the same code runs
on gloo/CPU in the unit tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank ``rank``: blocks of ceil(n/world), so
    the concatenation order of a recording's clips (training.py:350) survives."""
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def shard_recordings(clips_per_recording, rank: int, world: int):
    """Assigns whole recordings to ranks, balancing clip counts greedily in
    order; returns (first_recording, last_recording_exclusive, first_clip,
    last_clip_exclusive).  No recording is split, so every rank can decode its
    own recordings without an exchange."""
    total = sum(clips_per_recording)
    bounds, acc, r = [0], 0, 1
    for i, c in enumerate(clips_per_recording):
        acc += c
        while r < world and acc >= total * r / world:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world + 1:
        bounds.append(len(clips_per_recording))
    bounds[-1] = len(clips_per_recording)
    a, b = bounds[rank], bounds[rank + 1]
    return a, b, sum(clips_per_recording[:a]), sum(clips_per_recording[:b])


def all_gather_probs(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local: this rank's [n_local, T', n_cls] block under ``shard_range``.
    Returns the full [n_total, T', n_cls] tensor on every rank (one
    all-gather of equal-sized, zero-padded blocks)."""
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_total]


def all_reduce_mel_stats(s: torch.Tensor, s2: torch.Tensor, count: float, group=None):
    """(sum, sum of squares, count) per mel bin, float64 -> (mean, std) of the
    whole dataset on every rank: one all-reduce of 2 * n_mels + 1 doubles."""
    buf = torch.cat([s.double(), s2.double(), torch.tensor([count], dtype=torch.float64, device=s.device)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    m = s.numel()
    n = buf[-1]
    mean = buf[:m] / n
    var = (buf[m:2 * m] / n - mean * mean).clamp_min(0.0)
    return mean, var.sqrt()
