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

``predict_files_sharded`` is the file-level prediction loop (training.py:333-365)
over all ranks: whole recordings are dealt to ranks by their clip counts, every
rank runs ``predict.predict_files`` on its share, and the event lists are
gathered in file order (a few hundred bytes per event, once per call).
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


MEL_STD_FLOOR = 1e-12      # a bin that never varies (all-zero audio) is divided by this, not by zero


def mel_stats_to_mean_std(s: torch.Tensor, s2: torch.Tensor, count):
    """float64 (sum, sum of squares) per mel bin and the number of values per bin -> (mean, std), float64,
    population variance (the mean of squares minus the squared mean, clamped at zero)."""
    # a tensor divisor: dividing a device tensor by a Python number multiplies by its reciprocal instead, which
    # differs from the all-reduced path's tensor / tensor division in the last bit
    count = torch.as_tensor(count, dtype=torch.float64, device=s.device)
    mean = s / count
    var = (s2 / count - mean * mean).clamp_min(0.0)
    return mean, var.sqrt().clamp_min(MEL_STD_FLOOR)


def all_reduce_mel_stats(s: torch.Tensor, s2: torch.Tensor, count: float, group=None):
    """(sum, sum of squares, count) per mel bin, float64 -> (mean, std) of the
    whole dataset on every rank: one all-reduce of 2 * n_mels + 1 doubles."""
    buf = torch.cat([s.double(), s2.double(), torch.tensor([count], dtype=torch.float64, device=s.device)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    m = s.numel()
    return mel_stats_to_mean_std(buf[:m], buf[m:2 * m], buf[-1])


def predict_files_sharded(model, test_files, dataset_location, dataset, clip_length, *args, group=None,
                          audio_duration=None, predict_fn=None, **kwargs):
    """``predict.predict_files`` with the files split over the ranks of ``group``.

    Same positional arguments as ``predict_files`` (after ``clip_length``: n_fft, n_mels, hop_length, win_length,
    id2cls, block_length, ...).  Recordings stay whole (a recording's clips are decoded as one frame sequence,
    training.py:350), ranks get contiguous runs of files balanced by clip count, and every rank returns the
    complete event list in the order of ``test_files`` -- what the single-process call returns.
    ``return_probs`` is not supported here (frame predictions stay on the rank that made them).
    ``normalise=True`` without ``mean`` / ``std`` normalises with the statistics of ALL files: each rank sums its
    share and ``all_reduce_mel_stats`` adds the shares (every rank must make the call, files or not)."""
    import numpy as np

    from . import feature_extraction as fe
    if kwargs.get("return_probs"):
        raise ValueError("predict_files_sharded returns events only")
    if kwargs.get("normalise") and kwargs.get("mean") is None:
        # statistics of the whole file list, not of this rank's share: every rank adds its sums to the others'
        kwargs.setdefault("stats_group", True if group is None else group)
    if predict_fn is None:
        from .predict import predict_files as predict_fn
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    get_dur = audio_duration or fe._wav_duration
    clip_s = 0.001 * clip_length
    counts = [int(np.ceil(get_dur(dataset_location + dataset + '/' + f) / clip_s)) for f in test_files]
    a, b, _, _ = shard_recordings(counts, rank, world)
    mine = predict_fn(model, list(test_files[a:b]), dataset_location, dataset, clip_length, *args,
                      audio_duration=audio_duration, **kwargs)
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return [e for part in parts for e in part]
