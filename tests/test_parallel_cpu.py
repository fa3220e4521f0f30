"""Sharding helpers and the two exchange steps on a 2-rank gloo group (CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sound_event_detection_amd import parallel


def test_shard_range_covers_everything_in_order():
    for n in (0, 1, 7, 8, 9, 4096, 4097):
        for world in (1, 2, 3, 8):
            parts = [parallel.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b
            assert max(b - a for a, b in parts) <= -(-n // world) if n else True


def test_shard_recordings_keeps_recordings_whole():
    clips = [3, 1, 1, 24, 2, 2, 2, 9, 1, 5]
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            a, b, ca, cb = parallel.shard_recordings(clips, r, world)
            assert ca == sum(clips[:a]) and cb == sum(clips[:b])
            seen += list(range(a, b))
        assert seen == list(range(len(clips)))


def _worker(rank, world, port, n_total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        full = torch.rand(n_total, 25, 10, generator=g)
        lo, hi = parallel.shard_range(n_total, rank, world)
        got = parallel.all_gather_probs(full[lo:hi].clone(), n_total)
        assert torch.equal(got, full)
        mel = torch.rand(n_total, 16, 50, generator=g, dtype=torch.float64)
        loc = mel[lo:hi]
        mean, std = parallel.all_reduce_mel_stats(loc.sum((0, 2)), (loc * loc).sum((0, 2)), float(loc.shape[0] * 50))
        assert torch.allclose(mean, mel.mean((0, 2)), rtol=1e-12)
        assert torch.allclose(std, mel.std((0, 2), unbiased=False), rtol=1e-9)
        # sharded statistics = the single-rank statistics of the same per-shard sums, bit for bit in float64
        parts = [mel[slice(*parallel.shard_range(n_total, r, world))] for r in range(world)]
        s_all = sum(p.sum((0, 2)) for p in parts)
        s2_all = sum((p * p).sum((0, 2)) for p in parts)
        mean1, std1 = parallel.mel_stats_to_mean_std(s_all, s2_all, torch.tensor(float(n_total * 50), dtype=torch.float64))
        assert mean.dtype == torch.float64 and torch.equal(mean, mean1) and torch.equal(std, std1)
        # a bin that never varies is not divided by zero
        z = torch.zeros(4, dtype=torch.float64)
        _, std0 = parallel.all_reduce_mel_stats(z, z, 10.0)
        assert torch.all(std0 == parallel.MEL_STD_FLOOR)
    finally:
        dist.destroy_process_group()


def test_exchange_steps_on_two_gloo_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 7), nprocs=2, join=True)


def _fake_predict(model, files, dataset_location, dataset, clip_length, *args, audio_duration=None, **kw):
    # two events per file, named after it: enough to check order and completeness
    return [(f, 0.0, 1.0, "a") for f in files] + [(f, 2.0, 3.0, "b") for f in files]


def _sharded_worker(rank, world, port, files, durations):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = parallel.predict_files_sharded(None, files, "/data/", "desed_2022", 10000, 2048, 128, 320, 640, {}, 40,
                                             audio_duration=lambda p: durations[p.rsplit("/", 1)[1]],
                                             predict_fn=_fake_predict)
        counts = [int(np.ceil(durations[f] / 10.0)) for f in files]
        want = []
        for r in range(world):
            a, b, _, _ = parallel.shard_recordings(counts, r, world)
            want += _fake_predict(None, files[a:b], None, None, None)
        assert got == want
        assert sorted({e[0] for e in got}) == sorted(files)
    finally:
        dist.destroy_process_group()


def test_file_level_prediction_sharded_over_two_gloo_ranks():
    files = [f"f{i}.wav" for i in range(7)]
    durations = {f: 10.0 * (1 + (3 * i) % 5) - 0.5 for i, f in enumerate(files)}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, files, durations), nprocs=2, join=True)
