#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: clips start in pinned host memory, probabilities and events end there.
Chunks of clips are copied on a second stream while the previous chunk computes.

    python tools/bench_pcie.py --clips 4096 --chunk 4096 [--int16]

With --chunk equal to --clips a whole batch is copied while the previous one computes (the GRU kernels want the
full 4096 clips to fill the chip).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--int16", action="store_true", help="clips cross PCIe as 16-bit PCM (sedmi_melspec_i16)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from sound_event_detection_amd.pipeline import SedPipeline
    model, _ = bench.make_crnn(dev)
    pipe = SedPipeline(model, sr=bench.SR, n_fft=bench.N_FFT, n_mels=128, hop_length=bench.HOP, win_length=bench.WIN)
    host = bench.make_clips(args.clips, 0, dev).cpu()
    dtype = torch.float32
    if args.int16:
        host = (host / host.abs().max() * 0.5 * 32767).round().to(torch.int16)
        dtype = torch.int16
    host = host.pin_memory()
    n_chunks = args.clips // args.chunk
    bufs = [torch.empty((args.chunk, bench.N_SAMPLES), dtype=dtype, device=dev) for _ in range(2)]
    probs_host = torch.empty((args.clips, 250, bench.N_CLS), dtype=torch.float32).pin_memory()
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]

    def issue_copy(i):
        b, c = i & 1, i % n_chunks
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[b])                           # buffer b free again
            bufs[b].copy_(host[c * args.chunk:(c + 1) * args.chunk], non_blocking=True)
            ready[b].record(copy_stream)

    def run(n_steps):
        # chunk i + 1 is on its way across PCIe before chunk i's kernels are enqueued; buffers alternate
        total = n_steps * n_chunks
        issue_copy(0)
        for i in range(total):
            b, c = i & 1, i % n_chunks
            if i + 1 < total:
                issue_copy(i + 1)
            torch.cuda.current_stream(dev).wait_event(ready[b])
            out = pipe.run(bufs[b])
            probs_host[c * args.chunk:(c + 1) * args.chunk].copy_(out["probs"], non_blocking=True)
            done[b].record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)

    for b in range(2):
        done[b].record(torch.cuda.current_stream(dev))
    run(1)
    t0 = time.perf_counter()
    run(args.steps)
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps(dict(clips=args.clips, chunk=args.chunk, ms_per_step=1e3 * dt, clips_per_s=args.clips / dt,
                          h2d_gbps=args.clips * bench.N_SAMPLES * host.element_size() / dt / 1e9, input=str(dtype))))


if __name__ == "__main__":
    main()
