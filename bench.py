#!/usr/bin/env python3
"""Throughput benchmark of the SED inference hot path on MI355X.

A step = one pass of the whole path (mel frontend -> baseline CRNN -> threshold
/ event decode) over one batch of synthetic 10 s, 16 kHz clips that is already
resident in HBM.  `value` = clips / second, whole job (all ranks).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Clips are independent, so every rank runs the same per-GPU batch (weak
scaling) and the timed step has no collective: each rank decodes the events of
its own clips, as the reference decodes file by file.  --gather-probs adds an
all-gather of the frame probabilities over RCCL to the step, for a caller that
wants every prediction on every rank.

Rank 0 prints ONE JSON line carrying, besides the contract fields,
  roofline      the dominant kernel's achieved algorithmic rate vs its MI355X
                peak, timed with HIP events on the launch stream inside this
                run (all kernels are listed under "kernels");
  cpu_baseline  the CPU oracle (same arithmetic as the reference's CPU path)
                timed on this host on a bounded sample of the same clips.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, N_FFT, N_MELS, HOP, WIN, N_SAMPLES, N_CLS = 16000, 2048, 128, 320, 640, 160000, 10
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense f32-input MFMA (= f32 vector) peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense f16/bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0            # MI355X HBM3E peak


def make_model(device):
    from sound_event_detection_amd import synth
    from sound_event_detection_amd.nnet import CRNN
    model = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=N_CLS, attention=True, n_RNN_cell=128,
                 activation='glu', rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
                 nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6,
                 dropout_recurrent=0)
    sd = synth.synth_crnn_state_dict(0, nclass=N_CLS)
    model.load_state_dict(sd)
    model.eval()
    return model.to(device), sd


def make_clips(n_clips, rank, device):
    """n_clips distinct clips in HBM: 64 generator clips (noise + tone bursts,
    sound_event_detection_amd.synth) tiled and each perturbed with its own
    seeded noise, so no two clips are equal and the CPU sample can read back
    exactly the bytes the GPU processed."""
    from sound_event_detection_amd import synth
    base_n = min(64, n_clips)
    base = torch.from_numpy(synth.synth_clips(rank * 64, base_n)).to(device)
    reps = (n_clips + base_n - 1) // base_n
    y = base.repeat(reps, 1)[:n_clips].contiguous()
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    chunk = 256
    for i in range(0, n_clips, chunk):
        y[i:i + chunk] += 0.02 * torch.randn(y[i:i + chunk].shape, device=device, generator=g)
    return y


def cpu_baseline(y_sample: np.ndarray, sd, batch=32):
    """The oracle's mel -> CRNN -> decode on the host, batched as evaluate()
    batches (evaluation.py:12-24), all torch threads."""
    from oracle import crnn as ocrnn, decode as odec, mel as omel
    id2cls = {i: f"cls{i}" for i in range(N_CLS)}
    n = len(y_sample)
    t0 = time.perf_counter()
    feats = np.stack([omel.clip_feature(c, SR, N_FFT, N_MELS, HOP, WIN) for c in y_sample])
    t1 = time.perf_counter()
    probs = []
    for i in range(0, n, batch):
        probs.append(ocrnn.crnn_forward(sd, torch.from_numpy(feats[i:i + batch])))
    probs = torch.cat(probs).numpy()
    t2 = time.perf_counter()
    n_ev = 0
    for i in range(n):
        n_ev += len(odec.get_prediction_from_raw_output(probs[i], id2cls, 10.0, 40, f"clip{i}", 0.5, False))
    t3 = time.perf_counter()
    return dict(value=n / (t3 - t0), unit="clips/s", cores=int(torch.get_num_threads()), kind="port",
                sample=f"{n} of the benchmark's clips, batch {batch}; mel {t1 - t0:.2f}s (numpy, 1 thread) + "
                       f"CRNN {t2 - t1:.2f}s (torch CPU, {torch.get_num_threads()} threads) + decode {t3 - t2:.2f}s; "
                       f"host has {os.cpu_count()} logical CPUs"), probs, n_ev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=4096, help="clips per GPU per step")
    ap.add_argument("--cpu-clips", type=int, default=192, help="clips of the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=["split_f16", "f32"], default=None,
                    help="arithmetic of the conv / GRU-input contractions (default: the library default, split_f16)")
    ap.add_argument("--gather-probs", action="store_true",
                    help="end every step with an all-gather of the frame probabilities (N > 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from sound_event_detection_amd import _lib
    from sound_event_detection_amd.pipeline import SedPipeline

    model, sd = make_model(device)
    if args.precision:
        model.conv_precision = args.precision
    pipe = SedPipeline(model, sr=SR, n_fft=N_FFT, n_mels=N_MELS, hop_length=HOP, win_length=WIN)
    y = make_clips(args.clips, rank, device)
    gather = world > 1 and args.gather_probs
    gathered = None

    def step():
        nonlocal gathered
        b = pipe.run(y)
        if gather:
            if gathered is None:
                gathered = torch.empty((world,) + tuple(b["probs"].shape), dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(gathered, b["probs"])
        return b

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernels = _lib.profile_collect()
    _lib.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    n_events = int(b["n_events"].item())

    if rank == 0:
        total_clips = args.clips * world * args.steps
        for k in kernels:
            k["avg_ms"] = k["ms"] / max(k["launches"], 1)
            k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            k["gbps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        kernels.sort(key=lambda k: -k["ms"])
        dom = kernels[0]
        # f32 MFMA contractions (conv blocks, GRU) are priced against the matrix peak, byte movers against HBM
        mfma_bound = dom["flops"] / max(dom["bytes"], 1.0) > PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)
        if mfma_bound:
            # "*_split" kernels run on the f16 matrix cores and spend three MFMAs per float32-grade
            # product (hi*hi + hi*lo + lo*hi): `achieved` stays the ALGORITHMIC rate, so its ceiling
            # against the f16 peak is 1/3; `mfma_issue_frac` is the share of the f16 peak the
            # instruction stream actually sustains.
            split = "split" in dom["name"]
            peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
            roof = dict(kernel=dom["name"], bound="mfma", achieved=dom["tflops"], peak=peak,
                        unit="TFLOP/s", frac=dom["tflops"] / peak, traffic=None,
                        avg_launch_ms=dom["avg_ms"], launches=dom["launches"],
                        flop_per_launch=dom["flops"] / dom["launches"],
                        mfma_per_product=3 if split else 1,
                        mfma_issue_frac=(3 if split else 1) * dom["tflops"] / peak,
                        # tools/mfma_peak on this pool: a register-resident v_mfma_f32_32x32x16_f16 loop holds
                        # 2.48 PFLOP/s on constant operands but 1.63-1.72 on random ones (the chip lowers its
                        # clock under load), which caps split-f16 work at ~0.55-0.57 PFLOP/s algorithmic
                        peak_sustained_random_operands=1700.0 if split else None)
        else:
            roof = dict(kernel=dom["name"], bound="hbm", achieved=dom["gbps"], peak=PEAK_HBM_GBPS, unit="GB/s",
                        frac=dom["gbps"] / PEAK_HBM_GBPS, traffic=None, avg_launch_ms=dom["avg_ms"],
                        launches=dom["launches"], bytes_per_launch=dom["bytes"] / dom["launches"])
        # HBM bytes per launch of that kernel from the PMC passes kept under profiles/ (rocprofv3
        # cannot run inside this process); null if that kernel has not been profiled
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                rec = json.load(f).get(dom["name"])
            if rec:
                roof["traffic"] = rec["hbm_bytes_per_launch"]          # bytes per launch, PMC-measured
                roof["traffic_detail"] = {k: rec[k] for k in ("fetch_size_kib_raw", "write_size_kib", "note")}
                roof["algorithmic_bytes_per_launch"] = dom["bytes"] / dom["launches"]
        except (OSError, ValueError):
            pass
        cpu = None
        if args.cpu_clips > 0:
            ns = min(args.cpu_clips, args.clips)
            cpu, cpu_probs, _ = cpu_baseline(y[:ns].cpu().numpy(), sd)
            cpu["max_abs_prob_diff_vs_gpu"] = float(np.abs(b["probs"][:ns].cpu().numpy() - cpu_probs).max())
        out = {
            "metric": "clips/sec (10 s clips, mel+CRNN+decode)",
            "value": total_clips / elapsed,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (conv / GRU-input contractions as split-f16 x3 on the f16 matrix cores, f32 accumulate)"
                     if model.conv_precision == "split_f16" else "f32",
            "data": "synthetic",
            "config": {"workload": f"mel frontend + DCASE-baseline CRNN + threshold/run-length decode, "
                                   f"{args.clips} synthetic 10 s 16 kHz clips per GPU resident in HBM "
                                   f"(n_fft 2048, win 640, hop 320, 128 mels, 10 classes)",
                       "clips_per_gpu": args.clips, "parallelism": f"clip-sharded x{world}",
                       "all_gather_probs": bool(gather), "events_last_step": n_events},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": [dict(name=k["name"], launches=k["launches"], ms_total=round(k["ms"], 3),
                             tflops=round(k["tflops"], 2), gbps=round(k["gbps"], 1)) for k in kernels],
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
