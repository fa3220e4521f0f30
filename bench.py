#!/usr/bin/env python3
"""Throughput benchmark of the SED inference hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic input that is
already resident in HBM.  `value` = clips / second, whole job (all ranks).

    python bench.py                                   # 1 GPU, headline workload (--config crnn)
    python bench.py --gpus 4                          # starts 4 fresh rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

--config selects the workload (default: the one the headline metric is quoted on):
  crnn              mel frontend + DCASE-baseline CRNN + decode, 4096 10 s 16 kHz clips per GPU
  frontend          mel frontend only, 4096 clips per GPU
  rcnn              ShakeRCNN + decode on 128 clips of 1000 x 768 synthetic features per GPU (the shape of the
                    AST embeddings the reference trained it on; feature_extraction.py:23,57)
  transformer       ShakeTransformer + decode, same input
  basic             mel (64 bins) + BasicRCNN + decode from audio, 1024 clips per GPU
  rcnn-long         ShakeRCNN on 4 one-minute recordings per GPU (6000 frames -> 1500 GRU steps each): the latency
                    case, a handful of long recordings
  rcnn-long-64      the same with 64 one-minute recordings per GPU: the throughput case (a recurrence step costs
                    little more for 64 recordings than for 4)
  transformer-long  ShakeTransformer on 2 four-minute recordings per GPU (24000 frames -> 6000 tokens each)

Clips are independent, so every rank runs the same per-GPU batch (weak scaling) and
decodes the events of its own clips.  With more than one rank every step ends with the
all-gather of the float32 frame predictions over RCCL (10 KB per 10 s clip), issued
asynchronously so that it runs beside the next step's kernels; --no-gather leaves it out.

Rank 0 prints ONE JSON line carrying, besides the contract fields,
  roofline      the dominant kernel's achieved algorithmic rate vs its MI355X peak, timed with HIP events
                on the launch stream inside this run (all kernels are listed under "kernels");
  cpu_baseline  the CPU oracle (same arithmetic and primitives as the reference's CPU path) timed on this
                host on a bounded sample of the same input, in the ways the reference runs it (N = 1 only; with
                more ranks pass --cpu-clips to ask for it).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, N_FFT, HOP, WIN, N_SAMPLES, N_CLS = 16000, 2048, 320, 640, 160000, 10
PEAK_F32_VALU_TFLOPS = 157.3      # MI355X dense f32 vector (= f32-input MFMA) peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense f16/bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0            # MI355X HBM3E peak

# SEDMI_BENCH_REHEARSAL=cpu: the rank bookkeeping of a multi-process run (rendezvous, device check, both collectives,
# per-rank times, the JSON line) on gloo / CPU tensors with a stand-in step that runs NO kernel.  It exists so that
# the N > 1 code path can be executed where there is no multi-GPU machine (tests/test_bench_cpu.py); its line is marked
# and carries no performance figure.
REHEARSAL = os.environ.get("SEDMI_BENCH_REHEARSAL") == "cpu"

CONFIGS = {
    # name: (clips per GPU, kind)
    "crnn": 4096, "frontend": 4096, "rcnn": 128, "transformer": 128, "basic": 1024, "rcnn-long": 4,
    "rcnn-long-64": 64, "transformer-long": 2,
}


# ------------------------------------------------------------------------------------------------ inputs / models
def make_crnn(device):
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


def make_other(kind, device):
    from sound_event_detection_amd import models, synth
    if kind == "basic":
        sd = synth.synth_shake_state_dict(synth.basic_rcnn_state_dict_spec(N_CLS, 64), 0)
        m = models.BasicRCNN(N_CLS, 64)
    elif kind == "rcnn":
        sd = synth.synth_shake_state_dict(synth.shake_rcnn_state_dict_spec(N_CLS), 0)
        m = models.ShakeRCNN(N_CLS, 0)
    else:
        sd = synth.synth_shake_state_dict(synth.shake_transformer_state_dict_spec(N_CLS), 0)
        m = models.ShakeTransformer(N_CLS, 0.0)
    m.load_state_dict(sd)
    m.eval()
    return m.to(device), sd


def make_clips(n_clips, rank, device, amplitude=1.0):
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
    if amplitude != 1.0:
        y *= amplitude
    return y


def make_feats(n_clips, frames, rank, device):
    """[n_clips, frames, 768] synthetic features: 4 generator clips tiled, each with its own seeded noise."""
    from sound_event_detection_amd import synth
    base_n = min(4, n_clips)
    base = torch.from_numpy(synth.synth_shake_feats(3 + rank, base_n, frames)).to(device)
    x = base.repeat((n_clips + base_n - 1) // base_n, 1, 1)[:n_clips].contiguous()
    g = torch.Generator(device=device).manual_seed(2000 + rank)
    for i in range(0, n_clips, 16):
        x[i:i + 16] += 0.05 * torch.randn(x[i:i + 16].shape, device=device, generator=g)
    return x


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_crnn(y_sample: np.ndarray, sd, with_model=True, norm_stats=None):
    """The oracle's mel -> CRNN -> decode on the host, three ways: batched as evaluate() batches
    (evaluation.py:12-24) on all threads, and clip by clip as the prediction loop runs it
    (training.py:345-348: batch 1) on all threads and on one thread.  Returns the dict, the
    batched probabilities and the CPU event list."""
    from oracle import crnn as ocrnn, decode as odec, mel as omel
    id2cls = {i: f"cls{i}" for i in range(N_CLS)}
    n = len(y_sample)
    threads = int(torch.get_num_threads())
    t0 = time.perf_counter()
    if norm_stats is None:
        feats = np.stack([omel.clip_feature(c, SR, N_FFT, 128, HOP, WIN) for c in y_sample])
    else:       # --normalise: log-mel, then (x - mean) / std with the statistics the GPU step used
        mean, std = (np.asarray(v, dtype=np.float32) for v in norm_stats)
        feats = np.stack([omel.normalise(omel.power_to_db(omel.melspectrogram(c, SR, N_FFT, 128, HOP, WIN)), mean, std)
                          for c in y_sample])
        feats = feats.reshape(len(feats), feats.shape[2], feats.shape[1])          # feature_extraction.py:49-50
    t_mel = time.perf_counter() - t0
    if not with_model:
        return dict(value=n / t_mel, unit="clips/s", cores=1, kind="port", cpu=cpu_model_name(),
                    sample=f"{n} of the benchmark's clips; numpy float64 rFFT + float32 mel projection, 1 thread "
                           f"({t_mel:.2f} s); host has {os.cpu_count()} logical CPUs"), feats, None
    t0 = time.perf_counter()
    probs = torch.cat([ocrnn.crnn_forward(sd, torch.from_numpy(feats[i:i + 32])) for i in range(0, n, 32)]).numpy()
    t_b32 = time.perf_counter() - t0
    t0 = time.perf_counter()
    events = []
    for i in range(n):
        events += odec.get_prediction_from_raw_output(probs[i], id2cls, 10.0, 40, f"clip{i}", 0.5, False)
    t_dec = time.perf_counter() - t0
    # as the reference runs it: one clip per forward
    n1 = min(n, 32)
    t0 = time.perf_counter()
    for i in range(n1):
        ocrnn.crnn_forward(sd, torch.from_numpy(feats[i:i + 1]))
    t_b1 = (time.perf_counter() - t0) / n1
    n1t = min(n, 12)
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        for i in range(n1t):
            ocrnn.crnn_forward(sd, torch.from_numpy(feats[i:i + 1]))
        t_b1_1t = (time.perf_counter() - t0) / n1t
    finally:
        torch.set_num_threads(threads)
    per_mel, per_dec = t_mel / n, t_dec / n
    out = dict(
        value=n / (t_mel + t_b32 + t_dec), unit="clips/s", cores=threads, kind="port", cpu=cpu_model_name(),
        logical_cpus=os.cpu_count(),
        sample=f"{n} of the benchmark's clips; mel {t_mel:.2f} s (numpy, 1 thread) + CRNN {t_b32:.2f} s (torch CPU, "
               f"batch 32, {threads} threads) + decode {t_dec:.2f} s",
        batch1_all_threads=dict(value=1.0 / (per_mel + t_b1 + per_dec), unit="clips/s", cores=threads,
                                sample=f"model on {n1} clips one per forward (training.py:345-348)"),
        batch1_one_thread=dict(value=1.0 / (per_mel + t_b1_1t + per_dec), unit="clips/s", cores=1,
                               sample=f"model on {n1t} clips one per forward, torch.set_num_threads(1)"),
        stage_seconds_per_clip=dict(mel=per_mel, crnn_batch32=t_b32 / n, crnn_batch1=t_b1, crnn_batch1_one_thread=t_b1_1t,
                                    decode=per_dec))
    return out, probs, events


def cpu_baseline_other(kind, x_sample, sd, alphas, frames_out):
    """Oracle forward + decode of the other models on a few inputs (torch CPU, all threads)."""
    from oracle import decode as odec, shake as oshake
    id2cls = {i: f"cls{i}" for i in range(N_CLS)}
    fwd = {"rcnn": lambda x: oshake.shake_rcnn_forward(sd, x, alphas),
           "transformer": lambda x: oshake.shake_transformer_forward(sd, x, alphas),
           "basic": lambda x: oshake.basic_rcnn_forward(sd, x)}[kind]
    n = len(x_sample)
    t0 = time.perf_counter()
    logits = fwd(torch.from_numpy(x_sample)).numpy()
    t_model = time.perf_counter() - t0
    t0 = time.perf_counter()
    events = []
    for i in range(n):
        events += odec.get_prediction_from_raw_output(logits[i], id2cls, 1e9, 40, f"clip{i}", 0.5, True)
    t_dec = time.perf_counter() - t0
    return dict(value=n / (t_model + t_dec), unit="clips/s", cores=int(torch.get_num_threads()), kind="port",
                cpu=cpu_model_name(), logical_cpus=os.cpu_count(),
                sample=f"{n} of the benchmark's inputs in one batch: model {t_model:.2f} s (torch CPU) + decode "
                       f"{t_dec:.2f} s"), logits, events


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """--gpus N without a launcher: start N fresh rank processes (this process has not touched the GPU and never
    will) and pass on rank 0's output and the worst exit code."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ------------------------------------------------------------------------------------------------ rank bookkeeping
def device_identity(rank, local_rank):
    """What tells this rank's GPU from every other rank's: host name + PCI address (+ the device UUID)."""
    if REHEARSAL:       # no GPU: the process stands in for its device
        return dict(rank=rank, local_rank=local_rank, host=socket.gethostname(), pci=f"pid:{os.getpid()}", uuid="",
                    name="cpu (rehearsal)", cus=0)
    p = torch.cuda.get_device_properties(local_rank)
    pci = f"{int(getattr(p, 'pci_domain_id', 0)):04x}:{int(getattr(p, 'pci_bus_id', -1)):02x}:{int(getattr(p, 'pci_device_id', -1)):02x}"
    return dict(rank=rank, local_rank=local_rank, host=socket.gethostname(), pci=pci, uuid=str(getattr(p, "uuid", "")),
                name=p.name, cus=int(p.multi_processor_count))


def check_ranks(identities, n_gpus, world_size):
    """Refuses (SystemExit) a job whose ranks do not sit on n_gpus DIFFERENT devices: a line that says n_gpus = 8
    must come from 8 ranks on 8 GPUs.  identities: one device_identity() dict per rank, as all-gathered."""
    if world_size != n_gpus or len(identities) != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but the process group has {world_size} ranks and "
                         f"{len(identities)} reported a device: refusing to print a line")
    if sorted(i["rank"] for i in identities) != list(range(n_gpus)):
        raise SystemExit(f"bench.py: ranks seen {sorted(i['rank'] for i in identities)} are not 0..{n_gpus - 1}")
    seen = {}
    for i in identities:
        key = (i["host"], i["pci"], i["uuid"])
        if key in seen:
            raise SystemExit(f"bench.py: ranks {seen[key]} and {i['rank']} run on the same device {key}: refusing to "
                             f"print a {n_gpus}-GPU line measured on fewer GPUs")
        seen[key] = i["rank"]
    return [dict(rank=i["rank"], host=i["host"], pci=i["pci"], uuid=i["uuid"], name=i["name"], cus=i["cus"])
            for i in sorted(identities, key=lambda i: i["rank"])]


def gather_rank_times(dist, elapsed, steps, world, device):
    """(slowest rank's elapsed seconds, every rank's ms per step): value is computed from the slowest rank."""
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    rank_ms = [1e3 * elapsed / steps]
    if dist is not None:
        every = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, t)
        rank_ms = [1e3 * float(v) / steps for v in every.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), rank_ms


class AsyncGather:
    """Snapshot + asynchronous all-gather of a step's frame predictions: the collective runs on the backend's own
    stream beside the next step's kernels; slot i is reused two steps later, after waiting for the collective that
    read it."""

    def __init__(self, dist, world, device, collectives, backend):
        self.dist, self.world, self.device, self.collectives, self.backend = dist, world, device, collectives, backend
        self.snap, self.gathered, self.works, self.count = [None, None], [None, None], [None, None], 0

    def __call__(self, p):
        i = self.count % 2
        self.count += 1
        if self.snap[i] is None:
            self.snap[i] = torch.empty_like(p)
            # the ranks' blocks concatenated along the first axis: the form every backend takes
            self.gathered[i] = torch.empty((self.world * p.shape[0],) + tuple(p.shape[1:]), dtype=p.dtype, device=self.device)
            if i == 0:
                self.collectives.append(dict(name="all_gather of the float32 frame predictions", backend=self.backend,
                                             bytes_per_rank_per_step=p.numel() * 4, calls_per_step=1, blocking=False))
        if self.works[i] is not None:
            self.works[i].wait()
        self.snap[i].copy_(p)
        self.works[i] = self.dist.all_gather_into_tensor(self.gathered[i], self.snap[i], async_op=True)

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()


def stats_collective(backend):
    return dict(name="all_reduce(sum) of the float64 mel statistics (2 * n_mels + 1 doubles)", backend=backend,
                bytes_per_rank_per_step=8 * (2 * 128 + 1), calls_per_step=1, blocking=True)


def rank_fields(world, dist, ranks_seen, local_rank, rank_ms, collectives):
    """What proves a line: the process group's size, each rank's device (all different, or no line is printed), each
    rank's own time per step (value uses the slowest), the collectives a step ran."""
    return {
        "world_size": world if dist is None else dist.get_world_size(),
        "ranks_seen": ranks_seen if ranks_seen is not None else [check_ranks([device_identity(0, local_rank)], 1, 1)[0]],
        "ms_per_step_ranks": {"min": min(rank_ms), "max": max(rank_ms), "all": [round(v, 3) for v in rank_ms]},
        "collectives": collectives,
    }


def rehearse(args, world, rank, dist, ranks_seen):
    """The multi-rank bookkeeping of main() around a stand-in step (see REHEARSAL): same helpers, no kernel."""
    from sound_event_detection_amd import parallel
    device = torch.device("cpu")
    steps = args.steps if args.steps is not None else 2
    clips = args.clips if args.clips is not None else 8
    collectives = [stats_collective("gloo")] if world > 1 and args.normalise else []
    gather = AsyncGather(dist, world, device, collectives, "gloo") if world > 1 and not args.no_gather else None
    g = torch.Generator().manual_seed(rank)

    def step():
        probs = torch.rand(clips, 250, N_CLS, generator=g)
        if args.normalise and dist is not None:
            mel = torch.rand(clips, 128, 501, generator=g, dtype=torch.float64)
            parallel.all_reduce_mel_stats(mel.sum((0, 2)), (mel * mel).sum((0, 2)), float(clips * 501))
        if gather is not None:
            gather(probs)
        return probs

    def fence():
        if gather is not None:
            gather.wait()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed, rank_ms = gather_rank_times(dist, time.perf_counter() - t0, steps, world, device)
    if gather is not None:      # every rank's block arrived in rank order
        last = gather.gathered[(gather.count - 1) % 2].view(world, clips, 250, N_CLS)
        assert torch.equal(last[rank], gather.snap[(gather.count - 1) % 2])
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL (no kernel ran): rank bookkeeping on gloo / CPU", "value": None, "unit": "clips/s",
                          "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
                          "rehearsal": True, "config": {"name": args.config, "clips_per_gpu": clips},
                          **rank_fields(world, dist, ranks_seen, 0, rank_ms, collectives)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="crnn")
    ap.add_argument("--clips", type=int, default=None, help="clips (recordings) per GPU per step")
    ap.add_argument("--frames", type=int, default=None, help="feature frames per clip (rcnn / transformer configs)")
    ap.add_argument("--cpu-clips", type=int, default=None, help="size of the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=["split_f16", "f32"], default=None,
                    help="CRNN arithmetic (default: the library default, split_f16)")
    ap.add_argument("--no-exact-leg", action="store_true",
                    help="crnn config: skip the extra exact-float32 pass behind the timed region (value_f32_exact)")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: leave the all-gather of the frame predictions out of the step")
    ap.add_argument("--normalise", action="store_true",
                    help="crnn config: log-mel + mean/std normalisation with the statistics of the step's "
                         "clips (the stages the reference does not have; off = the reference's features).  Every "
                         "step then runs mel -> float64 statistics -> [N > 1: all-reduce over RCCL] -> normalise -> "
                         "model -> decode")
    args = ap.parse_args()
    if args.normalise and args.config != "crnn":
        raise SystemExit("bench.py: --normalise applies to the crnn config only")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line whose n_gpus "
                         f"differs from what was asked for")
    if REHEARSAL:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    sync = (lambda: None) if REHEARSAL else (lambda: torch.cuda.synchronize(device))
    dist = None
    ranks_seen = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        identities = [None] * world
        dist.all_gather_object(identities, device_identity(rank, local_rank))
        ranks_seen = check_ranks(identities, args.gpus, dist.get_world_size())     # every rank exits on a bad layout

    cfg = args.config
    if REHEARSAL:
        return rehearse(args, world, rank, dist, ranks_seen)
    from sound_event_detection_amd import _lib, feature_extraction as fe
    from sound_event_detection_amd.pipeline import SedPipeline

    kind = cfg.split("-")[0]
    clips = args.clips if args.clips is not None else CONFIGS[cfg]
    steps = args.steps if args.steps is not None else (3 if cfg in ("crnn", "frontend", "basic") else 2)
    alphas = [0.4, 0.5, 0.6]          # fixed ShakeShake mixing weights (the reference draws them per forward)
    frames = args.frames
    dtype = "f32"
    model = sd = x = y = pipe = None
    norm_kw = dict(log=True, normalise=True, group=True if world > 1 else None) if args.normalise else {}
    if kind in ("crnn", "frontend"):
        y = make_clips(clips, rank, device)
        model, sd = make_crnn(device)
        if args.precision:
            model.conv_precision = args.precision
        pipe = SedPipeline(model, sr=SR, n_fft=N_FFT, hop_length=HOP, win_length=WIN, **norm_kw)
        shape = f"{clips} synthetic 10 s 16 kHz clips per GPU resident in HBM (n_fft 2048, win 640, hop 320, 128 mels"
        if kind == "crnn":
            workload = f"mel frontend + DCASE-baseline CRNN + threshold/run-length decode, {shape}, {N_CLS} classes)"
            if args.normalise:
                workload += ("; with log-mel + per-bin mean/std normalisation from the step's own clips"
                             + (f" summed over the {world} ranks (all-reduce)" if world > 1 else ""))
            if model.conv_precision == "split_f16":
                dtype = "f32 (contractions as split-f16 x3 on the f16 matrix cores, f32 accumulate)"
            step_fn = lambda: pipe.run(y)                                            # noqa: E731
        else:
            workload = f"mel frontend only, {shape})"
            mel_out = torch.empty((clips, 128, 1 + N_SAMPLES // HOP), dtype=torch.float32, device=device)
            step_fn = lambda: dict(mel=fe.melspectrogram(y, SR, N_FFT, 128, HOP, WIN, out=mel_out))   # noqa: E731
    elif kind == "basic":
        y = make_clips(clips, rank, device, 0.25)
        model, sd = make_other("basic", device)
        pipe = SedPipeline(model, sr=SR, n_fft=N_FFT, hop_length=HOP, win_length=WIN)
        workload = (f"mel frontend (64 bins) + BasicRCNN + decode, {clips} synthetic 10 s 16 kHz clips per GPU resident "
                    f"in HBM (n_fft 2048, win 640, hop 320), {N_CLS} classes")
        dtype = "f32 (3x3 convs and GRU input projection as split-f16 x3, f32 accumulate)"
        step_fn = lambda: pipe.run(y)                                                # noqa: E731
    else:
        frames = frames or {"rcnn": 1000, "transformer": 1000, "rcnn-long": 6000, "rcnn-long-64": 6000,
                            "transformer-long": 24000}[cfg]
        x = make_feats(clips, frames, rank, device)
        model, sd = make_other(kind, device)
        pipe = SedPipeline(model, sr=SR, n_fft=N_FFT, hop_length=160, win_length=WIN)
        name = "ShakeRCNN" if kind == "rcnn" else "ShakeTransformer"
        workload = (f"{name} + decode on {clips} x ({frames} x 768) synthetic features per GPU resident in HBM "
                    f"(stand-in for the AST embeddings the reference feeds it; {frames // 4} "
                    f"{'GRU steps' if kind == 'rcnn' else 'tokens'} per clip), {N_CLS} classes")
        dtype = "f32 (contractions as split-f16 x3 on the f16 matrix cores, f32 accumulate)"
        step_fn = lambda: pipe.run_features(x, alphas=alphas)                        # noqa: E731

    collectives = [stats_collective("nccl (RCCL)")] if world > 1 and args.normalise else []
    gather = (AsyncGather(dist, world, device, collectives, "nccl (RCCL)")
              if world > 1 and not args.no_gather and kind != "frontend" else None)

    def step():
        b = step_fn()
        if gather is not None:
            gather(b["probs"])
        return b

    def fence():
        if gather is not None:
            gather.wait()
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        b = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernels = _lib.profile_collect()
    _lib.profile_enable(False)

    elapsed, rank_ms = gather_rank_times(dist, elapsed, steps, world, device)
    if pipe is not None and "n_events" in b:
        pipe.check(b)               # a kernel that reported a failed run makes this raise: no line from NaN logits
    n_events = int(b["n_events"].item()) if "n_events" in b else None

    if rank == 0:
        total_clips = clips * world * steps
        for k in kernels:
            k["avg_ms"] = k["ms"] / max(k["launches"], 1)
            k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            k["gbps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        kernels.sort(key=lambda k: -k["ms"])
        dom = kernels[0]
        intensity = dom["flops"] / max(dom["bytes"], 1.0)
        split = "split" in dom["name"] or dom["name"].startswith("shake_") or dom["name"].startswith("basic_")
        if dom["name"] == "mel_kernel":
            # fused frontend: 35 flop/B, above the f32 vector balance (157.3 TFLOP/s : 8 TB/s = 20 flop/B), so the
            # f32 vector rate binds (5.0 M clips/s) before HBM does (8.9 M clips/s).  The contract's roofline
            # object knows "hbm" and "mfma" only: the HBM view goes there, the binding vector bound beside it
            roof = dict(kernel=dom["name"], bound="hbm", achieved=dom["gbps"], peak=PEAK_HBM_GBPS, unit="GB/s",
                        frac=dom["gbps"] / PEAK_HBM_GBPS, traffic=None, avg_launch_ms=dom["avg_ms"],
                        launches=dom["launches"], bytes_per_launch=dom["bytes"] / dom["launches"],
                        binding_bound="f32 vector ALU", valu_achieved_tflops=dom["tflops"],
                        valu_peak_tflops=PEAK_F32_VALU_TFLOPS, valu_frac=dom["tflops"] / PEAK_F32_VALU_TFLOPS)
        elif intensity > PEAK_F32_VALU_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9):
            # "*_split" kernels run on the f16 matrix cores and spend three MFMAs per float32-grade
            # product (hi*hi + hi*lo + lo*hi): `achieved` stays the ALGORITHMIC rate, so its ceiling
            # against the f16 peak is 1/3; `mfma_issue_frac` is the share of the f16 peak the
            # instruction stream actually sustains.
            peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_VALU_TFLOPS
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
                # PMC-measured bytes per launch, scaled from the clips a launch covered in the profiled run to the
                # clips a launch covers here (every one of these kernels moves bytes in proportion to its clips)
                here = clips * steps / max(dom["launches"], 1)
                scale = here / rec.get("clips_per_launch", here)
                roof["traffic"] = rec["hbm_bytes_per_launch"] * scale
                roof["clips_per_launch"] = here
                roof["traffic_detail"] = {k: rec[k] for k in ("fetch_size_kib_raw", "write_size_kib", "note")}
                roof["traffic_detail"]["scaled_by"] = scale
                roof["algorithmic_bytes_per_launch"] = dom["bytes"] / dom["launches"]
        except (OSError, ValueError, KeyError):
            pass

        cpu = acceptance = None
        extra = {}
        n_cpu = args.cpu_clips
        if world > 1 and n_cpu is None:
            # the CPU leg is timed at N = 1 only: under a launcher every rank usually gets a single OpenMP thread,
            # which would turn a 25 s check into minutes while the other ranks wait at the final barrier
            n_cpu = 0
        if kind == "crnn":
            n_cpu = 192 if n_cpu is None else n_cpu
            if n_cpu > 0:
                from oracle import acceptance as oacc
                ns = min(n_cpu, clips)
                stats = (b["mel_mean"].cpu().numpy(), b["mel_std"].cpu().numpy()) if args.normalise else None
                cpu, cpu_probs, cpu_events = cpu_baseline_crnn(y[:ns].cpu().numpy(), sd, norm_stats=stats)
                gpu_probs = b["probs"][:ns].cpu().numpy()
                cpu["max_abs_prob_diff_vs_gpu"] = float(np.abs(gpu_probs - cpu_probs).max())
                n_diff, n_bad, near = oacc.frame_differences(cpu_probs.reshape(-1, N_CLS), gpu_probs.reshape(-1, N_CLS),
                                                             0.5, 1e-3, False)
                id2cls = {i: f"cls{i}" for i in range(N_CLS)}
                gpu_events = [e for e in pipe.events(b, id2cls, [f"clip{i}" for i in range(clips)])
                              if int(e[0][4:]) < ns]
                acceptance = dict(sample_clips=ns, frames_differing=n_diff, frames_differing_out_of_tolerance=n_bad,
                                  frames_within_tolerance_of_threshold=int(near.sum()), tolerance=1e-3, events_cpu=len(cpu_events),
                                  events_gpu=len(gpu_events), event_lists_identical=gpu_events == cpu_events)
            if model.conv_precision == "split_f16" and not args.no_exact_leg and world == 1:
                # the same path in exact-float32 arithmetic, outside the timed region above
                model.conv_precision = "f32"
                pipe.run(y)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                pipe.run(y)
                torch.cuda.synchronize(device)
                extra["value_f32_exact"] = clips * world / (time.perf_counter() - t1)
                model.conv_precision = "split_f16"
        elif kind == "frontend":
            n_cpu = 64 if n_cpu is None else n_cpu
            if n_cpu > 0:
                ns = min(n_cpu, clips)
                cpu, feats, _ = cpu_baseline_crnn(y[:ns].cpu().numpy(), None, with_model=False)
                got = b["mel"][:ns].cpu().numpy().reshape(feats.shape)
                cpu["max_rel_mel_diff_vs_gpu"] = float(np.abs(got - feats).max() / np.abs(feats).max())
        elif kind == "basic":
            n_cpu = 16 if n_cpu is None else n_cpu
            if n_cpu > 0:
                from oracle import mel as omel
                ns = min(n_cpu, clips)
                ys = y[:ns].cpu().numpy()
                t1 = time.perf_counter()
                feats = np.stack([omel.clip_feature(c, SR, N_FFT, 64, HOP, WIN) for c in ys])
                t_mel = time.perf_counter() - t1
                cpu, logits, _ = cpu_baseline_other("basic", feats, sd, None, 250)
                cpu["value"] = ns / (ns / cpu["value"] + t_mel)
                cpu["sample"] += f" + mel {t_mel:.2f} s (numpy, 1 thread)"
                cpu["max_abs_logit_diff_vs_gpu"] = float(np.abs(b["probs"][:ns].cpu().numpy() - logits).max())
        else:
            n_cpu = (2 if frames <= 6000 else 1) if n_cpu is None else n_cpu
            if n_cpu > 0:
                ns = min(n_cpu, clips)
                cpu, logits, _ = cpu_baseline_other(kind, x[:ns].cpu().numpy(), sd, alphas, frames // 4)
                cpu["max_abs_logit_diff_vs_gpu"] = float(np.abs(b["probs"][:ns].cpu().numpy() - logits).max())

        out = {
            "metric": "clips/sec (10 s clips, mel+CRNN+decode)" if cfg == "crnn" else f"clips/sec ({cfg})",
            "value": total_clips / elapsed,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "name": cfg, "clips_per_gpu": clips,
                       "parallelism": f"clip-sharded x{world}",
                       "all_gather": "float32 frame predictions, asynchronous, every step" if gather is not None else "none",
                       "events_last_step": n_events},
            **rank_fields(world, dist, ranks_seen, local_rank, rank_ms, collectives),
            **extra,
            "roofline": roof,
            "cpu_baseline": cpu,
            "event_acceptance": acceptance,
            "kernels": [dict(name=k["name"], launches=k["launches"], ms_total=round(k["ms"], 3),
                             tflops=round(k["tflops"], 2), gbps=round(k["gbps"], 1)) for k in kernels],
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
