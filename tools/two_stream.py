"""Experiment: the bench batch as one pass on one stream versus two half batches on two streams."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from sound_event_detection_amd.pipeline import SedPipeline

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, sd = bench.make_model(dev)
kw = dict(sr=bench.SR, n_fft=bench.N_FFT, n_mels=bench.N_MELS, hop_length=bench.HOP, win_length=bench.WIN)
y = bench.make_clips(4096, 0, dev)
one = SedPipeline(model, **kw)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipes = [SedPipeline(model, **kw) for _ in range(parts)]
streams = [torch.cuda.Stream(dev) for _ in range(parts)]
ys = [c.contiguous() for c in y.chunk(parts)]

def run_one():
    one.run(y)

def run_two():
    for p, s, yy in zip(pipes, streams, ys):
        with torch.cuda.stream(s):
            p.run(yy)

for name, fn in (("one", run_one), ("multi", run_two), ("one", run_one), ("multi", run_two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(name, parts, round(dt * 1e3, 2), "ms", round(4096 / dt), "clips/s", flush=True)
ref = one.run(y)["probs"].clone()
run_two(); torch.cuda.synchronize()
got = torch.cat([p._buf[next(iter(p._buf))]["probs"] for p in pipes])
print("max diff", float((ref - got).abs().max()))
