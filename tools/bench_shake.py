#!/usr/bin/env python3
"""Per-kernel timing of the Shake models on full-size clips (1000 x 768 AST-shaped features).

    python tools/bench_shake.py --model rcnn --clips 32 --steps 3
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["rcnn", "transformer", "basic"], default="rcnn")
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from sound_event_detection_amd import models
    if args.model == "rcnn":
        m = models.ShakeRCNN(10, 0)
        m.load_state_dict(synth.synth_shake_state_dict(synth.shake_rcnn_state_dict_spec(10), 0))
    elif args.model == "basic":         # mel input: 501 frames x 64 bins (training.py:201)
        m = models.BasicRCNN(10, 64)
        m.load_state_dict(synth.synth_shake_state_dict(synth.basic_rcnn_state_dict_spec(10, 64), 0))
    else:
        m = models.ShakeTransformer(10, 0)
        m.load_state_dict(synth.synth_shake_state_dict(synth.shake_transformer_state_dict_spec(10), 0))
    m.eval()
    m = m.cuda()
    if args.model == "basic":
        base = torch.from_numpy(synth.synth_shake_feats(3, min(args.clips, 4), min(args.frames, 501), 64)).cuda()
    else:
        base = torch.from_numpy(synth.synth_shake_feats(3, min(args.clips, 4), args.frames)).cuda()
    x = base.repeat((args.clips + base.shape[0] - 1) // base.shape[0], 1, 1)[:args.clips].contiguous()
    al = [0.4, 0.5, 0.6]
    fwd = (lambda: m(x)) if args.model == "basic" else (lambda: m(x, al))
    fwd()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ks = _lib.profile_collect()
    _lib.profile_enable(False)
    ks.sort(key=lambda k: -k["ms"])
    out = dict(model=args.model, clips=args.clips, frames=args.frames, ms_per_step=1e3 * dt, clips_per_s=args.clips / dt,
               kernels=[dict(name=k["name"], launches=k["launches"], ms_per_step=round(k["ms"] / args.steps, 3),
                             tflops=round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] else 0,
                             gbps=round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1) if k["ms"] else 0) for k in ks])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
