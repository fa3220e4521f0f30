#!/usr/bin/env python3
"""Diagnostic: where does conv_split_kernel spend its time?  Needs a library built with -DSPLIT_TIMING
(SEDMI_LIB_PATH points at it); prints, per conv block, the mean s_memtime ticks per tile in each section."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from sound_event_detection_amd import _lib  # noqa: E402
from sound_event_detection_amd.pipeline import SedPipeline  # noqa: E402

SECTIONS = ["prologue", "chunk sync", "stage+sync", "main loop", "bn", "phase sync", "z split+sync", "glu gemm",
            "epilogue"]


def main():
    dev = torch.device("cuda", 0)
    model, _ = bench.make_crnn(dev)
    pipe = SedPipeline(model, sr=bench.SR, n_fft=bench.N_FFT, n_mels=128, hop_length=bench.HOP,
                       win_length=bench.WIN)
    y = bench.make_clips(512, 0, dev)
    h = C.CDLL(_lib.LIB_PATH)
    buf = (C.c_ulonglong * 72)()
    pipe.run(y)
    h.sedmi_debug_split_ticks(buf)
    _lib.profile_enable(True)
    pipe.run(y)
    torch.cuda.synchronize()
    prof = {k["name"]: k["ms"] / max(k["launches"], 1) for k in _lib.profile_collect()}
    h.sedmi_debug_split_ticks(buf)
    names = ["conv_split_kernel<16,32,64>", "conv_split_kernel<32,64,32>", "conv_split_kernel<64,128,16>", "conv_split_kernel<128,128,8>",
             "conv_split_kernel<128,128,4>", "conv_split_kernel<128,128,2>"]
    for layer in range(6):
        row = [buf[layer * 12 + i] for i in range(12)]
        ms = prof.get(names[layer], 0.0)
        print(f"{names[layer]}: launch {ms * 1e3:.1f} us, span {row[11] - row[10]} ticks -> "
              f"{(row[11] - row[10]) / max(ms, 1e-9) / 1e3:.1f} ticks/us")
        n = max(row[9], 1)
        tot = sum(row[:9])
        print(f"block {layer + 1}: tiles {row[9]}, ticks/tile {tot / n:.1f}")
        for name, v in zip(SECTIONS, row[:9]):
            print(f"    {name:14s} {v / n:9.1f}  {100.0 * v / max(tot, 1):5.1f}%")


if __name__ == "__main__":
    main()
