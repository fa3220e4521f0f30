#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate
runs of `bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0`).

    make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream: a torch elementwise add in the same pass (reads 2 x 160000 KiB, reports ~160014) calibrates the
factor 2 that is applied here.  Keys are the kernel names bench.py prints (= the kernels' own names with their distinguishing template arguments).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def bench_name(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"conv_split_kernel<(\d+), (\d+), (\d+),", k)
    if m:
        return "conv_split_kernel<%s,%s,%s>" % m.groups()
    m = re.match(r"linear_split_kernel<768, (\d+), (\d+)>", k)
    if m:
        return "linear_split_kernel<768,%s,%s>" % m.groups()
    m = re.match(r"gru_split_kernel<(true|false)>", k)
    if m:
        return "gru_split_kernel<%s>" % m.group(1)
    for a in ("mel_kernel", "conv0_split_kernel", "conv0_kernel", "gru0_fused_kernel", "gru_kernel", "head_small_kernel",
              "head_kernel", "head_combine_kernel"):
        if k.startswith(a):
            return a
    return None


PROFILED_CLIPS = 1024 * 2          # the profiled command: one warm-up step and one timed step of 1024 clips


def mean_per_launch(path, counter, counts=None):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = bench_name(r["Kernel_Name"])
        if n:
            acc[n][0] += 1
            acc[n][1] += float(r["Counter_Value"])
    if counts is not None:
        counts.update({k: n for k, (n, _) in acc.items()})
    return {k: v / n for k, (n, v) in acc.items()}


def main():
    launches = {}
    fetch = mean_per_launch(sys.argv[1], "FETCH_SIZE", launches)
    write = mean_per_launch(sys.argv[2], "WRITE_SIZE")
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on bench.py --clips 1024 --steps 1 --warmup 1; "
            "clips_per_launch = the 2048 clips of that run over the kernel's launches (a kernel that covers half the "
            "image width per launch counts half the clips); "
            "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request; calibrated on a torch add in the same pass)")
    out = {k: dict(hbm_bytes_per_launch=1024.0 * (2.0 * fetch[k] + write.get(k, 0.0)), fetch_size_kib_raw=fetch[k],
                   write_size_kib=write.get(k, 0.0), clips_per_launch=PROFILED_CLIPS / launches[k], note=note)
           for k in fetch}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:32s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
