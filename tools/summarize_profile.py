#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small tables kept under profiles/.

usage: summarize_profile.py stats  <kernel_stats.csv>            -> per-kernel calls / avg us / share
       summarize_profile.py pmc    <counter_collection.csv> NAME -> per-kernel mean counter value per launch
       summarize_profile.py pmcall <counter_collection.csv>      -> per-kernel means of every counter in the file
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^()]*?>)?)\(", name)
    return (m.group(1) if m else name)[:60]


def stats(path):
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>12s} {'total_ms':>10s} {'share%':>7s}")
    for r in rows:
        print(f"{short(r['Name']):60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:12.1f} "
              f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):7.2f}")


def pmc(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{'kernel':60s} {'launches':>8s} {counter + '/launch':>22s} {'avg_us':>12s}")
    for k, (n, v, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} {n:8d} {v / n:22.1f} {us / n:12.1f}")


def pmcall(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    names = []
    for r in csv.DictReader(open(path)):
        k, c = short(r["Kernel_Name"]), r["Counter_Name"]
        if c not in names:
            names.append(c)
        a = acc[k][c]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if c == names[0]:
            dur[k][0] += 1
            dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{'kernel':48s} {'n':>4s} {'avg_us':>9s} " + " ".join(f"{c[-22:]:>22s}" for c in names))
    for k in sorted(acc, key=lambda k: -dur[k][1]):
        n, us = dur[k]
        print(f"{k[:48]:48s} {n:4d} {us / max(n, 1):9.1f} " +
              " ".join(f"{acc[k][c][1] / max(acc[k][c][0], 1):22.0f}" for c in names))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmcall":
        pmcall(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
