#!/bin/bash
# Every bench.py workload, one JSON line each, into gpurun_out/<tag>_bench_<config>.json
# usage: tools/bench_all.sh <tag> [configs...]
set -u
tag=${1:-r2}; shift || true
configs=${*:-crnn frontend rcnn transformer basic rcnn-long transformer-long}
mkdir -p gpurun_out
for c in $configs; do
    echo "== $c"
    timeout -k 10 300 python bench.py --config "$c" > "gpurun_out/${tag}_bench_${c}.json" 2> "gpurun_out/${tag}_bench_${c}.err" || { echo "FAILED $c"; tail -5 "gpurun_out/${tag}_bench_${c}.err"; exit 1; }
    python - "gpurun_out/${tag}_bench_${c}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], d["cpu_baseline"]
print(f'{d["config"]["name"]}: {d["value"]:.1f} {d["unit"]}, {d["ms_per_step"]:.2f} ms/step; dominant {r["kernel"]} {r["achieved"]:.1f} {r["unit"]} = {r["frac"]:.3f} of peak; cpu {c["value"]:.2f} ({c["cores"]} cores)' if c else d["value"])
for k in ("value_f32_exact", "event_acceptance"):
    if d.get(k): print("  ", k, d[k])
if c:
    print("  ", {k: v for k, v in c.items() if k.startswith("max_") or k.startswith("batch1")})
for k in d["kernels"][:8]: print("    ", k)
PY
done
