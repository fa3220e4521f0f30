#!/bin/bash
# usage: tools/bench_variants.sh <lib.so>...   -- one short bench per library build, prints the top kernels
for lib in "$@"; do
  echo "== $lib"
  SEDMI_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --cpu-clips 0 --steps 2 > gpurun_out/bv.json || exit 1
  python - <<'PY'
import json
d = json.load(open("gpurun_out/bv.json"))
print(round(d["value"]), round(d["ms_per_step"], 2))
for k in d["kernels"][:13]:
    print("   ", k["name"], round(k["ms_total"] / k["launches"] * 1e3, 1), "us")
PY
done
