#!/bin/bash
# usage: tools/shake_conv_variants.sh <lib.so>...   -- ShakeRCNN step time and its conv kernels per library build
for lib in "$@"; do
  SEDMI_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/bench_shake.py --model rcnn --clips 128 --steps 2 > gpurun_out/sv.json || exit 1
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sv.json"))
print(sys.argv[1], round(d["ms_per_step"], 2), "ms;",
      ", ".join(f'{k["name"]} {k["ms_per_step"]}' for k in d["kernels"] if "gate" in k["name"] or "conv" in k["name"]))
PY
done
