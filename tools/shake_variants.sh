#!/bin/bash
# usage: tools/shake_variants.sh <lib.so>...   -- Shake model timings per library build (GEMM-like kernels only)
for lib in "$@"; do
  for model in rcnn transformer; do
    SEDMI_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/bench_shake.py --model $model --clips 128 --steps 2 > gpurun_out/sv.json || exit 1
    python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sv.json"))
print(sys.argv[1], d["model"], round(d["ms_per_step"], 2), "ms;",
      ", ".join(f'{k["name"]} {k["ms_per_step"]} ({k["tflops"]} TF)' for k in d["kernels"]
                if any(t in k["name"] for t in ("proj", "ffn", "fc"))))
PY
  done
done
