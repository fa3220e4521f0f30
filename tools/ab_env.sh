#!/bin/bash
# usage: tools/ab_env.sh VAR=value [VAR=value ...]  -- headline bench once per setting (and once with none), prints the
# per-kernel ms per 4096-clip step; for A/B runs of the library's environment switches on one box
run() {
  timeout -k 10 120 python bench.py --cpu-clips 0 --no-exact-leg --steps 5 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$1 failed"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$1" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], "ms/step %.2f"%d['ms_per_step'], " ".join(f"{k['name'].replace('conv_split_kernel','cs').replace('conv_glut_kernel','cg').replace('_kernel','')}={k['ms_total']/d['steps']:.2f}" for k in d['kernels'][:13]))
PY
}
mkdir -p gpurun_out
run default
for kv in "$@"; do export "$kv"; run "$kv"; unset "${kv%%=*}"; done
run default
