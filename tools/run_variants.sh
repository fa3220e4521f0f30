#!/bin/bash
# usage: tools/run_variants.sh base v1 v2 ...   -- headline bench per library build variants/libsedmi_<name>.so
# ("base" = the in-tree library); prints ms per 4096-clip step of every kernel
for v in "$@"; do
  if [ $v = base ]; then unset SEDMI_LIB_PATH; else export SEDMI_LIB_PATH=$PWD/variants/libsedmi_$v.so; fi
  timeout -k 10 120 python bench.py --cpu-clips 0 --no-exact-leg --steps 5 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v failed"; tail -3 gpurun_out/var_$v.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/var_{sys.argv[1]}.json"))
print(sys.argv[1], "ms/step %.2f"%d['ms_per_step'], " ".join(f"{k['name'].replace('conv_split','cs').replace('_kernel','')}={k['ms_total']/d['steps']:.2f}" for k in d['kernels'][:12]))
PY
done
