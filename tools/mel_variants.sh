#!/bin/bash
# usage: tools/mel_variants.sh base v1 v2 ...  -- frontend bench (mel only, 4096 clips) per library build
# variants/libsedmi_<name>.so ("base" = the in-tree library); prints the mel kernel's ms per 4096 clips
for v in "$@"; do
  if [ $v = base ]; then unset SEDMI_LIB_PATH; else export SEDMI_LIB_PATH=$PWD/variants/libsedmi_$v.so; fi
  timeout -k 10 120 python bench.py --config frontend --cpu-clips 8 --steps 10 > gpurun_out/melvar_$v.json 2> gpurun_out/melvar_$v.err || { echo "$v failed"; tail -3 gpurun_out/melvar_$v.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/melvar_{sys.argv[1]}.json"))
print(sys.argv[1], "ms/step %.3f"%d['ms_per_step'], "rel err vs cpu %.2e" % d['cpu_baseline']['max_rel_mel_diff_vs_gpu'])
PY
done
