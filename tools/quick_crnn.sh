#!/bin/bash
# quick check of a conv/GRU change: the CRNN parity tests, then the headline bench with its kernel table
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_crnn.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 python bench.py --cpu-clips 0 --no-exact-leg --steps 5 > gpurun_out/quick.json 2> gpurun_out/quick.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/quick.json"))
print("ms/step %.2f"%d['ms_per_step'], " ".join(f"{k['name'].replace('conv_split','cs')}={k['ms_total']/d['steps']:.2f}" for k in d['kernels'][:12]))
PY
