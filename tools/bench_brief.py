"""Print the headline numbers of bench.py JSON lines: python tools/bench_brief.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(d["config"].get("name", "?"), round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
    for k in d.get("kernels", [])[:10]:
        print("   ", k["name"], k["launches"], k["ms_total"], k.get("tflops"), k.get("gbps"))
