#!/bin/bash
# Turns the raw output of tools/profile_round.sh (+ an optional SQ counter pass) into the tracked summaries.
# usage: bash tools/publish_profiles.sh <prof_dir> <round tag, e.g. r1> [sq_pmc_dir]
set -e
P=$1; R=$2; SQ=$3
python tools/make_traffic.py $P/fetch/p_counter_collection.csv $P/write/p_counter_collection.csv profiles/traffic.json > /dev/null
python tools/summarize_profile.py stats $P/stats/s_kernel_stats.csv > profiles/${R}_kernel_stats.txt
cp $P/stats/s_kernel_stats.csv profiles/${R}_kernel_stats_raw.csv
{ echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0"
  echo "# FETCH_SIZE is in KiB and, on gfx950, reads exactly 1/2 of the bytes of a wide coalesced stream: the torch"
  echo "# vectorized add in this same run reads 2 x 160000 KiB and reports ~160014 KiB. Double it before use."
  python tools/summarize_profile.py pmc $P/fetch/p_counter_collection.csv FETCH_SIZE; } > profiles/${R}_pmc_fetch_size.txt
{ echo "# rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 (KiB per launch)"
  python tools/summarize_profile.py pmc $P/write/p_counter_collection.csv WRITE_SIZE; } > profiles/${R}_pmc_write_size.txt
cp $P/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
# the bench line was produced before traffic.json was refreshed: patch its traffic field from the new file
python - "$P/bench.json" "profiles/${R}_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = json.load(open("profiles/traffic.json"))
rec = t.get(d["roofline"]["kernel"])
if rec:
    here = d["roofline"].get("clips_per_launch") or rec.get("clips_per_launch")
    d["roofline"]["traffic"] = rec["hbm_bytes_per_launch"] * here / rec.get("clips_per_launch", here)
    d["roofline"]["traffic_detail"] = {k: rec[k] for k in ("fetch_size_kib_raw", "write_size_kib", "note")}
json.dump(d, open(sys.argv[2], "w"))
PY
if [ -n "$SQ" ]; then
  { echo "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0"
    echo "# SQ_WAVE/WAIT/ACTIVE in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs"
    python tools/summarize_profile.py pmcall $SQ/p_counter_collection.csv; } > profiles/${R}_pmc_sq_waits.txt
fi
