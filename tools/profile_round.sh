#!/bin/bash
# Collects the round's profiles on the GPU box: kernel-trace stats, two PMC passes (HBM traffic), bench lines.
# usage (on the box, from the repository root): bash tools/profile_round.sh <out_dir under gpurun_out>
set -e
OUT=${1:-gpurun_out/prof}
mkdir -p $OUT/stats $OUT/fetch $OUT/write
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -f csv -- python3 bench.py --steps 4 --warmup 0 --cpu-clips 0 --no-exact-leg > $OUT/bench_under_rocprof.json 2> $OUT/stats/log.txt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -f csv -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 --no-exact-leg > /dev/null 2> $OUT/fetch/log.txt
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -f csv -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 --no-exact-leg > /dev/null 2> $OUT/write/log.txt
ls $OUT/stats $OUT/fetch $OUT/write
