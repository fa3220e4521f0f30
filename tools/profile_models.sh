#!/bin/bash
# Kernel-trace statistics of the other models' benches (on the GPU box, from the repository root).
# usage: bash tools/profile_models.sh <out_dir under gpurun_out>
set -e
OUT=${1:-gpurun_out/prof_models}
export TMPDIR=/tmp
for model in rcnn transformer basic; do
  clips=128; [ $model = basic ] && clips=1024
  mkdir -p $OUT/$model
  timeout -k 10 200 python3 tools/bench_shake.py --model $model --clips $clips --steps 3 > $OUT/$model/bench.json
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/$model -o s -f csv -- python3 tools/bench_shake.py --model $model --clips $clips --steps 3 > $OUT/$model/bench_under_rocprof.json 2> $OUT/$model/log.txt
done
ls $OUT/*
