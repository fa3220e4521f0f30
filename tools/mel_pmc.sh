cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
grep -c "" gpurun_out/counters_list.txt
P=gpurun_out/mel_pmc; mkdir -p $P
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $P/a -o p -f csv -- python3 bench.py --config frontend --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 > $P/a.json 2> $P/a.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU --kernel-trace -d $P/b -o p -f csv -- python3 bench.py --config frontend --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 > $P/b.json 2> $P/b.log || exit 1
echo done
