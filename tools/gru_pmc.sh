#!/bin/bash
# SQ counter passes over the headline bench at 1024 clips; prints the rows of the GRU kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=gpurun_out/gru_pmc; rm -rf $P; mkdir -p $P
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $P/a -o p -f csv -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 --no-exact-leg > $P/a.json 2> $P/a.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace -d $P/b -o p -f csv -- python3 bench.py --clips 1024 --steps 1 --warmup 1 --cpu-clips 0 --no-exact-leg > $P/b.json 2> $P/b.log || exit 1
for x in a b; do python tools/summarize_profile.py pmcall $P/$x/p_counter_collection.csv | grep -E "^kernel|gru|mel_kernel|conv_split_kernel<64"; done
