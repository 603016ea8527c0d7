#!/bin/bash
# SQ counters of the bench kernels: instruction mix, LDS activity / bank
# conflicts / issue stalls, wave cycles -- two --pmc passes (8 SQ counters per pass), each with
# --kernel-trace only (no sys / hip / memory-copy tracing next to --pmc).
# usage: bash tools/sq_counters.sh <subdir of gpurun_out> ; then tools/sq_summarize.py
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra"
timeout 900 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $out -o insts -- $cmd > $out/insts.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $out -o cycles -- $cmd > $out/cycles.log 2>&1
ls $out | head -20
tail -2 $out/insts.log $out/cycles.log
