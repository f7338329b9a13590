#!/bin/bash
# Round 4 collection on the GPU box (gpurun): rocprofv3 kernel-trace stats of the DEFAULT bench command (it carries
# BASELINE configs 2, 4 and 5), the PMC passes of the kernels the timed regions launch (the fused step of config 2,
# the three-launch step of config 4), and the bench lines of the same build without the profiler.
# Counters go in their own runs with --kernel-trace only (no sys/hip/hsa traces).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04
rm -rf $O
mkdir -p $O
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run stats   --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline
run fetch   --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 scripts/prof_step.py 100008,9,16 5 wide 12500000
run write   --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 scripts/prof_step.py 100008,9,16 5 wide 12500000
run sq_s    --pmc $SQ --kernel-trace --output-format csv -d $O/sq_s -- python3 scripts/prof_step.py 100008,9,16 5 wide
run sq_s2   --pmc $SQ2 --kernel-trace --output-format csv -d $O/sq_s2 -- python3 scripts/prof_step.py 100008,9,16 5 wide
run sq_e    --pmc $SQ --kernel-trace --output-format csv -d $O/sq_e -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide
run fetchL  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetchL -- python3 scripts/prof_step.py 100008,33,64 3 wide 12500000
run writeL  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/writeL -- python3 scripts/prof_step.py 100008,33,64 3 wide 12500000
run sq_l    --pmc $SQ --kernel-trace --output-format csv -d $O/sq_l -- python3 scripts/prof_step.py 100008,33,64 3 wide
run sq_l2   --pmc $SQ2 --kernel-trace --output-format csv -d $O/sq_l2 -- python3 scripts/prof_step.py 100008,33,64 3 wide
python3 scripts/pmc_summary.py $O/fetch $O/write $O/sq_s $O/sq_s2 $O/sq_e $O/fetchL $O/writeL $O/sq_l $O/sq_l2 > $O/pmc_summary.txt 2>&1
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
grep -v "at::native" $O/pmc_summary.txt | tail -n 90
