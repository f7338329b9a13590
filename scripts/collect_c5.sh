#!/bin/bash
# Runs on the GPU box (gpurun): BASELINE config 5 (variable coefficient, 1e6 elements) under rocprofv3 --
# kernel-trace stats of `bench.py --config 5`, SQ instruction / stall / LDS counters of the lane kernel, and
# FETCH_SIZE / WRITE_SIZE with BOTH calibration probes (full-line stream, half-line row chunks) in the same
# passes.  Counters in their own runs with --kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r03}
O=gpurun_out/prof_${R}_c5
rm -rf $O
mkdir -p $O
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run stats --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline
run sq_a  --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_a -- python3 scripts/prof_c5.py
run sq_b  --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_b -- python3 scripts/prof_c5.py
run sq_c  --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/sq_c -- python3 scripts/prof_c5.py
run fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 scripts/prof_c5.py
run write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 scripts/prof_c5.py
run fetch_em --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_em -- python3 scripts/prof_c5.py 1000008 5 wide element
run write_em --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_em -- python3 scripts/prof_c5.py 1000008 5 wide element
run sq_em --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq_em -- python3 scripts/prof_c5.py 1000008 5 wide element
run tcc   --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- python3 scripts/prof_c5.py
python3 scripts/pmc_summary.py $O/sq_a $O/sq_b $O/sq_c $O/fetch $O/write $O/fetch_em $O/write_em $O/sq_em $O/tcc > $O/pmc_summary.txt 2>&1
cp $O/stats/*/*kernel_stats.csv $O/bench_c5_kernel_stats.csv 2>/dev/null
grep -v "at::native" $O/pmc_summary.txt | tail -n 150
python3 bench.py --config 5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
