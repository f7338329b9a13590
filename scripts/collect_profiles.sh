#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of bench.py and the PMC passes
# (HBM traffic, calibrated; VALU / MFMA / LDS activity) that DESIGN.md and bench.py cite.
# Counters go in their own runs with --kernel-trace only (no sys/hip/hsa traces).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r02}
O=gpurun_out/prof_$R
rm -rf $O
mkdir -p $O
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run stats   --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline
run fetch   --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide 12500000
run write   --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide 12500000
run fetchL  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetchL -- python3 scripts/prof_enhance.py 100000,33,64 3 0 narrow
run writeL  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/writeL -- python3 scripts/prof_enhance.py 100000,33,64 3 0 narrow
run sq_s    --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_s -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide
run sq_s2   --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_s2 -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide
run sq_s1e7 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_s1e7 -- python3 scripts/prof_enhance.py 10000008,9,16 5 0 wide
run sq_s1m  --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_s1m -- python3 scripts/prof_enhance.py 1000000,9,16 5 0 narrow
run sq_w9   --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_w9 -- python3 scripts/prof_enhance.py 1000000,9,16 3 2 narrow
run sq_lm   --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_lm -- python3 scripts/prof_enhance.py 100000,33,64 3 2 narrow
run sq_lm2  --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_lm2 -- python3 scripts/prof_enhance.py 100000,33,64 3 2 narrow
run fetchLm --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetchLm -- python3 scripts/prof_enhance.py 100000,33,64 3 2 narrow
run writeLm --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/writeLm -- python3 scripts/prof_enhance.py 100000,33,64 3 2 narrow
run statsL  --kernel-trace --stats --output-format csv -d $O/statsL -- python3 bench.py --degree 32 --colloc 64 --elements 100000 --domain narrow --no-cpu-baseline
run sq_d9   --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_d9 -- python3 scripts/prof_enhance.py 100008,9,16 3 1 wide
run sq_d33  --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_d33 -- python3 scripts/prof_enhance.py 100000,33,64 2 1 narrow
run sq_l    --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_l -- python3 scripts/prof_enhance.py 100000,33,64 3 0 narrow
run sq_l2   --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_l2 -- python3 scripts/prof_enhance.py 100000,33,64 3 0 narrow
run fetchS  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetchS -- python3 scripts/prof_shared.py 10000000,9,16 5 narrow
run writeS  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/writeS -- python3 scripts/prof_shared.py 10000000,9,16 5 narrow
python3 scripts/pmc_summary.py $O/fetch $O/write $O/fetchL $O/writeL $O/fetchS $O/writeS $O/sq_s $O/sq_s2 $O/sq_s1m $O/sq_s1e7 $O/sq_w9 $O/sq_l $O/sq_l2 $O/sq_lm $O/sq_lm2 $O/sq_d9 $O/sq_d33 > $O/pmc_summary.txt 2>&1
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
cp $O/statsL/*/*kernel_stats.csv $O/bench_deg32_kernel_stats.csv 2>/dev/null
tail -n 60 $O/pmc_summary.txt
# bench lines of the same build without the profiler attached
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python3 bench.py --degree 32 --colloc 64 --elements 100000 --domain narrow --no-cpu-baseline > $O/bench_deg32.json 2> $O/bench_deg32.err; echo "bench32 rc=$?"
python3 bench.py --elements 10000008 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_1e7.json 2> $O/bench_1e7.err; echo "bench1e7 rc=$?"
python3 bench.py --solver dual --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_dual_deg8.json 2> $O/bench_dual_deg8.err; echo "bench dual8 rc=$?"
python3 bench.py --solver dual --degree 32 --colloc 64 --elements 100000 --domain narrow --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_dual_deg32.json 2> $O/bench_dual_deg32.err; echo "bench dual32 rc=$?"
python3 scripts/make_traffic.py $O > $O/traffic.json 2> $O/traffic.err; echo "traffic rc=$?"
