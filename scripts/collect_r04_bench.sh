#!/bin/bash
# Round 4, second half: the bench lines of the final build without the profiler attached (after
# profiles/instruction_counts.json and profiles/traffic.json have been regenerated from scripts/collect_r04.sh).
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04
mkdir -p $O
# bench lines of the same build without the profiler attached
python3 bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python3 bench.py --config 4 --steps 20 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python3 bench.py --config 5 --steps 20 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python3 bench.py --elements 10000008 --steps 20 --warmup 3 > $O/bench_1e7.json 2> $O/bench_1e7.err; echo "bench 1e7 rc=$?"
LSSVR_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err; echo "bench gloo2 rc=$?"
LSSVR_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 3 --no-embedded --no-cpu-baseline > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo "bench rccl1 rc=$?"
