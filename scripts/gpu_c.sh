#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fem_eval.py -m gpu -x -q > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_c.log
LSSVR_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "bench gloo2 rc=$?"; tail -3 gpurun_out/bench_g2.err
LSSVR_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_rccl1.json 2> gpurun_out/bench_rccl1.err; echo "bench rccl1 rc=$?"; tail -3 gpurun_out/bench_rccl1.err
