#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_enhance_large.py tests/test_gpu_hetero.py tests/test_gpu_enhance.py tests/test_gpu_shared.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-400 > gpurun_out/pytest_f.log; tail -30 gpurun_out/pytest_f.log
timeout -k 10 200 python scripts/domain_ab.py 100000,33,64 2>&1 | tail -4
timeout -k 10 200 python scripts/domain_ab.py 1000000,33,64 2>&1 | tail -4
