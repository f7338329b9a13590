#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_enhance_dual.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300 | tail
sed -i 's/for R in .*; do/for R in 3; do/' scripts/gpu_d6.sh
sed -i 's/LSSVR_DUAL_REFINE=$R //' scripts/gpu_d6.sh
bash scripts/gpu_d6.sh 2>&1 | grep refine
