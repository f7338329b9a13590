#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" > gpurun_out/pytest_e.log; tail -30 gpurun_out/pytest_e.log
for cfg in "--degree 8 --colloc 16" "--degree 32 --colloc 64 --elements 100000 --domain narrow"; do
timeout -k 10 300 python bench.py --solver dual $cfg --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/bench_dual.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['elements_total'], 'dual value %.3e el/s'%d['value'], 'kernel us %.1f'%r['kernel_us_avg'], 'frac %.3f'%r['frac'], 'flops/el', r['flops_per_element'], d['accuracy'])"
done
