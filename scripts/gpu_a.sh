#!/bin/bash
# round-2 GPU pass A: full GPU suite, measured bars, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 300 python scripts/measure_bars.py > gpurun_out/bars.log 2>&1; echo "bars rc=$?"
timeout -k 10 400 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_a.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_us_avg'], d['roofline']['frac'], d.get('narrow_domain'), d.get('accuracy'))
PY
