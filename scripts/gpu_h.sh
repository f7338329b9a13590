#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-300 > gpurun_out/pytest_h.log; tail -12 gpurun_out/pytest_h.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
