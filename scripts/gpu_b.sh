#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for s in 100008,9,16 1000000,9,16 10000008,9,16; do timeout -k 10 200 python scripts/domain_ab.py $s > gpurun_out/dom_$s.log 2>&1; tail -4 gpurun_out/dom_$s.log; done
O=gpurun_out/prof_b; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_s -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide > $O/sq_s.log 2>&1; echo rc=$?
python3 scripts/pmc_summary.py $O/sq_s
