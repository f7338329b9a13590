#!/bin/bash
# Config 5's variable-coefficient lane kernel under rocprofv3: durations + two SQ counter passes (scripts/prof_c5.py).
# usage (GPU box): bash scripts/pmc_c5_quick.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c5q_${1:-x}
rm -rf $O; mkdir -p $O
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run sq_a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_a -- python3 scripts/prof_c5.py
run sq_b --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_b -- python3 scripts/prof_c5.py
python3 scripts/pmc_summary.py $O/sq_a $O/sq_b > $O/pmc_summary.txt 2>&1
grep -A 9 "enhance_small_kernel" $O/pmc_summary.txt
