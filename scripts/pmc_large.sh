#!/bin/bash
# PMC passes for the degree-32 kernel only (rocprofv3, counters in their own runs).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-pmc_large}
CFG=${2:-100000,33,64}
mkdir -p $O
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run sq_l    --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_l -- python3 scripts/prof_enhance.py $CFG 3 0 narrow
run sq_l2   --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_l2 -- python3 scripts/prof_enhance.py $CFG 3 0 narrow
run sq_l3   --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/sq_l3 -- python3 scripts/prof_enhance.py $CFG 3 0 narrow
python3 scripts/pmc_summary.py $O/sq_l $O/sq_l2 $O/sq_l3 > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
