#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_g; rm -rf $O; mkdir -p $O
for sol in 0; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_l$sol -- python3 scripts/prof_enhance.py 1000000,33,64 3 $sol narrow > $O/sq_l$sol.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq_m$sol -- python3 scripts/prof_enhance.py 1000000,33,64 3 $sol narrow > $O/sq_m$sol.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/sq_n$sol -- python3 scripts/prof_enhance.py 1000000,33,64 3 $sol narrow > $O/sq_n$sol.log 2>&1; echo rc=$?
done
python3 scripts/pmc_summary.py $O/sq_l0 $O/sq_m0 $O/sq_n0
