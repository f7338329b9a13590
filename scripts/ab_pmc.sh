#!/bin/bash
# A/B of two libraries on the GPU box: stamped kernel durations (scripts/ab_kernel.py) and the
# executed instruction counts of the lane kernel (rocprofv3 --pmc, one pass per library).
# usage: scripts/ab_pmc.sh <tag> <libA.so> <libB.so>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; A=$2; B=$3
O=gpurun_out/ab_$tag
rm -rf $O; mkdir -p $O
python scripts/ab_kernel.py $A $B -- 100008,9,16 1000000,9,16 100000,33,64 > $O/ab.txt 2>&1
for lib in $A $B; do
  name=$(basename $lib .so)
  export LSSVR_HIP_LIB=$PWD/$lib
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$name -- python3 scripts/prof_enhance.py 100008,9,16 5 0 wide > $O/pmc_$name.log 2>&1
  unset LSSVR_HIP_LIB
done
python3 scripts/pmc_summary.py $O/pmc_* > $O/pmc_summary.txt 2>&1
cat $O/ab.txt; grep -E "==|SQ_INSTS|SQ_WAVES|mean" $O/pmc_summary.txt
