#!/bin/bash
# A/B of library builds on the config-4 step (wide mesh): rocprofv3 --kernel-trace durations per kernel, one process per
# library, two rounds.  usage (GPU box): bash scripts/ab_large.sh lib1.so lib2.so ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_large; rm -rf $O; mkdir -p $O
for rnd in 0 1; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    export LSSVR_HIP_LIB=$PWD/$lib
    rocprofv3 --kernel-trace --output-format csv -d $O/${name}_$rnd -- python3 scripts/prof_step.py 100008,33,64 10 wide > $O/${name}_$rnd.log 2>&1
    unset LSSVR_HIP_LIB
    echo "round $rnd $name"
    python3 scripts/kernel_trace_summary.py $O/${name}_$rnd/*/*kernel_trace.csv | grep -E "solve4|moments" | cut -c1-60,88-170
  done
done
