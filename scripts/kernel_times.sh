#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of scripts/prof_enhance.py for one or more libraries
# usage: scripts/kernel_times.sh <tag> "<prof_enhance args>" lib1.so [lib2.so ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; args=$2; shift 2
O=gpurun_out/kt_$tag; rm -rf $O; mkdir -p $O
for lib in "$@"; do
  name=$(basename $lib .so)
  export LSSVR_HIP_LIB=$PWD/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 scripts/prof_enhance.py $args > $O/$name.log 2>&1
  unset LSSVR_HIP_LIB
  echo "== $name"; python3 - "$O/$name" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lssvr" in r["Name"]:
            print("  %-70s calls %4s avg %9.1f us min %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
