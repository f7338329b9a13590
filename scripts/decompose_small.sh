#!/bin/bash
# Builds the LSSVR_DECOMP variants of the lane kernel (enhance_small_cheb.hpp) next to the shipped build:
# build/decomp/lib_{full,empty,loads,nostore,noarith}.so -- only enhance_small_a.o differs.
# Timed on the GPU box by scripts/ab_kernel.py (hipExt-stamped launches, interleaved rounds):
#   python scripts/ab_kernel.py build/decomp/lib_*.so -- 100008,9,16
set -e
cd "$(dirname "$0")/../hybrid_fem_lssvr_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../build/decomp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
OBJS=$(ls *.o | grep -v '^enhance_small_a.o$')
cp liblssvr_hip.so ../../build/decomp/lib_full.so
for v in empty:1 loads:2 nostore:3 noarith:4 directstore:5; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc $FL -DLSSVR_DECOMP=$d -c enhance_small_a.hip -o /tmp/esa_decomp_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/decomp/lib_$n.so $OBJS /tmp/esa_decomp_$n.o
done
ls -la ../../build/decomp
