"""CPU: the lane kernels' cold-path solvers (cheb_slow_build, cheb_ridge_solve: run-time loops on a per-lane scratch
buffer of ChebSlow<M>::kSize doubles) compiled for the HOST from the shipped header text and run under
AddressSanitizer + UBSan on a heap buffer of exactly kSize doubles, M = 3 .. 22.  (GPU sanitizers are not available on
the pool; round 4 met a memory fault in the M = 4 instantiation on the GPU -- DESIGN.md section 2c -- and this is the
check that its indexing is clean at source level.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybrid_fem_lssvr_amd", "csrc")

HARNESS = r'''#include <cmath>
#include <cstdio>
#include <cstdlib>
#define __device__
#define __host__
#define __forceinline__ inline
#include "%(tables)s"
namespace lssvr {
constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
inline double rcp_newton(double x) { return 1.0 / x; }
using std::fma;
using std::fabs;
%(body)s
template <int M>
void run() {
  using L = ChebSlow<M>;
  constexpr int MR = M - 2;
  double* buf = (double*)malloc(sizeof(double) * L::kSize);      // exactly kSize: any other index is reported
  for (int rep = 0; rep < 3; ++rep) {
    for (int d = 0; d < 2 * MR - 1; ++d) buf[L::kMom + d] = (d == 0) ? 16.0 : ((d & 1) ? 1e-13 * d : 3.0 - 8.0 / (d * d - 1.0));
    for (int i = 0; i < MR; ++i) buf[L::kRhs + i] = 0.1 * (i + 1);
    const double ta = -1.0 + 1e-16 * rep, tb = 1.0;
    const bool ok = cheb_ridge_solve<M>(buf, ta, tb, 0.3, -0.2, 1e8);
    for (int i = 0; i < MR; ++i) buf[L::kRhs + i] = 0.1 * (i + 1);
    cheb_slow_build<M>(buf, ta, tb, 0.3, -0.2, 1e-3);
    if (rep == 0) std::printf("M=%%d kSize=%%d ok=%%d\n", M, L::kSize, (int)ok);
  }
  free(buf);
}
}  // namespace lssvr
int main() {
  using namespace lssvr;
  run<3>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<9>(); run<10>(); run<11>(); run<12>();
  run<13>(); run<14>(); run<15>(); run<16>(); run<17>(); run<18>(); run<19>(); run<20>(); run<21>(); run<22>();
  return 0;
}
'''


def test_cold_path_solvers_index_inside_their_scratch(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    src = open(os.path.join(CSRC, "enhance_small_cheb.hpp")).read()
    body = src[src.index("template <int M>\nstruct ChebSlow {"):src.index("// REFINE: the build with the near-square refinement loop")]
    assert "cheb_ridge_solve" in body and "cheb_slow_build" in body
    cpp = tmp_path / "cold_path_host.cpp"
    cpp.write_text(HARNESS % {"tables": os.path.join(CSRC, "cheb_tables.hpp"), "body": body})
    exe = tmp_path / "cold_path_host"
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        str(cpp), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("M=")]
    assert len(lines) == 20 and all(ln.endswith("ok=1") for ln in lines), r.stdout
