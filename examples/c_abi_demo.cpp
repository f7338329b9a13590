// Stand-alone host program against the C ABI only (include/lssvr_hip.h): no Python, no torch.
// It reproduces the reference's demo (Dual.py:206-217: 25 nodes on [-1,1], M = 8, gamma = 1e4,
// 12 collocation points, 201 test points) with device buffers from hipMalloc:
//   lssvr_p1_assemble -> lssvr_tridiag_dirichlet_solve -> lssvr_enhance -> lssvr_eval
// and prints the relative L2 error against sin(pi x) (3.255e-06).
//
// build:  hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp \
//               -L hybrid_fem_lssvr_amd/csrc -llssvr_hip -Wl,-rpath,$PWD/hybrid_fem_lssvr_amd/csrc \
//               -o examples/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>
#include "lssvr_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define LSSVR_OK_OR_DIE(x)                                                         \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ < 0) {                                                                 \
      std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, lssvr_last_error());         \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

template <typename T>
static T* dev_alloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) std::exit(4);
  return static_cast<T*>(p);
}

int main() {
  if (lssvr_version() != LSSVR_ABI_VERSION) return 1;
  const int64_t ne = 24;
  const int M = 8, n_colloc = 12, P = 201;
  const double gamma = 1e4, lo = -1.0, hi = 1.0, pi = 3.14159265358979323846;
  const double rhs[2] = {pi * pi, pi};                       // poisson_rhs, Dual.py:11-12

  std::vector<double> nodes(ne + 1), xq(P);
  for (int64_t i = 0; i <= ne; ++i) nodes[i] = i * ((hi - lo) / ne) + lo;   // np.linspace
  nodes[ne] = hi;
  for (int i = 0; i < P; ++i) xq[i] = i * ((hi - lo) / (P - 1)) + lo;
  xq[P - 1] = hi;

  double* x = dev_alloc<double>(ne + 1);
  double* u = dev_alloc<double>(ne + 1);
  double* diag = dev_alloc<double>(ne + 1);
  double* off = dev_alloc<double>(ne);
  double* load = dev_alloc<double>(ne + 1);
  double* W = dev_alloc<double>(ne * M);
  int32_t* status = dev_alloc<int32_t>(ne);
  double* dxq = dev_alloc<double>(P);
  double* duq = dev_alloc<double>(P);
  int64_t* elem = dev_alloc<int64_t>(P);
  void* work = nullptr;
  HIP_OK(hipMalloc(&work, (size_t)lssvr_tridiag_work_bytes(ne)));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  HIP_OK(hipMemcpyAsync(x, nodes.data(), (ne + 1) * sizeof(double), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dxq, xq.data(), P * sizeof(double), hipMemcpyHostToDevice, stream));

  // solve_fem (Dual.py:110-137)
  LSSVR_OK_OR_DIE(lssvr_p1_assemble(x, ne, 2, LSSVR_RHS_SIN, rhs, nullptr, nullptr, diag, off, load,
                                    nullptr, nullptr, stream));
  LSSVR_OK_OR_DIE(lssvr_tridiag_dirichlet_solve(diag, off, load, ne, 0.0, 0.0, u, work, stream));
  // solve_lssvr_subproblems (Dual.py:139-169)
  LSSVR_OK_OR_DIE(lssvr_enhance(x, u, ne, 0, ne, lo, hi, 0.0, 0.0, M, n_colloc, gamma, LSSVR_RHS_SIN,
                                rhs, nullptr, LSSVR_SOLVER_PRIMAL, W, status, nullptr, stream));
  // evaluate_solution (Dual.py:176-203)
  LSSVR_OK_OR_DIE(lssvr_eval(x, W, ne, M, dxq, P, duq, elem, stream));

  std::vector<double> uq(P), uh(ne + 1);
  std::vector<int64_t> el(P);
  std::vector<int32_t> st(ne);
  HIP_OK(hipMemcpyAsync(uq.data(), duq, P * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(el.data(), elem, P * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(st.data(), status, ne * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(uh.data(), u, (ne + 1) * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  // the same enhancement as a BOUND step (lssvr_step_plan_*: assembly + enhancement in one launch, arguments
  // validated once): W2 must equal W bit for bit
  double* W2 = dev_alloc<double>(ne * M);
  lssvr_step_plan* plan = nullptr;
  LSSVR_OK_OR_DIE(lssvr_step_plan_create(&plan, x, u, ne, 0, ne, lo, hi, 0.0, 0.0, M, n_colloc, gamma, rhs, 2, diag,
                                         off, load, W2, status, nullptr));
  for (int rep = 0; rep < 3; ++rep) LSSVR_OK_OR_DIE(lssvr_step_plan_launch(plan, stream));
  std::vector<double> w1(ne * M), w2(ne * M);
  HIP_OK(hipMemcpyAsync(w1.data(), W, ne * M * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(w2.data(), W2, ne * M * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  LSSVR_OK_OR_DIE(lssvr_step_plan_destroy(plan));
  const bool plan_equal = std::memcmp(w1.data(), w2.data(), ne * M * sizeof(double)) == 0;

  double num = 0, den = 0, nodal = 0;
  for (int i = 0; i < P; ++i) {
    const double ex = std::sin(pi * xq[i]);
    num += (uq[i] - ex) * (uq[i] - ex);
    den += ex * ex;
  }
  for (int64_t i = 0; i <= ne; ++i) nodal = std::fmax(nodal, std::fabs(uh[i] - std::sin(pi * nodes[i])));
  int fallback = 0;
  for (int64_t i = 0; i < ne; ++i) fallback += st[i] != 0;
  const double rel = std::sqrt(num / den);
  std::printf("c_abi_demo: rel-L2 vs sin(pi x) = %.4e (expect 3.255e-06), max nodal error = %.4e "
              "(expect 3.274e-06), fallback elements = %d, element of x=0: %lld (expect 11)\n",
              rel, nodal, fallback, (long long)el[100]);
  std::printf("c_abi_demo: bound step (lssvr_step_plan_launch x 3) %s lssvr_enhance\n", plan_equal ? "bit-equal to" : "DIFFERS from");
  const bool ok = std::fabs(rel - 3.255e-6) < 5e-9 && std::fabs(nodal - 3.274e-6) < 2e-9 &&
                  fallback == 0 && el[100] == 11 && el[0] == 0 && el[P - 1] == ne - 1 && plan_equal;
  std::printf(ok ? "OK\n" : "MISMATCH\n");
  return ok ? 0 : 5;
}
