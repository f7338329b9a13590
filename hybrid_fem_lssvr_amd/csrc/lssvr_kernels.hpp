// Internal launch interface between the C-ABI layer (capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/lssvr_hip.h"
#include "../../include/lssvr_hip_bench.h"   // (measurement entries of the same library)
#include "lssvr_device.hpp"

namespace lssvr {

constexpr int kSmallMaxM = 22;   // lane-per-element path: (M-2)(M-1)/2 Gram entries in VGPRs + AGPRs
// Coefficient tiles up to this many doubles (16 MiB) are stored WRITE-THROUGH (system-scope stores: sc0 sc1):
// an eagerly launched kernel ends with a write-back of its dirty lines before the next dispatch may start, and
// at BASELINE's 1e5 elements (7.2 MB of W) that tail was 0.6 of the kernel's 9.7 us; written through, the lines
// drain while other waves still compute.  Larger outputs keep non-temporal stores (measured: equal from 3e5 to
// 3e6 elements, write-through 2-4 % slower at 1e7).
constexpr int64_t kWriteThroughMaxDoubles = int64_t(1) << 21;
constexpr int kLargeMaxM = 33;   // wave-per-element MFMA path: M-2 bubble coefficients + rhs <= 32

struct EnhanceArgs {
  const double* x;
  const double* u;
  int64_t ne, elem_offset, ne_global;
  double gxmin, gxmax, bc_left, bc_right, gamma;
  double inv_gamma;         // 1 / gamma, rounded on the host (the scalar-gamma path divides nowhere)
  int M, n;
  int refine;               // lane kernel, Poisson rows: refinement steps of the near-square regime (0: none)
  int rhs_id;
  double rhs_amp, rhs_omega;
  const double* rhs_values;
  const double* a_values;   // non-null => variable-coefficient rows
  const double* da_values;
  // tabulated arrays (rhs_values, a_values, da_values): entry (element e of the launch, point k)
  // is t[e * tab_es + k * tab_ps] -- element-major (n, 1) or point-major (1, launch count)
  int64_t tab_es, tab_ps;
  // heterogeneous launches (lssvr_enhance_subset): local element k of the launch is mesh
  // element elem_ids[k] (NULL: k itself) -- node / nodal-value / gamma_values / status / W
  // rows are addressed by the MESH index, the tabulated arrays (rhs_values, a_values,
  // da_values) by k; gamma_values[mesh index] replaces gamma when non-NULL; W rows are ldw
  // doubles apart (0 = M)
  // An id outside [0, ne_mesh) touches NOTHING (no load, no store; status has no slot for
  // it): the element is skipped and counted in fail_count.
  const int64_t* elem_ids;
  int64_t ne_mesh;          // elements of the mesh the ids index (== ne when elem_ids is NULL)
  const double* gamma_values;
  int64_t ldw;
  double* W;
  int32_t* status;
  int32_t* fail_count;
  TrigTables trig;          // sin / cos polynomial coefficients (SGPR operands), set by capi.hip
};

// Optional per-launch profiling: when both events are set the kernel goes through
// hipExtLaunchKernelGGL, which stamps them with the dispatch's own begin/end times.
struct LaunchOpts {
  hipEvent_t start = nullptr;
  hipEvent_t stop = nullptr;
};

template <typename K, typename... Args>
inline hipError_t launch(K kernel, dim3 grid, dim3 block, hipStream_t s, const LaunchOpts* o,
                         Args... args) {
  if (o && o->start && o->stop)
    hipExtLaunchKernelGGL(kernel, grid, block, 0, s, o->start, o->stop, 0, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
  return hipGetLastError();
}

// XCDs (L2 domains) of the current device, hipDeviceAttributeNumberOfXccs (8 on MI355X); cached
// per device, 1 when the runtime cannot say.  Used only for L2-affine workgroup numbering.
inline unsigned xcd_count() {
  static thread_local int cached_dev = -1;
  static thread_local unsigned cached = 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || v < 1) v = 1;
    cached = (unsigned)v;
    cached_dev = dev;
  }
  return cached;
}

// compute units of the current device (256 on MI355X), cached like xcd_count(); sizes persistent grids
inline unsigned cu_count() {
  static thread_local int cached_dev = -1;
  static thread_local unsigned cached = 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 1;
    cached = (unsigned)v;
    cached_dev = dev;
  }
  return cached;
}

hipError_t enhance_small(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o = nullptr);
hipError_t enhance_large(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o = nullptr);
// Poisson rows, any M <= 33: Chebyshev-moment Gram (enhance_large_cheb.hip, enhance_large_parity.hip): a
// sequence of kernels with a workspace of enhance_moment_ws_bytes(ne, M, n) bytes in between
int enhance_refine_steps(int M, int n);
int enhance_small_refine_steps(int M, int n);   // the lane kernel's rule (M <= kSmallMaxM)
int64_t enhance_moment_ws_bytes(int64_t ne, int M, int n);
hipError_t enhance_large_split(const EnhanceArgs& a, void* work, hipStream_t s, const LaunchOpts* o = nullptr);
// the well-posed regime (n >= 2 (M-2)) of the two-kernel path: parity-split solve (enhance_large_parity.hip)
constexpr int kMomentWsStride = 96;   // workspace doubles per element: m_0..m_60, a, b, g_l, r_0..r_30, g_r
bool enhance_parity_applies(int M, int n);
hipError_t launch_solve4_parity(const EnhanceArgs& a, const double* ws, hipStream_t s, hipEvent_t ev_stop);
hipError_t enhance_dual(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o = nullptr);
constexpr int kSharedMaxM = 33;  // shared-operator path (uniform meshes): coefficients in VGPRs
hipError_t enhance_shared(const EnhanceArgs& a, const double* op, hipStream_t s,
                          const LaunchOpts* o = nullptr);

hipError_t colloc_points(const double* x, int64_t ne, int n, double* xc, hipStream_t s, bool point_major = false);

struct QuadRule {
  double xi[5];
  double wt[5];
};

struct P1Args {
  const double* x;
  int64_t ne;
  int nquad;
  int rhs_id;
  double rhs_amp, rhs_omega;
  const double* rhs_quad;
  const double* a_quad;
  double* diag;
  double* off;
  double* load;
  double* kloc;
  double* floc;
};
hipError_t p1_assemble(const P1Args& a, hipStream_t s);
// assembly + enhancement of the same mesh in ONE launch (lane-per-element path, in-kernel rhs)
hipError_t step_small(const EnhanceArgs& e, const P1Args& a, hipStream_t s,
                      const LaunchOpts* o = nullptr);
constexpr int kStepVarcoefFusedMaxM = 12;   // lssvr_step_varcoef: one launch up to here, two above
hipError_t step_small_vc(const EnhanceArgs& e, const P1Args& a, hipStream_t s, const LaunchOpts* o = nullptr);
hipError_t quad_points(const double* x, int64_t ne, int nquad, double* xq, hipStream_t s);

int64_t tridiag_work_bytes(int64_t ne);
hipError_t tridiag_dirichlet_solve(const double* diag, const double* off, const double* load,
                                   int64_t ne, double u0, double u1, double* u, void* work,
                                   hipStream_t s);

int64_t flux_work_bytes(int64_t ne);
hipError_t flux_dirichlet_solve(const double* kloc, const double* load, int64_t ne, double u0,
                                double u1, double* u, void* work, hipStream_t s);

hipError_t eval_points(const double* x, const double* W, int64_t ne, int M, const double* xq,
                       int64_t P, double* uq, int64_t* elem, hipStream_t s);

hipError_t flux_aggregate(const double* kloc, const double* load, int64_t ne, bool first_global,
                          void* work, double* agg3, hipStream_t s);
hipError_t flux_finish(const double* kloc, const double* load, int64_t ne, bool first_global,
                       bool last_global, const void* work, const double* prefix3,
                       const double* grand3, double u0, double u1, double* u, hipStream_t s);

hipError_t eval_error(const double* x, const double* W, int64_t ne, int M, const double* xq,
                      int64_t P, double amp, double omega, double* out, hipStream_t s);

hipError_t fp64_probe(double* out, int blocks, int iters, int use_mfma, hipStream_t s);
hipError_t stream_probe(const double* src, double* dst, int64_t n, hipStream_t s);
hipError_t row_chunk_probe(const double* src, double* dst, int64_t nrows, int rowlen, int chunk, hipStream_t s);

}  // namespace lssvr
