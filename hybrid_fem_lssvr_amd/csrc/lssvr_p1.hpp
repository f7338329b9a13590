// Element-local P1 stiffness / load (Dual.py:117-128) as device code shared by the
// stand-alone assembly kernel (fem_eval.hip) and the fused step kernel (enhance_small.hip).
#pragma once
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

// ---------------------------------------------------------------------------
// Gauss-Legendre rules on [0,1] (abscissa xi, weight w; weights sum to 1)
// ---------------------------------------------------------------------------
inline bool quad_rule(int nq, QuadRule& q) {
  static const double X1[] = {0.5};
  static const double W1[] = {1.0};
  static const double X2[] = {0.21132486540518711775, 0.78867513459481288225};
  static const double W2[] = {0.5, 0.5};
  static const double X3[] = {0.11270166537925831148, 0.5, 0.88729833462074168852};
  static const double W3[] = {0.27777777777777777778, 0.44444444444444444444,
                              0.27777777777777777778};
  static const double X4[] = {0.069431844202973712388, 0.33000947820757186760,
                              0.66999052179242813240, 0.93056815579702628761};
  static const double W4[] = {0.17392742256872692869, 0.32607257743127307131,
                              0.32607257743127307131, 0.17392742256872692869};
  static const double X5[] = {0.046910077030668003601, 0.23076534494715845448, 0.5,
                              0.76923465505284154552, 0.95308992296933199640};
  static const double W5[] = {0.11846344252809454376, 0.23931433524968323402,
                              0.28444444444444444444, 0.23931433524968323402,
                              0.11846344252809454376};
  const double* X[] = {X1, X2, X3, X4, X5};
  const double* W[] = {W1, W2, W3, W4, W5};
  if (nq < 1 || nq > 5) return false;
  for (int i = 0; i < 5; ++i) {
    q.xi[i] = i < nq ? X[nq - 1][i] : 0.0;
    q.wt[i] = i < nq ? W[nq - 1][i] : 0.0;
  }
  return true;
}

// ---------------------------------------------------------------------------
// element-local P1 stiffness / load + gather-assembly of the tridiagonal bands
// ---------------------------------------------------------------------------
// One thread per NODE i: it evaluates the element to its right (i) and the
// element to its left (i-1) and sums their contributions, so the scatter of
// Dual.py:127-128 becomes a race-free gather (no atomics, bitwise reproducible).
struct ElemLocal {
  double k, fl, fr;
};

template <bool SIN>
__device__ __forceinline__ ElemLocal p1_element(const P1Args& p, const QuadRule& q, int64_t e) {
  const double a = p.x[e];
  const double h = p.x[e + 1] - a;
  double sl = 0.0, sr = 0.0, am = 0.0;
  for (int k = 0; k < p.nquad; ++k) {
    const double xi = q.xi[k];
    double f;
    if constexpr (SIN) {
      const double xq = a + h * xi;
      f = p.rhs_amp * sin_reduced(p.rhs_omega * xq);
    } else {
      f = p.rhs_quad[e * p.nquad + k];
    }
    sl += (q.wt[k] * (1.0 - xi)) * f;
    sr += (q.wt[k] * xi) * f;
    if (p.a_quad) am += q.wt[k] * p.a_quad[e * p.nquad + k];
  }
  ElemLocal r;
  r.k = (p.a_quad ? am : 1.0) / h;
  r.fl = h * sl;
  r.fr = h * sr;
  return r;
}

// The three bands of a small mesh (<= kWriteThroughMaxDoubles nodes) are stored write-through, like the lane
// kernels' coefficient tiles: in the fused step the assembly blocks are the grid's tail, so their dirty lines
// would all wait for the end-of-kernel release.
__device__ __forceinline__ void band_store(const P1Args& p, double* dst, double v) {
  if (p.ne <= kWriteThroughMaxDoubles) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *dst = v;
}

// One thread per NODE i (see above): both adjacent elements, race-free gather.
template <bool SIN>
__device__ __forceinline__ void p1_node(const P1Args& p, const QuadRule& q, int64_t i) {
  double d = 0.0, l = 0.0;
  if (i < p.ne) {
    const ElemLocal r = p1_element<SIN>(p, q, i);
    d += r.k;
    l += r.fl;
    band_store(p, &p.off[i], -r.k);
    if (p.kloc) p.kloc[i] = r.k;
    if (p.floc) {
      p.floc[2 * i] = r.fl;
      p.floc[2 * i + 1] = r.fr;
    }
  }
  if (i > 0) {
    const ElemLocal r = p1_element<SIN>(p, q, i - 1);
    d += r.k;
    l += r.fr;
  }
  band_store(p, &p.diag[i], d);
  band_store(p, &p.load[i], l);
}

}  // namespace lssvr
