// Per-element LSSVR enhancement, small-degree path (2 <= M <= 14): ONE ELEMENT PER
// LANE, the whole (M-2)x(M-2) system register-resident.
//
// What one lane computes (DESIGN.md "per-element solve"; oracle restatement:
// oracle/lssvr_oracle.py::solve_bc_eliminated; reference: Dual.py:20-98):
//   1. element data a=x[e], b=x[e+1], g_l, g_r (Dual.py:143-151, 65-75);
//   2. numpy's domain map and collocation abscissae, two-rounding arithmetic
//      (Dual.py:40,56 -> linspace / mapparms / mapdomain);
//   3. for every collocation point: f(x_k), the row rho_j = L''_{j+2}(t_k)
//      (Gegenbauer recurrence) and the rank-1 updates G += rho rho^T,
//      r += rho * phi -- the Legendre Gram contraction over collocation points;
//   4. the two boundary rows L_p(t_a), L_p(t_b), eliminated analytically:
//      w_{0,1} = d - C v;
//   5. S = G + eps (I + C^T C), Jacobi scaling, Cholesky, two triangular solves;
//   6. status / linear-interpolant fallback (Dual.py:164-169).
// The 256 x M coefficient tile of a workgroup is transposed through LDS so that
// the store to W[ne, M] (row-major, 8*M B per element) is fully coalesced.
//
// Why lane-per-element rather than a wave-cooperative factorisation: the system
// is 7x7 at degree 8; 28 Gram entries + 7 rhs fit in ~90 VGPRs, every operation
// is lane-local (no cross-lane traffic, no LDS in the loop, no divergence), and
// all 64 lanes do useful FP64 work.  Measured numbers: DESIGN.md.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

template <int M, int RHS, bool VC>
__global__ __launch_bounds__(kBlock) void enhance_small_kernel(EnhanceArgs p) {
  constexpr int MR = M - 2;
  constexpr int NT = MR * (MR + 1) / 2;
  __shared__ double tile[kBlock * M];

  const int tid = threadIdx.x;
  const int64_t e = (int64_t)blockIdx.x * kBlock + tid;
  double w[M];
  int st = LSSVR_ST_OK;
#pragma unroll
  for (int i = 0; i < M; ++i) w[i] = 0.0;

  if (e < p.ne) {
    const double a = p.x[e];
    const double b = p.x[e + 1];
    const int64_t eg = e + p.elem_offset;
    // Dual.py:65-75: Dirichlet value only on a global-boundary element whose end
    // point equals the global end point exactly
    const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[e];
    const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[e + 1];

    const DomainMap dm = map_params(a, b);
    const int n = p.n;
    const double step = dm.oldlen / (double)(n - 1);
    const double scl2 = dm.scl * dm.scl;
    const double inv_scl2 = rcp_newton(scl2);
    const double eps = rcp_newton(p.gamma * (scl2 * scl2));   // 1 / (gamma * scl^4)

    // --- boundary rows (Dual.py:61-76): B = [L_p(t_a); L_p(t_b)], eliminated as
    // w_{0,1} = d - C v with B1 = [[1, ta], [1, tb]], B1^{-1} = [[tb, -ta], [-1, 1]]/(tb - ta).
    // The variable-coefficient rows need C inside the Gram loop; the Poisson rows do
    // not, so there the block runs after the loop (64 fewer live VGPRs in the loop).
    double d0 = 0.0, d1 = 0.0;
    double C0[MR > 0 ? MR : 1], C1[MR > 0 ? MR : 1];
    auto boundary_rows = [&]() {
      const double ta = dm.off + dm.scl * a;
      const double tb = dm.off + dm.scl * b;
      double La[M], Lb[M];
      legendre_p<M>(ta, La);
      legendre_p<M>(tb, Lb);
      const double idet = rcp_newton(tb - ta);
      d0 = (tb * gl - ta * gr) * idet;
      d1 = (gr - gl) * idet;
#pragma unroll
      for (int j = 0; j < MR; ++j) {
        C0[j] = (tb * La[j + 2] - ta * Lb[j + 2]) * idet;
        C1[j] = (Lb[j + 2] - La[j + 2]) * idet;
      }
    };
    if constexpr (VC || MR == 0) boundary_rows();

    if constexpr (MR == 0) {
      w[0] = d0;
      w[1] = d1;
      if (!(isfinite(d0) && isfinite(d1))) st = LSSVR_ST_FALLBACK;
    } else {
      // --- Gram contraction over the collocation points ------------------------
      double G[NT], rv[MR];
#pragma unroll
      for (int i = 0; i < NT; ++i) G[i] = 0.0;
#pragma unroll
      for (int i = 0; i < MR; ++i) rv[i] = 0.0;

      for (int k = 0; k < n; ++k) {
        const double xk = linspace_at(a, b, dm.oldlen, step, k, n);
        const double tk = dm.off + dm.scl * xk;
        double fk;
        if constexpr (RHS == LSSVR_RHS_SIN) {
          fk = p.rhs_amp * sin_reduced(p.rhs_omega * xk);
        } else {
          fk = p.rhs_values[e * n + k];
        }
        double rho[MR];
        legendre_d2<MR>(tk, rho);
        double phi = -(fk * inv_scl2);
        if constexpr (VC) {
          const double ak = p.a_values[e * n + k];
          const double bk = p.da_values[e * n + k] / dm.scl;
          double r1[MR + 1];
          legendre_d1<MR + 1>(tk, r1);        // r1[m] = L'_{m+1}; need L'_{j+2} = r1[j+1]
#pragma unroll
          for (int j = 0; j < MR; ++j) rho[j] = fma(ak, rho[j], bk * (r1[j + 1] - C1[j]));
          phi = -fma(bk, d1, fk * inv_scl2);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) G[tri(i, j)] = fma(rho[i], rho[j], G[tri(i, j)]);
          rv[i] = fma(rho[i], phi, rv[i]);
        }
      }

      if constexpr (!VC) boundary_rows();
      // --- S = G + eps (I + C^T C),  rhs = r + eps C^T d --------------------------
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          double cc = fma(C0[i], C0[j], C1[i] * C1[j]);
          if (i == j) cc += 1.0;
          G[tri(i, j)] = fma(eps, cc, G[tri(i, j)]);
        }
        rv[i] = fma(eps, fma(C0[i], d0, C1[i] * d1), rv[i]);
      }

      // --- Jacobi scaling + Cholesky (lower, in place; diagonal holds 1/L_jj) ----
      bool ok = true;
      double dj[MR];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        ok = ok && (G[tri(i, i)] > 0.0);
        dj[i] = rsqrt_newton(G[tri(i, i)]);
      }
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) G[tri(i, j)] *= dj[i] * dj[j];
        rv[i] *= dj[i];
      }
#pragma unroll
      for (int j = 0; j < MR; ++j) {
        const double piv = G[tri(j, j)];
        ok = ok && (piv > 0.0) && (piv < 1.0e300);
        const double linv = rsqrt_newton(piv);
        G[tri(j, j)] = linv;
#pragma unroll
        for (int i = j + 1; i < MR; ++i) G[tri(i, j)] *= linv;
#pragma unroll
        for (int c = j + 1; c < MR; ++c) {
#pragma unroll
          for (int i = c; i < MR; ++i)
            G[tri(i, c)] = fma(-G[tri(i, j)], G[tri(c, j)], G[tri(i, c)]);
        }
      }
      // forward  L y = rhs
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        double s = rv[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s = fma(-G[tri(i, j)], rv[j], s);
        rv[i] = s * G[tri(i, i)];
      }
      // backward L^T z = y
#pragma unroll
      for (int i = MR - 1; i >= 0; --i) {
        double s = rv[i];
#pragma unroll
        for (int j = i + 1; j < MR; ++j) s = fma(-G[tri(j, i)], rv[j], s);
        rv[i] = s * G[tri(i, i)];
      }
      // v = D z,  w_{0,1} = d - C v
      double w0 = d0, w1 = d1;
#pragma unroll
      for (int j = 0; j < MR; ++j) {
        const double v = rv[j] * dj[j];
        w[j + 2] = v;
        w0 = fma(-C0[j], v, w0);
        w1 = fma(-C1[j], v, w1);
        ok = ok && isfinite(v);
      }
      w[0] = w0;
      w[1] = w1;
      ok = ok && isfinite(w0) && isfinite(w1);
      if (!ok) st = LSSVR_ST_FALLBACK;
    }

    if (st != LSSVR_ST_OK) {
      // Dual.py:164-169: linear interpolant of (g_l, g_r) as a Legendre series
#pragma unroll
      for (int i = 0; i < M; ++i) w[i] = 0.0;
      w[0] = 0.5 * (gl + gr);
      w[1] = 0.5 * (gr - gl);
      if (p.fail_count) atomicAdd(p.fail_count, 1);
    }
    if (p.status) p.status[e] = st;
  }

  // --- coalesced store of the workgroup's 256 x M tile -------------------------
#pragma unroll
  for (int i = 0; i < M; ++i) tile[tid * M + i] = w[i];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kBlock * M;
  const int64_t total = p.ne * M;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int64_t idx = base + (int64_t)i * kBlock + tid;
    if (idx < total) p.W[idx] = tile[i * kBlock + tid];
  }
}

// ----------------------------------------------------------------------------
// dispatch
// ----------------------------------------------------------------------------
template <int M, int RHS, bool VC>
static hipError_t launch_small(const EnhanceArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)((a.ne + kBlock - 1) / kBlock);
  hipLaunchKernelGGL((enhance_small_kernel<M, RHS, VC>), dim3(blocks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

template <int RHS, bool VC>
static hipError_t dispatch_m(const EnhanceArgs& a, hipStream_t s) {
  switch (a.M) {
#define LSSVR_CASE(MM) \
  case MM:             \
    return launch_small<MM, RHS, VC>(a, s);
    LSSVR_CASE(2)
    LSSVR_CASE(3)
    LSSVR_CASE(4)
    LSSVR_CASE(5)
    LSSVR_CASE(6)
    LSSVR_CASE(7)
    LSSVR_CASE(8)
    LSSVR_CASE(9)
    LSSVR_CASE(10)
    LSSVR_CASE(11)
    LSSVR_CASE(12)
    LSSVR_CASE(13)
    LSSVR_CASE(14)
#undef LSSVR_CASE
    default:
      return hipErrorInvalidValue;
  }
}

hipError_t enhance_small(const EnhanceArgs& a, hipStream_t s) {
  if (a.a_values) return dispatch_m<LSSVR_RHS_ARRAY, true>(a, s);
  if (a.rhs_id == LSSVR_RHS_SIN) return dispatch_m<LSSVR_RHS_SIN, false>(a, s);
  return dispatch_m<LSSVR_RHS_ARRAY, false>(a, s);
}

}  // namespace lssvr
