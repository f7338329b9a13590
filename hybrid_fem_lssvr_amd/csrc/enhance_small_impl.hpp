// Per-element LSSVR enhancement, lane-per-element path (2 <= M <= 22): ONE ELEMENT PER
// LANE, the whole (M-2)x(M-2) system register-resident (VGPRs up to M = 14, VGPRs + AGPRs
// at one wave per SIMD up to M = 22 -- still 4x (M = 20) to 1.7x (M = 22) faster than the
// wave-cooperative mapping of enhance_large.hip; the crossover is at M = 23).
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
//   5. S = G + eps (I + C^T C), LDL^T, two triangular solves;
//   6. status / linear-interpolant fallback (Dual.py:164-169).
// Each wave transposes its 64 x M coefficient tile through (wave-private) LDS so that
// the store to W[ne, M] (row-major, 8*M B per element) is fully coalesced.
//
// Why lane-per-element rather than a wave-cooperative factorisation: the system
// is 7x7 at degree 8; 28 Gram entries + 7 rhs fit in ~90 VGPRs, every operation
// is lane-local (no cross-lane traffic, no LDS in the loop, no divergence), and
// all 64 lanes do useful FP64 work.  Measured numbers: DESIGN.md.
#pragma once
#include "enhance_small_cheb.hpp"
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_p1.hpp"

namespace lssvr {

// LDS per wave: the 64 x M output tile, or (tabulated inputs) kStageK points of 64 rows per
// array at pitch kStageK + 1 -- whichever is larger; the two uses never overlap in time.
constexpr int kStageK = 8;
constexpr int kStageArr = 64 * (kStageK + 1);
template <int M, int RHS, bool VC>
constexpr int kSmallTilePerWave =
    (RHS == LSSVR_RHS_ARRAY && (VC ? 3 : 1) * kStageArr > 64 * M) ? (VC ? 3 : 1) * kStageArr : 64 * M;
// (RHS == LSSVR_RHS_ARRAY_PM: point-major tables, read directly -- no staging area)
// point-major tables: points fetched ahead of their use (measured at config 5, same box: 2 -> 102-104 us,
// 4 -> 101-109 us, 6 and 8 -> 149 us (256 registers); plain instead of non-temporal loads: no difference)
constexpr int kPrefetch = 4;

template <int M, int RHS, bool VC>
__device__ __forceinline__ void enhance_small_body(const EnhanceArgs& p, const unsigned block,
                                                   double* __restrict__ tile) {
  constexpr int MR = M - 2;
  constexpr int NT = MR * (MR + 1) / 2;

  const int tid = threadIdx.x;
  const int64_t e = (int64_t)block * kBlock + tid;
  double w[M];
  int st = LSSVR_ST_OK;
#pragma unroll
  for (int i = 0; i < M; ++i) w[i] = 0.0;

  const bool scattered = p.elem_ids != nullptr || (p.ldw != 0 && p.ldw != M);
  // Every lane runs the body (lanes past the end of the last wave on a duplicate of the last
  // element, their stores masked): the tabulated inputs are loaded cooperatively by the wave.
  bool live = e < p.ne;
  const int64_t ec = live ? e : p.ne - 1;                   // position in this launch
  const int lane = tid & 63;
  {
    int64_t id = ec;                                        // mesh index of this element
    if (p.elem_ids) {
      id = p.elem_ids[ec];
      if (id < 0 || id >= p.ne_mesh) {       // out-of-range id: nothing of the mesh is touched
        if (live && p.fail_count) atomicAdd(p.fail_count, 1);
        live = false;
        id = 0;
      }
    }
    const double a = p.x[id];
    const double b = p.x[id + 1];
    const int64_t eg = id + p.elem_offset;
    // Dual.py:65-75: Dirichlet value only on a global-boundary element whose end
    // point equals the global end point exactly
    const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[id];
    const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[id + 1];
    const double gamma = p.gamma_values ? p.gamma_values[id] : p.gamma;

    const DomainMap dm = map_params(a, b);
    const int n = p.n;
    const double step = dm.oldlen / (double)(n - 1);
    const double scl2 = dm.scl * dm.scl;
    const double inv_scl2 = rcp_newton(scl2);
    const double eps = rcp_newton(gamma * (scl2 * scl2));   // 1 / (gamma * scl^4)

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
        // Poisson rows work in the rescaled unknowns v' = D v (see legendre_d2_scaled):
        // C' = C D^-1, so w_{0,1} = d - C v = d - C' v'
        const double sc = VC ? idet : idet * (1.0 / d2_scale(j));
        C0[j] = (tb * La[j + 2] - ta * Lb[j + 2]) * sc;
        C1[j] = (Lb[j + 2] - La[j + 2]) * sc;
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

      // In-kernel rhs f = amp sin(fl(omega x_k)) without a full sin per point: the angles
      // advance by ~omega*step, so (s~, c~) = (sin, cos)(th0 + k dth) is carried by a rotation
      // (seeds: one sincos of th0 = fl(omega a) and one of dth = fl(omega step)), and the exact
      // argument numpy would use, arg_k = fl(omega x_k), is restored to first order:
      //   sin(arg_k) = s~ + c~ delta,  delta = (arg_k - th0) - k dth   (|delta| ~ |omega x| eps).
      // delta^2/2 and the rotation's rounding (<= ~n eps) are far below the 1e-13 parity bar;
      // a wave with any |delta| > 1e-7 (|omega x| > ~1e8) takes the per-point sin instead.
      // The rotation is linear, so the pair is carried pre-multiplied by kappa = -amp / scl^2:
      // phi_k = -f(x_k) / scl^2 comes out of the correction FMA directly.
      double rs = 0.0, rc = 1.0, sd = 0.0, cd = 1.0, th0 = 0.0, dth = 0.0, kappa = 0.0;
      if constexpr (RHS == LSSVR_RHS_SIN) {
        th0 = p.rhs_omega * a;
        dth = p.rhs_omega * step;
        sincos_tab(th0, rs, rc, p.trig);
        sincos_tab(dth, sd, cd, p.trig);
        kappa = -(p.rhs_amp * inv_scl2);
        rs *= kappa;
        rc *= kappa;
      }
      // Tabulated inputs (rhs_values, a_values, da_values: [element][point], a row per element):
      // a lane walking its own row touches 64 different cache lines per load instruction (6 %
      // of each used) and thrashes the L1 -- measured 9x slower than the in-kernel rhs.  The
      // wave loads kStageK points of its 64 rows at a time with consecutive lanes on consecutive
      // doubles (full 64-byte runs of every row) into LDS, row pitch kStageK + 1 (conflict-free
      // when every lane then reads its own row).
      [[maybe_unused]] double* const stg = tile + (tid >> 6) * kSmallTilePerWave<M, RHS, VC>;
      [[maybe_unused]] const int64_t e0 = (int64_t)block * kBlock + (tid & ~63);
      // one collocation point k with its tabulated values (fk, and for variable coefficients a_k, a'_k)
      auto point = [&](const int k, const double fk_tab, const double ak, const double dak) {
        const double xk = linspace_at(a, b, dm.oldlen, step, k, n);
        const double tk = dm.off + dm.scl * xk;
        double fk = 0.0, phi;
        if constexpr (RHS == LSSVR_RHS_SIN) {
          const double arg = p.rhs_omega * xk;
          const double delta = fma(-(double)k, dth, arg - th0);
          phi = fma(rc, delta, rs);
          if (__any(!(fabs(delta) < 1.0e-7))) phi = kappa * sin_tab(arg, p.trig);
          const double rs_next = fma(rs, cd, rc * sd);
          rc = fma(rc, cd, -(rs * sd));
          rs = rs_next;
        } else {
          fk = fk_tab;
          phi = -(fk * inv_scl2);
        }
        double rho[MR];
        if constexpr (VC) legendre_d2<MR>(tk, rho);
        else legendre_d2_scaled<MR>(tk, rho);
        if constexpr (VC) {
          // a'/scl as a'*(h/2): within an ulp of the division, 11 FP64 instructions fewer per point
          const double bk = dak * (0.5 * dm.oldlen);
          double r1[MR + 1];
          legendre_d1<MR + 1>(tk, r1);        // r1[m] = L'_{m+1}; need L'_{j+2} = r1[j+1]
#pragma unroll
          for (int j = 0; j < MR; ++j) rho[j] = fma(ak, rho[j], bk * (r1[j + 1] - C1[j]));
          phi = -fma(bk, d1, fk * inv_scl2);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            if (!VC && i == 0) continue;       // rho_0 = 3 for every point: G_00 = 9 n, set below
            G[tri(i, j)] = fma(rho[i], rho[j], G[tri(i, j)]);
          }
          rv[i] = fma(rho[i], phi, rv[i]);
        }
      };
      if constexpr (RHS == LSSVR_RHS_ARRAY_PM) {
        // POINT-MAJOR tables t[k * ne + e]: consecutive lanes read consecutive doubles (full 512-byte
        // runs per wave and instruction), no staging, no LDS; the values of the next kPrefetch points
        // are requested before the current kPrefetch are worked on (a point is ~110 instructions:
        // four of them cover the latency of HBM with two resident waves per SIMD).  Past the last
        // point the index is clamped (in bounds, value unused).
        const int64_t ps = p.tab_ps;
        const double* const tf = p.rhs_values + ec * p.tab_es;
        [[maybe_unused]] const double* const ta = VC ? p.a_values + ec * p.tab_es : nullptr;
        [[maybe_unused]] const double* const td = VC ? p.da_values + ec * p.tab_es : nullptr;
        double cf[kPrefetch], ca[kPrefetch], cdv[kPrefetch];
        auto fetch = [&](const int k, double& f_, double& a_, double& d_) {
          const int64_t g = (int64_t)min(k, n - 1) * ps;
          f_ = __builtin_nontemporal_load(tf + g);
          if constexpr (VC) {
            a_ = __builtin_nontemporal_load(ta + g);
            d_ = __builtin_nontemporal_load(td + g);
          } else {
            a_ = d_ = 0.0;
          }
        };
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) fetch(i, cf[i], ca[i], cdv[i]);
        for (int k = 0; k < n; k += kPrefetch) {
          double nf[kPrefetch], na[kPrefetch], nd[kPrefetch];
#pragma unroll
          for (int i = 0; i < kPrefetch; ++i) fetch(k + kPrefetch + i, nf[i], na[i], nd[i]);
#pragma unroll
          for (int i = 0; i < kPrefetch; ++i)
            if (k + i < n) point(k + i, cf[i], ca[i], cdv[i]);
#pragma unroll
          for (int i = 0; i < kPrefetch; ++i) {
            cf[i] = nf[i];
            ca[i] = na[i];
            cdv[i] = nd[i];
          }
        }
      } else {
        for (int k = 0; k < n; ++k) {
          if constexpr (RHS == LSSVR_RHS_ARRAY) {
            if ((k & (kStageK - 1)) == 0) {
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int i = 0; i < kStageK; ++i) {
                const int idx = i * 64 + lane;
                const int row = idx / kStageK, kk = idx % kStageK;
                const int64_t er = e0 + row;
                const bool in = (er < p.ne) && (k + kk < n);
                const int64_t g = in ? er * n + (k + kk) : 0;
                stg[row * (kStageK + 1) + kk] = in ? p.rhs_values[g] : 0.0;
                if constexpr (VC) {
                  stg[kStageArr + row * (kStageK + 1) + kk] = in ? p.a_values[g] : 0.0;
                  stg[2 * kStageArr + row * (kStageK + 1) + kk] = in ? p.da_values[g] : 0.0;
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
          double fk = 0.0, ak = 0.0, dak = 0.0;
          if constexpr (RHS == LSSVR_RHS_ARRAY) {
            fk = stg[lane * (kStageK + 1) + (k & (kStageK - 1))];
            if constexpr (VC) {
              ak = stg[kStageArr + lane * (kStageK + 1) + (k & (kStageK - 1))];
              dak = stg[2 * kStageArr + lane * (kStageK + 1) + (k & (kStageK - 1))];
            }
          }
          point(k, fk, ak, dak);
        }
      }
      if constexpr (!VC) G[0] = 9.0 * (double)n;

      if constexpr (!VC) boundary_rows();
      // --- S = G + eps (I + C^T C),  rhs = r + eps C^T d --------------------------
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          double cc = fma(C0[i], C0[j], C1[i] * C1[j]);
          if (i == j) cc += VC ? 1.0 : 1.0 / (d2_scale(i) * d2_scale(i));   // eps D^-2 on the diagonal
          G[tri(i, j)] = fma(eps, cc, G[tri(i, j)]);
        }
        rv[i] = fma(eps, fma(C0[i], d0, C1[i] * d1), rv[i]);
      }

      // --- LDL^T (lower, in place; unit L below the diagonal, diagonal holds 1/d_j).  No
      // square roots, no diagonal pre-scaling: elimination of an SPD matrix is invariant
      // under symmetric diagonal scaling up to rounding (measured: same <=2e-16 distance to
      // the 60-digit minimiser).  A zero / non-finite pivot turns into inf / NaN
      // in 1/d_j and reaches every later entry; a negative one is caught by the sign test.
      bool ok = true;
#pragma unroll
      for (int j = 0; j < MR; ++j) {
        ok = ok && (G[tri(j, j)] > 0.0);
        const double rinv = rcp_newton(G[tri(j, j)]);
        G[tri(j, j)] = rinv;
#pragma unroll
        for (int c = j + 1; c < MR; ++c) {
          const double lcj = G[tri(c, j)] * rinv;              // L_cj = a_cj / d_j
#pragma unroll
          for (int i = c; i < MR; ++i)
            G[tri(i, c)] = fma(-G[tri(i, j)], lcj, G[tri(i, c)]);
          G[tri(c, j)] = lcj;     // column j below the diagonal now holds L (rows <= c are done)
        }
      }
      // forward  L y = rhs
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        double s = rv[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s = fma(-G[tri(i, j)], rv[j], s);
        rv[i] = s;
      }
      // backward L^T z = D^-1 y
#pragma unroll
      for (int i = MR - 1; i >= 0; --i) {
        double s = rv[i] * G[tri(i, i)];
#pragma unroll
        for (int j = i + 1; j < MR; ++j) s = fma(-G[tri(j, i)], rv[j], s);
        rv[i] = s;
      }
      // w_{0,1} = d - C v
      double w0 = d0, w1 = d1;
#pragma unroll
      for (int j = 0; j < MR; ++j) {
        const double v = rv[j];                                   // v'_j (rescaled unknowns)
        w[j + 2] = VC ? v : v * (1.0 / d2_scale(j));
        w0 = fma(-C0[j], v, w0);
        w1 = fma(-C1[j], v, w1);
        ok = ok && (fabs(v) < 1.0e300);
      }
      w[0] = w0;
      w[1] = w1;
      ok = ok && (fabs(w0) < 1.0e300) && (fabs(w1) < 1.0e300);
      if (!ok) st = LSSVR_ST_FALLBACK;
    }

    if (st != LSSVR_ST_OK) {
      // Dual.py:164-169: linear interpolant of (g_l, g_r) as a Legendre series
#pragma unroll
      for (int i = 0; i < M; ++i) w[i] = 0.0;
      w[0] = 0.5 * (gl + gr);
      w[1] = 0.5 * (gr - gl);
      if (live && p.fail_count) atomicAdd(p.fail_count, 1);
    }
    if (live && p.status) {
      if (p.ne * (int64_t)M <= kWriteThroughMaxDoubles)          // small launch: write-through, like W below
        __hip_atomic_store(&p.status[id], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else p.status[id] = st;
    }
    if (live && scattered) {
      // heterogeneous launch: rows go to the mesh index, ldw apart (direct stores)
      double* const Wrow = p.W + id * (p.ldw ? p.ldw : (int64_t)M);
#pragma unroll
      for (int i = 0; i < M; ++i) Wrow[i] = w[i];
    }
  }
  if (scattered) return;

  // --- coalesced store: each wave transposes its own 64 x M tile through LDS --------
  // (wave-private, so no workgroup barrier: a wave that finishes early stores early;
  // LDS operations of one wave execute in order)
  double* const wt = tile + (tid >> 6) * kSmallTilePerWave<M, RHS, VC>;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the staging reads are done)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < M; ++i) wt[lane * M + i] = w[i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = ((int64_t)block * kBlock + (tid & ~63)) * M;
  const int64_t total = p.ne * M;
  if (total <= kWriteThroughMaxDoubles) {
    // small outputs: write-through stores (see kWriteThroughMaxDoubles)
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int64_t idx = base + (int64_t)i * 64 + lane;
      if (idx < total) __hip_atomic_store(&p.W[idx], wt[i * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int64_t idx = base + (int64_t)i * 64 + lane;
    // write-once output: non-temporal stores leave less for the end-of-kernel L2 write-back
    if (idx < total) __builtin_nontemporal_store(wt[i * 64 + lane], &p.W[idx]);
  }
}

// MINW = minimum waves per SIMD the register allocator must leave room for.  1 (the only value
// dispatched): no constraint -- 136 VGPRs at M = 9 with the trigonometric coefficients in SGPRs,
// three resident waves.
template <int M, int RHS, bool VC, int MINW>
__global__ __launch_bounds__(kBlock, MINW) void enhance_small_kernel(EnhanceArgs p) {
  if constexpr (VC) {
    __shared__ double tile[(kBlock / 64) * kSmallTilePerWave<M, RHS, VC>];
    enhance_small_body<M, RHS, VC>(p, blockIdx.x, tile);
  } else {
    // Poisson rows: the Chebyshev-moment form (enhance_small_cheb.hpp)
    __shared__ double tile[(kBlock / 64) * kChebTilePerWave<M, RHS>];
    enhance_small_body_cheb<M, RHS>(p, blockIdx.x, tile);
  }
}

// The Poisson lane kernel with the near-square refinement loop (launches with p.refine > 0 only)
template <int M, int RHS>
__global__ __launch_bounds__(kBlock) void enhance_small_refine_kernel(EnhanceArgs p) {
  __shared__ double tile[(kBlock / 64) * kChebTilePerWave<M, RHS>];
  enhance_small_body_cheb<M, RHS, true>(p, blockIdx.x, tile);
}

// One launch for a whole step of the hot path on one mesh: blocks [0, eblocks) run the
// per-element enhancement, the remaining blocks the element-local P1 assembly (one thread
// per node).  The two halves share nothing but the node array, so fusing them only removes
// a launch boundary and lets the short assembly run in the shadow of the enhancement.
template <int M>
__global__ __launch_bounds__(kBlock) void step_small_kernel(EnhanceArgs p, P1Args a, QuadRule q,
                                                             unsigned eblocks) {
  __shared__ double tile[kBlock * M];
  if (blockIdx.x < eblocks) {
    enhance_small_body_cheb<M, LSSVR_RHS_SIN>(p, blockIdx.x, tile);
  } else {
    const int64_t i = (int64_t)(blockIdx.x - eblocks) * kBlock + threadIdx.x;
    if (i <= a.ne) p1_node<true>(a, q, i);
  }
}

// The same fusion for BASELINE config 5: variable-coefficient enhancement (tabulated a, a', f) +
// the a-weighted P1 assembly from tabulated quadrature values, ONE grid.  Instantiated up to
// kStepVcMaxM (the direct-Gram bodies grow as M^2: above, lssvr_step_varcoef issues two launches).
constexpr int kStepVcMaxM = 12;
template <int M, int RHS>
__global__ __launch_bounds__(kBlock) void step_small_vc_kernel(EnhanceArgs p, P1Args a, QuadRule q,
                                                                unsigned eblocks) {
  __shared__ double tile[(kBlock / 64) * kSmallTilePerWave<M, RHS, true>];
  if (blockIdx.x < eblocks) {
    enhance_small_body<M, RHS, true>(p, blockIdx.x, tile);
  } else {
    const int64_t i = (int64_t)(blockIdx.x - eblocks) * kBlock + threadIdx.x;
    if (i <= a.ne) p1_node<false>(a, q, i);
  }
}

// ----------------------------------------------------------------------------
// dispatch
// ----------------------------------------------------------------------------
template <int M, int RHS, bool VC>
static hipError_t launch_small(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  const unsigned blocks = (unsigned)((a.ne + kBlock - 1) / kBlock);
  // (An occupancy-4 build -- __launch_bounds__(kBlock, 4): 128 VGPRs, a few spills -- used to take
  // over above 2e5 elements; since the Chebyshev-moment body needs 136 VGPRs it loses: 406 against
  // 388 us at 1e7 elements, 41.9 against 41.3 at 1e6, same run.)
  if constexpr (!VC && M >= kRefineMinM) {
    if (a.refine > 0) return launch(enhance_small_refine_kernel<M, RHS>, dim3(blocks), dim3(kBlock), s, o, a);
  }
  return launch(enhance_small_kernel<M, RHS, VC, 1>, dim3(blocks), dim3(kBlock), s, o, a);
}

template <int M>
static hipError_t launch_step(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,
                              hipStream_t s, const LaunchOpts* o) {
  const unsigned eb = (unsigned)((e.ne + kBlock - 1) / kBlock);
  const unsigned ab = (unsigned)((a.ne + 1 + kBlock - 1) / kBlock);
  return launch(step_small_kernel<M>, dim3(eb + ab), dim3(kBlock), s, o, e, a, q, eb);
}

template <int M>
static hipError_t launch_step_vc(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,
                                 hipStream_t s, const LaunchOpts* o) {
  if constexpr (M <= kStepVcMaxM) {
    const unsigned eb = (unsigned)((e.ne + kBlock - 1) / kBlock);
    const unsigned ab = (unsigned)((a.ne + 1 + kBlock - 1) / kBlock);
    if (e.tab_ps != 1)
      return launch(step_small_vc_kernel<M, LSSVR_RHS_ARRAY_PM>, dim3(eb + ab), dim3(kBlock), s, o, e, a, q, eb);
    return launch(step_small_vc_kernel<M, LSSVR_RHS_ARRAY>, dim3(eb + ab), dim3(kBlock), s, o, e, a, q, eb);
  } else {
    return hipErrorInvalidValue;
  }
}

// Each translation unit enhance_small_*.hip instantiates a range of M (the fully unrolled
// kernels are large: one TU per range keeps the build parallel) through this macro.
#define LSSVR_DEFINE_SMALL_RANGE(NAME, FOR_EACH_M)                                              \
  hipError_t enhance_small_##NAME(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {    \
    switch (a.M) {                                                                               \
      FOR_EACH_M(LSSVR_SMALL_CASE_ENH)                                                           \
      default:                                                                                   \
        return hipErrorInvalidValue;                                                             \
    }                                                                                            \
  }                                                                                              \
  hipError_t step_small_##NAME(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,         \
                               hipStream_t s, const LaunchOpts* o) {                             \
    switch (e.M) {                                                                               \
      FOR_EACH_M(LSSVR_SMALL_CASE_STEP)                                                          \
      default:                                                                                   \
        return hipErrorInvalidValue;                                                             \
    }                                                                                            \
  }                                                                                              \
  hipError_t step_small_vc_##NAME(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,      \
                                  hipStream_t s, const LaunchOpts* o) {                          \
    switch (e.M) {                                                                               \
      FOR_EACH_M(LSSVR_SMALL_CASE_STEP_VC)                                                       \
      default:                                                                                   \
        return hipErrorInvalidValue;                                                             \
    }                                                                                            \
  }

#define LSSVR_SMALL_CASE_ENH(MM)                                                     \
  case MM:                                                                           \
    if (a.a_values && a.tab_ps != 1) return launch_small<MM, LSSVR_RHS_ARRAY_PM, true>(a, s, o); \
    if (a.a_values) return launch_small<MM, LSSVR_RHS_ARRAY, true>(a, s, o);         \
    if (a.rhs_id == LSSVR_RHS_SIN) return launch_small<MM, LSSVR_RHS_SIN, false>(a, s, o); \
    if (a.tab_ps != 1) return launch_small<MM, LSSVR_RHS_ARRAY_PM, false>(a, s, o);  \
    return launch_small<MM, LSSVR_RHS_ARRAY, false>(a, s, o);
#define LSSVR_SMALL_CASE_STEP(MM) \
  case MM:                        \
    return launch_step<MM>(e, a, q, s, o);
#define LSSVR_SMALL_CASE_STEP_VC(MM) \
  case MM:                           \
    return launch_step_vc<MM>(e, a, q, s, o);

}  // namespace lssvr
