// Per-element LSSVR enhancement, lane-per-element path, POISSON rows (Dual.py:43-44: -u''):
// the CHEBYSHEV-MOMENT form of the Legendre Gram contraction.
//
// The PDE rows of the reference are pure polynomials of t: rho_j(t) = L''_{j+2}(t), degree j.
// Writing the bubble part in the Chebyshev basis,  sum_j v_j rho_j = sum_i z_i T_i  (v = Y z,
// Y the exact triangular connection matrix of cheb_tables.hpp), the Gram matrix of the
// collocation rows becomes
//     G_ik = sum_k' T_i(t_k') T_k(t_k') = 1/2 (m_{i+k} + m_{|i-k|}),   m_d = sum_k' T_d(t_k'),
// i.e. it is determined by 2 MR - 1 power sums instead of MR (MR+1)/2 products: per collocation
// point the kernel runs the two-term recurrence T_d = 2t T_{d-1} - T_{d-2} up to degree MR-1
// (one FMA per term; Legendre-type recurrences need two), accumulates m_1..m_{MR-1} (adds), the
// MR-1 products P_j = sum T_{MR-1} T_j that give the upper moments m_{MR-1+j} = 2 P_j - m_{MR-1-j},
// and the right-hand side r_i = sum T_i phi -- 3 MR + O(1) instructions per point instead of
// MR^2/2 + 3 MR.  Same QP, same minimiser (Dual.py:46-78; oracle restatement
// oracle/lssvr_oracle.py::solve_bc_eliminated; numpy prototype of exactly this arrangement
// scripts/proto/cheb_moment.py::solve_cheb_kernel: <= 7e-16 from the 60-digit minimiser over
// random M, n, gamma, h, x0 -- as accurate as the direct Gram in every regime).
//
// Boundary rows (Dual.py:61-76).  t_a = off + scl a and t_b = off + scl b are -1 and +1 up to
// the rounding of numpy's mapdomain, |1 + t_a|, |1 - t_b| ~ eps |x| / h.  With
//     e_a = 1 + t_a, e_b = 1 - t_b (exact), sigma = (e_a + e_b)/2, delta = (e_a - e_b)/2,
//     L_p(-1 + e) = (-1)^p (1 - a_p e + O((a_p e)^2)),  a_p = p (p+1) / 2,
// the eliminated rows  w_{0,1} = d - C v  are, to first order,
//     p even: C0 = 1 - a sigma, C1 = a delta;     p odd: C0 = (a-1) delta, C1 = 1 - (a-1) sigma.
// The neglected terms are (a e)^2 / 4 relative: the fast path is taken while
// a_max max(|e_a|, |e_b|) < 1e-6 (|x|/h up to ~1e8 at degree 8); beyond, the whole wave takes
// the exact Legendre recurrence and a dense ridge (cheb_slow_build: out of line, on scratch).
// The ridge eps (N + C_z^T C_z), N = Y^T Y, C_z = C Y, is built from compile-time tables
// (alpha, b, N of cheb_tables.hpp) with its first-order terms in sigma / delta.
// The whole system is carried scaled by 2 (S2 = m_{i+k} + m_{|i-k|} + 2 eps R) so that the 1/2
// of the product formula costs nothing.
//
// RIDGE-DOMINATED elements (gamma scl^4 below ridge_gamma_scl4(M), lssvr_device.hpp: coarse elements with a small gamma;
// no BASELINE configuration).  In the Chebyshev basis the ridge is eps (N + C_z^T C_z), N = Y^T Y, and
// once eps outweighs the Gram the solve inherits cond(Y)^2 (measured against the 60-digit minimiser,
// round 3's finding: 3e-13 at M = 9, 3e-11 at M = 22, where the float64 KKT solve holds 1e-15).  A wave
// with such an element runs cheb_ridge_solve: the SAME moments, mapped back to the Legendre-bubble
// basis, G_v = X^T G_T X (X = Y^-1 >= 0, exact table), where the ridge is eps (I + C^T C) again -- the
// accuracy of the direct Gram in that regime (<= 3e-15) at no cost to the hot path (out of line, on
// the cold path's scratch; numpy prototype scripts/proto/cheb_moment.py::solve_ridge_kernel).
#pragma once
#include "cheb_tables.hpp"
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

constexpr int kReseed = 64;   // in-kernel rhs: the (sin, cos) rotation is re-seeded every 64 points
constexpr int kPairMaxM = 12;   // in-kernel rhs: two collocation points per loop iteration up to here
constexpr int kRefineMinM = 14; // near-square refinement of the lane kernel: instantiated from here up

// MEASUREMENT HOOK: the body calls probe.mark(k) at its phase boundaries.  The shipped kernels pass NoProbe
// (nothing is emitted); scripts/micro/lane_phases.hip instantiates the SAME body with a probe that stamps
// s_memrealtime per wave: the per-phase timeline of DESIGN.md section 7.  (Round 3 carried five preprocessor
// variants of the body in this header instead; their numbers are profiles/r03_decompose_small.txt.)
struct NoProbe {
  __device__ __forceinline__ void mark(int) const {}
};
enum LanePhase : int { kPhEntry = 0, kPhArgs, kPhLoaded, kPhSeeded, kPhMoments, kPhSystem, kPhSolved, kPhStored, kPhCount };

template <int M, int RHS>
constexpr int kChebTilePerWave =
    (RHS == LSSVR_RHS_ARRAY && 64 * 9 > 64 * M) ? 64 * 9 : 64 * M;   // output tile / rhs staging


// ---------------------------------------------------------------------------------------------
// Cold path (a wave with any element beyond the first-order range, |x|/h >~ 1e8): exact boundary
// rows by the Legendre recurrence, C_z = C Y, dense ridge.  Out of line, on a per-lane scratch
// buffer, with run-time loops: inlined, its arrays cost the hot path 62 VGPRs and a resident wave
// per SIMD; like this it costs nothing but the call.
// ---------------------------------------------------------------------------------------------
template <int M>
struct ChebSlow {
  static constexpr int MR = M - 2;
  static constexpr int NT = MR * (MR + 1) / 2;
  static constexpr int kMom = 0;                    // in : m_0 .. m_{2MR-2}
  static constexpr int kRhs = kMom + 2 * MR - 1;    // in/out: rhs2[MR]
  static constexpr int kS = kRhs + MR;              // out: S2, packed lower triangle [NT]
  static constexpr int kD = kS + NT;                // out: d0, d1
  static constexpr int kC0 = kD + 2;                // out: C0[MR], C1[MR] (v-basis)
  static constexpr int kC1 = kC0 + MR;
  static constexpr int kZ0 = kC1 + MR;              // work: C_z = C Y
  static constexpr int kZ1 = kZ0 + MR;
  static constexpr int kRes = kZ1 + MR;             // cheb_ridge_solve's result: w[M]
  static constexpr int kFlag = kRes + M;            // this lane: 0 keeps its own result, +1 takes kRes, -1 kRes failed
  static constexpr int kSize = kFlag + 1;
};

template <int M>
__device__ __attribute__((noinline)) void cheb_slow_build(double* __restrict__ buf, double ta,
                                                          double tb, double gl, double gr,
                                                          double eps2) {
  using L = ChebSlow<M>;
  constexpr int MR = L::MR;
  const double idet = rcp_newton(tb - ta);
  const double d0 = (tb * gl - ta * gr) * idet;
  const double d1 = (gr - gl) * idet;
  buf[L::kD] = d0;
  buf[L::kD + 1] = d1;
  // L_p(ta), L_p(tb), p = 2 .. M-1:  (p+1) L_{p+1} = (2p+1) t L_p - p L_{p-1}
  double am1 = 1.0, a0 = ta, bm1 = 1.0, b0 = tb;
#pragma nounroll
  for (int pp = 1; pp < M - 1; ++pp) {
    const double inv = 1.0 / (double)(pp + 1);
    const double a1 = ((double)(2 * pp + 1) * ta * a0 - (double)pp * am1) * inv;
    const double b1 = ((double)(2 * pp + 1) * tb * b0 - (double)pp * bm1) * inv;
    am1 = a0; a0 = a1;
    bm1 = b0; b0 = b1;
    buf[L::kC0 + pp - 1] = (tb * a1 - ta * b1) * idet;      // column j = pp - 1 <-> degree pp + 1
    buf[L::kC1 + pp - 1] = (b1 - a1) * idet;
  }
#pragma nounroll
  for (int i = 0; i < MR; ++i) {
    double s0 = 0.0, s1 = 0.0;
#pragma nounroll
    for (int j = (i & 1); j <= i; j += 2) {
      s0 = fma(buf[L::kC0 + j], cheb::kY[j][i], s0);
      s1 = fma(buf[L::kC1 + j], cheb::kY[j][i], s1);
    }
    buf[L::kZ0 + i] = s0;
    buf[L::kZ1 + i] = s1;
  }
#pragma nounroll
  for (int i = 0; i < MR; ++i) {
    const double z0 = buf[L::kZ0 + i], z1 = buf[L::kZ1 + i];
#pragma nounroll
    for (int k = 0; k <= i; ++k) {
      double cc = fma(z0, buf[L::kZ0 + k], z1 * buf[L::kZ1 + k]);
      if (((i + k) & 1) == 0) cc += cheb::kN[i][k];
      buf[L::kS + tri(i, k)] = fma(eps2, cc, buf[L::kMom + i + k] + buf[L::kMom + i - k]);
    }
    buf[L::kRhs + i] = fma(eps2, fma(z0, d0, z1 * d1), buf[L::kRhs + i]);
  }
}

// ---------------------------------------------------------------------------------------------
// Cold path of a wave with a RIDGE-DOMINATED element (header comment): from the same moments and
// right-hand side, the system in the Legendre-bubble basis
//     S2_v = X^T (m_{i+k} + m_{|i-k|}) X + eps2 (I + C^T C),   rhs2_v = X^T r2 + eps2 C^T d
// with the exact boundary rows (Legendre recurrence, as cheb_slow_build), LDL^T and the two
// substitutions, all with run-time loops on the per-lane scratch buffer.  In: buf[kMom], buf[kRhs];
// out: buf[kRes .. kRes + M) = w; returns whether every pivot was positive.
// ---------------------------------------------------------------------------------------------
template <int M>
__device__ __attribute__((noinline)) bool cheb_ridge_solve(double* __restrict__ buf, double ta,
                                                           double tb, double gl, double gr,
                                                           double eps2) {
  using L = ChebSlow<M>;
  constexpr int MR = L::MR;
  const double idet = 1.0 / (tb - ta);
  const double d0 = (tb * gl - ta * gr) * idet;
  const double d1 = (gr - gl) * idet;
  double am1 = 1.0, a0 = ta, bm1 = 1.0, b0 = tb;
#pragma nounroll
  for (int pp = 1; pp < M - 1; ++pp) {
    const double inv = 1.0 / (double)(pp + 1);
    const double a1 = ((double)(2 * pp + 1) * ta * a0 - (double)pp * am1) * inv;
    const double b1 = ((double)(2 * pp + 1) * tb * b0 - (double)pp * bm1) * inv;
    am1 = a0; a0 = a1;
    bm1 = b0; b0 = b1;
    buf[L::kC0 + pp - 1] = (tb * a1 - ta * b1) * idet;
    buf[L::kC1 + pp - 1] = (b1 - a1) * idet;
  }
  // rhs2_v = X^T r2 (in place, descending: entry j reads entries <= j only) + eps2 C^T d
#pragma nounroll
  for (int j = MR - 1; j >= 0; --j) {
    double s = 0.0;
#pragma nounroll
    for (int i = (j & 1); i <= j; i += 2) s = fma(cheb::kX[i][j], buf[L::kRhs + i], s);
    buf[L::kRhs + j] = fma(eps2, fma(buf[L::kC0 + j], d0, buf[L::kC1 + j] * d1), s);
  }
  // S2_v column by column: a = G2 X[:, j] (kZ0, MR entries), then S[i][j] = X[:, i] . a, i >= j
#pragma nounroll
  for (int j = 0; j < MR; ++j) {
#pragma nounroll
    for (int i = 0; i < MR; ++i) {
      double s = 0.0;
#pragma nounroll
      for (int k = (j & 1); k <= j; k += 2) {
        const int dk = i > k ? i - k : k - i;
        s = fma(buf[L::kMom + i + k] + buf[L::kMom + dk], cheb::kX[k][j], s);
      }
      buf[L::kZ0 + i] = s;
    }
    const double c0j = buf[L::kC0 + j], c1j = buf[L::kC1 + j];
#pragma nounroll
    for (int i = j; i < MR; ++i) {
      double s = 0.0;
#pragma nounroll
      for (int k = (i & 1); k <= i; k += 2) s = fma(cheb::kX[k][i], buf[L::kZ0 + k], s);
      double cc = fma(buf[L::kC0 + i], c0j, buf[L::kC1 + i] * c1j);
      if (i == j) cc += 1.0;
      buf[L::kS + tri(i, j)] = fma(eps2, cc, s);
    }
  }
  // LDL^T in place (unit L below the diagonal, the diagonal holds 1/d_j), then L y = rhs, L^T v = D^-1 y
  bool ok = true;
#pragma nounroll
  for (int j = 0; j < MR; ++j) {
    const double dj = buf[L::kS + tri(j, j)];
    ok = ok && (dj > 0.0);
    const double rinv = 1.0 / dj;
    buf[L::kS + tri(j, j)] = rinv;
#pragma nounroll
    for (int c = j + 1; c < MR; ++c) {
      const double lcj = buf[L::kS + tri(c, j)] * rinv;
#pragma nounroll
      for (int i = c; i < MR; ++i)
        buf[L::kS + tri(i, c)] = fma(-buf[L::kS + tri(i, j)], lcj, buf[L::kS + tri(i, c)]);
      buf[L::kS + tri(c, j)] = lcj;
    }
  }
#pragma nounroll
  for (int i = 0; i < MR; ++i) {
    double s = buf[L::kRhs + i];
#pragma nounroll
    for (int j = 0; j < i; ++j) s = fma(-buf[L::kS + tri(i, j)], buf[L::kRhs + j], s);
    buf[L::kRhs + i] = s;
  }
  double w0 = d0, w1 = d1;
#pragma nounroll
  for (int i = MR - 1; i >= 0; --i) {
    double s = buf[L::kRhs + i] * buf[L::kS + tri(i, i)];
#pragma nounroll
    for (int j = i + 1; j < MR; ++j) s = fma(-buf[L::kS + tri(j, i)], buf[L::kRhs + j], s);
    buf[L::kRhs + i] = s;
    buf[L::kRes + 2 + i] = s;
    w0 = fma(-buf[L::kC0 + i], s, w0);
    w1 = fma(-buf[L::kC1 + i], s, w1);
  }
  buf[L::kRes] = w0;
  buf[L::kRes + 1] = w1;
#pragma nounroll
  for (int i = 0; i < M; ++i) ok = ok && (fabs(buf[L::kRes + i]) < 1.0e300);
  return ok;
}

// REFINE: the build with the near-square refinement loop (its own kernel, enhance_small_refine_kernel:
// compiled into the main kernel the loop cost the NORMAL regime 15-65 % at M = 16..22 through register
// allocation alone -- 61 -> 101 us at M = 20, n = 40 -- so launches with p.refine == 0 never see it).
template <int M, int RHS, bool REFINE = false, class Probe = NoProbe>
__device__ __forceinline__ void enhance_small_body_cheb(const EnhanceArgs& p, const unsigned block,
                                                        double* __restrict__ tile, const Probe probe = Probe{}) {
  constexpr int MR = M - 2;
  constexpr int TD = MR > 0 ? MR : 1;
  constexpr int NT = MR * (MR + 1) / 2;
  constexpr int kStageK = 8;

  const int tid = threadIdx.x;
  const int64_t e = (int64_t)block * kBlock + tid;
  double w[M];
  int st = LSSVR_ST_OK;
  // cold paths' per-lane scratch (never touched by a wave that stays on the fast path) and the lane's
  // verdict on the ridge-dominated solve parked in it (ChebSlow<M>::kFlag)
  [[maybe_unused]] double slowbuf[MR > 0 ? ChebSlow<M>::kSize : 1];
  [[maybe_unused]] double rflag = 0.0;
  probe.mark(kPhEntry);
  probe.mark(kPhArgs);
  const bool scattered = p.elem_ids != nullptr || (p.ldw != 0 && p.ldw != M);
  // Every lane runs the body (lanes past the end of the last wave on a duplicate of the last
  // element, their stores masked): tabulated inputs are loaded cooperatively by the wave.
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
    const double inv_gamma = p.gamma_values ? rcp_newton(p.gamma_values[id]) : p.inv_gamma;

    probe.mark(kPhLoaded);
    const DomainMap dm = map_params(a, b);
    const int n = p.n;
    const double step = dm.oldlen / (double)(n - 1);
    // 1 / scl^2 = (h/2)^2: within 2 ulp of 1 / fl(fl(2/h)^2), no division
    const double hh = 0.5 * dm.oldlen;
    const double inv_scl2 = hh * hh;
    const double eps2 = (2.0 * inv_gamma) * (inv_scl2 * inv_scl2);     // 2 / (gamma scl^4)

    // --- boundary rows, first order in (e_a, e_b) -------------------------------------------
    const double ta = dm.off + dm.scl * a;
    const double tb = dm.off + dm.scl * b;
    const double ea = 1.0 + ta, eb = 1.0 - tb;
    const double sig = 0.5 * (ea + eb), del = 0.5 * (ea - eb);
    const double idet = 0.5 * fma(sig, 1.0 + sig, 1.0);            // 1 / (tb - ta) = 1 / (2 - 2 sigma)
    constexpr double kAmax = 0.5 * (double)((M - 1) * M);
    // (a non-finite map -- degenerate element -- is not "slow": it ends in the status test)
    // ... or a RIDGE-DOMINATED element (header comment): the wave then takes the same cold branches
    // (exact boundary rows for every lane) and, inside them, the Legendre-bubble solve for those lanes
    const bool slow = kAmax * fmax(fabs(ea), fabs(eb)) >= 1.0e-6 || (MR > 0 && ridge_dominated(eps2, M));
    const bool any_slow = __any(slow);
    double d0 = (tb * gl - ta * gr) * idet;
    double d1 = (gr - gl) * idet;

    if constexpr (MR == 0) {
      if (any_slow) {
        const double idx = rcp_newton(tb - ta);
        d0 = (tb * gl - ta * gr) * idx;
        d1 = (gr - gl) * idx;
      }
    }

    if constexpr (MR == 0) {
      w[0] = d0;
      w[1] = d1;
      if (!(fabs(d0) < 1.0e300 && fabs(d1) < 1.0e300)) st = LSSVR_ST_FALLBACK;
    } else {
      // --- moments over the collocation points ----------------------------------------------
      double mom[TD], P[TD], rv[TD];
#pragma unroll
      for (int i = 0; i < MR; ++i) mom[i] = P[i] = rv[i] = 0.0;

      // In-kernel rhs f = amp sin(fl(omega x_k)) without a sin per point: (s~, c~) =
      // (sin, cos)(th0 + k dth) is carried by a rotation and numpy's argument rounding
      // arg_k = fl(omega x_k) is restored to first order, sin(arg_k) = s~ + c~ delta_k,
      // delta_k = (arg_k - th0) - k dth (|delta| ~ |omega x| eps; a wave with any |delta| > 1e-7
      // takes the per-point sin).  The pair is carried pre-multiplied by kappa = -2 amp / scl^2
      // (phi2_k = -2 f(x_k) / scl^2 comes out of the correction FMA) and re-seeded every kReseed
      // points, so the rotation's own rounding never exceeds ~64 eps.
      double rs = 0.0, rc = 1.0, sd = 0.0, cd = 1.0, th0 = 0.0, dth = 0.0, kappa = 0.0;
      if constexpr (RHS == LSSVR_RHS_SIN) {
        dth = p.rhs_omega * step;
        sincos_tab(dth, sd, cd, p.trig);
        kappa = -2.0 * (p.rhs_amp * inv_scl2);
      }
      [[maybe_unused]] const double fscale = -2.0 * inv_scl2;
      probe.mark(kPhSeeded);
      // Tabulated rhs ([element][point]): the wave loads kStageK points of its 64 rows at a time
      // with consecutive lanes on consecutive doubles into LDS at pitch kStageK + 1 (conflict-free
      // when every lane then reads its own row); a lane walking its own row would touch 64 cache
      // lines per load instruction.
      [[maybe_unused]] double* const stg = tile + (tid >> 6) * kChebTilePerWave<M, RHS>;
      [[maybe_unused]] const int64_t e0 = (int64_t)block * kBlock + (tid & ~63);
      if constexpr (RHS == LSSVR_RHS_ARRAY_PM) {
        // POINT-MAJOR table rhs_values[k * ne + e]: consecutive lanes read consecutive doubles, no
        // staging; the next kPrefetchCheb values are requested before the current ones are used.
        constexpr int kPF = 4;
        const int64_t ps = p.tab_ps;
        const double* const tf = p.rhs_values + ec * p.tab_es;
        double T[TD];
        auto accumulate = [&](const double xk, const double f) {
          const double tk = dm.off + dm.scl * xk;         // mapdomain, two roundings
          const double phi2 = f * fscale;
          T[0] = 1.0;
          if constexpr (MR > 1) T[1] = tk;
          const double tt = tk + tk;
#pragma unroll
          for (int d = 2; d < MR; ++d) T[d] = fma(tt, T[d - 1], -T[d - 2]);
          rv[0] += phi2;
#pragma unroll
          for (int d = 1; d < MR; ++d) {
            mom[d] += T[d];
            P[d] = fma(T[MR - 1], T[d], P[d]);
            rv[d] = fma(T[d], phi2, rv[d]);
          }
        };
        double cf[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) cf[i] = __builtin_nontemporal_load(tf + (int64_t)min(i, n - 1) * ps);
        for (int k = 0; k < n; k += kPF) {
          double nf[kPF];
#pragma unroll
          for (int i = 0; i < kPF; ++i) nf[i] = __builtin_nontemporal_load(tf + (int64_t)min(k + kPF + i, n - 1) * ps);
#pragma unroll
          for (int i = 0; i < kPF; ++i)
            if (k + i < n) accumulate((k + i == n - 1) ? b : (double)(k + i) * step + a, cf[i]);
#pragma unroll
          for (int i = 0; i < kPF; ++i) cf[i] = nf[i];
        }
      } else
      for (int k0 = 0; k0 < n; k0 += kReseed) {
        if constexpr (RHS == LSSVR_RHS_SIN) {
          const double x0 = (k0 == 0) ? a : fma((double)k0, step, a);
          th0 = p.rhs_omega * x0;
          sincos_tab(th0, rs, rc, p.trig);
          rs *= kappa;
          rc *= kappa;
        }
        const int k1 = min(k0 + kReseed, n);
        // one collocation point: abscissa xk (np.linspace), index k of the element, k0 of the chunk
        auto point = [&](const double xk, const int k) {
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
                stg[row * (kStageK + 1) + kk] = in ? p.rhs_values[er * n + (k + kk)] : 0.0;
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
          const double tk = dm.off + dm.scl * xk;         // mapdomain, two roundings
          double phi2;
          if constexpr (RHS == LSSVR_RHS_SIN) {
            const double arg = p.rhs_omega * xk;
            const double delta = fma(-(double)(k - k0), dth, arg - th0);
            phi2 = fma(rc, delta, rs);
            if (__any(!(fabs(delta) < 1.0e-7))) phi2 = kappa * sin_tab(arg, p.trig);
            // the rotation, written as three-operand FMAs into the carried registers themselves
            // (hipcc picks v_fmac + a copy per loop-carried value otherwise: 2 of 45 instructions)
            const double t1 = rc * sd, t2 = rs * sd;
            double rs_next, rc_next;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(rs_next) : "v"(rs), "v"(cd), "v"(t1));
            asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(rc_next) : "v"(rc), "v"(cd), "v"(t2));
            rs = rs_next;
            rc = rc_next;
          } else {
            phi2 = stg[lane * (kStageK + 1) + (k & (kStageK - 1))] * fscale;
          }
          // Chebyshev values T_0 .. T_{MR-1}, moments, upper-moment products, right-hand side
          double T[TD];
          T[0] = 1.0;
          if constexpr (MR > 1) T[1] = tk;
          const double tt = tk + tk;
#pragma unroll
          for (int d = 2; d < MR; ++d) T[d] = fma(tt, T[d - 1], -T[d - 2]);
          rv[0] += phi2;
#pragma unroll
          for (int d = 1; d < MR; ++d) {
            mom[d] += T[d];
            P[d] = fma(T[MR - 1], T[d], P[d]);
            rv[d] = fma(T[d], phi2, rv[d]);
          }
        };
        // np.linspace: fl(fl(k step) + a) for k < n-1, the last sample is b itself (step == 0 needs
        // h < 1e-320, where scl = 2/h overflows and the element ends in the linear fallback whatever
        // x_k is); the last point is peeled so that the loop carries no select
        const int kl = min(k1, n - 1);
        int k = k0;
        if constexpr (RHS == LSSVR_RHS_SIN && M <= kPairMaxM) {
          // TWO points per iteration, their dependent chains (abscissa -> t -> T_2 .. T_{MR-1}, 11 links of
          // 8.3 cycles at M = 9) side by side in one basic block: a wave issues in order, so with one point per
          // iteration every link of the chain stalled it (347 cycles per point for a lone wave, 301 with two on
          // the SIMD, against 45 instructions x 4.5).  No per-point branch in here: the wave qualifies when
          // |delta| < 1e-7 is guaranteed a priori -- |delta_k| <= eps |omega| (3 n step + 3 max|x|) by the
          // roundings of linspace / omega x_k / theta_0 / dtheta (eps = 2^-53), bounded below with 1e-15 |omega|
          // max(|a|, |b|, h) -- so for such waves the per-point test of `point` is never true and the two forms
          // perform the SAME operations in the SAME order: results are bit-equal.
          const double xmax = fmax(fmax(fabs(a), fabs(b)), fabs(dm.oldlen));
          const bool tame = fabs(p.rhs_omega) * xmax * 1.0e-15 < 1.0e-7;
          if (!__any(!tame)) {
            // points k (inside the element) and k + 1 (abscissa xB: the next linspace point, or b itself for the
            // last one -- np.linspace's endpoint, as in `point(b, n - 1)`)
            // (indices carried as doubles: exact, and no int -> double conversion in the loop)
            auto pair = [&](const double kd, const double xB) {
              const double jd = kd - (double)k0;
              const double xA = kd * step + a;
              const double tA = dm.off + dm.scl * xA;       // mapdomain, two roundings
              const double tB = dm.off + dm.scl * xB;
              const double argA = p.rhs_omega * xA;
              const double argB = p.rhs_omega * xB;
              const double dA = fma(-jd, dth, argA - th0);
              const double dB = fma(-(jd + 1.0), dth, argB - th0);
              const double phiA = fma(rc, dA, rs);
              const double rs1 = fma(rs, cd, rc * sd);
              const double rc1 = fma(rc, cd, -(rs * sd));
              const double phiB = fma(rc1, dB, rs1);
              rs = fma(rs1, cd, rc1 * sd);
              rc = fma(rc1, cd, -(rs1 * sd));
              double TA[TD], TB[TD];
              TA[0] = TB[0] = 1.0;
              if constexpr (MR > 1) {
                TA[1] = tA;
                TB[1] = tB;
              }
              const double ttA = tA + tA, ttB = tB + tB;
#pragma unroll
              for (int d = 2; d < MR; ++d) {
                TA[d] = fma(ttA, TA[d - 1], -TA[d - 2]);
                TB[d] = fma(ttB, TB[d - 1], -TB[d - 2]);
              }
              rv[0] = (rv[0] + phiA) + phiB;
#pragma unroll
              for (int d = 1; d < MR; ++d) {
                mom[d] = (mom[d] + TA[d]) + TB[d];
                P[d] = fma(TB[MR - 1], TB[d], fma(TA[MR - 1], TA[d], P[d]));
                rv[d] = fma(TB[d], phiB, fma(TA[d], phiA, rv[d]));
              }
            };
            double kd = (double)k0;
            // (four points per iteration: 164 registers, the same phase time -- not kept)
            for (; k + 2 <= kl; k += 2, kd += 2.0) pair(kd, (kd + 1.0) * step + a);
            if (k + 1 == kl && k1 == n) {                  // an even number of points: the last two as a pair as well
              pair(kd, b);
              k = n;
            }
          }
        }
        for (; k < kl; ++k) point((double)k * step + a, k);
        if (k1 == n && k < n) point(b, n - 1);
      }
      mom[0] = (double)n;
      probe.mark(kPhMoments);
      // all moments m_0 .. m_{2MR-2}:  m_{MR-1+j} = 2 P_j - m_{MR-1-j}
      double mm[2 * TD - 1];
#pragma unroll
      for (int d = 0; d < MR; ++d) mm[d] = mom[d];
#pragma unroll
      for (int j = 1; j < MR; ++j) mm[MR - 1 + j] = fma(2.0, P[j], -mom[MR - 1 - j]);

      // --- S2 = m_{i+k} + m_{|i-k|} + 2 eps (N + C_z^T C_z),  rhs2 = r2 + 2 eps C_z^T d ----------
      double G[NT];
      if (any_slow) {
        using L = ChebSlow<M>;
#pragma unroll
        for (int d = 0; d < 2 * MR - 1; ++d) slowbuf[L::kMom + d] = mm[d];
#pragma unroll
        for (int i = 0; i < MR; ++i) slowbuf[L::kRhs + i] = rv[i];
        // ridge-dominated lanes: the whole solve in the Legendre-bubble basis, result parked in kRes
        double flag = 0.0;
        if constexpr (M >= kRidgeMinM) {
          const bool ridge = ridge_dominated(eps2, M);
          if (__any(ridge)) {
            const bool rok = cheb_ridge_solve<M>(slowbuf, ta, tb, gl, gr, eps2);
            flag = ridge ? (rok ? 1.0 : -1.0) : 0.0;
#pragma unroll
            for (int i = 0; i < MR; ++i) slowbuf[L::kRhs + i] = rv[i];     // (transformed in place there)
          }
        }
        slowbuf[L::kFlag] = flag;
        cheb_slow_build<M>(slowbuf, ta, tb, gl, gr, eps2);
#pragma unroll
        for (int t = 0; t < NT; ++t) G[t] = slowbuf[L::kS + t];
#pragma unroll
        for (int i = 0; i < MR; ++i) rv[i] = slowbuf[L::kRhs + i];
        d0 = slowbuf[L::kD];
        d1 = slowbuf[L::kD + 1];
      } else {
        const double es = eps2 * sig, ed = eps2 * del;
        const double e_d0 = eps2 * d0, e_d1 = eps2 * d1;
        const double q_ev = eps2 * fma(del, d1, -(sig * d0));
        const double q_od = eps2 * fma(del, d0, -(sig * d1));
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
          for (int k = 0; k <= i; ++k) {
            const double Q = cheb::kAlpha[i] * cheb::kB[k] + cheb::kB[i] * cheb::kAlpha[k];
            double v = mm[i + k] + mm[i - k];
            if (((i + k) & 1) == 0) {
              const double R0 = cheb::kN[i][k] + cheb::kAlpha[i] * cheb::kAlpha[k];
              v = fma(R0, eps2, v);
              v = fma(-Q, es, v);
            } else {
              v = fma(Q, ed, v);
            }
            G[tri(i, k)] = v;
          }
          if ((i & 1) == 0) rv[i] = fma(cheb::kB[i], q_ev, fma(cheb::kAlpha[i], e_d0, rv[i]));
          else rv[i] = fma(cheb::kB[i], q_od, fma(cheb::kAlpha[i], e_d1, rv[i]));
        }
      }

      probe.mark(kPhSystem);
      // --- LDL^T (lower, in place; unit L below the diagonal, diagonal holds 1/d_j).  No
      // square roots, no pre-scaling (elimination of an SPD matrix is invariant under symmetric
      // diagonal scaling up to rounding).  A zero / non-finite pivot turns into inf / NaN in
      // 1/d_j and reaches every later entry; a negative one is caught by the sign test.
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
          G[tri(c, j)] = lcj;
        }
      }
      // x <- S^-1 x with the factors in G:  forward L y = x,  backward L^T z = D^-1 y
      auto ldl_solve = [&](double (&x)[TD]) {
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          double s = x[i];
#pragma unroll
          for (int j = 0; j < i; ++j) s = fma(-G[tri(i, j)], x[j], s);
          x[i] = s;
        }
#pragma unroll
        for (int i = MR - 1; i >= 0; --i) {
          double s = x[i] * G[tri(i, i)];
#pragma unroll
          for (int j = i + 1; j < MR; ++j) s = fma(-G[tri(j, i)], x[j], s);
          x[i] = s;
        }
      };
      ldl_solve(rv);
      // v = Y z (Legendre bubble coefficients), w_{0,1} = d - C v
      double C0[TD], C1[TD];
      if (any_slow) {
#pragma unroll
        for (int j = 0; j < MR; ++j) {
          C0[j] = slowbuf[ChebSlow<M>::kC0 + j];
          C1[j] = slowbuf[ChebSlow<M>::kC1 + j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < MR; ++j) {
          const double aj = cheb::kSlope[j];
          if ((j & 1) == 0) {
            C0[j] = fma(-aj, sig, 1.0);
            C1[j] = aj * del;
          } else {
            C0[j] = (aj - 1.0) * del;
            C1[j] = fma(-(aj - 1.0), sig, 1.0);
          }
        }
      }
      double w0 = d0, w1 = d1;
      auto to_legendre = [&]() {           // v = Y z,  w_{0,1} = d - C v
        w0 = d0;
        w1 = d1;
#pragma unroll
        for (int j = 0; j < MR; ++j) {
          double v = 0.0;
#pragma unroll
          for (int i = j; i < MR; i += 2) v = fma(cheb::kY[j][i], rv[i], v);
          w[j + 2] = v;
          w0 = fma(-C0[j], v, w0);
          w1 = fma(-C1[j], v, w1);
        }
      };
      to_legendre();
      // ---- NEAR-SQUARE REGIME (about as many equispaced points as bubble coefficients: the normal
      // equations lose up to ten digits): p.refine steps of the CORRECTED SEMI-NORMAL
      // equations -- the residual of S2 z = rhs2 taken through the ROWS,
      //     rho_i = sum_k T_i(t_k) (phi2_k - 2 sum_j z_j T_j(t_k))          (collocation part)
      //           + eps2 sum_j Y[j][i] (C0_j w_0 + C1_j w_1 - v_j)          (ridge part = -eps2 Y^T grad_v |w|^2/2)
      // never through the Gram matrix, solved with the factors already in G:  z += S^-1 rho.
      // What solve4_kernel<2> + residual_kernel do above M = 22 (round 2), here per lane (round 3).
      // A launch parameter (uniform); instantiated from M = 14 up (below, the normal equations are at
      // 2e-17 from n = M - 2 on) as a kernel of its own.
      if constexpr (REFINE) {
        for (int it = 0; it < p.refine; ++it) {
          double rho[TD];
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = (i & 1); j <= i; j += 2) s = fma(cheb::kY[j][i], fma(C0[j], w0, fma(C1[j], w1, -w[j + 2])), s);
            rho[i] = eps2 * s;
          }
#pragma nounroll
          for (int k = 0; k < n; ++k) {
            const double xk = (k == n - 1) ? b : (double)k * step + a;
            const double tk = dm.off + dm.scl * xk;
            double phi2;
            if constexpr (RHS == LSSVR_RHS_SIN) phi2 = kappa * sin_tab(p.rhs_omega * xk, p.trig);
            else phi2 = p.rhs_values[ec * p.tab_es + k * p.tab_ps] * fscale;
            double T[TD];
            T[0] = 1.0;
            if constexpr (MR > 1) T[1] = tk;
            const double tt = tk + tk;
            double Tz = rv[0];
            if constexpr (MR > 1) Tz = fma(tk, rv[1], Tz);
#pragma unroll
            for (int d = 2; d < MR; ++d) {
              T[d] = fma(tt, T[d - 1], -T[d - 2]);
              Tz = fma(T[d], rv[d], Tz);
            }
            const double ek = fma(-2.0, Tz, phi2);
#pragma unroll
            for (int d = 0; d < MR; ++d) rho[d] = fma(T[d], ek, rho[d]);
          }
          ldl_solve(rho);
#pragma unroll
          for (int i = 0; i < MR; ++i) rv[i] += rho[i];
          to_legendre();
        }
      }
#pragma unroll
      for (int j = 0; j < MR; ++j) ok = ok && (fabs(w[j + 2]) < 1.0e300);
      w[0] = w0;
      w[1] = w1;
      ok = ok && (fabs(w0) < 1.0e300) && (fabs(w1) < 1.0e300);
      if (any_slow) {
        // A ridge-dominated lane: its coefficients are the ones parked in slowbuf[kRes] and its status is
        // that solve's; they OVERWRITE the lane's row at the store (below) instead of being selected into
        // w[] here -- a select (or a branch with a join on w[]) costs every launch 39 vector instructions
        // per wave (3 %, measured with SQ_INSTS_VALU).  The asm keeps this a branch.
        asm volatile("" ::: "memory");
        rflag = slowbuf[ChebSlow<M>::kFlag];
        if (rflag != 0.0) ok = rflag > 0.0;
      }
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
    probe.mark(kPhSolved);
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
      if constexpr (MR > 0) {
        if (rflag > 0.0) {                     // ridge-dominated lane: the parked result (see above)
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < M; ++i) Wrow[i] = slowbuf[ChebSlow<M>::kRes + i];
        }
      }
    }
  }
  if (scattered) return;

  // --- coalesced store: each wave transposes its own 64 x M tile through LDS --------
  // (wave-private, so no workgroup barrier: a wave that finishes early stores early;
  // LDS operations of one wave execute in order)
  double* const wt = tile + (tid >> 6) * kChebTilePerWave<M, RHS>;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the staging reads are done)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < M; ++i) wt[lane * M + i] = w[i];
  if constexpr (MR > 0) {
    if (rflag > 0.0) {                         // ridge-dominated lane: the parked result (see above)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < M; ++i) wt[lane * M + i] = slowbuf[ChebSlow<M>::kRes + i];
    }
  }
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
    probe.mark(kPhStored);
    return;
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int64_t idx = base + (int64_t)i * 64 + lane;
    // write-once output: non-temporal stores leave less for the end-of-kernel L2 write-back
    if (idx < total) __builtin_nontemporal_store(wt[i * 64 + lane], &p.W[idx]);
  }
  probe.mark(kPhStored);
}

}  // namespace lssvr
