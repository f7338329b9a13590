// Per-element LSSVR enhancement, DUAL form -- the formulation BASELINE.json's north_star names:
// the Legendre kernel Gram matrix over the collocation points and the dense solve
// (K + I/gamma) alpha = y, one independent system per element (dual form;
// the QP is Dual.py:46-78).
//
//   rows      a_k = -L''(t_k)   (Poisson; -a_k L'' - (a'_k/scl) L' for variable coefficients),
//             B   = [L(t_a); L(t_b)]                     (the two boundary rows, Dual.py:61-76)
//   system    [[A A^T + eps I, A B^T], [B A^T, B B^T]] [lam; mu] = [f~; g],  w = A^T lam + B^T mu,
//             eps = 1 / (gamma scl^4).
//
// The 2 x 2 boundary block B B^T is well conditioned and is taken as the FIRST (block) pivot,
// analytically: with Pi = I - B^T (B B^T)^-1 B the Schur complement is the kernel matrix of the
// projected feature map,
//     K = (A Pi)(A Pi)^T,     (K + eps I) lam = f~ - A w_bc,     w = w_bc + (A Pi)^T lam,
//     w_bc = B^T (B B^T)^-1 g    (the minimum-norm coefficients that satisfy the boundary rows),
// an n x n SPD system -- K_ij = sum_p phi_p(x_i) phi_p(x_j) is north_star's Gram matrix with the
// boundary-projected PDE feature map.  cond(K + eps I) ~ |A|^2 gamma scl^4 reaches 1e22 on
// BASELINE's meshes (plain Cholesky breaks down), so the solve is, per SURVEY.md
// section 7 "Hard parts":
//   * symmetric Jacobi equilibration  d_i = (K_ii + eps)^-1/2,
//   * LU with PARTIAL PIVOTING (row per lane: the pivot search is a wave max-reduction, the pivot
//     row travels through LDS, the elimination is one FMA per entry and lane),
//   * safeguarded iterative refinement (up to 3 steps) with the residual in OPERATOR form,
//         r = f' - A'(w) - eps lam,     lam += (K + eps I)^-1 r,     w += A'^T dlam,
//     where w is CARRIED (never recomputed from lam: the rounding of A'^T lam, |lam| ~ gamma scl^4
//     |residual|, is what limits the plain dual form to 1e-10 on BASELINE config 1), and a final
//     re-projection of w onto the boundary rows.
// numpy prototype of exactly this arrangement: scripts/proto/dual_projected.py (float64, relative
// L2 to the 60-digit minimiser: 1e-16 on config 1, <= 1e-12 at degree 32 / 64 points, 5e-12 where
// n = M - 2 = 31 and the primal normal equations are at 3e-6).
//
// Mapping: LPE = 16 / 32 / 64 lanes per element (the smallest that holds max(n, M)), 64 / LPE
// elements per wave, one wave per workgroup, wave-private LDS.  Lane r of an element is collocation
// row r: it keeps its row of A' (M values) and its row of K (n values) in registers.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {
using namespace wave;

namespace {

constexpr int kDualMaxM = 33;
constexpr int kDualMaxN = 64;
#ifndef LSSVR_DUAL_REFINE
#define LSSVR_DUAL_REFINE 3      // (measurement builds lower it)
#endif
constexpr int kRefine = LSSVR_DUAL_REFINE;

// three-term recurrence coefficients as compile-time literals (the rows are built once per
// element in straight-line code, so literals cost no long-lived registers):
//   L_p = alL(p) t L_{p-1} - beL(p) L_{p-2};   q_m = L''_{m+2} = al2(m) t q_{m-1} - be2(m) q_{m-2};
//   r_m = L'_{m+1} = al1(m) t r_{m-1} - be1(m) r_{m-2}
__host__ __device__ constexpr double alL(int m) { return (double)(2 * m - 1) / (double)m; }
__host__ __device__ constexpr double beL(int m) { return (double)(m - 1) / (double)m; }
__host__ __device__ constexpr double al2(int m) { return (double)(2 * m + 3) / (double)m; }
__host__ __device__ constexpr double be2(int m) { return (double)(m + 3) / (double)m; }
__host__ __device__ constexpr double al1(int m) { return (double)(2 * m + 1) / (double)m; }
__host__ __device__ constexpr double be1(int m) { return (double)(m + 1) / (double)m; }

// per-element LDS (doubles).  ZS: rows of A' (row stride MP + 1, odd: conflict-free row writes),
// later the transposed products for w = A'^T lam (row stride LPE + 1).
template <int LPE, int MP>
struct DualLds {
  static constexpr int kZStride = MP + 1;
  static constexpr int kTStride = LPE + 1;
  static constexpr int kZS = (LPE * kZStride > MP * kTStride) ? LPE * kZStride : MP * kTStride;
  static constexpr int kLa = kZS;                 // L_p(t_a), p < MP
  static constexpr int kLb = kLa + MP;            // L_p(t_b)
  static constexpr int kProw = kLb + MP;          // pivot row ring: 2 x (LPE + 2)  [.. , rhs]
  static constexpr int kX = kProw + 2 * (LPE + 2);  // solution broadcast
  static constexpr int kD = kX + LPE;             // equilibration scales
  static constexpr int kW = kD + LPE;             // w (bubble part), p < MP
  static constexpr int kSize = ((kW + MP + 1) / 2) * 2;
};

// value of `v` in lane `src` (group-relative) of this lane's LPE-lane group
template <int LPE>
__device__ __forceinline__ double group_bcast(double v, int src, int gbase) {
  if constexpr (LPE == 64) {
    return readlane_f64(v, __builtin_amdgcn_readfirstlane(src));
  } else {
    return __shfl(v, gbase + src);
  }
}

template <int LPE>
__device__ __forceinline__ unsigned long long group_max_u64(unsigned long long k) {
#pragma unroll
  for (int o = 1; o < LPE; o <<= 1) {
    const unsigned long long other = __shfl_xor(k, o);
    k = other > k ? other : k;
  }
  return k;
}

template <int LPE>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = 1; o < LPE; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace

#ifndef LSSVR_DUAL_MINW
#define LSSVR_DUAL_MINW(LPE) ((LPE) == 16 ? 3 : (LPE) == 32 ? 2 : 1)
#endif
// LPE lanes per element, MP = padded number of Legendre coefficients (compile-time loop bound),
// RHS as in the primal kernels, VC = variable-coefficient rows.
template <int LPE, int MP, int RHS, bool VC>
__global__ __launch_bounds__(64, LSSVR_DUAL_MINW(LPE)) void enhance_dual_kernel(EnhanceArgs p, int nrefine) {
  using L = DualLds<LPE, MP>;
  constexpr int EPW = 64 / LPE;
  __shared__ double2_t lds2[EPW * L::kSize / 2];
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPE, r = lane % LPE, gbase = grp * LPE;
  double* const S = reinterpret_cast<double*>(lds2) + grp * L::kSize;
  double* const Zs = S;
  double* const LaS = S + L::kLa;
  double* const LbS = S + L::kLb;
  double* const Prow = S + L::kProw;
  double* const Ds = S + L::kD;
  double* const Ws = S + L::kW;
  const int M = p.M, n = p.n;

  const int64_t e_raw = (int64_t)blockIdx.x * EPW + grp;
  const bool live = e_raw < p.ne;
  const int64_t e = live ? e_raw : p.ne - 1;       // idle groups of the last wave: duplicate, stores masked
  const double a = p.x[e];
  const double b = p.x[e + 1];
  const int64_t eg = e + p.elem_offset;
  const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[e];
  const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[e + 1];
  const double gamma = p.gamma_values ? p.gamma_values[e] : p.gamma;
  const DomainMap dm = map_params(a, b);
  const double step = dm.oldlen / (double)(n - 1);
  const double scl2 = dm.scl * dm.scl;
  const double inv_scl2 = rcp_newton(scl2);
  const double eps = rcp_newton(gamma * (scl2 * scl2));

  // ---- boundary rows B = [L_p(t_a); L_p(t_b)] -> LDS; Q = B B^T and its inverse ---------------
  const double ta = dm.off + dm.scl * a;
  const double tb_ = dm.off + dm.scl * b;
  if (r < 2) {
    const double t = r == 0 ? ta : tb_;
    double* const dst = r == 0 ? LaS : LbS;
    double Lm2 = 1.0, Lm1 = t;
    dst[0] = 1.0;
    if (MP > 1) dst[1] = (M > 1) ? t : 0.0;
#pragma unroll
    for (int pp = 2; pp < MP; ++pp) {
      const double Lp = fma(alL(pp) * t, Lm1, -(beL(pp) * Lm2));
      dst[pp] = (pp < M) ? Lp : 0.0;               // padding columns are zero
      Lm2 = Lm1;
      Lm1 = Lp;
    }
  }
  wave_lds_sync();
  double q00 = 0.0, q01 = 0.0, q11 = 0.0;
#pragma unroll
  for (int pp = 0; pp < MP; ++pp) {
    const double la = LaS[pp], lb = LbS[pp];
    q00 = fma(la, la, q00);
    q01 = fma(la, lb, q01);
    q11 = fma(lb, lb, q11);
  }
  const double qdet = rcp_newton(fma(q00, q11, -(q01 * q01)));
  const double qi00 = q11 * qdet, qi01 = -q01 * qdet, qi11 = q00 * qdet;
  const double g0 = fma(qi00, gl, qi01 * gr), g1 = fma(qi01, gl, qi11 * gr);   // (B B^T)^-1 g

  // ---- row r of A: collocation point r (rows r >= n are padding: zero row, unit diagonal) ------
  const bool is_pt = r < n;
  const double xk = (is_pt && r == n - 1) ? b : (double)(is_pt ? r : 0) * step + a;
  const double tk = dm.off + dm.scl * xk;
  double ftil = 0.0;
  if (is_pt) {
    double fk;
    if constexpr (RHS == LSSVR_RHS_SIN) fk = p.rhs_amp * sin_reduced(p.rhs_omega * xk);
    else fk = p.rhs_values[e * p.tab_es + r * p.tab_ps];
    ftil = fk * inv_scl2;
  }
  double arow[MP];
  {
    const double sgn = is_pt ? -1.0 : 0.0;         // A = -L'' (padding rows: zero)
    double ak = 1.0, bk = 0.0;
    if constexpr (VC) {
      if (is_pt) {
        ak = p.a_values[e * p.tab_es + r * p.tab_ps];
        bk = p.da_values[e * p.tab_es + r * p.tab_ps] * (0.5 * dm.oldlen);     // a'/scl
      }
    }
    // q_m = L''_{m+2}, r1_m = L'_{m+1}
    double q2 = 0.0, q1 = 0.0, r2 = 0.0, r1 = 0.0;
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) {
      double d2 = 0.0, d1 = 0.0;
      if (pp >= 2) {
        const int m = pp - 2;
        d2 = (m == 0) ? 3.0 : (m == 1) ? 15.0 * tk : fma(al2(m) * tk, q1, -(be2(m) * q2));
        q2 = q1;
        q1 = d2;
      }
      if constexpr (VC) {
        if (pp >= 1) {
          const int m = pp - 1;
          d1 = (m == 0) ? 1.0 : (m == 1) ? 3.0 * tk : fma(al1(m) * tk, r1, -(be1(m) * r2));
          r2 = r1;
          r1 = d1;
        }
      }
      double v = VC ? fma(ak, d2, bk * d1) : d2;
      arow[pp] = (pp < M) ? sgn * v : 0.0;
    }
  }
  // projection A' = A - (A B^T)(B B^T)^-1 B and f' = f~ - A w_bc
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int pp = 0; pp < MP; ++pp) {
    s0 = fma(arow[pp], LaS[pp], s0);
    s1 = fma(arow[pp], LbS[pp], s1);
  }
  const double c0 = fma(qi00, s0, qi01 * s1), c1 = fma(qi01, s0, qi11 * s1);
#pragma unroll
  for (int pp = 0; pp < MP; ++pp) arow[pp] = fma(-c1, LbS[pp], fma(-c0, LaS[pp], arow[pp]));
  const double fp = ftil - fma(s0, g0, s1 * g1);

  // ---- K = A' A'^T: rows through LDS (broadcast reads), one dot product per entry ------------
#pragma unroll
  for (int pp = 0; pp < MP; ++pp) Zs[r * L::kZStride + pp] = arow[pp];
  wave_lds_sync();
  double krow[LPE];
#pragma unroll
  for (int c = 0; c < LPE; ++c) {
    double acc = 0.0;
    if (c < n) {
      const double* __restrict__ zc = Zs + c * L::kZStride;
#pragma unroll
      for (int pp = 0; pp < MP; ++pp) acc = fma(arow[pp], zc[pp], acc);
    }
    krow[c] = acc;
  }
  // (K + eps I), Jacobi equilibration
  double dr = 1.0;
#pragma unroll
  for (int c = 0; c < LPE; ++c)
    if (c == r) {
      const double kd = is_pt ? krow[c] + eps : 1.0;
      krow[c] = kd;
      dr = is_pt ? rsqrt_newton(kd) : 1.0;
    }
  Ds[r] = dr;
  wave_lds_sync();       // (also: every lane is done reading Zs)
#pragma unroll
  for (int c = 0; c < LPE; ++c) krow[c] = (c < n && is_pt) ? krow[c] * dr * Ds[c] : ((c == r) ? 1.0 : 0.0);

  // ---- LU with partial pivoting, rhs carried; refinement re-uses the factors -------------------
  // pivstep: step at which this row was the pivot (-1: still active).  After the elimination the
  // row holds U[pivstep][c] for c >= pivstep and the multipliers L for c < pivstep.
  int pivstep = is_pt ? -1 : 0x7fff;
  double rinv_own = 1.0;
  int pvec = 0;                                    // lane j of the group keeps the pivot lane of step j
  double y = dr * fp;
  // Only the STEPS are guarded by n (uniform branches); inside a step every one of the LPE columns
  // is processed: columns >= n are exact zeros in every row (padding), so they stay zero, and
  // guarding each column made the LPE = 64 kernel 63 000 instructions long (long-branch
  // expansion, hundreds of spilled SGPRs -- and wrong results from that build).
  // (ns: an opaque copy of n per use -- otherwise hipcc hoists the LPE uniform conditions j < n
  // out of the refinement loop and spills them)
  auto eliminate = [&]() {
    int ns = n;
    asm volatile("" : "+s"(ns));
#pragma unroll
    for (int j = 0; j < LPE; ++j) {
      if (j < ns) {
        const bool active = pivstep < 0;
        unsigned long long key = 0;
        if (active)
          key = (__builtin_bit_cast(unsigned long long, fabs(krow[j])) & ~127ull) | 64ull | (unsigned)r;
        key = group_max_u64<LPE>(key);
        const int P = (int)(key & 63ull);
        double* const pr = Prow + (j & 1) * (LPE + 2);
        if (r == P) {
#pragma unroll
          for (int c = j; c < LPE; ++c) pr[c] = krow[c];
          pr[LPE] = y;
          pivstep = j;
        }
        if (r == j) pvec = P;
        wave_lds_sync();
        const double rinv = rcp_newton(pr[j]);
        if (r == P) rinv_own = rinv;
        const bool act = pivstep < 0;
        const double m = act ? krow[j] * rinv : 0.0;
#pragma unroll
        for (int c = j + 1; c < LPE; ++c) krow[c] = fma(-m, pr[c], krow[c]);
        y = fma(-m, pr[LPE], y);
        krow[j] = act ? m : krow[j];
      }
    }
  };
  // forward substitution with the stored multipliers (refinement): v -> L^-1 P v, row-indexed
  auto forward = [&](double v) -> double {
    int ns = n;
    asm volatile("" : "+s"(ns));
#pragma unroll
    for (int j = 0; j < LPE; ++j) {
      if (j < ns) {
        const int P = __shfl(pvec, gbase + j);
        const double vj = group_bcast<LPE>(v, P, gbase);
        const double m = (pivstep > j && pivstep < 0x7fff) ? krow[j] : 0.0;
        v = fma(-m, vj, v);
      }
    }
    return v;
  };
  // back substitution U x = v; returns x_r (column-indexed: lane r gets the unknown of column r)
  auto backward = [&](double v) -> double {
    double xr = 0.0;
    int ns = n;
    asm volatile("" : "+s"(ns));
#pragma unroll
    for (int j = LPE - 1; j >= 0; --j) {
      if (j < ns) {
        const int P = __shfl(pvec, gbase + j);
        const double xj = group_bcast<LPE>(v * rinv_own, P, gbase);     // lane P: pivstep == j
        if (r == j) xr = xj;
        const double u = (pivstep < j) ? krow[j] : 0.0;
        v = fma(-u, xj, v);
      }
    }
    return xr;
  };
  // dw_p = sum_r A'[r][p] dl_r for p < MP -> lane p (LPE >= M); through LDS, row stride LPE + 1
  auto at_times = [&](double dl) -> double {
    wave_lds_sync();
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) Zs[pp * L::kTStride + r] = arow[pp] * dl;
    wave_lds_sync();
    double acc = 0.0;
    if (r < MP) {
      const double* __restrict__ row = Zs + r * L::kTStride;
#pragma unroll
      for (int c = 0; c < LPE; ++c) acc += row[c];        // (padding rows contribute exact zeros)
    }
    return acc;
  };

  eliminate();
  double lam = dr * backward(y);                   // unscale: lam = D x
  if (!is_pt) lam = 0.0;
  double wv = at_times(lam);                       // lane p: bubble part of w_p (carried)
  // equilibrated residual of the pair (lam, w) in operator form, and its squared norm
  auto residual = [&](double lam_, double wv_, double& nrm) -> double {
    wave_lds_sync();
    if (r < MP) Ws[r] = wv_;
    wave_lds_sync();
    // f' - eps lam - A' w in compensated (double-double) arithmetic: the terms are ~|A'| |w| while
    // the residual of a converged pair is ~eps |lam|; with a plain float64 sum the refinement stalls
    // at ~u |A'| / sqrt(eps) (1e-10 at gamma scl^4 ~ 1e15), with error-free products and sums at ~u.
    double hi = fp, lo = 0.0;
    auto acc = [&](double x, double y) {          // (hi, lo) += x * y exactly
      const double pr_ = x * y;
      const double pe = fma(x, y, -pr_);
      const double t = hi + pr_;
      const double bb = t - hi;
      lo += ((hi - (t - bb)) + (pr_ - bb)) + pe;
      hi = t;
    };
    acc(-eps, lam_);
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) acc(-arow[pp], Ws[pp]);
    double res = hi + lo;
    res = is_pt ? dr * res : 0.0;
    nrm = group_sum<LPE>(res * res);
    return res;
  };
  // Safeguarded refinement: a step is kept only if it lowers the residual norm.  On coarse meshes
  // at degree 32 (cond ~ 1e21) the factors are noise in the near-null directions of K and an
  // unguarded step can amplify rounding in A'^T dlam (measured: 4e-12 -> 2e-7 -> 1e-2 over four
  // unguarded steps on one element of the 24-element fixture mesh; every other element improves).
  double nrm = 0.0;
  double res = residual(lam, wv, nrm);
#pragma unroll 1
  for (int it = 0; it < nrefine; ++it) {
    double dl = dr * backward(forward(res));
    if (!is_pt) dl = 0.0;
    const double lam_c = lam + dl;
    const double wv_c = wv + at_times(dl);
    double nrm_c = 0.0;
    const double res_c = residual(lam_c, wv_c, nrm_c);
    const bool better = nrm_c < 4.0 * nrm;       // (NaN: rejected; a step may raise the norm 2x: at the
                                                 // rounding floor the norm is noise, a diverging step gains 100x)
    // a rejected step would be recomputed identically, a step that gains less than 2x in the norm
    // is at the rounding floor: the wave stops refining once none of its elements gains any more
    const bool gains = better && (nrm_c < 0.25 * nrm);
    lam = better ? lam_c : lam;
    wv = better ? wv_c : wv;
    res = better ? res_c : res;
    nrm = better ? nrm_c : nrm;
    if (!__any(gains)) break;
  }
  // ---- w = w_bc + w', re-projected onto the boundary rows -------------------------------------
  double wp = 0.0;
  if (r < MP) wp = fma(LaS[r], g0, fma(LbS[r], g1, wv));
  const double ra = group_sum<LPE>((r < MP) ? LaS[r] * wp : 0.0) - gl;      // B w - g
  const double rb = group_sum<LPE>((r < MP) ? LbS[r] * wp : 0.0) - gr;
  const double k0 = fma(qi00, ra, qi01 * rb), k1 = fma(qi01, ra, qi11 * rb);
  if (r < MP) wp = fma(-k1, LbS[r], fma(-k0, LaS[r], wp));
  const double bad = group_sum<LPE>((r < M && !(fabs(wp) < 1.0e300)) ? 1.0 : 0.0);
  const bool ok = bad == 0.0;
  if (live) {
    double* const Wrow = p.W + e * (p.ldw ? p.ldw : (int64_t)M);
    double out = wp;
    if (!ok) out = (r == 0) ? 0.5 * (gl + gr) : (r == 1) ? 0.5 * (gr - gl) : 0.0;
    if (r < M) Wrow[r] = out;
    if (r == 0) {
      if (p.status) p.status[e] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
      if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
    }
  }
}

// =============================================================================================
// n > 32 or M > 32 (BASELINE config 4: 64 points), round 3: one element per wave, TWO waves per SIMD.
// The generic kernel above at LPE = 64 holds a row of K (128 VGPRs) AND a row of A' (72) per lane, sends
// the pivot row through LDS and searches the pivot with twelve 64-bit ds_bpermute: 256 VGPRs + 136 AGPRs,
// one wave per SIMD, 23 400 instructions per element at the lone-wave issue rate (one instruction per
// ~9 cycles): 12.4 ms per 1e5 elements (here: 17 500 at two waves per SIMD, 4.6 ms).  Same algorithm here -- block-pivoted boundary rows, Jacobi
// equilibration, LU with PARTIAL PIVOTING, safeguarded refinement with the compensated operator residual --
// arranged for the register file:
//   * the rows of A' live in LDS (row stride 33, odd: conflict-free for a lane reading its own row and for
//     a lane reading a column); the Gram takes four columns of A' at a time against 64 accumulators;
//   * the pivot row never travels: every lane reads it out of lane P into the SGPR operands of its FMAs
//     (v_readfirstlane under EXEC = {P}, six columns per asm statement: 4.6 ms; with v_readlane, which
//     costs twice the issue time, 5.2 ms), no LDS round trip and no wave sync inside an elimination step
//     (measured alternative, DESIGN section 3.2b: the row through LDS with the publish overlapped by the
//     next pivot search, three or sixteen broadcast reads in flight: 6.0 / 6.1 ms);
//   * pivot search by a DPP max-reduction on 32-bit keys (exponent + 19 mantissa bits + lane);
//   * <= 256 registers, so two waves share a SIMD and fill each other's issue gaps; 19 KB of LDS per wave,
//     eight waves per CU.
// (Measured and rejected in between, DESIGN section 3.2b: one element per WORKGROUP of four waves, 16 columns
// per wave -- 14.2 ms with one barrier per elimination step, 14.6 ms blocked by column panel; the chains of
// the substitutions and the residual are serial across the waves and the barriers couple their issue.)
// =============================================================================================
namespace {
// maximum of a 32-bit key over the 64 lanes, as a wave-uniform value: four DPP row shifts leave the
// maximum of every 16-lane row in its lane 15; the four row results are combined on the scalar unit.
// ~12 instructions and no LDS round trip (the 64-bit shuffle reduction above: 12 ds_bpermute).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
  const unsigned r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  return max(max(r0, r1), max(r2, r3));
}
// The values of six doubles in lane P, as scalars.  v_readfirstlane_b32 under EXEC = {P} costs half the issue
// time of v_readlane_b32 with an SGPR lane select (1.9 against 3.6 ns per wave instruction and SIMD,
// scripts/micro/valu_rates.hip); EXEC is saved and restored inside the statement (nothing else is
// clobbered: the mask is built by s_bfm_b64, which does not write SCC -- the compiler keeps loop
// comparisons live in SCC across the statement), and the trailing s_nop covers the two wait states between a
// VALU write of an SGPR and a VALU read of it.  For wave-uniform control flow only (EXEC is replaced, not masked).
__device__ __forceinline__ void lane_values6(const double (&in)[6], double (&out)[6], int P) {
  unsigned lo[6], hi[6], olo[6], ohi[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, in[q]);
    lo[q] = (unsigned)u;
    hi[q] = (unsigned)(u >> 32);
  }
  unsigned long long sv;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_bfm_b64 exec, 1, %[P]\n\t"            // one bit at position P; unlike s_lshl_b64 it leaves SCC alone
      "v_readfirstlane_b32 %[a0], %[x0]\n\t"
      "v_readfirstlane_b32 %[b0], %[y0]\n\t"
      "v_readfirstlane_b32 %[a1], %[x1]\n\t"
      "v_readfirstlane_b32 %[b1], %[y1]\n\t"
      "v_readfirstlane_b32 %[a2], %[x2]\n\t"
      "v_readfirstlane_b32 %[b2], %[y2]\n\t"
      "v_readfirstlane_b32 %[a3], %[x3]\n\t"
      "v_readfirstlane_b32 %[b3], %[y3]\n\t"
      "v_readfirstlane_b32 %[a4], %[x4]\n\t"
      "v_readfirstlane_b32 %[b4], %[y4]\n\t"
      "v_readfirstlane_b32 %[a5], %[x5]\n\t"
      "v_readfirstlane_b32 %[b5], %[y5]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "s_nop 1"
      : [sv] "=&s"(sv), [a0] "=&s"(olo[0]), [b0] "=&s"(ohi[0]), [a1] "=&s"(olo[1]), [b1] "=&s"(ohi[1]),
        [a2] "=&s"(olo[2]), [b2] "=&s"(ohi[2]), [a3] "=&s"(olo[3]), [b3] "=&s"(ohi[3]), [a4] "=&s"(olo[4]),
        [b4] "=&s"(ohi[4]), [a5] "=&s"(olo[5]), [b5] "=&s"(ohi[5])
      : [P] "s"(__builtin_amdgcn_readfirstlane(P)), [x0] "v"(lo[0]), [y0] "v"(hi[0]), [x1] "v"(lo[1]), [y1] "v"(hi[1]), [x2] "v"(lo[2]),
        [y2] "v"(hi[2]), [x3] "v"(lo[3]), [y3] "v"(hi[3]), [x4] "v"(lo[4]), [y4] "v"(hi[4]), [x5] "v"(lo[5]),
        [y5] "v"(hi[5]));
#pragma unroll
  for (int q = 0; q < 6; ++q) out[q] = __builtin_bit_cast(double, ((unsigned long long)ohi[q] << 32) | olo[q]);
}
constexpr int kW64MP = 36;             // padded number of Legendre coefficients in the row build (M <= 33)
constexpr int kW64MU = kDualMaxM;      // columns of A' that can be non-zero
constexpr int kW64ZS = 33;             // row stride of A' in LDS
}  // namespace

template <int RHS, bool VC>
__global__ __launch_bounds__(64, 2) void enhance_dual_w64_kernel(EnhanceArgs p, int nrefine) {
  constexpr int MP = kW64MP, MU = kW64MU, ZS = kW64ZS, N = 64;
  __shared__ double Zs[N * ZS];
  __shared__ double LaS[MP], LbS[MP];
  __shared__ double Ds[N];              // equilibration scales
  __shared__ double Bs[N];              // broadcast operand of A'^T dl
  __shared__ double Ws[MP];             // carried w for the residual
  const int l = threadIdx.x & 63;
  const int M = p.M, n = p.n;
  const int64_t e = blockIdx.x;                     // grid = ne waves
  const double a = p.x[e];
  const double b = p.x[e + 1];
  const int64_t eg = e + p.elem_offset;
  const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[e];
  const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[e + 1];
  const double gamma = p.gamma_values ? p.gamma_values[e] : p.gamma;
  const DomainMap dm = map_params(a, b);
  const double step = dm.oldlen / (double)(n - 1);
  const double scl2 = dm.scl * dm.scl;
  const double inv_scl2 = rcp_newton(scl2);
  const double eps = rcp_newton(gamma * (scl2 * scl2));
  const bool is_pt = l < n;

  // ---- boundary rows, Q = B B^T, row l of A, projection, f' -> LDS ---------------------------------
  double qi00, qi01, qi11, g0, g1, fp;
  {
    const double ta = dm.off + dm.scl * a;
    const double tb_ = dm.off + dm.scl * b;
    if (l < 2) {
      const double t = l == 0 ? ta : tb_;
      double* const dst = l == 0 ? LaS : LbS;
      double Lm2 = 1.0, Lm1 = t;
      dst[0] = 1.0;
      dst[1] = (M > 1) ? t : 0.0;
#pragma unroll
      for (int pp = 2; pp < MP; ++pp) {
        const double Lp = fma(alL(pp) * t, Lm1, -(beL(pp) * Lm2));
        dst[pp] = (pp < M) ? Lp : 0.0;
        Lm2 = Lm1;
        Lm1 = Lp;
      }
    }
    wave_lds_sync();
    double q00 = 0.0, q01 = 0.0, q11 = 0.0;
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) {
      const double la = LaS[pp], lb = LbS[pp];
      q00 = fma(la, la, q00);
      q01 = fma(la, lb, q01);
      q11 = fma(lb, lb, q11);
    }
    const double qdet = rcp_newton(fma(q00, q11, -(q01 * q01)));
    qi00 = q11 * qdet;
    qi01 = -q01 * qdet;
    qi11 = q00 * qdet;
    g0 = fma(qi00, gl, qi01 * gr);
    g1 = fma(qi01, gl, qi11 * gr);
    const double xk = (is_pt && l == n - 1) ? b : (double)(is_pt ? l : 0) * step + a;
    const double tk = dm.off + dm.scl * xk;
    double ftil = 0.0;
    if (is_pt) {
      double fk;
      if constexpr (RHS == LSSVR_RHS_SIN) fk = p.rhs_amp * sin_reduced(p.rhs_omega * xk);
      else fk = p.rhs_values[e * p.tab_es + l * p.tab_ps];
      ftil = fk * inv_scl2;
    }
    double arow[MP];
    {
      const double sgn = is_pt ? -1.0 : 0.0;
      double ak = 1.0, bk = 0.0;
      if constexpr (VC) {
        if (is_pt) {
          ak = p.a_values[e * p.tab_es + l * p.tab_ps];
          bk = p.da_values[e * p.tab_es + l * p.tab_ps] * (0.5 * dm.oldlen);
        }
      }
      double q2 = 0.0, q1 = 0.0, r2 = 0.0, r1 = 0.0;
#pragma unroll
      for (int pp = 0; pp < MP; ++pp) {
        double d2 = 0.0, d1 = 0.0;
        if (pp >= 2) {
          const int m = pp - 2;
          d2 = (m == 0) ? 3.0 : (m == 1) ? 15.0 * tk : fma(al2(m) * tk, q1, -(be2(m) * q2));
          q2 = q1;
          q1 = d2;
        }
        if constexpr (VC) {
          if (pp >= 1) {
            const int m = pp - 1;
            d1 = (m == 0) ? 1.0 : (m == 1) ? 3.0 * tk : fma(al1(m) * tk, r1, -(be1(m) * r2));
            r2 = r1;
            r1 = d1;
          }
        }
        const double v = VC ? fma(ak, d2, bk * d1) : d2;
        arow[pp] = (pp < M) ? sgn * v : 0.0;
      }
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) {
      s0 = fma(arow[pp], LaS[pp], s0);
      s1 = fma(arow[pp], LbS[pp], s1);
    }
    const double c0 = fma(qi00, s0, qi01 * s1), c1 = fma(qi01, s0, qi11 * s1);
#pragma unroll
    for (int pp = 0; pp < MU; ++pp) Zs[l * ZS + pp] = fma(-c1, LbS[pp], fma(-c0, LaS[pp], arow[pp]));   // (columns >= M <= 33: zero)
    fp = ftil - fma(s0, g0, s1 * g1);
  }
  wave_lds_sync();

  // ---- K = A' A'^T, row l: four columns of A' at a time against all 64 rows -----------------------------
  // krow[c] += sum over the chunk of A'[l][pp] A'[c][pp]: the lane's own four values in registers, the other
  // operand by broadcast reads; 64 independent accumulation chains (one per column) keep the FP64 pipe busy
  // and every entry still sums its products in the order pp = 0, 1, 2, ...  Columns are skipped in blocks of
  // eight beyond n (rows >= n of A' are zero anyway).
  double krow[N];
  double kdiag = 0.0;
  {
    int ns = n;
    asm volatile("" : "+s"(ns));
#pragma unroll
    for (int c = 0; c < N; ++c) krow[c] = 0.0;
    constexpr int kQ = 4;
#pragma unroll
    for (int p0 = 0; p0 < MU; p0 += kQ) {
      double ar[kQ];
#pragma unroll
      for (int q = 0; q < kQ; ++q)
        if (p0 + q < MU) {
          ar[q] = Zs[l * ZS + p0 + q];
          kdiag = fma(ar[q], ar[q], kdiag);
        }
#pragma unroll
      for (int c0 = 0; c0 < N; c0 += 8) {
        if (c0 < ns) {
#pragma unroll
          for (int c = c0; c < c0 + 8; ++c) {
            const double* __restrict__ zc = Zs + c * ZS;
#pragma unroll
            for (int q = 0; q < kQ; ++q)
              if (p0 + q < MU) krow[c] = fma(ar[q], zc[p0 + q], krow[c]);
          }
        }
      }
    }
  }
  // (K + eps I), Jacobi equilibration (kdiag is K_ll with the rounding of krow[l]: same products, same order)
  const double dr = is_pt ? rsqrt_newton(kdiag + eps) : 1.0;
  Ds[l] = dr;
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double kc = krow[c] + ((c == l) ? eps : 0.0);
    krow[c] = (c < n && is_pt) ? kc * dr * Ds[c] : ((c == l) ? 1.0 : 0.0);
  }

  // ---- LU with partial pivoting, rhs carried; refinement re-uses the factors ---------------------------
  // pivstep: step at which this row was the pivot (-1: still active).  After the elimination the row holds
  // U[pivstep][c] for c >= pivstep and the multipliers L for c < pivstep.  Only the STEPS are guarded by n
  // (columns >= n are exact zeros in every row).
  int pivstep = is_pt ? -1 : 0x7fff;
  double rinv_own = 1.0;               // 1 / pivot of the step at which this row was the pivot
  int pvec = 0;                        // lane j keeps the pivot lane of step j
  double y = dr * fp;
  {
    int ns = n;
    asm volatile("" : "+s"(ns));
    static_for<0, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j < ns) {
        // pivot search on the top 31 bits of |a_lj| (exponent + 19 mantissa bits), lane in the low six
        const bool active = pivstep < 0;
        const unsigned hi = (unsigned)(__builtin_bit_cast(unsigned long long, fabs(krow[j])) >> 32);
        const unsigned key = active ? (0x80000000u | ((hi >> 1) & ~63u) | (unsigned)l) : 0u;
        const int P = (int)(wave_max_u32(key) & 63u);
        const double rinv = rcp_newton(readlane_f64(krow[j], P));
        if (l == P) {
          pivstep = j;
          rinv_own = rinv;
        }
        if (l == j) pvec = P;
        const bool act = pivstep < 0;
        const double m = act ? krow[j] * rinv : 0.0;
        if (act) krow[j] = m;
        // the pivot row, read out of lane P in batches of kB columns: v_readfirstlane under EXEC = {P}
        // (lane_values6), v_readlane for the ragged last batch
        constexpr int kB = 6;
        const double nm = -m;
#pragma unroll
        for (int c0 = j + 1; c0 < N; c0 += kB) {
          double u[kB];
          if (c0 + kB <= N) {
            double in[kB];
#pragma unroll
            for (int q = 0; q < kB; ++q) in[q] = krow[c0 + q];
            lane_values6(in, u, P);
          } else {
#pragma unroll
            for (int q = 0; q < kB; ++q)
              if (c0 + q < N) u[q] = readlane_f64(krow[c0 + q], P);
          }
#pragma unroll
          for (int q = 0; q < kB; ++q)
            if (c0 + q < N) krow[c0 + q] = fma(nm, u[q], krow[c0 + q]);
        }
        y = fma(nm, readlane_f64(y, P), y);
      }
    });
  }
  // forward substitution with the stored multipliers (refinement): v -> L^-1 P v, row-indexed
  auto forward = [&](double v) -> double {
    int ns = n;
    asm volatile("" : "+s"(ns));
    static_for<0, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j < ns) {
        const int P = __builtin_amdgcn_readlane(pvec, j);
        const double vj = readlane_f64(v, P);      // (v_readfirstlane under EXEC = {P} is slower here: a serial chain)
        const double m = (pivstep > j) ? krow[j] : 0.0;   // (padding rows: krow[j] == 0 for j < n)
        v = fma(-m, vj, v);
      }
    });
    return v;
  };
  // back substitution U x = v; returns x_l (column-indexed: lane l gets the unknown of column l)
  auto backward = [&](double v) -> double {
    double xl = 0.0;
    int ns = n;
    asm volatile("" : "+s"(ns));
    static_for<0, N>([&](auto jc) {
      constexpr int j = N - 1 - decltype(jc)::value;
      if (j < ns) {
        const int P = __builtin_amdgcn_readlane(pvec, j);
        const double xj = readlane_f64(v * rinv_own, P);              // lane P: pivstep == j
        if (l == j) xl = xj;
        const double u = (pivstep < j) ? krow[j] : 0.0;
        v = fma(-u, xj, v);
      }
    });
    return xl;
  };
  // lane p < 33: sum_r A'[r][p] dl_r (column p of A' in LDS: consecutive lanes, consecutive addresses)
  auto at_times = [&](double dl) -> double {
    wave_lds_sync();
    Bs[l] = dl;
    wave_lds_sync();
    const int pc = l < MU ? l : 0;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int c = 0; c < N; c += 2) {                                  // (padding rows are zero)
      a0 = fma(Zs[c * ZS + pc], Bs[c], a0);
      a1 = fma(Zs[(c + 1) * ZS + pc], Bs[c + 1], a1);
    }
    return l < MU ? a0 + a1 : 0.0;
  };
  // equilibrated residual of the pair (lam, w) in operator form, and its squared norm:
  // f' - eps lam - A' w in compensated (double-double) arithmetic (see the generic kernel)
  auto residual = [&](double lam_, double wv_, double& nrm_) -> double {
    wave_lds_sync();
    if (l < MP) Ws[l] = wv_;
    wave_lds_sync();
    double hi = fp, lo = 0.0;
    auto acc = [&](double x, double yv) {          // (hi, lo) += x * y exactly
      const double pr_ = x * yv;
      const double pe = fma(x, yv, -pr_);
      const double t = hi + pr_;
      const double bb = t - hi;
      lo += ((hi - (t - bb)) + (pr_ - bb)) + pe;
      hi = t;
    };
    acc(-eps, lam_);
#pragma unroll
    for (int pp = 0; pp < MU; ++pp) acc(-Zs[l * ZS + pp], Ws[pp]);
    double r_ = hi + lo;
    r_ = is_pt ? dr * r_ : 0.0;
    nrm_ = group_sum<64>(r_ * r_);
    return r_;
  };

  double lam = dr * backward(y);                   // unscale: lam = D x
  if (!is_pt) lam = 0.0;
  double wv = at_times(lam);                       // lane p: bubble part of w_p (carried)
  double nrm = 0.0;
  double res = residual(lam, wv, nrm);
#pragma unroll 1
  for (int it = 0; it < nrefine; ++it) {           // safeguarded refinement, as in the generic kernel
    double dl = dr * backward(forward(res));
    if (!is_pt) dl = 0.0;
    const double lam_c = lam + dl;
    const double wv_c = wv + at_times(dl);
    double nrm_c = 0.0;
    const double res_c = residual(lam_c, wv_c, nrm_c);
    const bool better = nrm_c < 4.0 * nrm;
    const bool gains = better && (nrm_c < 0.25 * nrm);
    lam = better ? lam_c : lam;
    wv = better ? wv_c : wv;
    res = better ? res_c : res;
    nrm = better ? nrm_c : nrm;
    if (!gains) break;                              // (nrm is wave-uniform: one element per wave)
  }

  // ---- w = w_bc + w', re-projected onto the boundary rows; store ------------------------------------
  double wp = 0.0;
  if (l < MP) wp = fma(LaS[l], g0, fma(LbS[l], g1, wv));
  const double ra = group_sum<64>((l < MP) ? LaS[l] * wp : 0.0) - gl;
  const double rb = group_sum<64>((l < MP) ? LbS[l] * wp : 0.0) - gr;
  const double k0 = fma(qi00, ra, qi01 * rb), k1 = fma(qi01, ra, qi11 * rb);
  if (l < MP) wp = fma(-k1, LbS[l], fma(-k0, LaS[l], wp));
  const double bad = group_sum<64>((l < M && !(fabs(wp) < 1.0e300)) ? 1.0 : 0.0);
  const bool ok = bad == 0.0;
  double* const Wrow = p.W + e * (p.ldw ? p.ldw : (int64_t)M);
  double out = wp;
  if (!ok) out = (l == 0) ? 0.5 * (gl + gr) : (l == 1) ? 0.5 * (gr - gl) : 0.0;
  if (l < M) Wrow[l] = out;
  if (l == 0) {
    if (p.status) p.status[e] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
    if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
  }
}

static hipError_t launch_dual_w64(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.ne > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.ne), block(64);
  const int nref = kRefine;
  if (a.a_values) return launch(enhance_dual_w64_kernel<LSSVR_RHS_ARRAY, true>, grid, block, s, o, a, nref);
  if (a.rhs_id == LSSVR_RHS_SIN) return launch(enhance_dual_w64_kernel<LSSVR_RHS_SIN, false>, grid, block, s, o, a, nref);
  return launch(enhance_dual_w64_kernel<LSSVR_RHS_ARRAY, false>, grid, block, s, o, a, nref);
}

template <int LPE, int MP>
static hipError_t launch_dual(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  constexpr int EPW = 64 / LPE;
  const int64_t blocks = (a.ne + EPW - 1) / EPW;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(64);
  const int nref = kRefine;
  if (a.a_values) return launch(enhance_dual_kernel<LPE, MP, LSSVR_RHS_ARRAY, true>, grid, block, s, o, a, nref);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_dual_kernel<LPE, MP, LSSVR_RHS_SIN, false>, grid, block, s, o, a, nref);
  return launch(enhance_dual_kernel<LPE, MP, LSSVR_RHS_ARRAY, false>, grid, block, s, o, a, nref);
}

hipError_t enhance_dual(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M > kDualMaxM || a.n > kDualMaxN || a.elem_ids) return hipErrorInvalidValue;
  const int need = a.n > a.M ? a.n : a.M;
  if (need <= 16) return launch_dual<16, 16>(a, s, o);
  if (need <= 32) return launch_dual<32, 32>(a, s, o);
  // above 32 rows: the register-lean wave-per-element kernel (round 3; the generic kernel at LPE = 64, 12.4 ms
  // against 4.6 at BASELINE config 4, is no longer instantiated)
  return launch_dual_w64(a, s, o);
}

}  // namespace lssvr
