// Per-element LSSVR enhancement, large-degree path for POISSON rows (23 <= M <= 33; any M >= 2 on
// request): the Chebyshev-moment form of the Legendre Gram contraction (DESIGN.md section 2b,
// enhance_small_cheb.hpp) in a two-phase wave mapping.
//
//   G_ik = sum_k' T_i(t_k') T_k(t_k') = 1/2 (m_{i+k} + m_{|i-k|}),   m_d = sum_k' T_d(t_k'), d <= 60:
// the 31 x 31 Gram matrix of a degree-32 element is determined by 61 power sums -- O(n M) work per
// element instead of the O(n M^2) of the direct contraction (which enhance_large.hip runs on the f64
// matrix cores and which stays in the library for variable-coefficient rows and as
// LSSVR_SOLVER_PRIMAL_WAVE, the A/B reference: DESIGN.md section 7).
//
// One wave works on 16 consecutive elements.
//   Phase 1 (moments): lane = (element i1 = lane >> 2, point slice s = lane & 3).  The lane walks the
//     collocation points k = s, s+4, ... of its element: abscissa and t_k in numpy's arithmetic,
//     f(x_k) by the rotation-carried (sin, cos) pair with numpy's argument rounding restored to first
//     order (enhance_small_cheb.hpp), T_0..T_30 by the two-term recurrence with two registers of
//     state, and accumulates m_1..m_30 (adds), the squares T_j^2 (j = 16..30) and neighbour products
//     T_j T_{j+1} (j = 15..29) that give the upper moments (T_j^2 = (T_2j + T_0)/2, T_j T_{j+1} =
//     (T_{2j+1} + T_1)/2), and the right-hand side r_i = sum T_i phi: 91 accumulators, ~135
//     instructions per point, no cross-lane traffic in the loop.  Two xor-shuffles combine the four
//     slices; the 92 numbers per element go to LDS.  The phase is the same for every M (degree 30
//     always: the padding costs nothing that the old 32-column padding did not).
//   Phase 2 (solve), 8 rounds of two elements: lane (c = lane & 31, h = lane >> 5) builds column c
//     of  S2 = m_{i+c} + m_{|i-c|} + 2 eps (N + C_z^T C_z)  straight from the moments (two
//     conflict-free LDS reads, one table load and four FMAs per entry; first-order boundary rows
//     and compile-time ridge tables as in the lane kernel), the right-hand side rides along as
//     row / column 31, then the DPP-broadcast LDL^T of lssvr_wave.hpp (factor frozen in registers),
//     back-substitution, v = Y z through LDS, w_{0,1} by half-wave reductions.
// No MFMA, no operand staging, no accumulator transposition: the LDS traffic of the old front end
// (35 % of its LDS cycles were bank conflicts) is gone with it.
#include "cheb_tables.hpp"
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {

using namespace wave;

namespace {

constexpr int kEPW = 16;                 // elements per wave
constexpr int kReseedLarge = 64;         // rotation-carried rhs: re-seeded every 64 points of a slice
constexpr int kTop = 30;                 // highest Chebyshev degree of a row (M = 33)
constexpr int kMomStride = 64;           // m_0 .. m_60 (+ 3 pad)
constexpr int kRStride = 32;             // r_0 .. r_30 (+ 1 pad)
constexpr int kMomDoubles = kEPW * kMomStride;
constexpr int kRDoubles = kEPW * kRStride;
constexpr int kHalfDoubles2 = 3 * 64;    // per half: E (C rows, 64) | F (C_z rows, 64) | Z (64)
constexpr int kLdsDoubles = kMomDoubles + kRDoubles + 2 * kHalfDoubles2;

// zero-padded device copies of the constant tables for run-time indexed reads
struct ChebDeviceTables {
  double Y[32][64];      // Y[j][i] = cheb::kY[j][i]  (v = Y z), zero beyond 30
  double N[32][32];      // N = Y^T Y
  double alpha[32], b[32], slope[32];
};

constexpr ChebDeviceTables make_cheb_device_tables() {
  ChebDeviceTables t{};
  for (int j = 0; j < 31; ++j) {
    for (int i = 0; i < 31; ++i) {
      t.Y[j][i] = cheb::kY[j][i];
      t.N[j][i] = cheb::kN[j][i];
    }
    t.alpha[j] = cheb::kAlpha[j];
    t.b[j] = cheb::kB[j];
    t.slope[j] = cheb::kSlope[j];
  }
  return t;
}

__device__ const ChebDeviceTables kTab = make_cheb_device_tables();

}  // namespace

template <int RHS>
__global__ __launch_bounds__(64, 2) void enhance_large_cheb_kernel(EnhanceArgs p, unsigned nxcd) {
  __shared__ double2_t lds2[kLdsDoubles / 2];
  double* const lds = reinterpret_cast<double*>(lds2);
  double* const Mom = lds;
  double* const Rv = lds + kMomDoubles;
  const int lane = threadIdx.x & 63;
  const int M = p.M, MR = M - 2, n = p.n;

  // XCD-aware numbering: consecutive 16-element blocks go to consecutive workgroups OF ONE XCD, so
  // the lines of x / u are fetched by one L2 (enhance_large.hip measured 2.5 MB instead of 7.2 MB)
  const unsigned xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
  const int64_t blk = (int64_t)xcd * (gridDim.x / nxcd) + slot;
  const int64_t E0 = blk * kEPW;
  if (E0 >= p.ne) return;

  // =========================== phase 1: moments ============================================
  {
    const int i1 = lane >> 2, s = lane & 3;
    const int64_t e1 = (E0 + i1 < p.ne) ? E0 + i1 : p.ne - 1;     // tail: duplicates (never stored)
    int64_t id = e1;
    if (p.elem_ids) {
      id = p.elem_ids[e1];
      if (id < 0 || id >= p.ne_mesh) id = 0;                      // (reported in phase 2)
    }
    const double a = p.x[id];
    const double b = p.x[id + 1];
    const DomainMap dm = map_params(a, b);
    const double step = dm.oldlen / (double)(n - 1);
    const double hh = 0.5 * dm.oldlen;
    const double inv_scl2 = hh * hh;                              // 1 / scl^2 within 2 ulp

    double mom[kTop + 1], sq[15], nb[15], rr[kTop + 1];
#pragma unroll
    for (int d = 0; d <= kTop; ++d) mom[d] = rr[d] = 0.0;
#pragma unroll
    for (int j = 0; j < 15; ++j) sq[j] = nb[j] = 0.0;

    double rs = 0.0, rc = 1.0, sd = 0.0, cd = 1.0, th0 = 0.0, dth4 = 0.0, kappa = 0.0;
    if constexpr (RHS == LSSVR_RHS_SIN) {
      dth4 = 4.0 * (p.rhs_omega * step);
      sincos_tab(dth4, sd, cd, p.trig);
      kappa = -2.0 * (p.rhs_amp * inv_scl2);                      // phi2 = -2 f / scl^2
    }
    [[maybe_unused]] const double fscale = -2.0 * inv_scl2;
    const int iters = (n + 3) >> 2;
    for (int j0 = 0; j0 < iters; j0 += kReseedLarge) {
      if constexpr (RHS == LSSVR_RHS_SIN) {
        const double x0 = fma((double)(s + 4 * j0), step, a);
        th0 = p.rhs_omega * x0;
        sincos_tab(th0, rs, rc, p.trig);
        rs *= kappa;
        rc *= kappa;
      }
      const int j1 = min(j0 + kReseedLarge, iters);
      for (int j = j0; j < j1; ++j) {
        const int k = s + 4 * j;
        const bool valid = k < n;
        // np.linspace / mapdomain, two roundings each; last point = b
        const double xk = (k == n - 1) ? b : (double)k * step + a;
        const double tk = dm.off + dm.scl * xk;
        double phi2;
        if constexpr (RHS == LSSVR_RHS_SIN) {
          const double arg = p.rhs_omega * xk;
          const double delta = fma(-(double)(j - j0), dth4, arg - th0);
          phi2 = fma(rc, delta, rs);
          if (__any(valid && !(fabs(delta) < 1.0e-7))) phi2 = kappa * sin_tab(arg, p.trig);
          const double rs_next = fma(rs, cd, rc * sd);
          rc = fma(rc, cd, -(rs * sd));
          rs = rs_next;
        } else {
          phi2 = valid ? p.rhs_values[e1 * n + k] * fscale : 0.0;
        }
        // a slice past the end contributes zeros: T_0 = 0 makes the whole recurrence vanish
        const double seed = valid ? 1.0 : 0.0;
        phi2 *= seed;
        double Tm2 = seed, Tm1 = tk * seed;
        const double tt = tk + tk;
        rr[0] += phi2;
        mom[1] += Tm1;
        rr[1] = fma(Tm1, phi2, rr[1]);
#pragma unroll
        for (int d = 2; d <= kTop; ++d) {
          const double Td = fma(tt, Tm1, -Tm2);
          mom[d] += Td;
          rr[d] = fma(Td, phi2, rr[d]);
          if (d - 1 >= 15) nb[d - 1 - 15] = fma(Tm1, Td, nb[d - 1 - 15]);     // T_{d-1} T_d
          if (d >= 16) sq[d - 16] = fma(Td, Td, sq[d - 16]);                   // T_d^2
          Tm2 = Tm1;
          Tm1 = Td;
        }
      }
    }
    // combine the four slices of an element
#pragma unroll
    for (int d = 1; d <= kTop; ++d) {
      mom[d] += __shfl_xor(mom[d], 1);
      mom[d] += __shfl_xor(mom[d], 2);
    }
#pragma unroll
    for (int d = 0; d <= kTop; ++d) {
      rr[d] += __shfl_xor(rr[d], 1);
      rr[d] += __shfl_xor(rr[d], 2);
    }
#pragma unroll
    for (int j = 0; j < 15; ++j) {
      sq[j] += __shfl_xor(sq[j], 1);
      sq[j] += __shfl_xor(sq[j], 2);
      nb[j] += __shfl_xor(nb[j], 1);
      nb[j] += __shfl_xor(nb[j], 2);
    }
    // all 61 moments -> LDS (each slice lane writes a quarter), m_{2j} = 2 sum T_j^2 - m_0,
    // m_{2j+1} = 2 sum T_j T_{j+1} - m_1
    double* const mo = Mom + i1 * kMomStride;
    double* const ro = Rv + i1 * kRStride;
    const double m0 = (double)n;
    if (s == 0) {
      mo[0] = m0;
#pragma unroll
      for (int d = 1; d <= 15; ++d) mo[d] = mom[d];
#pragma unroll
      for (int i = 0; i <= 7; ++i) ro[i] = rr[i];
    } else if (s == 1) {
#pragma unroll
      for (int d = 16; d <= 30; ++d) mo[d] = mom[d];
#pragma unroll
      for (int i = 8; i <= 15; ++i) ro[i] = rr[i];
    } else if (s == 2) {
#pragma unroll
      for (int j = 16; j <= 30; ++j) mo[2 * j] = fma(2.0, sq[j - 16], -m0);
#pragma unroll
      for (int i = 16; i <= 23; ++i) ro[i] = rr[i];
    } else {
#pragma unroll
      for (int j = 15; j <= 29; ++j) mo[2 * j + 1] = fma(2.0, nb[j - 15], -mom[1]);
#pragma unroll
      for (int i = 24; i <= 30; ++i) ro[i] = rr[i];
      mo[61] = mo[62] = mo[63] = 0.0;
      ro[31] = 0.0;
    }
  }
  wave_lds_sync();

  // =========================== phase 2: solve, two elements per round ======================
  const int c = lane & 31, h = lane >> 5;
  double* const Eh = lds + kMomDoubles + kRDoubles + h * kHalfDoubles2;   // exact C rows (cold path)
  double* const Fh = Eh + 64;                                             // C_z rows (cold path)
  double* const Z = Fh + 64;                                              // 64 entries, [32, 64) stay 0
  Z[32 + c] = 0.0;
  const bool in_sys = c < MR;
  const double alpha_c = in_sys ? kTab.alpha[c] : 0.0;
  const double b_c = in_sys ? kTab.b[c] : 0.0;
  const double slope_c = kTab.slope[c < 31 ? c : 30];
  const bool c_even = (c & 1) == 0;

#pragma unroll 1
  for (int q = 0; q < kEPW / 2; ++q) {
    const int loc = 2 * q + h;
    const int64_t e_raw = E0 + loc;
    if (E0 + 2 * q >= p.ne) break;                       // (uniform: both halves past the end)
    bool live = e_raw < p.ne;
    const int64_t e = live ? e_raw : p.ne - 1;
    int64_t id = e;
    if (p.elem_ids) {
      id = p.elem_ids[e];
      if (id < 0 || id >= p.ne_mesh) {     // out-of-range id: nothing of the mesh is touched
        if (live && c == 0 && p.fail_count) atomicAdd(p.fail_count, 1);
        live = false;
        id = 0;
      }
    }
    const double a = p.x[id];
    const double b = p.x[id + 1];
    const int64_t eg = id + p.elem_offset;
    const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[id];
    const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[id + 1];
    const double inv_gamma = p.gamma_values ? rcp_newton(p.gamma_values[id]) : p.inv_gamma;
    const DomainMap dm = map_params(a, b);
    const double hh = 0.5 * dm.oldlen;
    const double inv_scl2 = hh * hh;
    const double eps2 = (2.0 * inv_gamma) * (inv_scl2 * inv_scl2);

    // ---- boundary rows to first order (enhance_small_cheb.hpp); exact recurrence when the wave
    // holds an element beyond the first-order range
    const double ta = dm.off + dm.scl * a;
    const double tb = dm.off + dm.scl * b;
    const double ea = 1.0 + ta, eb = 1.0 - tb;
    const double sig = 0.5 * (ea + eb), del = 0.5 * (ea - eb);
    const double amax = 0.5 * (double)((M - 1) * M);
    const bool slow = amax * fmax(fabs(ea), fabs(eb)) >= 1.0e-6;
    const bool any_slow = __any(slow);
    double idet = 0.5 * fma(sig, 1.0 + sig, 1.0);
    if (any_slow) idet = rcp_newton(tb - ta);
    const double d0 = (tb * gl - ta * gr) * idet;
    const double d1 = (gr - gl) * idet;
    // this lane's own (C0_c, C1_c) of w_{0,1} = d - C v (v-basis)
    double C0c, C1c;
    if (c_even) {
      C0c = fma(-slope_c, sig, 1.0);
      C1c = slope_c * del;
    } else {
      C0c = (slope_c - 1.0) * del;
      C1c = fma(-(slope_c - 1.0), sig, 1.0);
    }
    double C0z = 0.0, C1z = 0.0;                     // cold path: this lane's column of C_z = C Y
    if (any_slow) {
      // exact L_{c+2}(ta), L_{c+2}(tb) by the Legendre recurrence, latch at degree c + 2
      double am1 = 1.0, a0 = ta, bm1 = 1.0, b0 = tb, La2 = 0.0, Lb2 = 0.0;
      for (int m = 1; m <= MR; ++m) {
        const double inv = 1.0 / (double)(m + 1);
        const double a1 = ((double)(2 * m + 1) * ta * a0 - (double)m * am1) * inv;
        const double b1 = ((double)(2 * m + 1) * tb * b0 - (double)m * bm1) * inv;
        am1 = a0; a0 = a1;
        bm1 = b0; b0 = b1;
        if (m == c + 1) {
          La2 = a1;
          Lb2 = b1;
        }
      }
      C0c = (tb * La2 - ta * Lb2) * idet;
      C1c = (Lb2 - La2) * idet;
      if (!in_sys) C0c = C1c = 0.0;
      wave_lds_sync();
      Eh[2 * c] = C0c;
      Eh[2 * c + 1] = C1c;
      wave_lds_sync();
      // C_z[., c] = sum_{j <= c, j = c mod 2} C[., j] Y[j][c]
      for (int t = 0; t < 16; ++t) {
        const int j = c - 2 * t;
        if (j >= 0 && in_sys) {
          const double y = kTab.Y[j][c];
          C0z = fma(Eh[2 * j], y, C0z);
          C1z = fma(Eh[2 * j + 1], y, C1z);
        }
      }
      Fh[2 * c] = C0z;
      Fh[2 * c + 1] = C1z;
      wave_lds_sync();
    }
    if (!in_sys) C0c = C1c = 0.0;

    // ---- right-hand side entry of this lane's column, then column c of S2 -------------------------
    const double* const mo = Mom + loc * kMomStride;
    const double* const ro = Rv + loc * kRStride;
    double rhs_c;
    if (any_slow) {
      rhs_c = fma(eps2, fma(C0z, d0, C1z * d1), ro[c]);
    } else {
      const double e_d = eps2 * (c_even ? d0 : d1);
      const double q_c = eps2 * (c_even ? fma(del, d1, -(sig * d0)) : fma(del, d0, -(sig * d1)));
      rhs_c = fma(b_c, q_c, fma(alpha_c, e_d, ro[c]));
    }
    if (!in_sys) rhs_c = 0.0;
    wave_lds_sync();                 // previous round's readers of Z are done
    Z[c] = rhs_c;
    wave_lds_sync();
    double col[kLP];
    {
      const double es = eps2 * sig, ed = eps2 * del;
      // coefficients of alpha_i / b_i in the first-order ridge, by the parity of the ROW
      const double u1 = fma(eps2, alpha_c, -(es * b_c)), u2 = -(es * alpha_c);   // same parity
      const double u3 = ed * b_c, u4 = ed * alpha_c;                             // opposite parity
      const double Xe = c_even ? u1 : u3, Ye = c_even ? u2 : u4;                 // even rows
      const double Xo = c_even ? u3 : u1, Yo = c_even ? u4 : u2;                 // odd rows
      const int cc = c < 31 ? c : 0;
#pragma unroll
      for (int i = 0; i < kLP - 1; ++i) {
        const int lo_idx = (i >= cc) ? i - cc : cc - i;
        double v = mo[i + cc] + mo[lo_idx];
        const double nic = kTab.N[i][cc];
        if (any_slow) {
          v = fma(eps2, fma(Fh[2 * i], C0z, fma(Fh[2 * i + 1], C1z, nic)), v);
        } else {
          v = fma(eps2, nic, v);
          v = fma(cheb::kAlpha[i], (i & 1) ? Xo : Xe, v);
          v = fma(cheb::kB[i], (i & 1) ? Yo : Ye, v);
        }
        // lane 31 carries the right-hand side as a column; padding columns are inert
        col[i] = (c == kRhsRow) ? Z[i] : (in_sys ? v : 0.0);
      }
      col[kRhsRow] = rhs_c;          // ... and every column carries it as row 31
    }

    // ---- LDL^T factor + solve of the MR x MR block ------------------------------------------
    bool lane_ok;
    const double z = ldlt_solve_dpp(col, Z, c, MR, lane_ok);
    // v = Y z (bubble Legendre coefficients): v_j = sum_{i >= j, i = j mod 2} Y[j][i] z_i
    wave_lds_sync();
    Z[c] = in_sys ? z : 0.0;
    wave_lds_sync();
    double v = 0.0;
    {
      const int jr = c < 31 ? c : 30;
#pragma unroll
      for (int t = 0; t < 16; ++t) v = fma(kTab.Y[jr][c + 2 * t], Z[c + 2 * t], v);
      if (!in_sys) v = 0.0;
    }
    const double w0 = d0 - half_sum(C0c * v);
    const double w1 = d1 - half_sum(C1c * v);
    const double bad = half_sum((lane_ok && fabs(v) < 1.0e300) ? 0.0 : 1.0);
    const bool ok = (bad == 0.0) && (fabs(w0) < 1e300) && (fabs(w1) < 1e300);

    // ---- store: lane c -> W[e][c+2]; lane 0 also writes w0, w1 -----------------------------
    if (live) {
      double* const Wrow = p.W + id * (p.ldw ? p.ldw : (int64_t)M);
      if (in_sys) Wrow[c + 2] = ok ? v : 0.0;
      if (c == 0) {
        Wrow[0] = ok ? w0 : 0.5 * (gl + gr);
        Wrow[1] = ok ? w1 : 0.5 * (gr - gl);
        if (p.status) p.status[id] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
        if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
      }
    }
  }
}

hipError_t enhance_large_cheb(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M - 2 + 1 > kLP || a.a_values) return hipErrorInvalidValue;
  const unsigned nxcd = xcd_count();
  int64_t blocks = (a.ne + kEPW - 1) / kEPW;
  blocks = (blocks + nxcd - 1) / nxcd * nxcd;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(64);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_large_cheb_kernel<LSSVR_RHS_SIN>, grid, block, s, o, a, nxcd);
  return launch(enhance_large_cheb_kernel<LSSVR_RHS_ARRAY>, grid, block, s, o, a, nxcd);
}

}  // namespace lssvr
