// Per-element LSSVR enhancement, DUAL form -- the formulation BASELINE.json's north_star
// names: the Legendre kernel Gram matrix over the collocation (and boundary) rows and the
// dense solve (K + I/gamma) alpha = y (dual form):
//
//   Z = [Ahat; B]  ((n+2) x M),   K = Z Z^T + diag(eps I_n, 0, 0),   K [lam; mu] = [ftil; g],
//   w = Z^T [lam; mu],            eps = 1 / (gamma scl^4),  Ahat = -L''(t_k), B = L(t_a), L(t_b).
//
// K[i][j] = sum_p Z_ip Z_jp is exactly north_star's K(x_i,x_j) = sum_p phi_p(x_i) phi_p(x_j)
// with the PDE rows' feature map phi_p = -L_p''.  Same wave mapping as enhance_large.hip
// (two elements per wave, f64 MFMA for the Gram, LDL^T in registers) with the roles of the
// two indices swapped: the contraction runs over the M Legendre indices.
//
// Accuracy gate: K is well conditioned when n + 2 <= M (few collocation points,
// the regime the primal normal equations cannot resolve) and badly conditioned otherwise
// (cond ~ |A A^T| / eps): there the primal solver is the accurate one (SURVEY.md App. B.3:
// dual LU 1.8e-10 on config 1).  Limits: n <= 29, M <= 32, Poisson rows.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {
using namespace wave;

template <int RHS>
__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void enhance_dual_kernel(EnhanceArgs p,
                                                                              RecTables tb) {
  __shared__ double2_t lds2[kWavesPerBlock * kWaveDoubles / 2];
  double* const lds = reinterpret_cast<double*>(lds2);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  double* const VtA = lds + wave * kWaveDoubles;
  double* const VtB = VtA + kHalfDoubles;
  double* const Vt = h ? VtB : VtA;
  double* const Gb = Vt;
  double* const Lm = Vt;
  double* const Zs = Vt + kVDoubles;             // 32-entry broadcast vector
  const int M = p.M, n = p.n;
  const int NS = n + 2;                          // system size, <= 31
  const bool need11 = NS > 16;
  const int64_t npair = (p.ne + 1) >> 1;

  for (int64_t pr = (int64_t)blockIdx.x * kWavesPerBlock + wave; pr < npair;
       pr += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t e_raw = 2 * pr + h;
    const bool live = e_raw < p.ne;
    const int64_t e = live ? e_raw : p.ne - 1;
    const double a = p.x[e];
    const double b = p.x[e + 1];
    const int64_t eg = e + p.elem_offset;
    const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[e];
    const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[e + 1];
    const DomainMap dm = map_params(a, b);
    const double step = dm.oldlen / (double)(n - 1);
    const double scl2 = dm.scl * dm.scl;
    const double inv_scl2 = rcp_newton(scl2);
    const double eps = rcp_newton(p.gamma * (scl2 * scl2));

    // ---- row c of Z: collocation point c (< n), boundary rows n, n+1, zero padding above ----
    const bool is_pt = c < n;
    const bool is_bc = (c == n) || (c == n + 1);
    const double xk = linspace_at(a, b, dm.oldlen, step, is_pt ? c : 0, n);
    const double xr = is_pt ? xk : (c == n ? a : b);
    const double t = dm.off + dm.scl * xr;       // t_k, or t(xmin) / t(xmax) for the boundary rows
    double rhs = 0.0;
    if (is_pt) {
      double fk;
      if constexpr (RHS == LSSVR_RHS_SIN) fk = p.rhs_amp * sin_reduced(p.rhs_omega * xk);
      else fk = p.rhs_values[e * n + c];
      rhs = fk * inv_scl2;
    } else if (c == n) {
      rhs = gl;
    } else if (c == n + 1) {
      rhs = gr;
    }
    const double s_pt = is_pt ? -1.0 : 0.0;      // Ahat = -L''
    const double s_bc = is_bc ? 1.0 : 0.0;

    // Z[c][pp] for pp = 0..M-1: value = s_pt * L''_pp(t) + s_bc * L_pp(t); `emit` consumes it
    auto for_each_entry = [&](auto&& emit) {
      double Lm2 = 1.0, Lm1 = t;                 // L_0, L_1
      double q2 = 0.0, q1 = 0.0;                 // L''_{pp-2}, L''_{pp-1} seeds (L''_0 = L''_1 = 0)
      emit(0, s_bc * 1.0);
      if (M > 1) emit(1, s_bc * t);
      for (int pp = 2; pp < M; ++pp) {
        const double Lp = fma(tb.alL[pp] * t, Lm1, -(tb.beL[pp] * Lm2));
        double q;
        if (pp == 2) q = 3.0;
        else if (pp == 3) q = 15.0 * t;
        else q = fma(tb.al2[pp - 2] * t, q1, -(tb.be2[pp - 2] * q2));
        emit(pp, fma(s_pt, q, s_bc * Lp));
        Lm2 = Lm1;
        Lm1 = Lp;
        q2 = q1;
        q1 = q;
      }
    };

    wave_lds_sync();
    for_each_entry([&](int pp, double v) { Vt[c * kSV + pp] = v; });
    for (int pp = M; pp < kLP; ++pp) Vt[c * kSV + pp] = 0.0;
    wave_lds_sync();

    // ---- K = Z Z^T on the matrix cores (contraction over the Legendre index) ----------------
    double4_t accA00 = {0, 0, 0, 0}, accA10 = {0, 0, 0, 0}, accA11 = {0, 0, 0, 0};
    double4_t accB00 = {0, 0, 0, 0}, accB10 = {0, 0, 0, 0}, accB11 = {0, 0, 0, 0};
    {
      const int ar = (lane & 15) * kSV + (lane >> 4);
#pragma unroll
      for (int s = 0; s < kLP / 4; ++s) {
        const double a0 = VtA[ar + 4 * s];
        const double b0 = VtB[ar + 4 * s];
        accA00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, accA00, 0, 0, 0);
        accB00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b0, accB00, 0, 0, 0);
        if (need11) {
          const double a1 = VtA[ar + 16 * kSV + 4 * s];
          const double b1 = VtB[ar + 16 * kSV + 4 * s];
          accA10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, accA10, 0, 0, 0);
          accB10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, accB10, 0, 0, 0);
          accA11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, accA11, 0, 0, 0);
          accB11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, accB11, 0, 0, 0);
        }
      }
    }
    wave_lds_sync();
    {
      const int col = lane & 15, rb = lane >> 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rb + 4 * q;
        VtA[row * kSG + col] = accA00[q];
        VtA[(16 + row) * kSG + col] = accA10[q];
        VtA[col * kSG + 16 + row] = accA10[q];
        VtA[(16 + row) * kSG + 16 + col] = accA11[q];
        VtB[row * kSG + col] = accB00[q];
        VtB[(16 + row) * kSG + col] = accB10[q];
        VtB[col * kSG + 16 + row] = accB10[q];
        VtB[(16 + row) * kSG + 16 + col] = accB11[q];
      }
    }
    wave_lds_sync();
    // + eps on the collocation block's diagonal (the "+ I/gamma"), rhs as row 31
    Gb[c * kSG + c] += is_pt ? eps : 0.0;
    Gb[kRhsRow * kSG + c] = rhs;           // row 31 of every column ...
    Gb[c * kSG + kRhsRow] = rhs;           // ... and, symmetrically, column 31 (lane 31's column)
    wave_lds_sync();
    double col[kLP];
#pragma unroll
    for (int i = 0; i < kLP; ++i) col[i] = Gb[i * kSG + c];
    wave_lds_sync();

    bool piv_ok;
    const double sol = ldlt_solve(col, Lm, Zs, c, NS, piv_ok);    // lam_c (c < n), mu (c = n, n+1)

    // ---- w = Z^T [lam; mu]: contributions T[pp][c] = Z[c][pp] sol_c, then lane pp sums row pp -----
    wave_lds_sync();
    for_each_entry([&](int pp, double v) { Vt[pp * kSV + c] = v * sol; });
    wave_lds_sync();
    double w = 0.0;
    if (c < M) {
#pragma unroll
      for (int i = 0; i < kLP; i += 2) {
        const double2_t t2 = *reinterpret_cast<const double2_t*>(&Vt[c * kSV + i]);
        w += t2[0];
        w += t2[1];
      }
    }
    const double bad = half_sum((fabs(w) < 1.0e300) ? 0.0 : 1.0);
    const bool ok = piv_ok && (bad == 0.0);
    if (live) {
      double* const Wrow = p.W + e * M;
      double out = w;
      if (!ok) out = (c == 0) ? 0.5 * (gl + gr) : (c == 1) ? 0.5 * (gr - gl) : 0.0;
      if (c < M) Wrow[c] = out;
      if (c == 0) {
        if (p.status) p.status[e] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
        if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
      }
    }
  }
}

hipError_t enhance_dual(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M > kLP || a.n + 2 > kRhsRow || a.a_values) return hipErrorInvalidValue;
  static const RecTables tables = make_rec_tables();
  const int64_t npair = (a.ne + 1) / 2;
  int64_t blocks = (npair + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = 256 * 2 * 8;
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks), block(kWavesPerBlock * 64);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_dual_kernel<LSSVR_RHS_SIN>, grid, block, s, o, a, tables);
  return launch(enhance_dual_kernel<LSSVR_RHS_ARRAY>, grid, block, s, o, a, tables);
}

}  // namespace lssvr
