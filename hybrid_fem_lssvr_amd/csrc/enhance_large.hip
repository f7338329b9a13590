// Per-element LSSVR enhancement, large-degree path (15 <= M <= 33): TWO ELEMENTS
// PER WAVE (one per 32-lane half), the Legendre Gram contraction on the f64
// matrix cores.
//
// Same mathematics as enhance_small.hip (DESIGN.md "per-element solve"); what
// changes is the mapping.  With MR = M-2 <= 31 bubble coefficients the augmented
// row [rho_0 .. rho_{MR-1}, phi] of a collocation point has <= 32 entries, so
//
//   [G r; r^T *] = sum_k [rho;phi]_k [rho;phi]_k^T            (32 x 32, contraction over k)
//
// is three 16x16 tiles of v_mfma_f64_16x16x4_f64 per 4 collocation points.
// Lane (c = lane&31, h = lane>>5) works for element 2*pair + h:
//   1. as collocation point c of a 32-point chunk: abscissa, f, Gegenbauer
//      recurrence; every value goes straight to the LDS Vandermonde block
//      Vt_h[col][point] (col stride 34 doubles: conflict-free for the point-major
//      writes AND for the MFMA operand reads, DESIGN.md "LDS layouts");
//   2. the whole wave runs 8 k-steps x 3 MFMAs per chunk for element A, then for
//      element B (operands by ds_read_b64, accumulators stay in registers);
//   3. accumulators -> LDS (aliasing Vt) -> lane c owns column c of S (32 rows in
//      registers); the right-hand side rides along as row/column MR, so forward
//      substitution is free;
//   4. right-looking Cholesky, both elements in lock step: pivot by v_readlane,
//      column j through LDS (it is also the stored factor, read back as b128
//      broadcasts), rank-1 update in registers -- no lane masks anywhere;
//   5. backward substitution through a 32-entry LDS vector, w_{0,1} by a
//      half-wave shuffle reduction.
// Waves of a workgroup are independent (wave-private LDS, no __syncthreads), so
// one wave's MFMA phase overlaps its neighbours' VALU phases.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {

using namespace wave;

template <int RHS, bool VC>
__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void enhance_large_kernel(EnhanceArgs p,
                                                                               RecTables tb) {
  __shared__ double2_t lds2[kWavesPerBlock * kWaveDoubles / 2];
  double* const lds = reinterpret_cast<double*>(lds2);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  double* const VtA = lds + wave * kWaveDoubles;   // element A (half 0)
  double* const VtB = VtA + kHalfDoubles;          // element B (half 1)
  double* const Vt = h ? VtB : VtA;                // this lane's element
  double* const Gb = Vt;                           // aliases Vt once the MFMAs have read it
  double* const Lm = Vt;                           // aliases Gb once the columns are in registers
  double* const E0 = Vt + kVDoubles;               // C0_j (j<MR), d0 at MR, 0 beyond
  double* const E1 = E0 + kLP;                     // C1_j, d1, 0
  double* const Z = E1 + kLP;                      // back-substitution broadcast vector
  const int M = p.M, MR = M - 2, n = p.n;
  constexpr int ntile = 2;            // the rhs column sits at index 31: always both tile rows
  const bool need11 = MR > 16;        // tile (1,1) holds nothing but padding otherwise
  const int64_t npair = (p.ne + 1) >> 1;

  for (int64_t pr = (int64_t)blockIdx.x * kWavesPerBlock + wave; pr < npair;
       pr += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t e_raw = 2 * pr + h;
    const bool live = e_raw < p.ne;               // odd ne: half 1 of the last pair idles
    const int64_t e = live ? e_raw : p.ne - 1;    // ... on a duplicate, stores masked
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

    // ---- boundary rows: lane c holds L_{c+2}(ta), L_{c+2}(tb) of its element ----------
    const double ta = dm.off + dm.scl * a;
    const double tbb = dm.off + dm.scl * b;
    double La2, Lb2;
    {
      const double sa = 0.5 * (1.0 + ta), sb = 0.5 * (1.0 - tbb);
      const double big = (double)((MR + 1) * (MR + 2));
      const bool series_ok = (fabs(sa) * big < 1e-3) && (fabs(sb) * big < 1e-3);
      if (__all(series_ok)) {
        const int pp = c + 2;
        La2 = legendre_near_one(pp, sa);
        if (pp & 1) La2 = -La2;
        Lb2 = legendre_near_one(pp, sb);
      } else {
        // general recurrence at both end points, latch at degree c+2
        double am1 = 1.0, a0 = ta, bm1 = 1.0, b0 = tbb;
        La2 = 0.0;
        Lb2 = 0.0;
        for (int m = 1; m <= MR; ++m) {
          const double inv = 1.0 / (double)(m + 1);
          const double a1 = ((double)(2 * m + 1) * ta * a0 - (double)m * am1) * inv;
          const double b1 = ((double)(2 * m + 1) * tbb * b0 - (double)m * bm1) * inv;
          am1 = a0; a0 = a1;
          bm1 = b0; b0 = b1;
          if (m == c + 1) {
            La2 = a1;
            Lb2 = b1;
          }
        }
      }
    }
    const double idet = rcp_newton(tbb - ta);
    const double d0 = (tbb * gl - ta * gr) * idet;
    const double d1 = (gr - gl) * idet;
    {
      double e0 = (tbb * La2 - ta * Lb2) * idet;
      double e1 = (Lb2 - La2) * idet;
      if (c == kRhsRow) {
        e0 = d0;
        e1 = d1;
      } else if (c >= MR) {
        e0 = 0.0;
        e1 = 0.0;
      }
      E0[c] = e0;
      E1[c] = e1;
    }

    // ---- Gram contraction on the matrix cores ----------------------------------------
    double4_t accA00 = {0, 0, 0, 0}, accA10 = {0, 0, 0, 0}, accA11 = {0, 0, 0, 0};
    double4_t accB00 = {0, 0, 0, 0}, accB10 = {0, 0, 0, 0}, accB11 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < n; k0 += kCH) {
      wave_lds_sync();   // previous chunk's operand reads (and the E0/E1 writes) are done
      {
        const int k = k0 + c;
        const bool valid = k < n;
        const double xk = linspace_at(a, b, dm.oldlen, step, valid ? k : 0, n);
        const double tk = dm.off + dm.scl * xk;
        double fk;
        if constexpr (RHS == LSSVR_RHS_SIN) {
          fk = p.rhs_amp * sin_reduced(p.rhs_omega * xk);
        } else {
          fk = valid ? p.rhs_values[e * n + k] : 0.0;
        }
        // a padding point contributes a zero row: zero seeds make the whole recurrence zero
        const double seed = valid ? 1.0 : 0.0;
        double phi = -(fk * inv_scl2) * seed;
        if constexpr (!VC) {
          double q2 = 3.0 * seed, q1 = 15.0 * tk * seed;
          if (MR > 0) Vt[0 * kSV + c] = q2;
          if (MR > 1) Vt[1 * kSV + c] = q1;
          for (int j = 2; j < MR; ++j) {
            const double q = fma(tb.al2[j] * tk, q1, -(tb.be2[j] * q2));
            q2 = q1;
            q1 = q;
            Vt[j * kSV + c] = q;
          }
        } else {
          const double ak = valid ? p.a_values[e * n + k] : 0.0;
          const double bk = valid ? p.da_values[e * n + k] / dm.scl : 0.0;
          phi = -fma(bk, d1, fk * inv_scl2) * seed;
          // q_j = L''_{j+2}(tk);  r1 = L'_{j+2}(tk) = C^{(3/2)}_{j+1};  rho_j = a q_j + b (r1 - C1_j)
          double q2 = 0.0, q1 = 0.0, r2 = 1.0, r1 = 3.0 * tk;
          for (int j = 0; j < MR; ++j) {
            double q;
            if (j == 0) q = 3.0;
            else if (j == 1) q = 15.0 * tk;
            else q = fma(tb.al2[j] * tk, q1, -(tb.be2[j] * q2));
            q2 = q1;
            q1 = q;
            Vt[j * kSV + c] = fma(ak, q, bk * (r1 - E1[j]));
            const double rn = fma(tb.al1[j + 2] * tk, r1, -(tb.be1[j + 2] * r2));
            r2 = r1;
            r1 = rn;
          }
        }
        for (int j = MR; j < kRhsRow; ++j) Vt[j * kSV + c] = 0.0;
        Vt[kRhsRow * kSV + c] = phi;        // rhs always rides in the last column
      }
      wave_lds_sync();
      const int ar = (lane & 15) * kSV + (lane >> 4);
#pragma unroll
      for (int s = 0; s < kCH / 4; ++s) {
        const double a0 = VtA[ar + 4 * s];
        const double b0 = VtB[ar + 4 * s];
        accA00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, accA00, 0, 0, 0);
        accB00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b0, accB00, 0, 0, 0);
        if (ntile == 2) {
          const double a1 = VtA[ar + 16 * kSV + 4 * s];
          const double b1 = VtB[ar + 16 * kSV + 4 * s];
          accA10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, accA10, 0, 0, 0);
          accB10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, accB10, 0, 0, 0);
          if (need11) {
            accA11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, accA11, 0, 0, 0);
            accB11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, accB11, 0, 0, 0);
          }
        }
      }
    }

    // ---- accumulators -> G_A, G_B [row][col] (full symmetric 32x32, stride 33) ----------
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
    // + eps on the diagonal of the MR x MR block (lane c owns G[c][c] of its element)
    Gb[c * kSG + c] += (c < MR) ? eps : 0.0;
    wave_lds_sync();

    // ---- S = G + eps (I + C^T C): lane c takes column c, all 32 rows ----------------------
    double col[kLP];
    {
      const double e0c = E0[c], e1c = E1[c];
#pragma unroll
      for (int i = 0; i < kLP; ++i) {
        const double add = fma(E0[i], e0c, E1[i] * e1c);
        col[i] = fma(eps, add, Gb[i * kSG + c]);
      }
    }
    wave_lds_sync();    // G is dead from here on; the factor (stride 34) reuses the region

    // ---- LDL^T factor + solve of the MR x MR block, rhs carried as row/column 31 ------------
    bool piv_ok;
    const double v = ldlt_solve(col, Lm, Z, c, MR, piv_ok);
    const double w0 = d0 - half_sum(E0[c] * v);
    const double w1 = d1 - half_sum(E1[c] * v);
    const double bad = half_sum((fabs(v) < 1.0e300) ? 0.0 : 1.0);
    const bool ok = piv_ok && (bad == 0.0) && (fabs(w0) < 1e300) && (fabs(w1) < 1e300);

    // ---- store: lane c -> W[e][c+2]; lane 0 also writes w0, w1 -----------------------------
    if (live) {
      double* const Wrow = p.W + e * M;
      if (c < MR) Wrow[c + 2] = ok ? v : 0.0;
      if (c == 0) {
        Wrow[0] = ok ? w0 : 0.5 * (gl + gr);
        Wrow[1] = ok ? w1 : 0.5 * (gr - gl);
        if (p.status) p.status[e] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
        if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
      }
    }
  }
}

hipError_t enhance_large(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M - 2 + 1 > kLP) return hipErrorInvalidValue;
  static const RecTables tables = make_rec_tables();
  const int64_t npair = (a.ne + 1) / 2;
  int64_t blocks = (npair + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = 256 * 2 * 8;            // 8 rounds of a full chip at 2 blocks per CU
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks), block(kWavesPerBlock * 64);
  if (a.a_values)
    return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, true>, grid, block, s, o, a, tables);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_large_kernel<LSSVR_RHS_SIN, false>, grid, block, s, o, a, tables);
  return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, false>, grid, block, s, o, a, tables);
}

}  // namespace lssvr
