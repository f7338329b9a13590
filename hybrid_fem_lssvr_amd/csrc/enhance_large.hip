// Per-element LSSVR enhancement, large-degree path (23 <= M <= 33; any M >= 2 on request):
// TWO ELEMENTS PER WAVE (one per 32-lane half), the Legendre Gram contraction on the f64
// matrix cores, three waves per SIMD.
//
// Same mathematics as enhance_small_impl.hpp (DESIGN.md "per-element solve"); what changes
// is the mapping.  With MR = M-2 <= 31 bubble coefficients the augmented row
// [rho_0 .. rho_{MR-1}, 0.., phi] of a collocation point has 32 entries, so
//
//   [G r; r^T *] = sum_k [rho;phi]_k [rho;phi]_k^T            (32 x 32, contraction over k)
//
// is three 16x16 tiles of v_mfma_f64_16x16x4_f64 per 4 collocation points.
// Lane (c = lane&31, h = lane>>5) works for element 2*pair + h:
//   1. as collocation point c of a 32-point chunk: abscissa, f, Gegenbauer recurrence.
//      The Vandermonde block goes through LDS in TWO HALVES of 16 columns (column stride 34
//      doubles: conflict-free for the point-major writes and for the operand reads): columns
//      0..15 are written, every lane picks up its 8 MFMA operands of them, then the recurrence
//      carries on (two registers of state) and columns 16..31 take the same 4.3 KB;
//   2. 8 k-steps x 3 MFMAs per chunk and element, operands and accumulators in registers;
//   3. accumulators -> LDS in two stages (tiles (0,0)+(1,0), then (1,1) in the place of (0,0))
//      -> lane c owns column c of S (32 rows in registers); the right-hand side rides along as
//      row/column 31, so forward substitution is free;
//   4. right-looking LDL^T, both elements in lock step, factor FROZEN IN REGISTERS (lane c
//      stops at step c); only the pivot rows travel through a two-row LDS ring
//      (lssvr_wave.hpp::ldlt_solve_frozen);
//   5. backward substitution out of the frozen columns, z_i broadcast through 32 doubles of
//      LDS; w_{0,1} by a half-wave shuffle reduction.
// LDS: 5 KB per element (10 KB per wave) -> 3 workgroups of 4 waves per CU; waves are
// independent (wave-private LDS, no __syncthreads), so one wave's MFMA phase overlaps its
// neighbours' VALU phases.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {

using namespace wave;

namespace {
constexpr int kSB = 34;                    // operand block: column stride (doubles)
constexpr int kBlk = 16 * kSB;             // 544 = 16 columns x 32 points = two 16 x 17 tiles
constexpr int kST = 17;                    // tile row stride
constexpr int kT1 = 16 * kST;              // offset of the second tile
constexpr int kHalf2 = kBlk + 2 * kLP + kLP;   // block | E (C0_i, C1_i interleaved) | Z
static_assert(2 * kT1 <= kBlk && 2 * kSL <= kBlk, "tiles / pivot ring alias the operand block");
}  // namespace

template <int RHS, bool VC>
__global__ __launch_bounds__(kWavesPerBlock * 64, 3) void enhance_large_kernel(EnhanceArgs p,
                                                                               RecTables tb) {
  __shared__ double2_t lds2[kWavesPerBlock * kHalf2];      // 2 halves x kHalf2 doubles per wave
  double* const lds = reinterpret_cast<double*>(lds2);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  double* const BfA = lds + wave * (2 * kHalf2);   // element A (half 0)
  double* const BfB = BfA + kHalf2;                // element B (half 1)
  double* const Bf = h ? BfB : BfA;                // this lane's element
  double* const E = Bf + kBlk;                     // E[2i] = C0_i, E[2i+1] = C1_i; (d0, d1) at i = 31
  double* const Z = E + 2 * kLP;                   // back-substitution broadcast vector
  const int M = p.M, MR = M - 2, n = p.n;
  const bool need11 = MR > 16;        // tile (1,1) holds nothing but padding otherwise
  const int64_t npair = (p.ne + 1) >> 1;

  // one element pair per wave, no persistent loop: with a loop the compiler hoists ~70
  // VGPRs of lane constants (LDS addresses, series factors, sin coefficients) out of it and
  // spills them at the 168-register budget of three waves per SIMD
  const int64_t pr = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (pr >= npair) return;
  {
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
    double e0c, e1c;                               // this lane's own (C0_c, C1_c)
    {
      e0c = (tbb * La2 - ta * Lb2) * idet;
      e1c = (Lb2 - La2) * idet;
      if (c == kRhsRow) {
        e0c = d0;
        e1c = d1;
      } else if (c >= MR) {
        e0c = 0.0;
        e1c = 0.0;
      }
      wave_lds_sync();       // previous pair's reads of E / Z are done
      double2_t ev = {e0c, e1c};
      *reinterpret_cast<double2_t*>(&E[2 * c]) = ev;
    }

    // ---- Gram contraction on the matrix cores ----------------------------------------
    double4_t accA00 = {0, 0, 0, 0}, accA10 = {0, 0, 0, 0}, accA11 = {0, 0, 0, 0};
    double4_t accB00 = {0, 0, 0, 0}, accB10 = {0, 0, 0, 0}, accB11 = {0, 0, 0, 0};
    const int ar = (lane & 15) * kSB + (lane >> 4);
    for (int k0 = 0; k0 < n; k0 += kCH) {
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
      double ak = 0.0, bk = 0.0;
      if constexpr (VC) {
        ak = valid ? p.a_values[e * n + k] : 0.0;
        bk = valid ? p.da_values[e * n + k] / dm.scl : 0.0;
        phi = -fma(bk, d1, fk * inv_scl2) * seed;
      }
      // recurrence state across the two column halves: q = L''_{j+2}, r = L'_{j+2}
      double q2 = 0.0, q1 = 0.0, r2 = seed, r1 = 3.0 * tk * seed;
      auto next_col = [&](int j) -> double {
        double q;
        if (j == 0) q = 3.0 * seed;
        else if (j == 1) q = 15.0 * tk * seed;
        else q = fma(tb.al2[j] * tk, q1, -(tb.be2[j] * q2));
        q2 = q1;
        q1 = q;
        double val = q;
        if constexpr (VC) {
          // rho_j = a q_j + b (L'_{j+2} - C1_j)
          val = fma(ak, q, bk * (r1 - E[2 * j + 1]) * seed);
          const double rn = fma(tb.al1[j + 2] * tk, r1, -(tb.be1[j + 2] * r2));
          r2 = r1;
          r1 = rn;
        }
        return (j < MR) ? val : 0.0;
      };

      wave_lds_sync();   // the previous chunk's operand reads (and the E writes) are done
#pragma nounroll
      for (int j = 0; j < 16; ++j) Bf[j * kSB + c] = next_col(j);
      wave_lds_sync();
      double a0[kCH / 4], b0[kCH / 4];
#pragma unroll
      for (int s = 0; s < kCH / 4; ++s) {
        a0[s] = BfA[ar + 4 * s];
        b0[s] = BfB[ar + 4 * s];
      }
      wave_lds_sync();
#pragma nounroll
      for (int j = 16; j < kRhsRow; ++j) Bf[(j - 16) * kSB + c] = next_col(j);
      Bf[(kRhsRow - 16) * kSB + c] = phi;        // rhs always rides in the last column
      wave_lds_sync();
#pragma unroll
      for (int s = 0; s < kCH / 4; ++s) {
        const double a1 = BfA[ar + 4 * s];
        const double b1 = BfB[ar + 4 * s];
        accA00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], a0[s], accA00, 0, 0, 0);
        accB00 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0[s], b0[s], accB00, 0, 0, 0);
        accA10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0[s], accA10, 0, 0, 0);
        accB10 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0[s], accB10, 0, 0, 0);
        if (need11) {
          accA11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, accA11, 0, 0, 0);
          accB11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, accB11, 0, 0, 0);
        }
      }
    }

    // ---- accumulators -> columns.  Stage 1: tiles (0,0) and (1,0), [row][col], stride 17 ----
    wave_lds_sync();
    const int tcol = lane & 15, trb = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = trb + 4 * q;
      BfA[row * kST + tcol] = accA00[q];
      BfA[kT1 + row * kST + tcol] = accA10[q];
      BfB[row * kST + tcol] = accB00[q];
      BfB[kT1 + row * kST + tcol] = accB10[q];
    }
    wave_lds_sync();
    // + eps on the diagonal of the MR x MR block (lane c owns G[c][c] of its element)
    if (c < 16 && c < MR) Bf[c * kST + c] += eps;
    wave_lds_sync();
    // S = G + eps (I + C^T C): lane c takes column c.  Rows 0..15: G[i][c] is tile (0,0)
    // [c][i] (symmetric) for c < 16 and tile (1,0) [c-16][i] for c >= 16.
    double col[kLP];
    {
      // tile (0,0) is symmetric, so lanes c < 16 read row c instead of column c: every lane
      // reads 16 consecutive doubles at c * 17 (kT1 = 16 * 17 makes the two cases one formula)
      const int base0 = c * kST;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double2_t ei = *reinterpret_cast<const double2_t*>(&E[2 * i]);
        const double add = fma(ei[0], e0c, ei[1] * e1c);
        col[i] = fma(eps, add, Bf[base0 + i]);
      }
    }
    // Stage 2: tile (1,1) takes the place of tile (0,0).  Rows 16..31: tile (1,0) [i-16][c]
    // for c < 16, tile (1,1) [i-16][c-16] for c >= 16.
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = trb + 4 * q;
      BfA[row * kST + tcol] = accA11[q];
      BfB[row * kST + tcol] = accB11[q];
    }
    wave_lds_sync();
    if (c >= 16 && c < MR) Bf[(c - 16) * kST + (c - 16)] += eps;
    wave_lds_sync();
    {
      const int base1 = (c < 16) ? kT1 + c : c - 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double2_t ei = *reinterpret_cast<const double2_t*>(&E[2 * (16 + i)]);
        const double add = fma(ei[0], e0c, ei[1] * e1c);
        col[16 + i] = fma(eps, add, Bf[base1 + i * kST]);
      }
    }
    wave_lds_sync();    // G is dead from here on; the pivot-row ring reuses the region

    // ---- LDL^T factor + solve of the MR x MR block, rhs carried as row/column 31 ------------
    bool piv_ok;
    const double v = ldlt_solve_frozen(col, Bf, Z, c, MR, piv_ok);
    const double w0 = d0 - half_sum(((c < MR) ? e0c : 0.0) * v);
    const double w1 = d1 - half_sum(((c < MR) ? e1c : 0.0) * v);
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
  const int64_t blocks = (npair + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(kWavesPerBlock * 64);
  if (a.a_values)
    return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, true>, grid, block, s, o, a, tables);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_large_kernel<LSSVR_RHS_SIN, false>, grid, block, s, o, a, tables);
  return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, false>, grid, block, s, o, a, tables);
}

}  // namespace lssvr
