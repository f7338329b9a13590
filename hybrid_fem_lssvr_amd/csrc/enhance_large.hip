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
// is three 16x16 tiles, each computed as 4x4 blocks by v_mfma_f64_4x4x4_4b_f64 (see below).
// Lane (c = lane&31, h = lane>>5) works for element 2*pair + h:
//   1. as collocation point c of a 32-point chunk: abscissa, f, Gegenbauer recurrence.
//      The Vandermonde block goes through LDS in TWO HALVES of 16 columns (column stride 34
//      doubles: conflict-free for the point-major writes and for the operand reads): columns
//      0..15 are written, every lane picks up its 8 MFMA operands of them, then the recurrence
//      carries on (two registers of state) and columns 16..31 take the same 4.3 KB;
//   2. 8 k-steps x 10 block MFMAs per chunk and element, operands and accumulators in registers;
//   3. accumulators -> LDS in two stages (tiles (0,0)+(1,0), then (1,1) in the place of (0,0))
//      -> lane c owns column c of S (32 rows in registers); the right-hand side rides along as
//      row/column 31, so forward substitution is free;
//   4. right-looking LDL^T, both elements in lock step, factor FROZEN IN REGISTERS (lane c
//      stops at step c); the pivot row reaches the lanes by DPP row broadcast inside the FMA
//      (lssvr_wave.hpp::ldlt_solve_dpp), no LDS;
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
#ifndef LSSVR_LARGE_WAVES
#define LSSVR_LARGE_WAVES 1
#endif
constexpr int kLargeWaves = LSSVR_LARGE_WAVES;   // waves per workgroup: they share nothing, and one-wave
                                                 // workgroups free their LDS soonest (-2 % at 1e6 elements)
constexpr int kSB = 34;                    // operand block: column stride (doubles)
constexpr int kBlk = 16 * kSB;             // 544 = 16 columns x 32 points = two 16 x 17 tiles
constexpr int kST = 17;                    // tile row stride
constexpr int kT1 = 16 * kST;              // offset of the second tile
constexpr int kHalf2 = kBlk + 2 * kLP + kLP;   // block | E (C0_i, C1_i interleaved) | Z
static_assert(2 * kT1 <= kBlk && 2 * kSL <= kBlk, "tiles / pivot ring alias the operand block");

// The rows use the RESCALED second-derivative family p_m = q_m / c_m (lssvr_device.hpp:
// p_m = A_m t p_{m-1} - p_{m-2}, one multiply less per term than the q recurrence and a single
// coefficient table, which the fully unrolled row loops read as s_load batches from the kernel
// arguments).  Working with p is the diagonal change of variables G' = D^-1 G D^-1, v' = D v,
// D = diag(c_m): the boundary block becomes C' = C D^-1, the ridge eps D^-2, and the final
// coefficients w = D^-1 v' -- all per-lane multiplications by kInvScale2[c].
struct LargeTables {
  double a2s[kLP];             // A_m = d2_coef(m), m >= 2
  double sc2[kLP];             // c_m = d2_scale(m)   (variable-coefficient rows need q_m itself)
  double al1[kLP + 2], be1[kLP + 2];   // L' family (variable-coefficient rows)
  double sinc[10];             // odd Taylor coefficients of sin, -1/21! .. 1/3! (sin_reduced's)
};

inline LargeTables make_large_tables() {
  LargeTables t{};
  for (int m = 0; m < kLP; ++m) {
    t.a2s[m] = m >= 2 ? d2_coef(m) : 0.0;
    t.sc2[m] = d2_scale(m);
  }
  for (int m = 1; m < kLP + 2; ++m) {
    t.al1[m] = (double)(2 * m + 1) / (double)m;
    t.be1[m] = (double)(m + 1) / (double)m;
  }
  const double sinc[10] = {-1.0 / 51090942171709440000.0, 1.0 / 121645100408832000.0,
                           -1.0 / 355687428096000.0,      1.0 / 1307674368000.0,
                           -1.0 / 6227020800.0,           1.0 / 39916800.0,
                           -1.0 / 362880.0,               1.0 / 5040.0,
                           -1.0 / 120.0,                  1.0 / 6.0};   // sin_reduced's literals
  for (int i = 0; i < 10; ++i) t.sinc[i] = sinc[i];
  return t;
}

// lssvr_device.hpp::sin_reduced with the polynomial coefficients read from the kernel arguments
// (SGPR operands) instead of 20 VGPRs of hoisted literals -- the kernel runs at 168 registers.
__device__ __forceinline__ double sin_reduced_tab(double arg, const double* __restrict__ sc) {
  constexpr double kInvPi = 0.31830988618379067154;
  constexpr double kPiHi = 3.14159265358979311600e+00;
  constexpr double kPiLo = 1.22464679914735317723e-16;
  if (!(fabs(arg) < 3.0e9)) return sin(arg);
  const double j = rint(arg * kInvPi);
  double r = fma(-j, kPiHi, arg);
  r = fma(-j, kPiLo, r);
  const double z = r * r;
  double p = sc[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) p = fma(p, z, sc[i]);
  const double s = fma(-(r * z), p, r);
  const long long ji = (long long)j;
  return (ji & 1) ? -s : s;
}


// ---- f64 MFMA in 4x4x4 blocks ------------------------------------------------------------
// Measured on gfx950 (scripts/probes/mfma_probe.cpp): v_mfma_f64_4x4x4_4b_f64 (four 4x4x4
// blocks, 256 FMAs) takes 8.2 ns per SIMD, v_mfma_f64_16x16x4_f64 (1024 FMAs) 42 ns -- 28 % more
// per FMA -- and both share the FP64 vector pipe.  Operand layout (probed with one-hot inputs):
//   A: lane 16 k + 4 blk + i = A_blk[i][k]   B: lane 16 k + 4 blk + j = B_blk[k][j]
//   D: lane 16 i + 4 blk + j = D_blk[i][j]
// A 16 x 16 tile of the Gram matrix is 4 x 4 blocks.  With A = the tile's row operand in the
// layout above (block row blk = rows 4 blk .. 4 blk + 3) and B = the column operand ROTATED by r
// quads inside every 16-lane row, one instruction yields the four blocks (blk, (blk + r) & 3).
// The rotated operand is simply read from the LDS operand block with the rotated lane -> column
// map (same conflict-free pattern as the plain read; a DPP row_ror of the register costs two
// VALU movs per operand, and the kernel is bound by VALU issue: measured 8 % slower).
// A symmetric tile needs r = 0, 1, 2 (10 distinct blocks incl. mirrors), the off-diagonal tile
// r = 0 .. 3: 10 instructions per 4 collocation points and element instead of 3 of the
// 16x16x4 kind -- 82 ns instead of 126.
#define LSSVR_MFMA4(acc, av, bv) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0)

// the ten accumulators of one element: sym[0..2] tile (0,0) r = 0,1,2; off[0..3] tile (0,1)
// (= tile (1,0) transposed); low[0..2] tile (1,1)
struct GramAcc {
  double sym[3], off[4], low[3];
};

#define LSSVR_IS8(b) 1.0 / d2_scale(b), 1.0 / d2_scale(b + 1), 1.0 / d2_scale(b + 2), 1.0 / d2_scale(b + 3), \
                     1.0 / d2_scale(b + 4), 1.0 / d2_scale(b + 5), 1.0 / d2_scale(b + 6), 1.0 / d2_scale(b + 7)
__device__ const double kInvScale2[kLP] = {LSSVR_IS8(0), LSSVR_IS8(8), LSSVR_IS8(16), LSSVR_IS8(24)};
#undef LSSVR_IS8
}  // namespace

template <int RHS, bool VC>
__global__ __launch_bounds__(kLargeWaves * 64, 3) void enhance_large_kernel(EnhanceArgs p,
                                                                               LargeTables tb,
                                                                               unsigned nxcd) {
  __shared__ double2_t lds2[kLargeWaves * kHalf2];      // 2 halves x kHalf2 doubles per wave
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
  // XCD-aware numbering: workgroups go round-robin over the 8 XCDs (each with its own L2), so
  // consecutive element pairs are handed to consecutive workgroups OF ONE XCD -- otherwise every
  // 128-byte line of x / u (8 pairs) is fetched by 8 different L2s (measured: 7.2 MB instead of
  // 2.5 MB of HBM reads per 1e5 elements).  gridDim.x is a multiple of nxcd (the device's XCD
  // count, hipDeviceAttributeNumberOfXccs; the mapping is a bijection for any nxcd, so a wrong
  // guess costs L2 locality, never correctness).
  const unsigned xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
  const int64_t per_xcd = (int64_t)(gridDim.x / nxcd) * kLargeWaves;
  const int64_t pr = (int64_t)xcd * per_xcd + (int64_t)slot * kLargeWaves + wave;
  if (pr >= npair) return;
  {
    const int64_t e_raw = 2 * pr + h;
    bool live = e_raw < p.ne;                     // odd ne: half 1 of the last pair idles
    const int64_t e = live ? e_raw : p.ne - 1;    // ... on a duplicate, stores masked
    int64_t id = e;                               // mesh index (lssvr_enhance_subset)
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
    const double gamma = p.gamma_values ? p.gamma_values[id] : p.gamma;
    const DomainMap dm = map_params(a, b);
    const double step = dm.oldlen / (double)(n - 1);
    const double scl2 = dm.scl * dm.scl;
    const double inv_scl2 = rcp_newton(scl2);
    const double eps = rcp_newton(gamma * (scl2 * scl2));

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
    const double isc = VC ? 1.0 : kInvScale2[c];   // 1 / c_c of this lane's column
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
      } else {
        e0c *= isc;             // C' = C D^-1
        e1c *= isc;
      }
      wave_lds_sync();       // previous pair's reads of E / Z are done
      double2_t ev = {e0c, e1c};
      *reinterpret_cast<double2_t*>(&E[2 * c]) = ev;
    }

    // ---- Gram contraction on the matrix cores ----------------------------------------
    GramAcc gA = {}, gB = {};
    // operand addresses: rotation r reads column 4 ((blk + r) & 3) + j of the 16-column block
    int ar[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      ar[r] = (4 * ((((lane >> 2) & 3) + r) & 3) + (lane & 3)) * kSB + (lane >> 4);
    for (int k0 = 0; k0 < n; k0 += kCH) {
      // (an opaque zero in the table index keeps the compiler from hoisting all 31 coefficient
      // loads out of the chunk loop, where they would occupy 62 SGPRs for the whole kernel)
      int zi = 0;
      asm volatile("" : "+s"(zi));
      const int k = k0 + c;
      const bool valid = k < n;
      const double xk = linspace_at(a, b, dm.oldlen, step, valid ? k : 0, n);
      const double tk = dm.off + dm.scl * xk;
      double fk;
      if constexpr (RHS == LSSVR_RHS_SIN) {
        fk = p.rhs_amp * sin_reduced_tab(p.rhs_omega * xk, tb.sinc + zi);
      } else {
        fk = valid ? p.rhs_values[e * p.tab_es + k * p.tab_ps] : 0.0;
      }
      // a padding point contributes a zero row: zero seeds make the whole recurrence zero
      const double seed = valid ? 1.0 : 0.0;
      double phi = -(fk * inv_scl2) * seed;
      double ak = 0.0, bk = 0.0;
      if constexpr (VC) {
        ak = valid ? p.a_values[e * p.tab_es + k * p.tab_ps] : 0.0;
        bk = valid ? p.da_values[e * p.tab_es + k * p.tab_ps] * (0.5 * dm.oldlen) : 0.0;      // a'/scl, no division
        phi = -fma(bk, d1, fk * inv_scl2) * seed;
      }
      // Recurrence state across the two column halves: p = L''_{j+2} / c_j, r = L'_{j+2}.
      // Columns j >= MR (padding, M < 33) carry on with the recurrence: their Gram rows and
      // columns never meet a pivot, a live column's solution or the rhs row (elimination
      // stops at MR, E is zero there), so they need no masking.
      double p2 = 0.0, p1 = 0.0, r2 = seed, r1 = 3.0 * tk * seed;
      const double* const a2s = tb.a2s + zi;
      const double* const sc2 = tb.sc2 + zi;
      const double* const al1 = tb.al1 + zi;
      const double* const be1 = tb.be1 + zi;
      auto next_col = [&](const int j) -> double {
        double pj;
        if (j == 0) pj = 3.0 * seed;
        else if (j == 1) pj = 15.0 * tk * seed;
        else pj = fma(a2s[j] * tk, p1, -p2);
        p2 = p1;
        p1 = pj;
        if constexpr (VC) {
          // rho_j = a q_j + b (L'_{j+2} - C1_j),  q_j = c_j p_j
          const double val = fma(ak, pj * sc2[j], bk * (r1 - E[2 * j + 1]) * seed);
          const double rn = fma(al1[j + 2] * tk, r1, -(be1[j + 2] * r2));
          r2 = r1;
          r1 = rn;
          return val;
        } else {
          return pj;
        }
      };

      wave_lds_sync();   // the previous chunk's operand reads (and the E writes) are done
#pragma unroll
      for (int j = 0; j < 16; ++j) Bf[j * kSB + c] = next_col(j);
      wave_lds_sync();
      // OPERAND READS: every one a ds_read_b64 of its own.  hipcc's load/store optimiser otherwise
      // pairs them (same base register, offsets 32 B or 64 doubles apart) into ds_read2_b64 /
      // ds_read2st64_b64, which bank modulo 32 dwords in 16-lane groups instead of modulo 64 in 32-lane
      // halves (MI355X_MICROARCH.md, LDS): the 16 columns of a group, 34 doubles apart, then fall on 8
      // bank pairs -- a 2-way conflict on EVERY operand read at 4 LDS cycles per value instead of 2
      // (round-2 PMC: SQ_LDS_BANK_CONFLICT 35 % of SQ_LDS_IDX_ACTIVE, the LDS array busy ~60 % of the
      // kernel).  A volatile access is never merged; stride 34 is conflict-free for ds_read_b64.
      // (the access keeps the LDS address space: a volatile generic pointer would become a flat load)
      using lds_cvd = const volatile __attribute__((address_space(3))) double;
      auto opA = [&](const int r, const int s) { return *(lds_cvd*)(BfA + ar[r] + 4 * s); };
      auto opB = [&](const int r, const int s) { return *(lds_cvd*)(BfB + ar[r] + 4 * s); };
      double a0[kCH / 4], b0[kCH / 4];
#pragma unroll
      for (int s = 0; s < kCH / 4; ++s) {
        a0[s] = opA(0, s);
        b0[s] = opB(0, s);
        const double a0r1 = opA(1, s), a0r2 = opA(2, s);
        const double b0r1 = opB(1, s), b0r2 = opB(2, s);
        LSSVR_MFMA4(gA.sym[0], a0[s], a0[s]);
        LSSVR_MFMA4(gB.sym[0], b0[s], b0[s]);
        LSSVR_MFMA4(gA.sym[1], a0[s], a0r1);
        LSSVR_MFMA4(gB.sym[1], b0[s], b0r1);
        LSSVR_MFMA4(gA.sym[2], a0[s], a0r2);
        LSSVR_MFMA4(gB.sym[2], b0[s], b0r2);
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 16; j < kRhsRow; ++j) Bf[(j - 16) * kSB + c] = next_col(j);
      Bf[(kRhsRow - 16) * kSB + c] = phi;        // rhs always rides in the last column
      wave_lds_sync();
      // tile (0,1) = rows 0..15 (kept operands) x columns 16..31 (this block), and tile (1,1)
#pragma unroll
      for (int s = 0; s < kCH / 4; ++s) {
        const double a1 = opA(0, s), a1r1 = opA(1, s);
        const double a1r2 = opA(2, s), a1r3 = opA(3, s);
        const double b1 = opB(0, s), b1r1 = opB(1, s);
        const double b1r2 = opB(2, s), b1r3 = opB(3, s);
        LSSVR_MFMA4(gA.off[0], a0[s], a1);
        LSSVR_MFMA4(gB.off[0], b0[s], b1);
        LSSVR_MFMA4(gA.off[1], a0[s], a1r1);
        LSSVR_MFMA4(gB.off[1], b0[s], b1r1);
        LSSVR_MFMA4(gA.off[2], a0[s], a1r2);
        LSSVR_MFMA4(gB.off[2], b0[s], b1r2);
        LSSVR_MFMA4(gA.off[3], a0[s], a1r3);
        LSSVR_MFMA4(gB.off[3], b0[s], b1r3);
        if (need11) {
          LSSVR_MFMA4(gA.low[0], a1, a1);
          LSSVR_MFMA4(gB.low[0], b1, b1);
          LSSVR_MFMA4(gA.low[1], a1, a1r1);
          LSSVR_MFMA4(gB.low[1], b1, b1r1);
          LSSVR_MFMA4(gA.low[2], a1, a1r2);
          LSSVR_MFMA4(gB.low[2], b1, b1r2);
        }
      }
    }

    // ---- accumulators -> columns.  Stage 1: tiles (0,0) and (1,0), [row][col], stride 17 ----
    wave_lds_sync();
    // D layout: lane 16 i + 4 blk + j holds element [4 blk + i][4 ((blk + r) & 3) + j] of its tile
    const int ti = lane >> 4, tblk = (lane >> 2) & 3, tj = lane & 3;
    const int trow = 4 * tblk + ti;
    int tcol[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tcol[r] = 4 * ((tblk + r) & 3) + tj;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      BfA[trow * kST + tcol[r]] = gA.sym[r];
      BfB[trow * kST + tcol[r]] = gB.sym[r];
    }
    BfA[tcol[1] * kST + trow] = gA.sym[1];        // mirrors of the r = 1 blocks
    BfB[tcol[1] * kST + trow] = gB.sym[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      BfA[kT1 + tcol[r] * kST + trow] = gA.off[r];      // tile (1,0) = tile (0,1) transposed
      BfB[kT1 + tcol[r] * kST + trow] = gB.off[r];
    }
    wave_lds_sync();
    // + eps on the diagonal of the MR x MR block (lane c owns G[c][c] of its element)
    const double epsd = eps * isc * isc;           // eps D^-2
    if (c < 16 && c < MR) Bf[c * kST + c] += epsd;
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
    for (int r = 0; r < 3; ++r) {
      BfA[trow * kST + tcol[r]] = gA.low[r];
      BfB[trow * kST + tcol[r]] = gB.low[r];
    }
    BfA[tcol[1] * kST + trow] = gA.low[1];
    BfB[tcol[1] * kST + trow] = gB.low[1];
    wave_lds_sync();
    if (c >= 16 && c < MR) Bf[(c - 16) * kST + (c - 16)] += epsd;
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
    bool lane_ok;
    const double v = ldlt_solve_dpp(col, Z, c, MR, lane_ok);
    const double w0 = d0 - half_sum(((c < MR) ? e0c : 0.0) * v);
    const double w1 = d1 - half_sum(((c < MR) ? e1c : 0.0) * v);
    const double bad = half_sum((lane_ok && fabs(v) < 1.0e300) ? 0.0 : 1.0);
    const bool ok = (bad == 0.0) && (fabs(w0) < 1e300) && (fabs(w1) < 1e300);

    // ---- store: lane c -> W[e][c+2]; lane 0 also writes w0, w1 -----------------------------
    if (live) {
      double* const Wrow = p.W + id * (p.ldw ? p.ldw : (int64_t)M);
      if (c < MR) Wrow[c + 2] = ok ? v * isc : 0.0;    // w = D^-1 v'
      if (c == 0) {
        Wrow[0] = ok ? w0 : 0.5 * (gl + gr);
        Wrow[1] = ok ? w1 : 0.5 * (gr - gl);
        if (p.status) p.status[id] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
        if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
      }
    }
  }
}

hipError_t enhance_large(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M - 2 + 1 > kLP) return hipErrorInvalidValue;
  static const LargeTables tables = make_large_tables();
  const int64_t npair = (a.ne + 1) / 2;
  int64_t blocks = (npair + kLargeWaves - 1) / kLargeWaves;
  const unsigned nxcd = xcd_count();
  blocks = (blocks + nxcd - 1) / nxcd * nxcd;     // XCD-aware numbering needs a multiple of nxcd
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(kLargeWaves * 64);
  if (a.a_values)
    return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, true>, grid, block, s, o, a, tables, nxcd);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_large_kernel<LSSVR_RHS_SIN, false>, grid, block, s, o, a, tables, nxcd);
  return launch(enhance_large_kernel<LSSVR_RHS_ARRAY, false>, grid, block, s, o, a, tables, nxcd);
}

}  // namespace lssvr
