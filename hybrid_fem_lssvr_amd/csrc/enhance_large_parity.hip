// Per-element LSSVR enhancement, large degree (23 <= M <= 33), Poisson rows, second kernel of the
// two-kernel path in the well-posed regime n >= 2 (M-2) (every BASELINE configuration): the
// PARITY-SPLIT solve of the Chebyshev-moment system.
//
// The collocation points of an element are np.linspace(a, b, n) (Dual.py:40): symmetric about the
// element centre, so in exact arithmetic every odd Chebyshev moment m_d = sum_k T_d(t_k) vanishes
// and  S2 = m_{i+c} + m_{|i-c|} + 2 eps (N + C_z^T C_z)  decouples into an even-index block
// (16 x 16 at M = 33) and an odd-index block (15 x 15):   S = P + E,
//   P = same-parity entries,  E = opposite-parity entries = the float64 asymmetry of the mapped
//   points t_k = off + scl x_k (relative size 1e-16 |x|/h: 1e-13 on [-1,1], 1e-11 on the wide
//   domain of BASELINE config 2) + the first-order asymmetry of the boundary rows.
// The reference's rows are those float64 points, so E is NOT dropped: the kernel factors P
// (two LDL^T of half the size in lock step: 256 broadcast-FMAs for four systems instead of 872)
// and iterates  z <- P^-1 (b - E z)  from z = P^-1 b until the step is below rounding:
// contraction rate rho = |P^-1 E| <= 1e-3 in this regime (cond(S) <= 3e3; measured
// scripts/proto/parity_split.py: 1e-13 .. 4e-4), one correction on ordinary meshes, each costs a
// structured matvec with the odd moments (31 LDS reads + 31 broadcast-FMAs per lane) and two
// substitutions with the kept factor.  Same minimiser as solve4_kernel: 1e-16 apart (tests).
//
// Mapping: a workgroup of four waves = sixteen elements; lane (g = lane >> 4, q = lane & 15) of a
// wave owns, of element g of its four: even column 2q in A[], odd column 2q+1 in B[] (16 + 15 rows
// in registers), the multipliers of both in the dead upper halves of the same registers.  Moments
// sit in LDS mirrored and with even and odd degrees apart, so m_{i+c} and m_{|i-c|} are ds_reads of
// 16 consecutive doubles at immediate offsets from per-lane bases: no index arithmetic per entry, no
// bank conflict.  Waves are persistent (one resident set per launch, the next quad's workspace rows
// prefetched into registers): the kernel is FP64-issue bound (~1 800 vector instructions per four
// elements at 4.6 cycles each with two waves per SIMD), not latency bound.
// MEASURED (MI355X, M = 33, 64 points, pair with moments_kernel): 192-197 us at 1e5 elements, 1.49-1.52 ms
// at 1e6, against 242 us / 2.04 ms with the full solve4_kernel and 395 us / 3.2-3.3 ms of the f64-MFMA kernel.
#include "cheb_tables.hpp"
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_wave.hpp"

namespace lssvr {

using namespace wave;

namespace {

constexpr int kWaves = 4;
constexpr int kWs = kMomentWsStride;
// LDS per element: me[0..60] = m_{2|k|}, k = -30..30 | mo[0..59] = m_{|2k+1|}, k = -30..29 (at 61) |
// a b g_l at 121..123 | r_0..r_30 at 124..154 | g_r at 155.  Even and odd moments apart and
// mirrored: the 16 lanes of a system read 16 CONSECUTIVE doubles at an immediate offset (no bank
// conflict, no index arithmetic); stride = 16 mod 32 doubles puts the two systems of a 32-lane half
// on complementary banks.
constexpr int kStride = 176;
constexpr int kOdd = 61;
constexpr int kTabDoubles = 4 * kPB * kPB;
constexpr int kMaxCorrections = 8;
constexpr int kResidentPerCu = 2;       // workgroups per CU the register budget admits (2 waves per SIMD)

struct ParityTables {
  double NE[kPB][kPB], NO[kPB][kPB];   // N[2p][2q], N[2p+1][2q+1]   (N = Y^T Y)
  double YE[kPB][kPB], YO[kPB][kPB];   // TRANSPOSED: YE[p][q] = Y[2q][2p], YO[p][q] = Y[2q+1][2p+1]  (v = Y z:
                                       // the 16 lanes q of a system read 16 consecutive doubles, no bank conflict)
  double alpha[32], b[32], slope[32];
};

constexpr ParityTables make_parity_tables() {
  ParityTables t{};
  for (int j = 0; j < 31; ++j) {
    for (int i = 0; i < 31; ++i) {
      if ((i & 1) != (j & 1)) continue;
      if (j & 1) {
        t.NO[j >> 1][i >> 1] = cheb::kN[j][i];
        t.YO[i >> 1][j >> 1] = cheb::kY[j][i];
      } else {
        t.NE[j >> 1][i >> 1] = cheb::kN[j][i];
        t.YE[i >> 1][j >> 1] = cheb::kY[j][i];
      }
    }
    t.alpha[j] = cheb::kAlpha[j];
    t.b[j] = cheb::kB[j];
    t.slope[j] = cheb::kSlope[j];
  }
  return t;
}

__device__ const ParityTables kPar = make_parity_tables();

// sum over the 16 lanes of a DPP row (every lane gets it): four row rotations, VALU only
template <int ROT>
__device__ __forceinline__ double row_ror(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, 0x120 + ROT, 0xf, 0xf, false);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), 0x120 + ROT, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double row_sum16(double v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  return v;
}

__global__ __launch_bounds__(64 * kWaves, 2) void solve4_parity_kernel(EnhanceArgs p,
                                                                        const double* __restrict__ ws) {
  __shared__ double lds_all[kTabDoubles + kWaves * 4 * kStride];
  double* const NEl = lds_all;
  double* const NOl = NEl + kPB * kPB;
  double* const YEl = NOl + kPB * kPB;
  double* const YOl = YEl + kPB * kPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* const lds = lds_all + kTabDoubles + wave * (4 * kStride);
#pragma unroll
  for (int t = 0; t < kTabDoubles / (64 * kWaves); ++t)
    lds_all[t * 64 * kWaves + tid] = (&kPar.NE[0][0])[t * 64 * kWaves + tid];
  // per-lane constants of the whole launch
  const int g = lane >> 4;
  int q = lane & 15;
  asm volatile("" : "+v"(q));
  const double* const el = lds + g * kStride;
  const int M = p.M, MR = M - 2, nE = (MR + 1) >> 1, nO = MR >> 1;
  const int cA = 2 * q, cB = 2 * q + 1;
  const bool inA = q < nE, inB = q < nO;
  const double alA = kPar.alpha[cA], bA = kPar.b[cA], slA = kPar.slope[cA];
  const double alB = kPar.alpha[cB], bB = kPar.b[cB], slB = kPar.slope[cB];
  const double amax = 0.5 * (double)((M - 1) * M);
  const int rowbase = lane & ~15;

  // Persistent waves: wave w of the launch takes the quads (four consecutive elements) w, w + W,
  // w + 2W, ...; the 96 workspace doubles of the NEXT quad's elements are fetched into registers
  // before the current quad is worked on and go to LDS (mirrored, even / odd apart) after it, so
  // the global-memory latency of the hand-over from moments_kernel is paid once per wave, not once
  // per quad (measured: 38 of 216 us at 1e5 elements were this prologue with one quad per wave).
  const int64_t nquads = (p.ne + 3) >> 2;
  const int64_t wstride = (int64_t)gridDim.x * kWaves;
  int64_t quad = (int64_t)blockIdx.x * kWaves + wave;
  double pre[6];
  auto fetch = [&](const int64_t qd) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int idx = k * 64 + lane;                  // 384 = 4 x 96 doubles
      const int eln = idx / kWs, j = idx % kWs;
      // (past the end: duplicates, stores masked)
      const int64_t e = (qd * 4 + eln < p.ne) ? qd * 4 + eln : p.ne - 1;
      pre[k] = ws[e * kWs + j];
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int idx = k * 64 + lane;
      const int eln = idx / kWs, j = idx % kWs;
      double* const L = lds + eln * kStride;
      const int k2 = j >> 1;
      const double v = pre[k];
      if (j > 60) {
        L[60 + j] = v;                                // a b g_l | r | g_r at 121..155
      } else if (j & 1) {
        L[kOdd + 30 + k2] = v;                        // m_{2k+1} = m_{-(2k+1)}
        L[kOdd + 29 - k2] = v;
      } else {
        L[30 + k2] = v;                               // m_{2k} = m_{-2k}
        L[30 - k2] = v;
      }
    }
  };
  if (quad < nquads) fetch(quad);
  __syncthreads();                                    // the tables are in LDS
  while (quad < nquads) {
    wave_lds_sync();                                    // the previous quad's LDS reads are done
    stash();
    wave_lds_sync();
    const int64_t E0 = quad * 4;
    quad += wstride;
    if (quad < nquads) fetch(quad);

    const int64_t e_raw = E0 + g;
    bool live = e_raw < p.ne;
    int64_t id = live ? e_raw : p.ne - 1;              // position in the launch ...
    if (p.elem_ids) {                                   // ... -> mesh index (lssvr_enhance_subset_ws)
      id = p.elem_ids[id];
      if (id < 0 || id >= p.ne_mesh) {       // out-of-range id: nothing of the mesh is touched
        if (live && q == 0 && p.fail_count) atomicAdd(p.fail_count, 1);
        live = false;
        id = 0;
      }
    }
    const double a = el[121], b = el[122], gl = el[123], gr = el[155];
    const double inv_gamma = p.gamma_values ? rcp_newton(p.gamma_values[id]) : p.inv_gamma;
    const DomainMap dm = map_params(a, b);
    const double hh = 0.5 * dm.oldlen;
    const double inv_scl2 = hh * hh;
    const double eps2 = (2.0 * inv_gamma) * (inv_scl2 * inv_scl2);
    if (ridge_dominated(eps2, M)) live = false;       // solved by ridge_fixup_kernel (enhance_large_cheb.hip)

    // ---- boundary rows to first order (enhance_small_cheb.hpp); exact recurrence when the wave
    // holds an element beyond the first-order range
    const double ta = dm.off + dm.scl * a;
    const double tb = dm.off + dm.scl * b;
    const double ea = 1.0 + ta, eb = 1.0 - tb;
    const double sig = 0.5 * (ea + eb), del = 0.5 * (ea - eb);
    const bool slow = amax * fmax(fabs(ea), fabs(eb)) >= 1.0e-6;
    const bool any_slow = __any(slow);
    double idet = 0.5 * fma(sig, 1.0 + sig, 1.0);
    if (any_slow) idet = rcp_newton(tb - ta);
    const double d0 = (tb * gl - ta * gr) * idet;
    const double d1 = (gr - gl) * idet;
    // (C0, C1) of w_{0,1} = d - C v for this lane's two columns (v-basis): even column, odd column
    double C0A = fma(-slA, sig, 1.0), C1A = slA * del;
    double C0B = (slB - 1.0) * del, C1B = fma(-(slB - 1.0), sig, 1.0);
    // Ridge part of S2 in rank-2 form, both paths:  eps2 N_ic + G1_i H1_c + G2_i H2_c  (same parity),
    //   first order:  G = (al_i, b_i),  H = (eps2 al_c - es b_c, -es al_c)
    //   exact:        G = eps2 (C0z_i, C1z_i),  H = (C0z_c, C1z_c),  C_z = C Y
    // The row factors G of row 2t / 2t+1 live in lane t and reach the columns by DPP broadcast, so
    // the cold path costs no extra code in the build.
    const double es = eps2 * sig, ed = eps2 * del;
    double G1A = alA, G2A = bA, H1A = fma(eps2, alA, -(es * bA)), H2A = -(es * alA);
    double G1B = alB, G2B = bB, H1B = fma(eps2, alB, -(es * bB)), H2B = -(es * alB);
    double k1A = eps2 * d0, k2A = eps2 * fma(del, d1, -(sig * d0));      // rhs = r + U1 k1 + U2 k2
    double k1B = eps2 * d1, k2B = eps2 * fma(del, d0, -(sig * d1));
    if (any_slow) {
      // exact L_{c+2}(ta), L_{c+2}(tb) by the Legendre recurrence, latched at degrees cA+2, cB+2
      double am1 = 1.0, a0 = ta, bm1 = 1.0, b0 = tb;
      double LaA = 0.0, LbA = 0.0, LaB = 0.0, LbB = 0.0;
      for (int m = 1; m <= MR; ++m) {
        const double inv = 1.0 / (double)(m + 1);
        const double a1 = ((double)(2 * m + 1) * ta * a0 - (double)m * am1) * inv;
        const double b1 = ((double)(2 * m + 1) * tb * b0 - (double)m * bm1) * inv;
        am1 = a0; a0 = a1;
        bm1 = b0; b0 = b1;
        if (m == cA + 1) { LaA = a1; LbA = b1; }
        if (m == cB + 1) { LaB = a1; LbB = b1; }
      }
      C0A = inA ? (tb * LaA - ta * LbA) * idet : 0.0;
      C1A = inA ? (LbA - LaA) * idet : 0.0;
      C0B = inB ? (tb * LaB - ta * LbB) * idet : 0.0;
      C1B = inB ? (LbB - LaB) * idet : 0.0;
      // C_z[., c] = sum_{j <= c, j = c mod 2} C[., j] Y[j][c]:  C[., 2p] lives in lane p (A side),
      // C[., 2p+1] in lane p (B side); Y[2p][2q] = YEl[q][p] (transposed table; 0 for p > q)
      double z0A_ = 0.0, z1A_ = 0.0, z0B_ = 0.0, z1B_ = 0.0;
      for (int t = 0; t < kPB; ++t) {
        const double yE = YEl[q * kPB + t], yO = YOl[q * kPB + t];
        z0A_ = fma(__shfl(C0A, rowbase + t), yE, z0A_);
        z1A_ = fma(__shfl(C1A, rowbase + t), yE, z1A_);
        z0B_ = fma(__shfl(C0B, rowbase + t), yO, z0B_);
        z1B_ = fma(__shfl(C1B, rowbase + t), yO, z1B_);
      }
      H1A = z0A_; H2A = z1A_; G1A = eps2 * z0A_; G2A = eps2 * z1A_;
      H1B = z0B_; H2B = z1B_; G1B = eps2 * z0B_; G2B = eps2 * z1B_;
      k1A = k1B = eps2 * d0;
      k2A = k2B = eps2 * d1;
    }
    if (!inA) C0A = C1A = 0.0;
    if (!inB) C0B = C1B = 0.0;

    // ---- right-hand sides of this lane's two columns
    const double* const rr = el + 124;
    const double rhsA = any_slow ? fma(H2A, k2A, fma(H1A, k1A, rr[cA])) : fma(G2A, k2A, fma(G1A, k1A, rr[cA]));
    const double rhsB = any_slow ? fma(H2B, k2B, fma(H1B, k1B, rr[cB])) : fma(G2B, k2B, fma(G1B, k1B, rr[cB]));

    // ---- the two columns of P (same-parity entries of S2) ------------------------------------
    // even block (row 2p, column 2q):   m_{2p+2q} + m_{|2p-2q|} = meP[p] + meM[p]
    // odd block (row 2p+1, col 2q+1):   m_{2p+2q+2} + m_{|2p-2q|} = meP[p+1] + meM[p]
    const double* const meP = el + 30 + q;
    const double* const meM = el + 30 - q;
    const double* const moP = el + kOdd + 30 + q;
    const double* const moM = el + kOdd + 30 - q;
    double A[kPB + 1], B[kPB + 1];
    {
      const double colB = inB ? 1.0 : 0.0;
      const int padrow = (nO < nE) ? nO : -1;
      asm volatile("s_nop 1" : "+v"(G1A), "+v"(G2A), "+v"(G1B), "+v"(G2B));     // DPP sources below
      // LDS operands one PAIR of rows ahead of their use: the reads of rows 2g+2, 2g+3 are issued before rows 2g,
      // 2g+1 are worked on, so their latency (two waves per SIMD, in-order issue) falls into 18 vector instructions
      // instead of being waited for at the head of every group of four rows.  Same values, same arithmetic order.
      struct Row { double m, pn, ne, no; };               // meM[t], meP[t + 1], NE[t][q], NO[t][q]
      auto load_row = [&](auto tc) -> Row {
        constexpr int t = decltype(tc)::value;
        return Row{meM[t], meP[t + 1], NEl[t * kPB + q], NOl[(t < kPB - 1 ? t : 0) * kPB + q]};
      };
      auto build_row = [&](auto tc, const double p, const Row& r) {
        constexpr int t = decltype(tc)::value;
        double v = p + r.m;
        v = fma(eps2, r.ne, v);
        fmac_rowbcast<t>(v, G1A, H1A);
        fmac_rowbcast<t>(v, G2A, H2A);
        A[t] = v;
        if constexpr (t < kPB - 1) {
          double w = r.pn + r.m;
          w = fma(eps2, r.no, w);
          fmac_rowbcast<t>(w, G1B, H1B);
          fmac_rowbcast<t>(w, G2B, H2B);
          // an odd number of bubble coefficients: the odd block is one short of the even one and
          // step nO pivots on a padding column -- that column is the unit vector, its row zero elsewhere
          B[t] = (t == padrow) ? ((q == t) ? 1.0 : 0.0) : w * colB;
        }
      };
      double pcur = meP[0];
      Row r0 = load_row(std::integral_constant<int, 0>{}), r1 = load_row(std::integral_constant<int, 1>{});
      static_for<0, kPB / 2>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        Row n0 = r0, n1 = r1;
        if constexpr (g + 1 < kPB / 2) {
          n0 = load_row(std::integral_constant<int, 2 * g + 2>{});
          n1 = load_row(std::integral_constant<int, 2 * g + 3>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        build_row(std::integral_constant<int, 2 * g>{}, pcur, r0);
        build_row(std::integral_constant<int, 2 * g + 1>{}, r0.pn, r1);
        __builtin_amdgcn_sched_barrier(0);
        pcur = r1.pn;
        r0 = n0;
        r1 = n1;
      });
      B[kPB - 1] = (q == kPB - 1) ? 1.0 : 0.0;
      A[kPB] = rhsA;
      B[kPB] = rhsB;
    }

    // ---- factor P (both blocks in lock step), z0 = P^-1 b -------------------------------------
    double nrA, nrB;
    ldlt_parity_factor(A, B, q, nE, nrA, nrB);
    const bool lane_ok = ((nrA < 0.0) || !inA) && ((nrB < 0.0) || !inB);
    double zA = A[kPB], zB = B[kPB];
    ldlt_parity_backward(A, B, q, nE, nrA, nrB, zA, zB);
    if (!inA) zA = 0.0;
    if (!inB) zB = 0.0;
    const double z0A = zA, z0B = zB;

    // ---- z <- z0 - P^-1 E z: the opposite-parity coupling -------------------------------------
    //   even row 2q  x odd column 2p+1:  m_{2q+2p+1} + m_{|2q-2p-1|} = moP[p] + moM[p]
    //   odd row 2q+1 x even column 2p:   m_{2q+2p+1} + m_{|2q+1-2p|} = moP[p] + moM[p-1]
    //   ridge part (first order):  ed (al_i b_c + b_i al_c);  cold path:  eps2 (C0z_i C0z_c + C1z_i C1z_c)
    // (only when it can reach the last bit: |ed| a_max^2 against moments of size n)
    const bool ridge_couples = any_slow || __any(fabs(ed) * (amax * amax) > 1.0e-18);
    const double scale = row_sum16(fabs(z0A) + fabs(z0B));
    double prev = scale;
    for (int it = 0; it < kMaxCorrections; ++it) {
      double srcA = zA, srcB = zB;
      asm volatile("s_nop 1" : "+v"(srcA), "+v"(srcB));
      double rA = 0.0, rB = 0.0;
      {
        // odd moments two PAIRS of rows ahead of their use (see the build above): moP[t], moM[t] of rows 2g+4, 2g+5
        // are read while rows 2g, 2g+1 are worked on
        struct Odd { double p0, m0, p1, m1; };            // moP[2g], moM[2g], moP[2g+1], moM[2g+1]
        auto load_odd = [&](auto gc) -> Odd {
          constexpr int g = decltype(gc)::value;
          return Odd{moP[2 * g], moM[2 * g], moP[2 * g + 1], moM[2 * g + 1]};
        };
        double mprev = (moM - 1)[0];
        Odd o0 = load_odd(std::integral_constant<int, 0>{}), o1 = load_odd(std::integral_constant<int, 1>{});
        static_for<0, kPB / 2>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          Odd o2 = o1;
          if constexpr (g + 2 < kPB / 2) o2 = load_odd(std::integral_constant<int, g + 2>{});
          __builtin_amdgcn_sched_barrier(0);
          fmac_rowbcast<2 * g>(rA, srcB, o0.p0 + o0.m0);
          fmac_rowbcast<2 * g>(rB, srcA, o0.p0 + mprev);
          if constexpr (2 * g + 1 < kPB - 1) fmac_rowbcast<2 * g + 1>(rA, srcB, o0.p1 + o0.m1);
          fmac_rowbcast<2 * g + 1>(rB, srcA, o0.p1 + o0.m0);
          __builtin_amdgcn_sched_barrier(0);
          mprev = o0.m1;
          o0 = o1;
          o1 = o2;
        });
      }
      if (ridge_couples) {
        // first order:  ed (al_i b_c + b_i al_c);   exact:  eps2 (C0z_i C0z_c + C1z_i C1z_c)
        const double q1A = any_slow ? H1A : G2A, q2A = any_slow ? H2A : G1A;
        const double q1B = any_slow ? H1B : G2B, q2B = any_slow ? H2B : G1B;
        const double s1E = row_sum16(q1A * zA), s2E = row_sum16(q2A * zA);
        const double s1O = row_sum16(q1B * zB), s2O = row_sum16(q2B * zB);
        const double f = any_slow ? 1.0 : ed;
        rA = fma(f, fma(G1A, s1O, G2A * s2O), rA);
        rB = fma(f, fma(G1B, s1E, G2B * s2E), rB);
      }
      ldlt_parity_forward(A, B, q, nE, rA, rB);
      ldlt_parity_backward(A, B, q, nE, nrA, nrB, rA, rB);
      const double nA = inA ? z0A - rA : 0.0, nB = inB ? z0B - rB : 0.0;
      const double step = row_sum16(fabs(nA - zA) + fabs(nB - zB));
      zA = nA;
      zB = nB;
      // the next step would be about step^2 / prev: stop once that is below rounding
      const bool more = step * step > 1.0e-17 * scale * prev;
      prev = step;
      if (!__any(more)) break;
    }

    // ---- v = Y z (bubble Legendre coefficients): v_{2q} = sum_p YE[q][p] z_{2p}, same for the odd ones
    double vA = 0.0, vB = 0.0;
    {
      double srcA = zA, srcB = zB;
      asm volatile("s_nop 1" : "+v"(srcA), "+v"(srcB));
      const double* const yE = YEl + q;
      const double* const yO = YOl + q;
      // (table rows two pairs ahead of their use, as above)
      struct Yrow { double e0, o0, e1, o1; };
      auto load_y = [&](auto gc) -> Yrow {
        constexpr int g = decltype(gc)::value;
        return Yrow{yE[(2 * g) * kPB], yO[(2 * g) * kPB], yE[(2 * g + 1) * kPB], yO[(2 * g + 1) * kPB]};
      };
      Yrow y0 = load_y(std::integral_constant<int, 0>{}), y1 = load_y(std::integral_constant<int, 1>{});
      static_for<0, kPB / 2>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        Yrow y2 = y1;
        if constexpr (g + 2 < kPB / 2) y2 = load_y(std::integral_constant<int, g + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        fmac_rowbcast<2 * g>(vA, srcA, y0.e0);
        fmac_rowbcast<2 * g>(vB, srcB, y0.o0);
        fmac_rowbcast<2 * g + 1>(vA, srcA, y0.e1);
        fmac_rowbcast<2 * g + 1>(vB, srcB, y0.o1);
        __builtin_amdgcn_sched_barrier(0);
        y0 = y1;
        y1 = y2;
      });
      if (!inA) vA = 0.0;
      if (!inB) vB = 0.0;
    }
    const double w0 = d0 - row_sum16(fma(C0A, vA, C0B * vB));
    const double w1 = d1 - row_sum16(fma(C1A, vA, C1B * vB));
    const unsigned long long badmask =
        __ballot(!(lane_ok && fabs(vA) < 1.0e300 && fabs(vB) < 1.0e300));
    const bool ok = (((badmask >> (lane & 48)) & 0xffffull) == 0) && (fabs(w0) < 1e300) && (fabs(w1) < 1e300);

    // ---- store: lane q -> W[e][2 + 2q], W[e][3 + 2q]; lane 0 also writes w0, w1 ----------------
    if (live) {
      double* const Wrow = p.W + id * (p.ldw ? p.ldw : (int64_t)M);
      if (inA) Wrow[cA + 2] = ok ? vA : 0.0;
      if (inB) Wrow[cB + 2] = ok ? vB : 0.0;
      if (q == 0) {
        Wrow[0] = ok ? w0 : 0.5 * (gl + gr);
        Wrow[1] = ok ? w1 : 0.5 * (gr - gl);
        if (p.status) p.status[id] = ok ? LSSVR_ST_OK : LSSVR_ST_FALLBACK;
        if (!ok && p.fail_count) atomicAdd(p.fail_count, 1);
      }
    }
  }  // quads of this wave
}

}  // namespace

// The parity split needs cond(S) small enough for the coupling iteration to contract fast: twice as
// many collocation points as bubble coefficients (cond <= 3e3, rho <= 1e-3 on any mesh float64 can
// hold); in between, the full four-systems-per-wave solve runs (enhance_large_cheb.hip).
bool enhance_parity_applies(int M, int n) { return M > kSmallMaxM && M <= kLargeMaxM && n >= 2 * (M - 2); }

hipError_t launch_solve4_parity(const EnhanceArgs& a, const double* ws, hipStream_t s, hipEvent_t ev_stop) {
  if (a.a_values || !ws) return hipErrorInvalidValue;
  int64_t blocks = (a.ne + 4 * kWaves - 1) / (4 * kWaves);
  const int64_t resident = (int64_t)cu_count() * kResidentPerCu;       // persistent: one resident set
  if (blocks > resident) blocks = resident;
  const dim3 grid((unsigned)blocks), block(64 * kWaves);
  if (ev_stop) hipExtLaunchKernelGGL(solve4_parity_kernel, grid, block, 0, s, nullptr, ev_stop, 0, a, ws);
  else hipLaunchKernelGGL(solve4_parity_kernel, grid, block, 0, s, a, ws);
  return hipGetLastError();
}

}  // namespace lssvr
