// Shared pieces of the wave-level kernels (enhance_large.hip, enhance_large_cheb.hip,
// enhance_large_parity.hip, enhance_dual.hip): LDS geometry, wave-level helpers, and the
// in-register LDL^T factor / solve routines -- a padded 32 x 32 system one column per lane, two
// systems per wave (ldlt_solve_dpp); two columns per lane, four systems per wave
// (ldlt_solve_dpp4); and its parity-split form (ldlt_parity_*).
#pragma once
#include <type_traits>
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {
namespace wave {

constexpr int kLP = 32;                 // padded size of the augmented system
constexpr int kCH = 32;                 // collocation points per chunk
constexpr int kSV = 34;                 // Vt column stride (doubles)
constexpr int kSG = 33;                 // G row stride (doubles)
constexpr int kSL = 34;                 // L row stride (even: 16-B aligned row pairs)
constexpr int kVDoubles = kLP * kSV;    // 1088 = 32*34 >= 32*33 (G and L alias Vt)
constexpr int kHalfDoubles = kVDoubles + 3 * kLP;   // Vt | E0 | E1 | Z
constexpr int kWaveDoubles = 2 * kHalfDoubles;
constexpr int kWavesPerBlock = 4;

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// recurrence coefficients, index m: q_m = al2[m] t q_{m-1} - be2[m] q_{m-2} (L'' family),
// r_m = al1[m] t r_{m-1} - be1[m] r_{m-2} (L' family)
// L_p = alL[p] t L_{p-1} - beL[p] L_{p-2} (Legendre values)
struct RecTables {
  double al2[kLP + 2], be2[kLP + 2], al1[kLP + 2], be1[kLP + 2], alL[kLP + 2], beL[kLP + 2];
};

inline RecTables make_rec_tables() {
  RecTables t{};
  for (int m = 1; m < kLP + 2; ++m) {
    t.al2[m] = (double)(2 * m + 3) / (double)m;
    t.be2[m] = (double)(m + 3) / (double)m;
    t.al1[m] = (double)(2 * m + 1) / (double)m;
    t.be1[m] = (double)(m + 1) / (double)m;
    t.alL[m] = (double)(2 * m - 1) / (double)m;
    t.beL[m] = (double)(m - 1) / (double)m;
  }
  return t;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, srclane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), srclane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// sum over the 32 lanes of a half
__device__ __forceinline__ double half_sum(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// L_p(t) for t = +-1 -/+ 2s, |s| tiny:  L_p(1-2s) = sum_k (-1)^k C(p,k) C(p+k,k) s^k
__device__ __forceinline__ double legendre_near_one(int p, double s) {
  constexpr double kInvSq[6] = {1.0, 1.0 / 4.0, 1.0 / 9.0, 1.0 / 16.0, 1.0 / 25.0, 1.0 / 36.0};
  double term = 1.0, sum = 1.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    term *= (-s * kInvSq[k]) * (double)((p - k) * (p + k + 1));
    sum += term;
  }
  return sum;
}


// ---------------------------------------------------------------------------
// LDL^T factor + solve of the padded symmetric system, one column per lane: lane (c, h) holds
// column c of the matrix of system h in col[0..31] (row index = register index); the right-hand
// side sits in row kRhsRow of every column (and, by symmetry, is column kRhsRow).  Only the
// leading nsys x nsys block is factorised.  Right-looking elimination without square roots:
// a_ic -= a_ji (a_jc / d_j).  A zero / negative / non-finite pivot makes 1/d_j inf/NaN or flips
// signs; every later entry inherits it, so the caller's single finiteness test on the solution
// detects a breakdown (SPD input => all d_j > 0 is also checked).
// ---------------------------------------------------------------------------
constexpr int kRhsRow = kLP - 1;

// ---------------------------------------------------------------------------
// The elimination with NO LDS AT ALL in the factorisation: the pivot row is broadcast by DPP.  gfx90a+ allows DPP on the 64-bit VOP2 v_fmac_f64 for row_newbcast (lane n of each
// 16-lane row feeds the whole row), so  a_ic -= a_ji t_c  is ONE instruction
//     v_fmac_f64_dpp col[i], P, ntc  row_newbcast:(i & 15)
// with P = the register col[j] seen across lanes (lane i holds a_ji).  A system spans two
// rows (32 lanes); gfx950's v_permlane16_swap turns two copies of P into t0 = the even row's
// data in both rows and t1 = the odd row's, so every lane finds a_ji in its own row.  The
// pivot reciprocal is taken element-wise on the copy whose lane (j & 15) holds d_j and
// reaches every lane through the same broadcast.  Everything runs with the full EXEC mask
// (a DPP source lane must be active); a column freezes because its multiplier is zeroed
// from its own pivot step on (lane c stops updating its column at step c, so after the
// elimination col[i], i > c, is the final a_ic = d_c L_ic, col[c] the pivot and col[kRhsRow] the
// forward-substituted right-hand side).  Versus publishing the pivot rows through an LDS ring
// (round 1, removed): 250 ds_read_b128 + 62 ds_read/ds_write_b64 fewer per element pair, 7 more
// VALU instructions per step, 5-7 % faster.
// ---------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double rowdata, double mul) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(rowdata), "v"(mul), "n"(I));
}

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Returns z_c; lane_ok = this lane's own pivot was positive (lanes >= nsys: true).
__device__ __forceinline__ double ldlt_solve_dpp(double (&col)[kLP], double* __restrict__ Z,
                                                 int c, int nsys, bool& lane_ok) {
  int cc = c;
  asm volatile("" : "+v"(cc));
  int ns = nsys;
  asm volatile("" : "+s"(ns));
  double nrinv_own = -1.0;          // -1/d_c of this lane's own pivot, latched at step c
  static_for<0, kLP - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < ns) {
      const unsigned long long pu = __builtin_bit_cast(unsigned long long, col[j]);
      const unsigned plo = (unsigned)pu, phi = (unsigned)(pu >> 32);
      const auto slo = __builtin_amdgcn_permlane16_swap(plo, plo, false, false);
      const auto shi = __builtin_amdgcn_permlane16_swap(phi, phi, false, false);
      double t0 = __builtin_bit_cast(double, ((unsigned long long)(unsigned)shi[0] << 32) | (unsigned)slo[0]);
      double t1 = __builtin_bit_cast(double, ((unsigned long long)(unsigned)shi[1] << 32) | (unsigned)slo[1]);
      double nrinv = -rcp_newton(j < 16 ? t0 : t1);        // lane (j & 15) of each row: -1/d_j
      if (cc == j) nrinv_own = nrinv;
      // DPP reads of a VGPR need two wait states after the VALU write (the compiler cannot
      // see the DPP inside the asm statements)
      asm volatile("s_nop 1" : "+v"(t0), "+v"(t1), "+v"(nrinv));
      double ntc = 0.0;
      fmac_rowbcast<(j & 15)>(ntc, nrinv, col[j]);          // -a_jc / d_j on every lane
      if (!(cc > j)) ntc = 0.0;                             // frozen columns stay as they are
      asm volatile("s_nop 0" : "+v"(ntc));
      static_for<j + 1, kLP>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fmac_rowbcast<(i & 15)>(col[i], i < 16 ? t0 : t1, ntc);
      });
    }
  });
  lane_ok = (nrinv_own < 0.0) || (c >= nsys);
  // backward substitution L^T z = D^-1 y out of the frozen columns (z_i through 32 doubles of LDS)
  double Y = col[kRhsRow];
  double rinv = -nrinv_own;
  if (c >= nsys) {
    Y = 0.0;
    rinv = 0.0;
  }
  wave_lds_sync();
  asm volatile("" : "+s"(ns));
#pragma unroll
  for (int i = kLP - 2; i >= 1; --i) {
    if (i < ns) {
      Z[c] = Y * rinv;
      wave_lds_sync();
      const double zi = Z[i];
      if (cc < i) Y = fma(-col[i], zi, Y);
    }
  }
  return (c < nsys) ? Y * rinv : 0.0;
}


// ---------------------------------------------------------------------------
// FOUR systems per wave: 16 lanes per system, TWO columns per lane (lane q of a 16-lane DPP row
// holds columns q and q + 16 of its system, rows in registers: A[i], B[i]).  The pivot row of
// step j, seen across the columns, is the register pair (A[j], B[j]) of the row's 16 lanes, so
//     a_ic -= a_ji t_c     is     v_fmac_f64_dpp  X[i], (i < 16 ? A[j] : B[j]), nt_X  row_newbcast:(i & 15)
// with no cross-row movement at all (ldlt_solve_dpp needs two v_permlane16_swap pairs per step
// because its systems span two DPP rows).  Per step the bookkeeping (pivot reciprocal, the two
// multipliers, the freeze masks) is shared by four systems instead of two, and once j >= 16 the
// first column of every lane is frozen, so only the second one is updated: 872 broadcast-FMAs
// and ~12 bookkeeping instructions per step for four systems, against 527 + 17 for two.
// The right-hand side rides along as row 31 of every column (forward substitution for free);
// the back-substitution broadcasts z_i the same way (no LDS).
// On exit zA / zB = solution components of columns q / q + 16 (0 beyond nsys).
//
// HAZARD (gfx950): a DPP operand read of a VGPR needs two wait states after a VALU write of that
// VGPR, and hipcc cannot see the DPP inside an asm statement.  Every DPP source here (tA, tB, the
// reciprocal, the scaled solution) is therefore passed through an `s_nop 1` statement right
// after it is produced and is not written again before its last DPP read.  Nothing but the
// register allocator could put a VALU write in between (a live-range split copy); that is checked
// on the ISA of every build by scripts/check_dpp_hazard.py (run from __graft_entry__.build()), and
// the parity tests tests/test_gpu_enhance_large.py (all M = 15..33) exercise the result.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ldlt_solve_dpp4(double (&A)[kLP], double (&B)[kLP], int q, int nsys,
                                                bool& lane_ok, double& zA, double& zB) {
  int qq = q;
  asm volatile("" : "+v"(qq));     // opaque copies: keep the lane masks / uniform conditions
  int ns = nsys;                   // from being hoisted out of the caller's round loop
  asm volatile("" : "+s"(ns));
  double nrA = -1.0, nrB = -1.0;   // -1/d of this lane's own pivots, latched at their steps
  static_for<0, kLP - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < ns) {
      double tA = A[j], tB = B[j];                          // row j across the columns
      double nr = -rcp_newton(j < 16 ? tA : tB);            // lane (j & 15): -1/d_j
      if constexpr (j < 16) {
        if (qq == j) nrA = nr;
      } else {
        if (qq == j - 16) nrB = nr;
      }
      asm volatile("s_nop 1" : "+v"(tA), "+v"(tB), "+v"(nr));
      double ntA = 0.0, ntB = 0.0;                          // -a_jc / d_j of the two columns
      if constexpr (j < 16) {
        fmac_rowbcast<(j & 15)>(ntA, nr, tA);
        if (!(qq > j)) ntA = 0.0;                           // frozen columns stay as they are
        fmac_rowbcast<(j & 15)>(ntB, nr, tB);
      } else {
        fmac_rowbcast<(j & 15)>(ntB, nr, tB);
        if (!(qq > j - 16)) ntB = 0.0;
      }
      asm volatile("s_nop 0" : "+v"(ntA), "+v"(ntB));
      static_for<j + 1, kLP>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (j < 16) fmac_rowbcast<(i & 15)>(A[i], i < 16 ? tA : tB, ntA);
        fmac_rowbcast<(i & 15)>(B[i], i < 16 ? tA : tB, ntB);
      });
    }
  });
  const int cA = q, cB = q + 16;
  lane_ok = ((nrA < 0.0) || (cA >= nsys)) && ((nrB < 0.0) || (cB >= nsys));
  // backward substitution L^T z = D^-1 y out of the frozen columns: column t keeps
  // Y_t = y_t - sum_{i>t} a_it z_i; the owner of column i publishes -z_i = Y_i (-1/d_i) by DPP
  double YA = A[kRhsRow], YB = B[kRhsRow];
  if (cA >= nsys) { YA = 0.0; nrA = 0.0; }
  if (cB >= nsys) { YB = 0.0; nrB = 0.0; }
  asm volatile("" : "+s"(ns));
  static_for<1, kLP - 1>([&](auto ir) {
    constexpr int i = kLP - 1 - decltype(ir)::value;        // i = 30 .. 1
    if (i < ns) {
      double nz = (i < 16) ? YA * nrA : YB * nrB;           // lane (i & 15): -z_i
      asm volatile("s_nop 1" : "+v"(nz));
      if constexpr (i < 16) {
        const double mA = (qq < i) ? A[i] : 0.0;            // a_it, t = column q < i
        fmac_rowbcast<(i & 15)>(YA, nz, mA);
      } else {
        fmac_rowbcast<(i & 15)>(YA, nz, A[i]);              // every first column is < i
        const double mB = (qq < i - 16) ? B[i] : 0.0;
        fmac_rowbcast<(i & 15)>(YB, nz, mB);
      }
    }
  });
  zA = -(YA * nrA);
  zB = -(YB * nrB);
}

}  // namespace wave
}  // namespace lssvr

namespace lssvr {
namespace wave {

// ---------------------------------------------------------------------------
// PARITY-SPLIT factorisation, four systems per wave (enhance_large_parity.hip): the S matrix of an
// element whose collocation points are symmetric about its centre is block diagonal in the
// even / odd Chebyshev indices up to rounding-level coupling, so the 31 x 31 LDL^T becomes a
// 16 x 16 and a 15 x 15 one in lock step: lane q of a 16-lane DPP row holds column q of the even
// block in A and column q of the odd block in B (rows in registers), the rhs entry of its
// columns in A[16] / B[16]:  136 + 120 broadcast-FMAs for four systems instead of 872.
//
//   step j:  t = A[j] (row j across the lanes; lane j: the pivot d_j);  nt_c = -a_jc / d_j (0 for
//            c <= j);  y_c += y_j nt_c;  a_ic += a_ji nt_c  (v_fmac_f64_dpp row_newbcast:i);  then
//            A[j] := nt in lanes c >= j -- the pivot row's upper part is dead, the multipliers
//            -L_cj take its place (row form, for further right-hand sides).  Afterwards register
//            A[i] holds, in lane c:  c < i: a_ic = d_c L_ic (column form, for the back-substitution),
//            c = i: 0,  c > i: -L_ci (row form, for forward substitutions) -- the whole factor in
//            the registers of the matrix; the substitutions mask the half they do not want.
// Rows and columns beyond the block sizes may hold ANY finite values: steps stop at ns, a lane
// c >= ns is never a broadcast source of anything a real lane keeps, rows >= ns feed nothing.
// The DPP hazard rule of ldlt_solve_dpp4 applies (s_nop 1 after every DPP source is produced;
// scripts/check_dpp_hazard.py proves it on the ISA).
// ---------------------------------------------------------------------------
constexpr int kPB = 16;                 // rows / columns of a parity block (padded)

__device__ __forceinline__ void ldlt_parity_factor(double (&A)[kPB + 1], double (&B)[kPB + 1],
                                                   int q, int nsteps, double& nrA, double& nrB) {
  int qq = q;
  asm volatile("" : "+v"(qq));
  int ns = nsteps;
  asm volatile("" : "+s"(ns));
  nrA = -1.0;
  nrB = -1.0;                      // -1/d of this lane's own pivots, latched at their step
  static_for<0, kPB>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < ns) {
      double tA = A[j], tB = B[j];
      double rA = -rcp_newton(tA), rB = -rcp_newton(tB);      // lane j: -1/d_j
      if (qq == j) {
        nrA = rA;
        nrB = rB;
      }
      asm volatile("s_nop 1" : "+v"(tA), "+v"(tB), "+v"(rA), "+v"(rB), "+v"(A[kPB]), "+v"(B[kPB]));
      double ntA = 0.0, ntB = 0.0;
      fmac_rowbcast<j>(ntA, rA, tA);
      fmac_rowbcast<j>(ntB, rB, tB);
      if (!(qq > j)) {                                        // frozen columns stay as they are
        ntA = 0.0;
        ntB = 0.0;
      }
      asm volatile("s_nop 0" : "+v"(ntA), "+v"(ntB));
      fmac_rowbcast<j>(A[kPB], A[kPB], ntA);                  // forward substitution rides along
      fmac_rowbcast<j>(B[kPB], B[kPB], ntB);
      static_for<j + 1, kPB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fmac_rowbcast<i>(A[i], tA, ntA);
        fmac_rowbcast<i>(B[i], tB, ntB);
      });
      if (qq >= j) {
        A[j] = ntA;
        B[j] = ntB;
      }
    }
  });
}

// L y = r for a further right-hand side (one entry per lane and block), in place.
__device__ __forceinline__ void ldlt_parity_forward(const double (&A)[kPB + 1], const double (&B)[kPB + 1],
                                                    int q, int nsteps, double& yA, double& yB) {
  int qq = q;
  asm volatile("" : "+v"(qq));
  int ns = nsteps;
  asm volatile("" : "+s"(ns));
  static_for<0, kPB - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < ns) {
      asm volatile("s_nop 1" : "+v"(yA), "+v"(yB));
      // lanes c >= j only (EXEC mask instead of zeroed multipliers: the source lane j stays active
      // and its own entry A[j] is 0)
      if (qq >= j) {
        fmac_rowbcast<j>(yA, yA, A[j]);                       // y_c -= L_cj y_j
        fmac_rowbcast<j>(yB, yB, B[j]);
      }
    }
  });
}

// L^T z = D^-1 y out of the column-form entries; y in, z out.
__device__ __forceinline__ void ldlt_parity_backward(const double (&A)[kPB + 1], const double (&B)[kPB + 1],
                                                     int q, int nsteps, double nrA, double nrB,
                                                     double& yA, double& yB) {
  int qq = q;
  asm volatile("" : "+v"(qq));
  int ns = nsteps;
  asm volatile("" : "+s"(ns));
  static_for<1, kPB>([&](auto ir) {
    constexpr int i = kPB - decltype(ir)::value;              // i = 15 .. 1
    if (i < ns) {
      double nzA = yA * nrA, nzB = yB * nrB;                  // lane i: -z_i
      asm volatile("s_nop 1" : "+v"(nzA), "+v"(nzB));
      if (qq <= i) {                                          // columns t <= i (lane i: A[i] = 0)
        fmac_rowbcast<i>(yA, nzA, A[i]);                      // Y_t -= a_it z_i
        fmac_rowbcast<i>(yB, nzB, B[i]);
      }
    }
  });
  yA = -(yA * nrA);
  yB = -(yB * nrB);
}

}  // namespace wave
}  // namespace lssvr
