// Device-side building blocks shared by the gfx950 kernels.
//
// Everything here is FP64.  The translation units are compiled with
// -ffp-contract=off: the places that mirror numpy's two-rounding arithmetic
// (linspace, mapdomain, legval) rely on it, and every fused multiply-add that
// is wanted is written as an explicit fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lssvr {

constexpr int kBlock = 256;  // 4 waves of 64: one wave per SIMD per workgroup

// ---------------------------------------------------------------------------
// numpy arithmetic mirrored on the device
// ---------------------------------------------------------------------------

// polyutils.mapparms(old=[a,b], new=[-1,1]): off = (b*(-1) - a*1)/oldlen, scl = 2/oldlen
struct DomainMap {
  double off, scl, oldlen;
};
__device__ __forceinline__ DomainMap map_params(double a, double b) {
  DomainMap m;
  m.oldlen = b - a;
  m.off = (-b - a) / m.oldlen;
  m.scl = 2.0 / m.oldlen;
  return m;
}

// np.linspace(a, b, n)[k]: fl(fl(k*step) + a), last sample forced to b
// (numpy/_core/function_base.py:140-175; step == 0 -> fl(fl(k/div)*delta) + a).
__device__ __forceinline__ double linspace_at(double a, double b, double delta, double step, int k,
                                              int n) {
  if (k == n - 1) return b;
  double y = (step == 0.0) ? ((double)k / (double)(n - 1)) * delta : (double)k * step;
  return y + a;
}

// ---------------------------------------------------------------------------
// sin for the in-kernel right-hand side
// ---------------------------------------------------------------------------
// |arg| >= 3e9: the library routines (Payne-Hanek reduction), kept OUT of line -- inlined they
// raise the register allocation of every kernel that might take this once-in-never branch.
__device__ __attribute__((noinline)) static double sin_huge(double arg) { return sin(arg); }
__device__ __attribute__((noinline)) static double cos_huge(double arg) { return cos(arg); }

// sin(arg) for |arg| < 2^30*pi: arg = j*pi + r, |r| <= pi/2, two-term FMA
// reduction, then the odd Taylor polynomial through r^21 (truncation
// (pi/2)^23/23! = 1.3e-18).  Larger arguments go to the ocml routine.
__device__ __forceinline__ double sin_reduced(double arg) {
  constexpr double kInvPi = 0.31830988618379067154;
  constexpr double kPiHi = 3.14159265358979311600e+00;
  constexpr double kPiLo = 1.22464679914735317723e-16;
  if (!(fabs(arg) < 3.0e9)) return sin_huge(arg);
  const double j = rint(arg * kInvPi);
  double r = fma(-j, kPiHi, arg);
  r = fma(-j, kPiLo, r);
  const double z = r * r;
  double p = -1.0 / 51090942171709440000.0;             // -1/21!
  p = fma(p, z, 1.0 / 121645100408832000.0);            //  1/19!
  p = fma(p, z, -1.0 / 355687428096000.0);              // -1/17!
  p = fma(p, z, 1.0 / 1307674368000.0);                 //  1/15!
  p = fma(p, z, -1.0 / 6227020800.0);                   // -1/13!
  p = fma(p, z, 1.0 / 39916800.0);                      //  1/11!
  p = fma(p, z, -1.0 / 362880.0);                       // -1/9!
  p = fma(p, z, 1.0 / 5040.0);                          //  1/7!
  p = fma(p, z, -1.0 / 120.0);                          // -1/5!
  p = fma(p, z, 1.0 / 6.0);                             //  1/3!  (sign folded below)
  // sin r = r - r^3/6 + ... = r - r*z*(1/6 - z/120 + ...)
  const double s = fma(-(r * z), p, r);
  const long long ji = (long long)j;
  return (ji & 1) ? -s : s;
}

// sin and cos of the same argument, same reduction as sin_reduced (|arg| < 3e9; callers
// fall back to the ocml routines beyond): odd / even Taylor polynomials on [-pi/2, pi/2]
// through r^21 / r^22 (truncation < 2e-18).
__device__ __forceinline__ void sincos_reduced(double arg, double& s_out, double& c_out) {
  constexpr double kInvPi = 0.31830988618379067154;
  constexpr double kPiHi = 3.14159265358979311600e+00;
  constexpr double kPiLo = 1.22464679914735317723e-16;
  if (!(fabs(arg) < 3.0e9)) {
    s_out = sin_huge(arg);
    c_out = cos_huge(arg);
    return;
  }
  const double j = rint(arg * kInvPi);
  double r = fma(-j, kPiHi, arg);
  r = fma(-j, kPiLo, r);
  const double z = r * r;
  double p = -1.0 / 51090942171709440000.0;             // -1/21!
  p = fma(p, z, 1.0 / 121645100408832000.0);
  p = fma(p, z, -1.0 / 355687428096000.0);
  p = fma(p, z, 1.0 / 1307674368000.0);
  p = fma(p, z, -1.0 / 6227020800.0);
  p = fma(p, z, 1.0 / 39916800.0);
  p = fma(p, z, -1.0 / 362880.0);
  p = fma(p, z, 1.0 / 5040.0);
  p = fma(p, z, -1.0 / 120.0);
  p = fma(p, z, 1.0 / 6.0);
  double s = fma(-(r * z), p, r);
  double q = 1.0 / 1124000727777607680000.0;            //  1/22!
  q = fma(q, z, -1.0 / 2432902008176640000.0);          // -1/20!
  q = fma(q, z, 1.0 / 6402373705728000.0);              //  1/18!
  q = fma(q, z, -1.0 / 20922789888000.0);               // -1/16!
  q = fma(q, z, 1.0 / 87178291200.0);                   //  1/14!
  q = fma(q, z, -1.0 / 479001600.0);                    // -1/12!
  q = fma(q, z, 1.0 / 3628800.0);                       //  1/10!
  q = fma(q, z, -1.0 / 40320.0);                        // -1/8!
  q = fma(q, z, 1.0 / 720.0);                           //  1/6!
  q = fma(q, z, -1.0 / 24.0);                           // -1/4!
  q = fma(q, z, 0.5);                                   //  1/2!   (sign folded below)
  double c = fma(-z, q, 1.0);
  const long long ji = (long long)j;
  if (ji & 1) {
    s = -s;
    c = -c;
  }
  s_out = s;
  c_out = c;
}

// ---------------------------------------------------------------------------
// The same sin / sincos with the polynomial coefficients read from a table that travels in
// the kernel arguments (EnhanceArgs::trig): uniform loads -> SGPR operands.  As literals the
// 21 coefficients are hoisted into 42 VGPRs for the whole kernel, which costs the
// lane-per-element kernels a resident wave per SIMD.
// ---------------------------------------------------------------------------
struct TrigTables {
  double s[10];   // -1/21! .. 1/3!   (sin_reduced's literals, same order)
  double c[11];   //  1/22! .. 1/2!   (sincos_reduced's)
};

inline TrigTables make_trig_tables() {
  TrigTables t{};
  const double s[10] = {-1.0 / 51090942171709440000.0, 1.0 / 121645100408832000.0,
                        -1.0 / 355687428096000.0,      1.0 / 1307674368000.0,
                        -1.0 / 6227020800.0,           1.0 / 39916800.0,
                        -1.0 / 362880.0,               1.0 / 5040.0,
                        -1.0 / 120.0,                  1.0 / 6.0};
  const double c[11] = {1.0 / 1124000727777607680000.0, -1.0 / 2432902008176640000.0,
                        1.0 / 6402373705728000.0,       -1.0 / 20922789888000.0,
                        1.0 / 87178291200.0,            -1.0 / 479001600.0,
                        1.0 / 3628800.0,                -1.0 / 40320.0,
                        1.0 / 720.0,                    -1.0 / 24.0,
                        0.5};
  for (int i = 0; i < 10; ++i) t.s[i] = s[i];
  for (int i = 0; i < 11; ++i) t.c[i] = c[i];
  return t;
}

// HUGE = false: arguments beyond 3e9 give NaN instead of the library call (the call, even out
// of line, costs the caller ~16 VGPRs; enhance_shared.hip documents the restriction).
template <bool HUGE = true>
__device__ __forceinline__ double sin_tab(double arg, const TrigTables& t) {
  constexpr double kInvPi = 0.31830988618379067154;
  constexpr double kPiHi = 3.14159265358979311600e+00;
  constexpr double kPiLo = 1.22464679914735317723e-16;
  if (!(fabs(arg) < 3.0e9)) {
    if constexpr (HUGE) return sin_huge(arg);
    else return __builtin_nan("");
  }
  const double j = rint(arg * kInvPi);
  double r = fma(-j, kPiHi, arg);
  r = fma(-j, kPiLo, r);
  const double z = r * r;
  double p = t.s[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) p = fma(p, z, t.s[i]);
  const double s = fma(-(r * z), p, r);
  const long long ji = (long long)j;
  return (ji & 1) ? -s : s;
}

template <bool HUGE = true>
__device__ __forceinline__ void sincos_tab(double arg, double& s_out, double& c_out,
                                           const TrigTables& t) {
  constexpr double kInvPi = 0.31830988618379067154;
  constexpr double kPiHi = 3.14159265358979311600e+00;
  constexpr double kPiLo = 1.22464679914735317723e-16;
  if (!(fabs(arg) < 3.0e9)) {
    if constexpr (HUGE) {
      s_out = sin_huge(arg);
      c_out = cos_huge(arg);
    } else {
      s_out = c_out = __builtin_nan("");
    }
    return;
  }
  const double j = rint(arg * kInvPi);
  double r = fma(-j, kPiHi, arg);
  r = fma(-j, kPiLo, r);
  const double z = r * r;
  double p = t.s[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) p = fma(p, z, t.s[i]);
  double s = fma(-(r * z), p, r);
  double q = t.c[0];
#pragma unroll
  for (int i = 1; i < 11; ++i) q = fma(q, z, t.c[i]);
  double c = fma(-z, q, 1.0);
  const long long ji = (long long)j;
  if (ji & 1) {
    s = -s;
    c = -c;
  }
  s_out = s;
  c_out = c;
}

// 1/sqrt(x) to ~1 ulp: hardware seed (v_rsq_f64, ~2^-26 rel.) + two Newton steps.
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps.
__device__ __forceinline__ double rcp_newton(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * fma(-x, y, 2.0);
  y = y * fma(-x, y, 2.0);
  return y;
}

// ---------------------------------------------------------------------------
// Legendre families by three-term recurrences (compile-time coefficients)
// ---------------------------------------------------------------------------
// q[m] = L''_{m+2}(t) = 3 C^{(5/2)}_m(t):  q0 = 3, q1 = 15 t,
//   m q_m = (2m+3) t q_{m-1} - (m+3) q_{m-2}
template <int MR>
__device__ __forceinline__ void legendre_d2(double t, double (&q)[MR]) {
  if constexpr (MR > 0) q[0] = 3.0;
  if constexpr (MR > 1) q[1] = 15.0 * t;
#pragma unroll
  for (int m = 2; m < MR; ++m) {
    const double al = (double)(2 * m + 3) / (double)m;
    const double be = (double)(m + 3) / (double)m;
    q[m] = fma(al * t, q[m - 1], -(be * q[m - 2]));
  }
}

// The same family rescaled so that the recurrence needs one multiply less per term:
//   q_m = c_m p_m,  c_0 = c_1 = 1,  c_m = c_{m-2} (m+3)/m   =>   p_m = A_m t p_{m-1} - p_{m-2},
//   A_m = (2m+3) c_{m-1} / (m c_m).
// Working with p instead of q is a diagonal change of variables of the (M-2) system:
// G' = D^-1 G D^-1, v' = D v, D = diag(c) (enhance_small.hip folds D into the boundary
// block and the final coefficients).
__host__ __device__ constexpr double d2_scale(int m) {
  double c = 1.0;
  for (int i = m; i >= 2; i -= 2) c *= (double)(i + 3) / (double)i;
  return c;
}
__host__ __device__ constexpr double d2_coef(int m) {
  return (double)(2 * m + 3) * d2_scale(m - 1) / ((double)m * d2_scale(m));
}
template <int MR>
__device__ __forceinline__ void legendre_d2_scaled(double t, double (&p)[MR]) {
  if constexpr (MR > 0) p[0] = 3.0;
  if constexpr (MR > 1) p[1] = 15.0 * t;
#pragma unroll
  for (int m = 2; m < MR; ++m) p[m] = fma(d2_coef(m) * t, p[m - 1], -p[m - 2]);
}

// r[m] = L'_{m+1}(t) = C^{(3/2)}_m(t):  r0 = 1, r1 = 3 t,
//   m r_m = (2m+1) t r_{m-1} - (m+1) r_{m-2}
template <int MD>
__device__ __forceinline__ void legendre_d1(double t, double (&r)[MD]) {
  if constexpr (MD > 0) r[0] = 1.0;
  if constexpr (MD > 1) r[1] = 3.0 * t;
#pragma unroll
  for (int m = 2; m < MD; ++m) {
    const double al = (double)(2 * m + 1) / (double)m;
    const double be = (double)(m + 1) / (double)m;
    r[m] = fma(al * t, r[m - 1], -(be * r[m - 2]));
  }
}

// L_p(t), p < M:  (p+1) L_{p+1} = (2p+1) t L_p - p L_{p-1}
template <int M>
__device__ __forceinline__ void legendre_p(double t, double (&L)[M]) {
  L[0] = 1.0;
  if constexpr (M > 1) L[1] = t;
#pragma unroll
  for (int p = 1; p < M - 1; ++p) {
    const double al = (double)(2 * p + 1) / (double)(p + 1);
    const double be = (double)p / (double)(p + 1);
    L[p + 1] = fma(al * t, L[p], -(be * L[p - 1]));
  }
}

// packed lower-triangular index, j <= i
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// ---------------------------------------------------------------------------
// RIDGE-DOMINATED elements of the Chebyshev-moment kernels (Poisson rows): gamma scl^4 below
// ridge_gamma_scl4(M), i.e. eps2 = 2 / (gamma scl^4) above ridge_eps2_threshold(M).  There the ridge
// eps (N + C_z^T C_z), N = Y^T Y, outweighs the moment Gram and the solve in the Chebyshev basis
// inherits cond(Y)^2 (3e-13 at M = 9, 3e-11 at M = 22, 6e-9 at M = 33 against the 60-digit minimiser,
// where the float64 KKT solve holds 1e-15): such elements are solved in the Legendre-bubble basis
// instead, from the same moments (cheb_ridge_solve / ridge_wave_solve).  The crossover of the two
// forms' measured errors (scripts/proto/cheb_moment.py ridge_sweep: smooth and rough right-hand
// sides) moves down with cond(Y_M)^2.  ONE predicate for every kernel: the two-kernel path relies
// on moments_kernel and the solve kernels taking the same decision for an element.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr double ridge_gamma_scl4(int M) {
  return (M <= 12) ? 3.0e-4 : (M <= 17) ? 1.0e-4 : 3.0e-5;
}
// (every branch a compile-time constant: with a run-time M the division would be executed per wave)
__host__ __device__ constexpr double ridge_eps2_threshold(int M) {
  return (M <= 12) ? 2.0 / ridge_gamma_scl4(12) : (M <= 17) ? 2.0 / ridge_gamma_scl4(17) : 2.0 / ridge_gamma_scl4(33);
}
// (inf / NaN -- gamma = 0, a degenerate element -- are not "ridge": they end in the status test.)
// M <= 4 never takes the ridge form: with at most two bubble coefficients Y is DIAGONAL (T_0 = rho_0 / 3,
// T_1 = rho_1 / 15), the two bases differ by a diagonal scaling and the Chebyshev-basis solve is as well
// conditioned as the other (measured 2e-16 at gamma scl^4 = 1e-12).  (The instantiation cheb_ridge_solve<4> also
// aborted with a memory-aperture violation on the MI355X for a reason its ISA does not show; it is not compiled.)
constexpr int kRidgeMinM = 5;
__device__ __forceinline__ bool ridge_dominated(double eps2, int M) {
  return M >= kRidgeMinM && eps2 > ridge_eps2_threshold(M) && eps2 < 1.0e300;
}

}  // namespace lssvr
