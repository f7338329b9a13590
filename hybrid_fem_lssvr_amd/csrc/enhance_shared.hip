// Shared-operator enhancement for UNIFORM meshes (SURVEY.md section 0 finding 8 / 8(d): "reported
// as a separate line if built"; never the headline path).
//
// In the h-free scaling the per-element system matrix depends on the
// element only through eps = 1/(gamma scl^4): on a uniform mesh every element has the same
// matrix, and the minimiser is LINEAR in the data,
//     w = sum_k P_k f~_k + P_l g_l + P_r g_r,      f~_k = f(x_k) / scl_e^2,
// with P = the (n+2) x M response table of ONE canonical element.  The table is built by the
// general per-element kernel itself (unit right-hand sides on a few elements of the same h,
// ops.build_shared_operator -- no second implementation of the algebra), so this kernel is a
// per-element matrix-vector product: ~(M + 12) instructions per collocation point instead of
// ~65 + a factorisation, 88 B of HBM traffic per element -- the one genuinely HBM-bound form
// of the path.  Everything that depends on the element itself stays exact per element: the
// abscissae x_k (numpy's linspace arithmetic), f(x_k) with numpy's argument rounding, scl_e,
// the boundary values and the Dual.py:65-75 rule.  What is shared is the operator, i.e. the
// (in-kernel f needs |omega x| < 3e9: beyond, the coefficients are NaN -> status FALLBACK)
// rounding of t_k = off + scl x_k of the canonical element instead of each element's own:
// ~(|x|/h) eps relative L2 (1e-11 on BASELINE config 2; tests/test_gpu_shared.py), inside
// north_star's 1e-10 on the BASELINE single-GPU meshes but outside this repository's 1e-13 bar
// for the general path -- hence a separate solver.
//
// One element per lane; the table rows reach the FMAs as SGPR operands (uniform address ->
// s_load batches), coefficients leave through the same wave-private LDS transposition as
// enhance_small_impl.hpp.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

template <int M>
constexpr int kSharedTilePerWave = (64 * 9 > 64 * M) ? 64 * 9 : 64 * M;   // output tile / rhs staging

template <int M, int RHS>
__global__ __launch_bounds__(kBlock) void enhance_shared_kernel(EnhanceArgs p,
                                                                const double* __restrict__ op) {
  __shared__ double tile[(kBlock / 64) * kSharedTilePerWave<M>];
  const int tid = threadIdx.x;
  const int64_t e = (int64_t)blockIdx.x * kBlock + tid;
  double w[M];
  int st = LSSVR_ST_OK;
#pragma unroll
  for (int i = 0; i < M; ++i) w[i] = 0.0;

  // every lane runs the body (the tail of the last wave on a duplicate of the last element,
  // stores masked): a tabulated rhs is loaded cooperatively by the wave
  const bool live = e < p.ne;
  const int64_t ec = live ? e : p.ne - 1;
  const int lane = tid & 63;
  {
    const double a = p.x[ec];
    const double b = p.x[ec + 1];
    const int64_t eg = ec + p.elem_offset;
    const double gl = (eg == 0 && a == p.gxmin) ? p.bc_left : p.u[ec];           // Dual.py:65-75
    const double gr = (eg == p.ne_global - 1 && b == p.gxmax) ? p.bc_right : p.u[ec + 1];
    const int n = p.n;
    const double oldlen = b - a;
    const double step = oldlen / (double)(n - 1);             // numpy's linspace step (exact division)
    // 1 / scl^2 = (h/2)^2: within an ulp of the general kernel's rcp(fl(2/h)^2), no division
    const double inv_scl2 = 0.25 * (oldlen * oldlen);

    // in-kernel rhs: (sin, cos)(th0 + k dth) carried by a rotation, pre-scaled by amp / scl^2
    // (enhance_small_impl.hpp).  While |omega x| stays below 64 over the element, numpy's
    // rounding of the argument fl(omega fl(x_k)) moves f by < 1e-14 relative and the rotated
    // value is used as it is; beyond, the argument is restored to first order per point.
    double rs = 0.0, rc = 1.0, sd = 0.0, cd = 1.0, th0 = 0.0, dth = 0.0, kappa = 0.0;
    bool small_args = false;
    if constexpr (RHS == LSSVR_RHS_SIN) {
      th0 = p.rhs_omega * a;
      dth = p.rhs_omega * step;
      sincos_tab<false>(th0, rs, rc, p.trig);
      if (__all(fabs(dth) < 0.5)) {
        // short Taylor pair for the step angle: truncation 0.5^15/15! < 3e-17
        const double z = dth * dth;
        double ps = -1.0 / 6227020800.0;                    // -1/13!
        ps = fma(ps, z, 1.0 / 39916800.0);
        ps = fma(ps, z, -1.0 / 362880.0);
        ps = fma(ps, z, 1.0 / 5040.0);
        ps = fma(ps, z, -1.0 / 120.0);
        ps = fma(ps, z, 1.0 / 6.0);
        sd = fma(-(dth * z), ps, dth);
        double pc = 1.0 / 87178291200.0;                    //  1/14!
        pc = fma(pc, z, -1.0 / 479001600.0);
        pc = fma(pc, z, 1.0 / 3628800.0);
        pc = fma(pc, z, -1.0 / 40320.0);
        pc = fma(pc, z, 1.0 / 720.0);
        pc = fma(pc, z, -1.0 / 24.0);
        pc = fma(pc, z, 0.5);
        cd = fma(-z, pc, 1.0);
      } else {
        sincos_tab<false>(dth, sd, cd, p.trig);
      }
      kappa = p.rhs_amp * inv_scl2;
      rs *= kappa;
      rc *= kappa;
      small_args = __all(fabs(th0) + (double)n * fabs(dth) < 64.0);
    }
    if (RHS == LSSVR_RHS_SIN && small_args) {
      for (int k = 0; k < n; ++k) {
        const double ft = rs;                                 // f(x_k) / scl^2
        const double rs_next = fma(rs, cd, rc * sd);
        rc = fma(rc, cd, -(rs * sd));
        rs = rs_next;
        const double* __restrict__ Pk = op + (int64_t)k * M;  // uniform: scalar loads
#pragma unroll
        for (int i = 0; i < M; ++i) w[i] = fma(Pk[i], ft, w[i]);
      }
    } else {
      // tabulated rhs: 8 points of the wave's 64 rows at a time through LDS, consecutive lanes
      // on consecutive doubles (enhance_small_impl.hpp; the area is the output tile's)
      constexpr int kK = 8;
      [[maybe_unused]] double* const stg = tile + (tid >> 6) * kSharedTilePerWave<M>;
      [[maybe_unused]] const int64_t e0 = (int64_t)blockIdx.x * kBlock + (tid & ~63);
      [[maybe_unused]] const bool pm = p.tab_ps != 1;      // point-major table: direct coalesced loads
      for (int k = 0; k < n; ++k) {
        double ft;
        if constexpr (RHS == LSSVR_RHS_ARRAY) {
          if (!pm && (k & (kK - 1)) == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < kK; ++i) {
              const int idx = i * 64 + lane;
              const int row = idx / kK, kk = idx % kK;
              const int64_t er = e0 + row;
              const bool in = (er < p.ne) && (k + kk < n);
              stg[row * (kK + 1) + kk] = in ? p.rhs_values[er * n + (k + kk)] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
        if constexpr (RHS == LSSVR_RHS_SIN) {
          const double xk = linspace_at(a, b, oldlen, step, k, n);
          const double arg = p.rhs_omega * xk;
          const double delta = fma(-(double)k, dth, arg - th0);
          ft = fma(rc, delta, rs);
          if (__any(!(fabs(delta) < 1.0e-7))) ft = kappa * sin_tab<false>(arg, p.trig);
          const double rs_next = fma(rs, cd, rc * sd);
          rc = fma(rc, cd, -(rs * sd));
          rs = rs_next;
        } else {
          ft = (pm ? p.rhs_values[ec * p.tab_es + k * p.tab_ps]
                   : stg[lane * (kK + 1) + (k & (kK - 1))]) * inv_scl2;
        }
        const double* __restrict__ Pk = op + (int64_t)k * M;
#pragma unroll
        for (int i = 0; i < M; ++i) w[i] = fma(Pk[i], ft, w[i]);
      }
    }
    const double* __restrict__ Pl = op + (int64_t)n * M;
    const double* __restrict__ Pr = Pl + M;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      w[i] = fma(Pl[i], gl, w[i]);
      w[i] = fma(Pr[i], gr, w[i]);
    }
    // the map is linear with a finite table: a non-finite input (f, g_l, g_r, h) reaches every
    // coefficient of its parity, so one test on w_0 + w_1 covers them all
    const bool ok = fabs(w[0] + (M > 1 ? w[1] : 0.0)) < 1.0e300;
    if (!ok) {                                                // Dual.py:164-169
      st = LSSVR_ST_FALLBACK;
#pragma unroll
      for (int i = 0; i < M; ++i) w[i] = 0.0;
      w[0] = 0.5 * (gl + gr);
      if constexpr (M > 1) w[1] = 0.5 * (gr - gl);
      if (live && p.fail_count) atomicAdd(p.fail_count, 1);
    }
    if (live && p.status) p.status[e] = st;
  }

  // coalesced store: each wave transposes its own 64 x M tile through wave-private LDS
  double* const wt = tile + (tid >> 6) * kSharedTilePerWave<M>;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the staging reads are done)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < M; ++i) wt[lane * M + i] = w[i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = ((int64_t)blockIdx.x * kBlock + (tid & ~63)) * M;
  const int64_t total = p.ne * M;
  if (total <= kWriteThroughMaxDoubles) {
    // small outputs: write-through stores (see kWriteThroughMaxDoubles)
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int64_t idx = base + (int64_t)i * 64 + lane;
      if (idx < total) __hip_atomic_store(&p.W[idx], wt[i * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int64_t idx = base + (int64_t)i * 64 + lane;
    if (idx < total) __builtin_nontemporal_store(wt[i * 64 + lane], &p.W[idx]);
  }
}

template <int M>
static hipError_t launch_shared(const EnhanceArgs& a, const double* op, hipStream_t s,
                                const LaunchOpts* o) {
  const unsigned blocks = (unsigned)((a.ne + kBlock - 1) / kBlock);
  if (a.rhs_id == LSSVR_RHS_SIN)
    return launch(enhance_shared_kernel<M, LSSVR_RHS_SIN>, dim3(blocks), dim3(kBlock), s, o, a, op);
  return launch(enhance_shared_kernel<M, LSSVR_RHS_ARRAY>, dim3(blocks), dim3(kBlock), s, o, a, op);
}

hipError_t enhance_shared(const EnhanceArgs& a, const double* op, hipStream_t s,
                          const LaunchOpts* o) {
  switch (a.M) {
#define LSSVR_SH(MM) \
  case MM:           \
    return launch_shared<MM>(a, op, s, o);
    LSSVR_SH(2) LSSVR_SH(3) LSSVR_SH(4) LSSVR_SH(5) LSSVR_SH(6) LSSVR_SH(7) LSSVR_SH(8) LSSVR_SH(9)
    LSSVR_SH(10) LSSVR_SH(11) LSSVR_SH(12) LSSVR_SH(13) LSSVR_SH(14) LSSVR_SH(15) LSSVR_SH(16)
    LSSVR_SH(17) LSSVR_SH(18) LSSVR_SH(19) LSSVR_SH(20) LSSVR_SH(21) LSSVR_SH(22) LSSVR_SH(23) LSSVR_SH(24)
    LSSVR_SH(25) LSSVR_SH(26) LSSVR_SH(27) LSSVR_SH(28) LSSVR_SH(29) LSSVR_SH(30) LSSVR_SH(31) LSSVR_SH(32)
    LSSVR_SH(33)
#undef LSSVR_SH
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace lssvr
