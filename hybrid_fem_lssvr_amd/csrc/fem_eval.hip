// Kernels either side of the enhancement: collocation / quadrature abscissae,
// element-local P1 assembly (Dual.py:117-128), evaluate_solution (Dual.py:176-203)
// and the FP64 peak probe used for the roofline.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"
#include "lssvr_p1.hpp"

namespace lssvr {

// ---------------------------------------------------------------------------
// np.linspace(x[e], x[e+1], n) for every element  (Dual.py:40)
// ---------------------------------------------------------------------------
template <bool PM>
__global__ __launch_bounds__(kBlock) void colloc_points_kernel(const double* __restrict__ x,
                                                                int64_t ne, int n,
                                                                double* __restrict__ xc) {
  const int64_t total = ne * n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    // output index i: element-major (e, k) = (i / n, i % n); point-major (k, e) = (i / ne, i % ne)
    const int64_t e = PM ? i % ne : i / n;
    const int k = (int)(PM ? i / ne : i - e * n);
    const double a = x[e], b = x[e + 1];
    const double delta = b - a;
    const double step = delta / (double)(n - 1);
    xc[i] = linspace_at(a, b, delta, step, k, n);
  }
}

hipError_t colloc_points(const double* x, int64_t ne, int n, double* xc, hipStream_t s, bool point_major) {
  if (ne == 0) return hipSuccess;
  const int64_t total = ne * n;
  const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock < 8192 ? (total + kBlock - 1) / kBlock : 8192);
  if (point_major)
    hipLaunchKernelGGL(colloc_points_kernel<true>, dim3(blocks), dim3(kBlock), 0, s, x, ne, n, xc);
  else
    hipLaunchKernelGGL(colloc_points_kernel<false>, dim3(blocks), dim3(kBlock), 0, s, x, ne, n, xc);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void quad_points_kernel(const double* __restrict__ x,
                                                              int64_t ne, int nq, QuadRule q,
                                                              double* __restrict__ xq) {
  const int64_t total = ne * nq;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t e = i / nq;
    const int k = (int)(i - e * nq);
    const double a = x[e];
    const double h = x[e + 1] - a;
    xq[i] = a + h * q.xi[k];
  }
}

hipError_t quad_points(const double* x, int64_t ne, int nquad, double* xq, hipStream_t s) {
  QuadRule q;
  if (!quad_rule(nquad, q)) return hipErrorInvalidValue;
  if (ne == 0) return hipSuccess;
  const int64_t total = ne * nquad;
  const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock < 8192 ? (total + kBlock - 1) / kBlock : 8192);
  hipLaunchKernelGGL(quad_points_kernel, dim3(blocks), dim3(kBlock), 0, s, x, ne, nquad, q, xq);
  return hipGetLastError();
}

template <bool SIN>
__global__ __launch_bounds__(kBlock) void p1_assemble_kernel(P1Args p, QuadRule q) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= p.ne;
       i += (int64_t)gridDim.x * kBlock)
    p1_node<SIN>(p, q, i);
}

hipError_t p1_assemble(const P1Args& a, hipStream_t s) {
  QuadRule q;
  if (!quad_rule(a.nquad, q)) return hipErrorInvalidValue;
  const int64_t nn = a.ne + 1;
  const unsigned blocks = (unsigned)((nn + kBlock - 1) / kBlock < 16384 ? (nn + kBlock - 1) / kBlock : 16384);
  if (a.rhs_id == LSSVR_RHS_SIN)
    hipLaunchKernelGGL(p1_assemble_kernel<true>, dim3(blocks), dim3(kBlock), 0, s, a, q);
  else
    hipLaunchKernelGGL(p1_assemble_kernel<false>, dim3(blocks), dim3(kBlock), 0, s, a, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// evaluate_solution (Dual.py:176-203)
// ---------------------------------------------------------------------------
// Element of a query point: the first j with x[j] <= xq <= x[j+1]
//   == clamp(#{nodes < xq} - 1, 0, ne-1)      (a point on an interior node takes
// the left element; outside the mesh the first / last element extrapolates).
__device__ __forceinline__ int64_t locate(const double* __restrict__ x, int64_t ne, double xq,
                                          double x0, double inv_h) {
  // uniform-mesh guess, verified against the actual nodes
  double g = (xq - x0) * inv_h;
  int64_t j = g > 0.0 ? (g < (double)(ne - 1) ? (int64_t)g : ne - 1) : 0;
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    const bool lo_ok = (j == 0) || (x[j] < xq);
    const bool hi_ok = (j == ne - 1) || (xq <= x[j + 1]);
    if (lo_ok && hi_ok) return j;
    j += lo_ok ? 1 : -1;
  }
  // general mesh: lower_bound over the ne+1 nodes
  int64_t lo = 0, hi = ne + 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (x[mid] < xq) lo = mid + 1; else hi = mid;
  }
  j = lo - 1;
  return j < 0 ? 0 : (j > ne - 1 ? ne - 1 : j);
}

// value of the hybrid solution at one point (Dual.py:182-201); j_out = element used (-1: NaN)
__device__ __forceinline__ double eval_point(const double* __restrict__ x,
                                             const double* __restrict__ W, int64_t ne, int M,
                                             double xi, double x0, double inv_h, int64_t& j_out) {
  if (xi != xi) {  // NaN: no branch of Dual.py:182-201 fires, the zero stays
    j_out = -1;
    return 0.0;
  }
  {
    const int64_t j = locate(x, ne, xi, x0, inv_h);
    j_out = j;
    const DomainMap dm = map_params(x[j], x[j + 1]);
    const double t = dm.off + dm.scl * xi;      // mapdomain, two roundings
    const double* c = W + j * M;
    double c0, c1;
    if (M == 1) {
      c0 = c[0];
      c1 = 0.0;
    } else if (M == 2) {
      c0 = c[0];
      c1 = c[1];
    } else {
      // numpy legval: c0 = c[-i] - (c1*(nd-1))/nd ; c1 = tmp + (c1*x*(2*nd-1))/nd
      int nd = M;
      c0 = c[M - 2];
      c1 = c[M - 1];
      for (int k = 3; k <= M; ++k) {
        const double tmp = c0;
        nd = nd - 1;
        c0 = c[M - k] - (c1 * (double)(nd - 1)) / (double)nd;
        c1 = tmp + ((c1 * t) * (double)(2 * nd - 1)) / (double)nd;
      }
    }
    return c0 + c1 * t;
  }
}

__global__ __launch_bounds__(kBlock) void eval_kernel(const double* __restrict__ x,
                                                       const double* __restrict__ W, int64_t ne,
                                                       int M, const double* __restrict__ xq,
                                                       int64_t P, double* __restrict__ uq,
                                                       int64_t* __restrict__ elem) {
  const double x0 = x[0];
  const double inv_h = (double)ne / (x[ne] - x0);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P;
       i += (int64_t)gridDim.x * kBlock) {
    int64_t j;
    uq[i] = eval_point(x, W, ne, M, xq[i], x0, inv_h, j);
    if (elem) elem[i] = j;
  }
}

// Error norms of the hybrid solution against amp*sin(omega x) (Dual.py:216-217 computes
// `computed_solution` and `exact_solution` on the test points):
// out[0] += sum (u - ex)^2, out[1] += sum ex^2, out[2] = max(out[2], max |u - ex|).
// One partial per workgroup (LDS tree), then three atomics per workgroup.
__global__ __launch_bounds__(kBlock) void eval_error_kernel(const double* __restrict__ x,
                                                             const double* __restrict__ W,
                                                             int64_t ne, int M,
                                                             const double* __restrict__ xq, int64_t P,
                                                             double amp, double omega,
                                                             double* __restrict__ out) {
  __shared__ double sh[3][kBlock];
  const double x0 = x[0];
  const double inv_h = (double)ne / (x[ne] - x0);
  double se = 0.0, sx = 0.0, mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P;
       i += (int64_t)gridDim.x * kBlock) {
    const double xi = xq[i];
    int64_t j;
    const double u = eval_point(x, W, ne, M, xi, x0, inv_h, j);
    if (j < 0) continue;
    const double ex = amp * sin_reduced(omega * xi);
    const double d = u - ex;
    se = fma(d, d, se);
    sx = fma(ex, ex, sx);
    mx = fmax(mx, fabs(d));
  }
  sh[0][threadIdx.x] = se;
  sh[1][threadIdx.x] = sx;
  sh[2][threadIdx.x] = mx;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
      sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], sh[0][0]);
    atomicAdd(&out[1], sh[1][0]);
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(&out[2]),
              (unsigned long long)__double_as_longlong(sh[2][0]));
  }
}

hipError_t eval_error(const double* x, const double* W, int64_t ne, int M, const double* xq,
                      int64_t P, double amp, double omega, double* out, hipStream_t s) {
  if (P == 0) return hipSuccess;
  const int64_t b = (P + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(eval_error_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(kBlock), 0, s, x,
                     W, ne, M, xq, P, amp, omega, out);
  return hipGetLastError();
}

hipError_t eval_points(const double* x, const double* W, int64_t ne, int M, const double* xq,
                       int64_t P, double* uq, int64_t* elem, hipStream_t s) {
  if (P == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((P + kBlock - 1) / kBlock < 16384 ? (P + kBlock - 1) / kBlock : 16384);
  hipLaunchKernelGGL(eval_kernel, dim3(blocks), dim3(kBlock), 0, s, x, W, ne, M, xq, P, uq, elem);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// FP64 peak probe
// ---------------------------------------------------------------------------
// use_mfma >= 100: VALU probe with only the first (use_mfma - 100) lanes of every wave active
// (does a partially masked wave64 FP64 instruction cost fewer cycles?)
__global__ __launch_bounds__(kBlock) void fp64_fma_masked_probe_kernel(double* out, int iters,
                                                                        int active) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  if ((threadIdx.x & 63) >= active) return;
  double a0 = 1.0 + tid * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
  double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
  const double m = 0.999999, c = 1e-7;
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
    a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
  }
  out[tid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

__global__ __launch_bounds__(kBlock) void fp64_fma_probe_kernel(double* out, int iters) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  double a0 = 1.0 + tid * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
  double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
  const double m = 0.999999, c = 1e-7;
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
    a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
  }
  out[tid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void fp64_mfma_probe_kernel(double* out, int iters) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + (tid & 63) * 1e-3, b = 1.0 - (tid & 63) * 1e-3;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[tid] = (c0[0] + c1[1]) + (c2[2] + c3[3]);
}

// the 4-block instruction the degree-32 kernel's Gram contraction uses (256 FMAs each)
__global__ __launch_bounds__(kBlock) void fp64_mfma4_probe_kernel(double* out, int iters) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  const double a = 1.0 + (tid & 63) * 1e-3, b = 1.0 - (tid & 63) * 1e-3;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[tid] = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
}

// Do the f64 MFMA and the FP64 vector FMA overlap on one SIMD?  Even workgroups run the MFMA
// loop, odd ones the FMA loop with 8x the iterations (same pipe time each); every SIMD hosts
// both kinds.  Overlapping pipes finish in ~max of the two single-kind runs, a shared pipe
// in ~their sum (scripts/maskprobe.py prints the three durations).
__global__ __launch_bounds__(kBlock) void fp64_mixed_probe_kernel(double* out, int iters) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.x & 1) {
    double a0 = 1.0 + tid * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
    double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double m = 0.999999, c = 1e-7;
    for (int i = 0; i < 8 * iters; ++i) {
      a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
      a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
    }
    out[tid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  } else {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + (tid & 63) * 1e-3, b = 1.0 - (tid & 63) * 1e-3;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[tid] = (c0[0] + c1[1]) + (c2[2] + c3[3]);
  }
}

// 8-byte-per-lane streaming copy with a known byte count: calibrates rocprofv3's
// FETCH_SIZE / WRITE_SIZE for the access width the enhancement kernels use.
__global__ __launch_bounds__(kBlock) void stream_copy_probe_kernel(const double* __restrict__ src,
                                                                    double* __restrict__ dst,
                                                                    int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    dst[i] = src[i] + 1.0;
}

hipError_t stream_probe(const double* src, double* dst, int64_t n, hipStream_t s) {
  const int64_t b = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(stream_copy_probe_kernel, dim3((unsigned)(b < 16384 ? b : 16384)),
                     dim3(kBlock), 0, s, src, dst, n);
  return hipGetLastError();
}

// The access pattern of the tabulated-input staging of the lane kernels (enhance_small_impl.hpp):
// a wave owns 64 consecutive rows of `rowlen` doubles and reads them CHUNK columns at a time,
// consecutive lanes on consecutive doubles of a row (CHUNK*8-byte runs, rows rowlen*8 bytes apart), the
// next CHUNK columns by later instructions.  Known byte count (nrows*rowlen*8 read, nrows*8 written):
// calibrates FETCH_SIZE for that pattern (CHUNK = 8: half-line requests) and measures the bandwidth
// the pattern itself can reach (the full-line stream probe above reaches 6.2 TB/s).
template <int CHUNK>
__global__ __launch_bounds__(kBlock) void row_chunk_probe_kernel(const double* __restrict__ src,
                                                                  double* __restrict__ dst,
                                                                  int64_t nrows, int rowlen) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv * 64 < nrows; wv += nwave) {
    const int64_t r0 = wv * 64;
    double acc = 0.0;
    for (int k0 = 0; k0 < rowlen; k0 += CHUNK) {
#pragma unroll
      for (int i = 0; i < CHUNK; ++i) {
        const int idx = i * 64 + lane;
        const int64_t row = r0 + idx / CHUNK;
        const int kk = k0 + idx % CHUNK;
        if (row < nrows && kk < rowlen) acc += src[row * rowlen + kk];
      }
      asm volatile("" : "+v"(acc));      // one batch lands before the next is issued, like the staging
    }
    if (r0 + lane < nrows) dst[r0 + lane] = acc;
  }
}

hipError_t row_chunk_probe(const double* src, double* dst, int64_t nrows, int rowlen, int chunk,
                           hipStream_t s) {
  const int64_t b = (nrows + kBlock - 1) / kBlock;
  const dim3 grid((unsigned)(b < 65536 ? b : 65536));
  if (chunk == 16)
    hipLaunchKernelGGL(row_chunk_probe_kernel<16>, grid, dim3(kBlock), 0, s, src, dst, nrows, rowlen);
  else if (chunk == 8)
    hipLaunchKernelGGL(row_chunk_probe_kernel<8>, grid, dim3(kBlock), 0, s, src, dst, nrows, rowlen);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t fp64_probe(double* out, int blocks, int iters, int use_mfma, hipStream_t s) {
  if (use_mfma == 3)
    hipLaunchKernelGGL(fp64_mfma4_probe_kernel, dim3(blocks), dim3(kBlock), 0, s, out, iters);
  else if (use_mfma == 2)
    hipLaunchKernelGGL(fp64_mixed_probe_kernel, dim3(blocks), dim3(kBlock), 0, s, out, iters);
  else if (use_mfma >= 100)
    hipLaunchKernelGGL(fp64_fma_masked_probe_kernel, dim3(blocks), dim3(kBlock), 0, s, out, iters,
                       use_mfma - 100);
  else if (use_mfma)
    hipLaunchKernelGGL(fp64_mfma_probe_kernel, dim3(blocks), dim3(kBlock), 0, s, out, iters);
  else
    hipLaunchKernelGGL(fp64_fma_probe_kernel, dim3(blocks), dim3(kBlock), 0, s, out, iters);
  return hipGetLastError();
}

}  // namespace lssvr
