// Development probe (not part of the library; see scripts/probes/README.md): v_mfma_f64_4x4x4_4b_f64 operand layout and rate
// versus v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(int hot, int which, double* out) {
  const int lane = threadIdx.x;
  double a = 1.0, b = 1.0;
  if (which == 0) a = (lane == hot) ? 1.0 : 0.0;
  else b = (lane == hot) ? 1.0 : 0.0;
  double c = 0.0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[lane] = c;
}

__global__ __launch_bounds__(256) void rate44(double* out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
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

__global__ __launch_bounds__(256) void rate16(double* out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
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

__global__ __launch_bounds__(256) void ratefma(double* out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  double a0 = 1.0 + tid * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
  double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
  const double m = 0.999999, c = 1e-7;
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
    a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
  }
  out[tid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

template <typename K>
static float timeit(K k, int blocks, double* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  double* out;
  hipMalloc(&out, 8192 * 256 * sizeof(double));
  std::vector<double> h(64);
  for (int which = 0; which < 2; ++which) {
    printf("== one-hot %s, other operand all ones: output lanes that see it\n", which ? "B" : "A");
    for (int hot = 0; hot < 64; ++hot) {
      hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, hot, which, out);
      hipMemcpy(h.data(), out, 64 * sizeof(double), hipMemcpyDeviceToHost);
      printf("hot %2d:", hot);
      for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf(" %d", l);
      printf("\n");
    }
  }
  const int blocks = 4096;
  for (int wpc = 0; wpc < 2; ++wpc) {
    const int B = wpc ? 1024 : blocks;   // 1024 blocks = one wave per SIMD
    const int it = wpc ? 8192 : 2048;
    float t44 = timeit(rate44, B, out, it);
    float t16 = timeit(rate16, B, out, it);
    float tf = timeit(ratefma, B, out, it * 4);
    const double nsimd = 1024.0, waves = B * 4.0;
    printf("blocks %d: 4x4x4_4b: %.3f ms -> %.1f ns per MFMA per SIMD (256 FMA each);  16x16x4: %.3f ms -> %.1f ns per MFMA per SIMD (1024 FMA each);  v_fma_f64: %.3f ms -> %.2f ns per wave-FMA per SIMD (64 FMA each)\n",
           B, t44, t44 * 1e6 / (waves / nsimd * it * 8.0), t16, t16 * 1e6 / (waves / nsimd * it * 4.0),
           tf, tf * 1e6 / (waves / nsimd * it * 4.0 * 8.0));
  }
  return 0;
}
