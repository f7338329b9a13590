// Microbenchmark (gfx950): what the MEMORY side delivers for BASELINE config 5's point-major table pattern on
// its own -- three tables t[k * ne + e], 16 points, one element per lane, every load a coalesced 512-byte wave
// request, 48 requests per wave -- at the resident-wave counts the enhancement kernel runs at (two per SIMD)
// and at full occupancy, with all 48 loads of a lane in flight at once or in groups of 12 (four points, as
// the kernel's register prefetch).  Output: us and TB/s of the 384 B per element read (+ 8 B written).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int GROUP, int LDS_BYTES>
__global__ __launch_bounds__(256) void pm_kernel(const double* __restrict__ f, const double* __restrict__ a,
                                                 const double* __restrict__ d, double* __restrict__ out, long ne) {
  __shared__ double pad[LDS_BYTES / 8];
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  double s = 0.0;
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += GROUP) {
    double v[3 * GROUP];
#pragma unroll
    for (int i = 0; i < GROUP; ++i) {
      v[3 * i] = __builtin_nontemporal_load(f + (long)(k0 + i) * ne + e);
      v[3 * i + 1] = __builtin_nontemporal_load(a + (long)(k0 + i) * ne + e);
      v[3 * i + 2] = __builtin_nontemporal_load(d + (long)(k0 + i) * ne + e);
    }
#pragma unroll
    for (int i = 0; i < 3 * GROUP; ++i) s += v[i];
  }
  if (s == 1.2345e300) pad[threadIdx.x] = s;
  out[e] = s;
}

template <int GROUP, int LDS_BYTES>
static void run(const char* name, const double* f, const double* a, const double* d, double* out, long ne) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  const unsigned grid = (unsigned)((ne + 255) / 256);
  for (int r = 0; r < 21; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((pm_kernel<GROUP, LDS_BYTES>), dim3(grid), dim3(256), 0, 0, f, a, d, out, ne);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  printf("%-72s median %6.1f us  min %6.1f us = %.2f TB/s\n", name, t[10], t[0], 392.0 * ne / (t[10] * 1e-6) / 1e12);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  const long ne = 1000000;
  double *f, *a, *d, *out;
  hipMalloc(&f, ne * 16 * 8); hipMalloc(&a, ne * 16 * 8); hipMalloc(&d, ne * 16 * 8); hipMalloc(&out, ne * 8);
  hipMemset(f, 0, ne * 128); hipMemset(a, 0, ne * 128); hipMemset(d, 0, ne * 128);
  run<16, 1024>("all 48 loads of a lane in flight, full occupancy", f, a, d, out, ne);
  run<4, 1024>("groups of 4 points (12 loads), full occupancy", f, a, d, out, ne);
  run<16, 80 * 1024>("all 48 loads in flight, 2 waves per SIMD (80 KB LDS per block)", f, a, d, out, ne);
  run<4, 80 * 1024>("groups of 4 points, 2 waves per SIMD", f, a, d, out, ne);
  run<2, 80 * 1024>("groups of 2 points, 2 waves per SIMD", f, a, d, out, ne);
  run<4, 40 * 1024>("groups of 4 points, 3 waves per SIMD (40 KB LDS per block)", f, a, d, out, ne);
  return 0;
}
