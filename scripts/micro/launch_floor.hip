// Microbenchmark (gfx950): what a kernel LAUNCH costs as the dispatch's own begin -> end stamps see it
// (hipExtLaunchKernelGGL events = what rocprofv3 --kernel-trace reports), for an EMPTY kernel, by grid
// size, static LDS and register budget -- the floor under the 9.5 us of the headline lane kernel at
// BASELINE config 2 (391 workgroups of 256 threads, 18 KB LDS, 136 VGPRs).  Also: a kernel that only
// streams out 7.2 MB (the config-2 coefficient write-out), and back-to-back launches per stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_DOUBLES>
__global__ __launch_bounds__(256) void empty_kernel(int* flag) {
  __shared__ double tile[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  if (flag && threadIdx.x == 1025) {            // never: keeps the LDS allocation
    tile[threadIdx.x % (LDS_DOUBLES > 0 ? LDS_DOUBLES : 1)] = 1.0;
    *flag = (int)tile[0];
  }
}

// 136 VGPRs like the lane kernel (an empty kernel otherwise asks for a handful)
__global__ __launch_bounds__(256) void empty_fat_kernel(double* sink) {
  __shared__ double tile[2304];
  double r[60];
#pragma unroll
  for (int i = 0; i < 60; ++i) asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0" : "=v"(((int*)&r[i])[0]), "=v"(((int*)&r[i])[1]));
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 60; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
  for (int i = 0; i < 60; ++i) s += r[i];
  if (s == 1.2345) { tile[threadIdx.x] = s; sink[0] = tile[0]; }
}

__global__ __launch_bounds__(256) void write_kernel(double* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const long j = (long)blockIdx.x * 256 * 9 + k * 256 + threadIdx.x;
    if (j < n) __builtin_nontemporal_store((double)i, &out[j]);
  }
}

template <typename F>
static void stamp(const char* name, F launch, int reps = 200) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    launch(e0, e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  printf("%-58s median %6.2f us  min %6.2f us\n", name, t[t.size() / 2], t[0]);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  int* flag = nullptr; double* out; const long n = 100008L * 9;
  hipMalloc(&out, n * 8 + 4096);
  hipStream_t s; hipStreamCreate(&s);
  const int grids[] = {1, 64, 256, 391, 512, 1024, 4096};
  char name[128];
  for (int g : grids) {
    snprintf(name, sizeof name, "empty kernel, %4d x 256 threads, no LDS", g);
    stamp(name, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(empty_kernel<0>, dim3(g), dim3(256), 0, s, a, b, 0, flag); });
  }
  stamp("empty kernel,  391 x 256 threads, 18 KB LDS", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(empty_kernel<2304>, dim3(391), dim3(256), 0, s, a, b, 0, flag); });
  stamp("empty kernel,  391 x 256 threads, 18 KB LDS, 120+ VGPRs", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(empty_fat_kernel, dim3(391), dim3(256), 0, s, a, b, 0, out); });
  stamp("7.2 MB write-out only, 391 x 256 threads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(write_kernel, dim3(391), dim3(256), 0, s, a, b, 0, out, n); });
  // back-to-back launches on one stream: time per launch from plain events around 200 launches
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; ++i) {
      if (mode == 0) hipLaunchKernelGGL(empty_kernel<0>, dim3(391), dim3(256), 0, s, flag);
      else hipLaunchKernelGGL(write_kernel, dim3(391), dim3(256), 0, s, out, n);
    }
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 back-to-back %s launches on one stream: %.2f us per launch\n", mode ? "7.2 MB write-out" : "empty", ms * 1e3f / 200);
  }
  return 0;
}
