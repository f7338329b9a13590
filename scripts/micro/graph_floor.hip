// Microbenchmark (gfx950): the per-launch floor of bench.py's TIMED REGION -- K dependent launches on one
// stream captured in a hipGraph and replayed, timed with an event pair around the replay -- for EMPTY kernels
// with the headline lane kernel's launch geometry (391 workgroups x 256 threads, 18 KB static LDS, 136 VGPRs,
// ~400 B of kernel arguments by value), for the one-wave-workgroup geometry (1563 x 64 threads, 4.6 KB LDS),
// and for the same launches issued eagerly.  What an empty kernel costs here is what NO kernel body can save.
// (profiles/r03_launch_floor.txt measured begin -> end stamps of isolated launches, 4.1 us, and 200 eager
// back-to-back launches, 3.1 us: the first includes the stamping, the second is the HOST's launch rate.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct FatArgs { double d[48]; void* p[4]; };     // 416 B by value, like EnhanceArgs + P1Args + QuadRule

template <int THREADS, int LDS_DOUBLES, bool FAT>
__global__ __launch_bounds__(THREADS) void empty_kernel(FatArgs a, double* sink) {
  __shared__ double tile[LDS_DOUBLES];
  if constexpr (FAT) {
    double r[60];
#pragma unroll
    for (int i = 0; i < 60; ++i) asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0" : "=v"(((int*)&r[i])[0]), "=v"(((int*)&r[i])[1]));
#pragma unroll
    for (int i = 0; i < 60; ++i) asm volatile("" : "+v"(r[i]));
    double s = a.d[3];
#pragma unroll
    for (int i = 0; i < 60; ++i) s += r[i];
    if (s == 1.2345) { tile[threadIdx.x % LDS_DOUBLES] = s; sink[0] = tile[0]; }
  } else {
    if (a.d[3] == 1.2345 && threadIdx.x == 9999) { tile[0] = 1.0; sink[0] = tile[0]; }
  }
}

// a body of N dependent FP64 FMAs per lane (no memory): the issue time of a lone wave
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fma_kernel(FatArgs a, double* sink, int n) {
  double x = a.d[0] + threadIdx.x, y = a.d[1];
  for (int i = 0; i < n; ++i) x = fma(x, y, 1.0);
  if (x == 1.2345) sink[0] = x;
}

template <typename F>
static void run(const char* name, F launch, int K) {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2000; ++i) launch(s);            // steady state
  hipStreamSynchronize(s);
  // eager
  std::vector<float> te, tg;
  for (int r = 0; r < 21; ++r) {
    hipEventRecord(e0, s);
    for (int i = 0; i < K; ++i) launch(s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); te.push_back(ms * 1e3f / K);
  }
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < K; ++i) launch(s);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  for (int r = 0; r < 21; ++r) {
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tg.push_back(ms * 1e3f / K);
  }
  std::sort(te.begin(), te.end()); std::sort(tg.begin(), tg.end());
  printf("%-64s K=%3d  graph replay %6.2f us/launch (min %5.2f)   eager %6.2f us/launch (min %5.2f)\n", name, K,
         tg[tg.size() / 2], tg[0], te[te.size() / 2], te[0]);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
}

int main() {
  double* sink; hipMalloc(&sink, 4096);
  FatArgs a{};
  for (int K : {20, 200}) {
    run("empty, 391 x 256 threads, 18 KB LDS, 136 VGPRs, 416 B kernarg",
        [&](hipStream_t s) { hipLaunchKernelGGL((empty_kernel<256, 2304, true>), dim3(391), dim3(256), 0, s, a, sink); }, K);
    run("empty, 391 x 256 threads, no LDS, few VGPRs",
        [&](hipStream_t s) { hipLaunchKernelGGL((empty_kernel<256, 1, false>), dim3(391), dim3(256), 0, s, a, sink); }, K);
    run("empty, 1563 x 64 threads, 4.6 KB LDS, 136 VGPRs",
        [&](hipStream_t s) { hipLaunchKernelGGL((empty_kernel<64, 576, true>), dim3(1563), dim3(64), 0, s, a, sink); }, K);
    run("empty, 1 x 64 threads",
        [&](hipStream_t s) { hipLaunchKernelGGL((empty_kernel<64, 1, false>), dim3(1), dim3(64), 0, s, a, sink); }, K);
  }
  for (int n : {300, 1258, 2516}) {
    char name[128];
    snprintf(name, sizeof name, "%d dependent FMAs per lane, 391 x 256 threads", n);
    run(name, [&](hipStream_t s) { hipLaunchKernelGGL((fma_kernel<256>), dim3(391), dim3(256), 0, s, a, sink, n); }, 20);
    snprintf(name, sizeof name, "%d dependent FMAs per lane, 1563 x 64 threads", n);
    run(name, [&](hipStream_t s) { hipLaunchKernelGGL((fma_kernel<64>), dim3(1563), dim3(64), 0, s, a, sink, n); }, 20);
    snprintf(name, sizeof name, "%d dependent FMAs per lane, 1024 x 64 threads (one wave per SIMD)", n);
    run(name, [&](hipStream_t s) { hipLaunchKernelGGL((fma_kernel<64>), dim3(1024), dim3(64), 0, s, a, sink, n); }, 20);
    snprintf(name, sizeof name, "%d dependent FMAs per lane, 2048 x 64 threads (two waves per SIMD)", n);
    run(name, [&](hipStream_t s) { hipLaunchKernelGGL((fma_kernel<64>), dim3(2048), dim3(64), 0, s, a, sink, n); }, 20);
  }
  return 0;
}
