// Microbenchmark (gfx950): cycles per DEPENDENT instruction for the chains the wave-level solvers are
// made of.  One workgroup of W waves per CU region; prints s_memtime deltas / chain length.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int I>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double rowdata, double mul) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rowdata), "v"(mul), "n"(I));
}

constexpr int N = 256;

template <int MODE>
__global__ void chain(double* out, long long* cyc, double seed) {
  double y = seed + threadIdx.x * 1e-3, a = 1.0000001, b = 1e-9, c = seed;
  double y2 = y + 1.0;
  asm volatile("" : "+v"(y), "+v"(y2));
  long long t0 = __builtin_readcyclecounter();
  asm volatile("" : "+v"(y), "+v"(y2), "+s"(t0));
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (MODE == 0) { y = fma(y, a, b); asm volatile("" : "+v"(y)); }  // plain dependent FMA
    if constexpr (MODE == 1) { fmac_rowbcast<3>(y, c, a); asm volatile("" : "+v"(y)); }   // DPP source constant, accumulator chain
    if constexpr (MODE == 2) { asm volatile("s_nop 1" : "+v"(y)); fmac_rowbcast<3>(y, y, b); }   // DPP source = previous result
    if constexpr (MODE == 3) { double nz = y * a; asm volatile("s_nop 1" : "+v"(nz)); fmac_rowbcast<3>(y, nz, b); }  // backward step
    if constexpr (MODE == 4) { y = fma(y, a, b); y2 = fma(y2, a, b); asm volatile("" : "+v"(y), "+v"(y2)); }   // two independent plain chains
    if constexpr (MODE == 5) { asm volatile("s_nop 1" : "+v"(y), "+v"(y2)); fmac_rowbcast<3>(y, y, b); fmac_rowbcast<5>(y2, y2, b); }
    if constexpr (MODE == 6) { y = y * a; asm volatile("" : "+v"(y)); }        // dependent mul
    if constexpr (MODE == 7) { y = __builtin_amdgcn_rcp(y) + 1.0; asm volatile("" : "+v"(y)); }   // rcp + add
  }
  asm volatile("" : "+v"(y), "+v"(y2));
  long long t1 = __builtin_readcyclecounter();
  asm volatile("" : "+v"(y), "+v"(y2), "+s"(t1));
  out[blockIdx.x * blockDim.x + threadIdx.x] = y + y2;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main(int argc, char** argv) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 64 * 8);
  const char* names[] = {"plain fma chain", "dpp fmac, acc chain", "dpp fmac, dpp src = prev (nop 1)", "mul -> nop -> dpp fmac", "2 plain chains (per pair)", "2 dpp-src chains (per pair)", "mul chain", "rcp+add chain (per 2 ops)"};
  for (int waves_per_simd : {1, 2, 3, 4}) {
    const int block = 256 * waves_per_simd;     // 4 SIMDs per CU
    (void)hipMemset(cyc, 0, 64 * 8);
#define RUN(M) hipLaunchKernelGGL(chain<M>, dim3(256 * 2), dim3(block > 1024 ? 1024 : block), 0, 0, out, cyc, 1.5)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7);
    (void)hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("waves/SIMD (block %d):", block);
    for (int m = 0; m < 8; ++m) printf("\n   %-36s %.1f cycles per step", names[m], (double)h[m] / N);
    printf("\n");
  }
  return 0;
}
