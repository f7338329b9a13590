// Development probe: semantics of v_permlane16_swap_b32 and v_fmac_f64_dpp row_newbcast on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned* o2, unsigned long long execmask) {
  const int lane = threadIdx.x;
  unsigned x = lane, y = lane;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o2[lane] = r[0];
  o2[64 + lane] = r[1];
  double rowdata = (double)lane, mul = 1.0, acc = 0.0, acc2 = 1000.0;
  asm volatile("s_nop 1" : "+v"(rowdata));
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rowdata), "v"(mul));
  out[lane] = acc;
  // source lane inactive: only lanes selected by execmask run the DPP op
  if ((execmask >> lane) & 1ull) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(rowdata), "v"(mul));
  }
  out[64 + lane] = acc2;
}
int main() {
  double* out; unsigned* o2;
  hipMalloc(&out, 128 * 8); hipMalloc(&o2, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, o2, 0xffffffffffffffdfull & ~(1ull << 21));
  double h[128]; unsigned g[128];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(g, o2, sizeof(g), hipMemcpyDeviceToHost);
  printf("swap r0:"); for (int i = 0; i < 64; ++i) printf(" %u", g[i]); printf("\nswap r1:"); for (int i = 0; i < 64; ++i) printf(" %u", g[64 + i]);
  printf("\nbcast5 :"); for (int i = 0; i < 64; ++i) printf(" %g", h[i]);
  printf("\nbcast5 with lanes 5 and 21 inactive (acc2 starts at 1000):"); for (int i = 0; i < 64; ++i) printf(" %g", h[64 + i]);
  printf("\n");
  return 0;
}
