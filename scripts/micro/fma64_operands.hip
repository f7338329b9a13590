// Microbenchmark (gfx950): what one FP64 vector instruction costs a SIMD as a function of WHERE ITS OPERANDS COME
// FROM and of how many waves share the SIMD.  Question behind it: the variable-coefficient
// lane kernel and moments_kernel issue one instruction per ~6 cycles whatever their occupancy, prefetch depth or
// table layout, the Poisson lane kernel 4.3 with three waves -- is a v_fma_f64 with three distinct VGPR-pair
// operands slower than one with a constant / SGPR / repeated operand?
// Every wave runs ITERS x 48 instructions of one kind on 8 independent accumulators and stamps the shader clock
// (s_memtime) around the loop; occupancy is set by the static LDS allocation (one round of resident waves).
// Output: shader cycles per wave instruction and SIMD = (wave's clock delta) / (instructions x waves per SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int kBody = 48;

template <int KIND, int LDS_BYTES>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters, double sval) {
  __shared__ double pad[LDS_BYTES / 8];
  const int l = threadIdx.x;
  pad[l] = (double)l;
  __syncthreads();
  double acc[8], r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = 1.0 + i + l * 1e-3;
    r[i] = 1e-9 * (i + 1) + 1e-12 * l;
  }
  double x = 1.0 + 1e-9 * l, y = 1e-12 * (l + 1);
  const double s = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(long long, sval) >> 32)) * 4294967296ll);
  int it2 = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < kBody; ++q) {
      double& a = acc[q & 7];
      if constexpr (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));                  // 3 distinct, x y fixed
      if constexpr (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(r[q % 7]), "v"(r[(q + 3) % 7])); // Gram-like
      if constexpr (KIND == 2) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a) : "v"(x));                          // 2 distinct
      if constexpr (KIND == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "s"(s));                  // SGPR pair
      if constexpr (KIND == 4) asm volatile("v_fma_f64 %0, %1, 2.0, %0" : "+v"(a) : "v"(x));                         // inline constant
      if constexpr (KIND == 5) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a) : "v"(x), "v"(y));                      // 2 reads
      if constexpr (KIND == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(x));                              // 2 reads
      if constexpr (KIND == 7) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a));                                   // 1 register
      if constexpr (KIND == 8) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));                 // VOP2 accumulate
      if constexpr (KIND == 9) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a) : "v"(x), "v"(y), "v"(r[q & 7]));   // 3 reads + separate dst
      if constexpr (KIND == 11) {                                                                                      // opcodes alternate: fma, add
        if (q & 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(x));
        else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
      }
      if constexpr (KIND == 12) {                                                                                      // fma, mul
        if (q & 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a) : "v"(x), "v"(y));
        else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
      }
      if constexpr (KIND == 13) {                                                                                      // the lane loop's mix: 4 fmac/fma : 2 add : 1 mul
        const int m = q % 7;
        if (m == 2 || m == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(x));
        else if (m == 6) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a) : "v"(x), "v"(y));
        else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
      }
      if constexpr (KIND == 14) {                                                                                      // the same mix with every add / mul written as an FMA
        const int m = q % 7;
        if (m == 2 || m == 5) asm volatile("v_fma_f64 %0, %1, 1.0, %0" : "+v"(a) : "v"(x));
        else if (m == 6) asm volatile("v_fma_f64 %0, %1, %2, 0" : "=v"(a) : "v"(x), "v"(y));
        else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
      }
      if constexpr (KIND == 15) {                                                                                      // FP64 stream with a scalar instruction after every 8th
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
        if ((q & 7) == 7) asm volatile("s_add_i32 %0, %0, 1" : "+s"(it2));
      }
      if constexpr (KIND == 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(x), "v"(y));   // DPP broadcast operand
      if constexpr (KIND == 17) {                                                                                      // the parity factorisation's mix: 3 DPP : 1 plain
        if ((q & 3) == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
        else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(x), "v"(y));
      }
      if constexpr (KIND == 10) {                                                                                      // 32-bit FMA for scale
        float& f = reinterpret_cast<float*>(&a)[0];
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(reinterpret_cast<float*>(&x)[0]), "v"(reinterpret_cast<float*>(&y)[0]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += acc[i];
  out[blockIdx.x * 64 + l] = sum + pad[(l * 7 + it2) & 63];
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int LDS_BYTES>
static void run(const char* what, int wps, int cus, int iters) {
  const int blocks = cus * 4 * wps;
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<double> per;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, LDS_BYTES>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) best = std::min(best, ms);
  }
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double instr = (double)iters * kBody;
  const double med = (double)h[blocks / 2] / (instr * wps), mx = (double)h[blocks - 1] / (instr * wps);
  // s_memtime = shader clock (scripts/micro/lane_phases.hip calibrates it against s_memrealtime)
  printf("%-52s %d waves/SIMD: %6.3f ns per instruction and SIMD (event), wave clock delta median/max %.4f/%.4f shader cycles (s_memtime)\n",
         what, wps, best * 1e6 / (instr * wps), med, mx);
  hipFree(out);
  hipFree(cyc);
}

template <int KIND>
static void sweep(const char* what, int cus, int iters) {
  run<KIND, 40 * 1024>(what, 1, cus, iters);
  run<KIND, 20 * 1024>(what, 2, cus, iters);
  run<KIND, 13 * 1024>(what, 3, cus, iters);
  run<KIND, 10 * 1024>(what, 4, cus, iters);
  run<KIND, 5 * 1024>(what, 8, cus, iters);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, runtime clock %.2f GHz; one round of resident waves, %d-instruction body\n", prop.name, cus,
         prop.clockRate * 1e-6, kBody);
  const int iters = 3000;
  for (int w = 0; w < 3; ++w) run<7, 20 * 1024>("(warm-up)", 2, cus, iters);
  sweep<0>("v_fma_f64 d += x*y   (3 VGPR pairs, x y fixed)", cus, iters);
  sweep<1>("v_fma_f64 d += r_i*r_j (3 VGPR pairs, Gram-like)", cus, iters);
  sweep<9>("v_fma_f64 d = x*y + c (3 VGPR pairs + separate dst)", cus, iters);
  sweep<8>("v_fmac_f64 d += x*y  (VOP2)", cus, iters);
  sweep<2>("v_fma_f64 d += x*x   (2 VGPR pairs)", cus, iters);
  sweep<3>("v_fma_f64 d += x*s   (SGPR pair)", cus, iters);
  sweep<4>("v_fma_f64 d += x*2.0 (inline constant)", cus, iters);
  sweep<7>("v_fma_f64 d = d*d+d  (1 VGPR pair)", cus, iters);
  sweep<5>("v_mul_f64 d = x*y", cus, iters);
  sweep<6>("v_add_f64 d += x", cus, iters);
  sweep<10>("v_fma_f32 d += x*y", cus, iters);
  sweep<11>("alternating v_fma_f64 / v_add_f64", cus, iters);
  sweep<12>("alternating v_fma_f64 / v_mul_f64", cus, iters);
  sweep<13>("lane-loop mix: 4 v_fmac : 2 v_add : 1 v_mul", cus, iters);
  sweep<14>("the same mix, add and mul written as v_fma_f64", cus, iters);
  sweep<16>("v_fmac_f64_dpp row_newbcast (DPP operand)", cus, iters);
  sweep<17>("3 v_fmac_f64_dpp : 1 v_fmac_f64", cus, iters);
  return 0;
}
