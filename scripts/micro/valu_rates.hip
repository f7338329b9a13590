// Microbenchmark (gfx950): issue cost per wave instruction of the operations the dual solver's LU is made
// of -- v_fma_f64 (VGPR and SGPR-pair operands), v_readlane_b32 with an SGPR lane select, the
// readlane/readlane/FMA column update, 32-bit VALU, and LDS broadcast reads -- at ONE and at TWO waves per
// SIMD (occupancy set by the static LDS allocation, as in enhance_dual_w64_kernel: 19 KB per wave).
// Output: ns per wave instruction per SIMD, and cycles at the clock the run reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

constexpr int kBody = 48;      // instructions (or column updates) per loop iteration

template <int KIND, int LDS_BYTES>
__global__ __launch_bounds__(64) void rate_kernel(double* out, int iters, int lanesel) {
  __shared__ double pad[LDS_BYTES / 8];
  const int l = threadIdx.x;
  pad[l] = (double)l;
  pad[l + 64] = 1.0 + l;
  __syncthreads();
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 1.0 + i + l * 1e-3;
  double x = 1.0 + 1e-9 * l, yv = 1e-12;
  int P = __builtin_amdgcn_readfirstlane(lanesel);
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {              // v_fma_f64, VGPR operands, 8 independent chains
#pragma unroll
      for (int k = 0; k < kBody; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[k & 7]) : "v"(x), "v"(yv));
    } else if constexpr (KIND == 1) {       // v_fma_f64 with an SGPR pair operand
      double s = __builtin_bit_cast(double, ((unsigned long long)__builtin_amdgcn_readfirstlane(0x3ff00000) << 32));
#pragma unroll
      for (int k = 0; k < kBody; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[k & 7]) : "s"(s), "v"(yv));
    } else if constexpr (KIND == 2) {       // v_readlane_b32, SGPR lane select, independent destinations
      int s0, s1, s2, s3;
#pragma unroll
      for (int k = 0; k < kBody; k += 4) {
        asm volatile("v_readlane_b32 %0, %4, %5\n v_readlane_b32 %1, %4, %5\n v_readlane_b32 %2, %4, %5\n v_readlane_b32 %3, %4, %5"
                     : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(((int*)&x)[0]), "s"(P));
      }
      asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3));
    } else if constexpr (KIND == 7) {       // v_readlane_b32, 16 distinct destination SGPRs per group
      int s[16];
#pragma unroll
      for (int k = 0; k < kBody; k += 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s[q]) : "v"(((int*)&acc[q & 7])[q & 1]), "s"(P));
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" :: "s"(s[q]));
      }
    } else if constexpr (KIND == 8) {       // v_readfirstlane_b32
      int s[16];
#pragma unroll
      for (int k = 0; k < kBody; k += 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s[q]) : "v"(((int*)&acc[q & 7])[q & 1]));
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" :: "s"(s[q]));
      }
    } else if constexpr (KIND == 3) {       // the LU column update: 2 x v_readlane_b32 + v_fma_f64 (SGPR pair), batches of 6
#pragma unroll
      for (int k = 0; k < kBody; k += 6) {
        double u[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int lo = __builtin_amdgcn_readlane(((int*)&acc[(k + q) & 7])[0], P);
          const int hi = __builtin_amdgcn_readlane(((int*)&acc[(k + q) & 7])[1], P);
          u[q] = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[(k + q) & 7] = __builtin_fma(yv, u[q], acc[(k + q) & 7]);
      }
    } else if constexpr (KIND == 4) {       // v_mov_b32 (32-bit VALU)
      int t = l;
#pragma unroll
      for (int k = 0; k < kBody; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(l));
      ((int*)&acc[0])[0] ^= t & 1;
    } else if constexpr (KIND == 5) {       // LDS broadcast ds_read_b128 (uniform address) + 2 FMAs on the result
      const double* q = pad + (P & 7) * 2;
#pragma unroll
      for (int k = 0; k < kBody; k += 2) {
        double a, b;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(__attribute__((ext_vector_type(2))) double*)&a) : "v"((unsigned)(unsigned long long)q), "i"((k & 30) * 8));
        (void)b;
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 1.2345e300) out[0] = s + pad[(l * 7) & 63];
}

template <int KIND, int LDS_BYTES>
static void run(const char* name, double instr_per_body, double clock_ghz) {
  double* out; hipMalloc(&out, 64);
  const int iters = 2000;
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int waves_per_cu = (160 * 1024) / LDS_BYTES >= 8 ? 8 : 4;
  const int grid = ncu * waves_per_cu;             // exactly one round of resident waves
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<KIND, LDS_BYTES>), dim3(grid), dim3(64), 0, 0, out, iters, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double per_simd = (double)waves_per_cu / 4.0 * iters * instr_per_body;      // wave instructions per SIMD
  const double ns = (t[3] * 1e6 - 5000.0) / per_simd;
  printf("%-64s %d waves/SIMD: %6.2f ns per wave instruction per SIMD = %5.2f cycles at %.2f GHz\n", name, waves_per_cu / 4, ns, ns * clock_ghz, clock_ghz);
  hipFree(out);
}

int main() {
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double ghz = khz * 1e-6;
  run<0, 20480>("v_fma_f64 (VGPR operands, 8 chains)", kBody, ghz);
  run<0, 40960>("v_fma_f64 (VGPR operands, 8 chains)", kBody, ghz);
  run<1, 20480>("v_fma_f64 (SGPR-pair operand)", kBody, ghz);
  run<2, 20480>("v_readlane_b32 (SGPR lane select)", kBody, ghz);
  run<2, 40960>("v_readlane_b32 (SGPR lane select)", kBody, ghz);
  run<3, 20480>("column update: 2 v_readlane_b32 + v_fma_f64 (per instruction)", kBody * 3, ghz);
  run<3, 40960>("column update: 2 v_readlane_b32 + v_fma_f64 (per instruction)", kBody * 3, ghz);
  run<7, 20480>("v_readlane_b32 (16 distinct destination SGPRs)", kBody, ghz);
  run<8, 20480>("v_readfirstlane_b32 (16 distinct destination SGPRs)", kBody, ghz);
  run<4, 20480>("v_add_u32 (dependent chain)", kBody, ghz);
  run<5, 20480>("ds_read_b128 broadcast (uniform address), per read", kBody / 2, ghz);
  run<5, 40960>("ds_read_b128 broadcast (uniform address), per read", kBody / 2, ghz);
  return 0;
}
