// Measurement build of the HEADLINE lane kernel (BASELINE config 2: degree 8 / 16 points, in-kernel rhs):
// the SAME body as the shipped enhance_small_kernel<9, RHS_SIN> (hybrid_fem_lssvr_amd/csrc/enhance_small_cheb.hpp),
// instantiated with a probe that stamps the 100 MHz constant clock (s_memrealtime) per wave at the body's phase
// boundaries.  K launches are captured in a hipGraph and replayed, as in bench.py's timed region; every launch
// stamps into its own slot, so the table shows, for a launch in the MIDDLE of the sequence:
//   ramp   -- when waves start, relative to the first wave of the launch (workgroup dispatch)
//   phases -- per-wave time from entry to: inputs loaded / moments done / system built / solved / stores issued
//   tail   -- last wave's last stamp -> first wave of the NEXT launch (store drain + kernel boundary)
// and the launch-to-launch period, which is what bench.py's ms_per_step measures.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I hybrid_fem_lssvr_amd/csrc
//        scripts/micro/lane_phases.hip -o scripts/micro/lane_phases      (usage: lane_phases [ne])
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "enhance_small_cheb.hpp"

using namespace lssvr;

constexpr int kSlots = kPhCount + 3;     // + shader clock at entry, shader clock at the end, HW_ID | XCC_ID << 32
struct StampProbe {
  unsigned long long* out;      // [wave][kSlots] of this launch
  __device__ __forceinline__ void mark(int k) const {
    if (k == kPhLoaded) __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0): the element's inputs have arrived
    if (k == kPhArgs) __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the kernel arguments have arrived
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
      out[(size_t)wave * kSlots + k] = t;
      if (k == kPhEntry || k == kPhStored) out[(size_t)wave * kSlots + kPhCount + (k == kPhStored)] = __builtin_amdgcn_s_memtime();
      if (k == kPhEntry) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[(size_t)wave * kSlots + kPhCount + 2] = hw | ((unsigned long long)xcc << 32);
      }
    }
  }
};

// LSSVR_PRELOAD (build with -mllvm -amdgpu-kernarg-preload-count=16): the pointers and counts the first loads need as
// LEADING SCALAR parameters (64 bytes = 16 SGPRs), which the command processor can place in SGPRs before the wave
// starts -- does the "kernel arguments arrived" phase go away?
#ifdef LSSVR_PRELOAD
#define LEAD const double* x_, const double* u_, long long ne_, const long long* ids_, const double* gam_, double* W_, int* st_, long long nem_,
#define FIX(p) p.x = x_; p.u = u_; p.ne = ne_; p.elem_ids = (const int64_t*)ids_; p.gamma_values = gam_; p.W = W_; p.status = st_; p.ne_mesh = nem_;
#define LEADARGS(a) a.x, a.u, (long long)a.ne, (const long long*)a.elem_ids, a.gamma_values, a.W, a.status, (long long)a.ne_mesh,
#else
#define LEAD
#define FIX(p)
#define LEADARGS(a)
#endif

__global__ __launch_bounds__(kBlock) void probed_kernel(LEAD EnhanceArgs p, unsigned long long* stamps) {
  FIX(p)
  __shared__ double tile[(kBlock / 64) * kChebTilePerWave<9, LSSVR_RHS_SIN>];
  enhance_small_body_cheb<9, LSSVR_RHS_SIN, false, StampProbe>(p, blockIdx.x, tile, StampProbe{stamps});
}

__global__ __launch_bounds__(kBlock) void plain_kernel(LEAD EnhanceArgs p) {
  FIX(p)
  __shared__ double tile[(kBlock / 64) * kChebTilePerWave<9, LSSVR_RHS_SIN>];
  enhance_small_body_cheb<9, LSSVR_RHS_SIN>(p, blockIdx.x, tile);
}

// ONE-WAVE WORKGROUPS (the round-3 review's first suggestion): the same body, 64 threads per workgroup.  The body
// indexes elements as block * kBlock + tid, so workgroup b works on a view of the arrays shifted by 64 (b & 3)
// elements with block index b >> 2: the same elements, the same arithmetic, 1 563 workgroups of one wave.
__global__ __launch_bounds__(64) void plain64_kernel(EnhanceArgs p) {
  __shared__ double tile[kChebTilePerWave<9, LSSVR_RHS_SIN>];
  const int64_t off = 64 * (int64_t)(blockIdx.x & 3);
  if (off + (int64_t)(blockIdx.x >> 2) * kBlock >= p.ne) return;
  p.x += off; p.u += off; p.W += off * 9; p.status += off;
  p.ne -= off; p.ne_mesh -= off; p.elem_offset += off;
  enhance_small_body_cheb<9, LSSVR_RHS_SIN>(p, blockIdx.x >> 2, tile);
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double pct(std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; }

int main(int argc, char** argv) {
  const long ne = argc > 1 ? atol(argv[1]) : 100008;
  const int K = 20;
  const int M = 9, n = 16;
  std::vector<double> x(ne + 1), u(ne + 1);
  const double half = ne / 24.0;
  for (long i = 0; i <= ne; ++i) { x[i] = i * ((2.0 * half) / ne) - half; u[i] = sin(M_PI * x[i]); }
  x[ne] = half; u[0] = u[ne] = 0.0;
  double *dx, *du, *dW; int* dst; unsigned long long* dstamp;
  const long waves = (ne + 63) / 64;
  hipMalloc(&dx, (ne + 1) * 8); hipMalloc(&du, (ne + 1) * 8); hipMalloc(&dW, ne * M * 8); hipMalloc(&dst, ne * 4);
  hipMalloc(&dstamp, (size_t)K * waves * kSlots * 8);
  hipMemcpy(dx, x.data(), (ne + 1) * 8, hipMemcpyHostToDevice);
  hipMemcpy(du, u.data(), (ne + 1) * 8, hipMemcpyHostToDevice);
  EnhanceArgs a{};
  a.x = dx; a.u = du; a.ne = ne; a.elem_offset = 0; a.ne_global = ne; a.gxmin = -half; a.gxmax = half;
  a.gamma = 1e4; a.inv_gamma = 1e-4; a.M = M; a.n = n; a.rhs_id = LSSVR_RHS_SIN; a.rhs_amp = M_PI * M_PI; a.rhs_omega = M_PI;
  a.tab_es = n; a.tab_ps = 1; a.ne_mesh = ne; a.W = dW; a.status = dst; a.trig = make_trig_tables();
  const unsigned blocks = (unsigned)((ne + kBlock - 1) / kBlock);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto graph_of = [&](bool probed) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < K; ++i) {
      if (probed) hipLaunchKernelGGL(probed_kernel, dim3(blocks), dim3(kBlock), 0, s, LEADARGS(a) a, dstamp + (size_t)i * waves * kSlots);
      else hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(kBlock), 0, s, LEADARGS(a) a);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    return ge;
  };
  hipGraphExec_t gp = graph_of(false), gs = graph_of(true);
  hipGraphExec_t g64;
  {
    hipGraph_t g;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(plain64_kernel, dim3(4 * blocks), dim3(64), 0, s, a);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&g64, g, nullptr, nullptr, 0);
  }
  for (int i = 0; i < 300; ++i) hipGraphLaunch(gp, s);        // steady state (~50 ms)
  hipStreamSynchronize(s);
  auto time_graph = [&](hipGraphExec_t g) {
    std::vector<double> t;
    for (int r = 0; r < 21; ++r) {
      hipEventRecord(e0, s); hipGraphLaunch(g, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3 / K);
    }
    return med(t);
  };
  printf("ne = %ld, %u workgroups of %d threads, K = %d launches per replay\n", ne, blocks, kBlock, K);
  printf("graph-replayed launch period, shipped body      : %.2f us\n", time_graph(gp));
  printf("graph-replayed launch period, body with stamps  : %.2f us\n", time_graph(gs));
  {
    std::vector<double> w256(ne * M), w64(ne * M);
    hipGraphLaunch(gp, s); hipStreamSynchronize(s);
    hipMemcpy(w256.data(), dW, ne * M * 8, hipMemcpyDeviceToHost);
    hipMemset(dW, 0, ne * M * 8);
    for (int i = 0; i < 50; ++i) hipGraphLaunch(g64, s);
    hipStreamSynchronize(s);
    const double t64 = time_graph(g64);
    hipMemcpy(w64.data(), dW, ne * M * 8, hipMemcpyDeviceToHost);
    printf("graph-replayed launch period, ONE-WAVE workgroups (%u x 64 threads): %.2f us   (256-thread form again: %.2f us; results %s)\n",
           4 * blocks, t64, time_graph(gp), w256 == w64 ? "bit-equal" : "DIFFER");
  }
  hipGraphLaunch(gs, s); hipStreamSynchronize(s);
  std::vector<unsigned long long> st((size_t)K * waves * kSlots);
  hipMemcpy(st.data(), dstamp, st.size() * 8, hipMemcpyDeviceToHost);
  auto at = [&](int l, long w, int k) { return (double)st[((size_t)l * waves + w) * kSlots + k] * 0.01; };   // us
  auto raw = [&](int l, long w, int k) { return st[((size_t)l * waves + w) * kSlots + k]; };
  // per launch: first start, last end
  std::vector<double> first(K), last(K);
  for (int l = 0; l < K; ++l) {
    double f = 1e300, e = 0;
    for (long w = 0; w < waves; ++w) { f = std::min(f, at(l, w, kPhEntry)); e = std::max(e, at(l, w, kPhStored)); }
    first[l] = f; last[l] = e;
  }
  std::vector<double> period, span, gap;
  for (int l = 5; l < K - 1; ++l) { period.push_back(first[l + 1] - first[l]); span.push_back(last[l] - first[l]); gap.push_back(first[l + 1] - last[l]); }
  printf("launch period from the stamps (first wave -> first wave of the next launch): median %.2f us\n", med(period));
  printf("  span  first wave entry -> last wave's stores issued : median %.2f us\n", med(span));
  printf("  gap   last stores issued -> next launch's first wave: median %.2f us   (store drain + kernel boundary)\n", med(gap));
  const int l = K / 2;
  std::vector<double> start, dur[kPhCount], endt;
  for (long w = 0; w < waves; ++w) {
    start.push_back(at(l, w, kPhEntry) - first[l]);
    for (int k = 1; k < kPhCount; ++k) dur[k].push_back(at(l, w, k) - at(l, w, k - 1));
    endt.push_back(at(l, w, kPhStored) - first[l]);
  }
  printf("launch %d of the replay, %ld waves (times in us):\n", l, waves);
  printf("  wave start after the launch's first wave : median %.2f  p90 %.2f  max %.2f\n", med(start), pct(start, 0.9), pct(start, 1.0));
  const char* names[kPhCount] = {"", "entry -> kernel arguments arrived (s_load)", "-> inputs loaded (4 global loads)", "element set-up (map, boundary data, rhs seeds)", "collocation loop (16 points: moments, rhs)",
                                 "system built from the moments", "LDL^T + substitutions + v = Y z", "LDS transposition + stores issued"};
  for (int k = 1; k < kPhCount; ++k) printf("  %-52s: median %.2f  p10 %.2f  p90 %.2f\n", names[k], med(dur[k]), pct(dur[k], 0.1), pct(dur[k], 0.9));
  printf("  wave end after the launch's first wave   : median %.2f  p90 %.2f  max %.2f\n", med(endt), pct(endt, 0.9), pct(endt, 1.0));
  // by start order: do late-starting waves run longer (two waves on a SIMD)?
  std::vector<std::pair<double, double>> sd;
  for (long w = 0; w < waves; ++w) sd.push_back({start[w], endt[w] - start[w]});
  std::sort(sd.begin(), sd.end());
  for (int q = 0; q < 4; ++q) {
    std::vector<double> d;
    for (size_t i = q * sd.size() / 4; i < (q + 1) * sd.size() / 4; ++i) d.push_back(sd[i].second);
    printf("  waves by start order, quarter %d: start %.2f .. %.2f us, lifetime median %.2f us\n", q + 1, sd[q * sd.size() / 4].first,
           sd[(q + 1) * sd.size() / 4 - 1].first, med(d));
  }
  // shader clock during the launch: s_memtime ticks per s_memrealtime tick (100 MHz), per wave
  std::vector<double> ghz;
  for (long w = 0; w < waves; ++w) {
    const double dt = at(l, w, kPhStored) - at(l, w, kPhEntry);
    if (dt > 0) ghz.push_back((double)(raw(l, w, kPhCount + 1) - raw(l, w, kPhCount)) / dt * 1e-3);
  }
  printf("  shader clock over the waves' lifetimes (s_memtime / s_memrealtime): median %.3f GHz  p10 %.3f  p90 %.3f\n", med(ghz), pct(ghz, 0.1), pct(ghz, 0.9));
  // placement: waves per SIMD (xcc, se, sh, cu, simd from HW_ID), and lifetime by the SIMD's wave count
  std::vector<long> key(waves);
  std::vector<int> cnt(1 << 20, 0);
  for (long w = 0; w < waves; ++w) {
    const unsigned long long v = raw(l, w, kPhCount + 2);
    const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 0xf;
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    key[w] = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd);
    cnt[key[w]]++;
  }
  int hist[8] = {0};
  long used = 0;
  for (int c : cnt) if (c > 0) { hist[std::min(c, 7)]++; ++used; }
  printf("  SIMDs in use: %ld;  SIMDs holding 1 / 2 / 3 / 4+ waves of this launch: %d / %d / %d / %d\n", used, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
  for (int c = 1; c <= 3; ++c) {
    std::vector<double> d, lp;
    for (long w = 0; w < waves; ++w) if (cnt[key[w]] == c) { d.push_back(endt[w] - start[w]); lp.push_back(dur[kPhMoments][w]); }
    if (!d.empty()) printf("  waves on a SIMD with %d wave(s): %zu, lifetime median %.2f us, collocation loop median %.2f us\n", c, d.size(), med(d), med(lp));
    if (!d.empty()) {
      printf("    phases (us):");
      for (int k = 1; k < kPhCount; ++k) {
        std::vector<double> ph;
        for (long w = 0; w < waves; ++w) if (cnt[key[w]] == c) ph.push_back(dur[k][w]);
        printf(" %.2f", med(ph));
      }
      printf("\n");
    }
  }
  return 0;
}
