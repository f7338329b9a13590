// v_rcp_f64 seed accuracy on gfx950 and the error after one / two Newton steps (relative, in ulps of 2^-53)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* o0, double* o1, double* o2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  o0[i] = y;
  y = y * fma(-v, y, 2.0);
  o1[i] = y;
  y = y * fma(-v, y, 2.0);
  o2[i] = y;
}
__global__ void k2(const double* x, double* o, int n) {   // one step in the residual form  y + y (1 - x y)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  o[i] = fma(y, fma(-v, y, 1.0), y);
}
int main() {
  const int n = 1 << 22;
  double *hx = new double[n], *h = new double[n];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); if (s & 1) hx[i] = -hx[i]; }
  double *x, *o0, *o1, *o2, *o3;
  (void)hipMalloc(&x, n * 8); (void)hipMalloc(&o0, n * 8); (void)hipMalloc(&o1, n * 8); (void)hipMalloc(&o2, n * 8); (void)hipMalloc(&o3, n * 8);
  (void)hipMemcpy(x, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, o0, o1, o2, n);
  k2<<<n / 256, 256>>>(x, o3, n);
  double* outs[4] = {o0, o1, o2, o3};
  const char* nm[4] = {"seed", "1 step y(2-xy)", "2 steps", "1 step y+y(1-xy)"};
  for (int t = 0; t < 4; ++t) {
    (void)hipMemcpy(h, outs[t], n * 8, hipMemcpyDeviceToHost);
    long double worst = 0;
    for (int i = 0; i < n; ++i) { long double e = fabsl((long double)h[i] * (long double)hx[i] - 1.0L); if (e > worst) worst = e; }
    printf("%-18s max |x y - 1| = %.3Le  (%.2Lf x 2^-53)\n", nm[t], worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
