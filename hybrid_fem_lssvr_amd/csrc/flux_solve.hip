// Dirichlet solve of the assembled P1 system by two nested prefix sums -- `enforce` +
// `solve` (Dual.py:129-130) specialised to what Dual.py:117-128 assembles.
//
// The P1 stiffness matrix of -(a u')' on a line is A = D^T K D (D = nodal difference,
// K = diag(k_e), k_e = abar_e / h_e).  With the element flux q_e = k_e (u_{e+1} - u_e), row i
// reads q_{i-1} - q_i = load_i, hence
//     q_e = q_0 - S_e,                 S_e = sum_{j=1..e} load_j,
//     u_m = u_0 + q_0 R_m - T_m,       R_m = sum_{e<m} 1/k_e,  T_m = sum_{e<m} S_e / k_e,
//     q_0 = (u_ne - u_0 + T_ne) / R_ne.
// No elimination, so no amplification by the condition number (~ne^2) of A: the forward
// error is that of three length-ne sums.  The three sums are ONE scan of the associative,
// non-commutative operator on (alpha, rho, gamma):
//     element e:  (l_e, 1/k_e, l_e/k_e),   l_0 = 0, l_e = load_e
//     (a1,r1,g1) then (a2,r2,g2)  ->  (a1+a2, r1+r2, g1+g2 + r2*a1)
// whose running value after element e is (S_e, R_{e+1}, T_{e+1}).
// Three launches: block aggregates, scan of the aggregates (one workgroup), block-local scan
// + output.  Each thread owns kItems consecutive elements.
// Sharded meshes: the operator being associative, a rank's shard contributes one aggregate
// (flux_aggregate); the exclusive combination of the lower ranks' aggregates and the grand
// total (24 bytes per rank through one all-gather) are all flux_finish needs -- the global
// solve shards with a single tiny collective.
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

namespace {

constexpr int kItems = 4;                   // measured at 1e7 elements: 4 -> 178 us, 2 -> 255, 8 -> 246-269, 16 -> 705
constexpr int kTile = kBlock * kItems;      // elements per workgroup

struct Agg {
  double a, r, g;
};

__device__ __forceinline__ Agg combine(const Agg& x, const Agg& y) {   // x first, then y
  Agg o;
  o.a = x.a + y.a;
  o.r = x.r + y.r;
  o.g = (x.g + y.g) + y.r * x.a;
  return o;
}

// first_global: local element 0 is the mesh's first element (its left node is the Dirichlet
// node, l_0 = 0); on a later shard of a sharded mesh the node's load counts.
__device__ __forceinline__ Agg elem_agg(const double* __restrict__ kloc,
                                        const double* __restrict__ load, int64_t e,
                                        bool first_global) {
  const double l = (e == 0 && first_global) ? 0.0 : load[e];
  const double rk = 1.0 / kloc[e];
  Agg o;
  o.a = l;
  o.r = rk;
  o.g = l * rk;
  return o;
}

// exclusive scan of the 256 per-thread aggregates of a workgroup (in order), returns the
// workgroup total in `total`
__device__ __forceinline__ Agg block_exclusive(const Agg& mine, Agg* sh, Agg& total) {
  const int tid = threadIdx.x;
  sh[tid] = mine;
  __syncthreads();
  // Hillis-Steele inclusive scan with the non-commutative operator (left operand = earlier)
  for (int off = 1; off < kBlock; off <<= 1) {
    Agg v = sh[tid];
    if (tid >= off) v = combine(sh[tid - off], v);
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
  }
  total = sh[kBlock - 1];
  Agg ex;
  ex.a = ex.r = ex.g = 0.0;
  if (tid > 0) ex = sh[tid - 1];
  __syncthreads();
  return ex;
}

__global__ __launch_bounds__(kBlock) void flux_block_agg_kernel(const double* __restrict__ kloc,
                                                                 const double* __restrict__ load,
                                                                 int64_t ne, bool first_global,
                                                                 Agg* __restrict__ bagg) {
  __shared__ Agg sh[kBlock];
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  Agg acc;
  acc.a = acc.r = acc.g = 0.0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t e = base + i;
    if (e < ne) acc = combine(acc, elem_agg(kloc, load, e, first_global));
  }
  Agg total;
  block_exclusive(acc, sh, total);
  if (threadIdx.x == 0) bagg[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the nb block aggregates in place; the grand total goes to
// bagg[nb]
__global__ __launch_bounds__(kBlock) void flux_scan_agg_kernel(Agg* __restrict__ bagg, int64_t nb,
                                                                double* __restrict__ agg_out) {
  __shared__ Agg sh[kBlock];
  Agg carry;
  carry.a = carry.r = carry.g = 0.0;
  for (int64_t base = 0; base < nb; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    Agg mine;
    mine.a = mine.r = mine.g = 0.0;
    if (i < nb) mine = bagg[i];
    Agg total;
    const Agg ex = block_exclusive(mine, sh, total);
    if (i < nb) bagg[i] = combine(carry, ex);
    carry = combine(carry, total);
  }
  if (threadIdx.x == 0) {
    bagg[nb] = carry;
    if (agg_out) {
      agg_out[0] = carry.a;
      agg_out[1] = carry.r;
      agg_out[2] = carry.g;
    }
  }
}

__global__ __launch_bounds__(kBlock) void flux_output_kernel(const double* __restrict__ kloc,
                                                              const double* __restrict__ load,
                                                              int64_t ne, bool first_global,
                                                              bool last_global,
                                                              const Agg* __restrict__ bagg,
                                                              int64_t nb,
                                                              const double* __restrict__ prefix3,
                                                              const double* __restrict__ grand3,
                                                              double u0, double u1,
                                                              double* __restrict__ u) {
  __shared__ Agg sh[kBlock];
  // running value before this shard and over the whole mesh (single shard: zero / own total)
  Agg pre;
  pre.a = pre.r = pre.g = 0.0;
  if (prefix3) {
    pre.a = prefix3[0];
    pre.r = prefix3[1];
    pre.g = prefix3[2];
  }
  Agg grand = bagg[nb];
  if (grand3) {
    grand.a = grand3[0];
    grand.r = grand3[1];
    grand.g = grand3[2];
  }
  const double q0 = ((u1 - u0) + grand.g) / grand.r;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  Agg loc[kItems];
  Agg acc;
  acc.a = acc.r = acc.g = 0.0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t e = base + i;
    if (e < ne) acc = combine(acc, elem_agg(kloc, load, e, first_global));
    loc[i] = acc;                      // inclusive within the thread
  }
  Agg total;
  const Agg ex = combine(pre, combine(bagg[blockIdx.x], block_exclusive(acc, sh, total)));
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t e = base + i;
    if (e < ne) {
      const Agg v = combine(ex, loc[i]);                 // (S_e, R_{e+1}, T_{e+1})
      u[e + 1] = (e + 1 == ne && last_global) ? u1 : (u0 + q0 * v.r) - v.g;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) u[0] = first_global ? u0 : (u0 + q0 * pre.r) - pre.g;
}

}  // namespace

int64_t flux_work_bytes(int64_t ne) {
  const int64_t nb = (ne + kTile - 1) / kTile;
  return (nb + 2) * (int64_t)sizeof(Agg) + 64;
}

hipError_t flux_aggregate(const double* kloc, const double* load, int64_t ne, bool first_global,
                          void* work, double* agg3, hipStream_t s) {
  const int64_t nb = (ne + kTile - 1) / kTile;
  Agg* bagg = reinterpret_cast<Agg*>(work);
  hipLaunchKernelGGL(flux_block_agg_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, kloc, load, ne,
                     first_global, bagg);
  hipLaunchKernelGGL(flux_scan_agg_kernel, dim3(1), dim3(kBlock), 0, s, bagg, nb, agg3);
  return hipGetLastError();
}

hipError_t flux_finish(const double* kloc, const double* load, int64_t ne, bool first_global,
                       bool last_global, const void* work, const double* prefix3,
                       const double* grand3, double u0, double u1, double* u, hipStream_t s) {
  const int64_t nb = (ne + kTile - 1) / kTile;
  const Agg* bagg = reinterpret_cast<const Agg*>(work);
  hipLaunchKernelGGL(flux_output_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, kloc, load, ne,
                     first_global, last_global, bagg, nb, prefix3, grand3, u0, u1, u);
  return hipGetLastError();
}

hipError_t flux_dirichlet_solve(const double* kloc, const double* load, int64_t ne, double u0,
                                double u1, double* u, void* work, hipStream_t s) {
  hipError_t e = flux_aggregate(kloc, load, ne, true, work, nullptr, s);
  if (e != hipSuccess) return e;
  return flux_finish(kloc, load, ne, true, true, work, nullptr, nullptr, u0, u1, u, s);
}

}  // namespace lssvr
