// Device tridiagonal solve for the assembled P1 system with Dirichlet data on
// both end dofs -- `enforce(A, b, D=basis.get_dofs())` + `solve(A, b)`
// (Dual.py:129-130).  SURVEY.md section 8(f) "next-1".
//
// Algorithm: recursive substructuring (static condensation).  Every kLc-th (8th)
// unknown is a separator; one thread condenses the kLc-1 unknowns between two
// separators onto them (two O(1)-state sweeps, nothing stored), the separators
// form a tridiagonal system kLc times smaller, which is solved the same way
// until <= kBase unknowns remain (parallel cyclic reduction in LDS).  Going back up, each thread
// re-solves its chunk with the now-known separator values.  The P1 matrix is
// SPD, so no pivoting is needed at any level (Schur complements of SPD are SPD).
#include "lssvr_device.hpp"
#include "lssvr_kernels.hpp"

namespace lssvr {

// Chunk length per level: 8 at every size (round 2, one run: 1e7 unknowns 432 us against 675 us
// with chunks of 16 and 818 us with 32 -- a thread's 64-byte run of each array is one batch, a
// 128-byte line is shared by two neighbouring lanes of the same load instruction; 4 is as fast,
// with twice the levels).  Each row costs ONE division (1/den) and three multiplications in every
// sweep -- with three divisions per row the sweeps were bound by the FP64 division chain:
// 101 -> 55 us at 1e5 unknowns, 177 -> 91 us at 1e6.
constexpr int kBase = 512;
static inline int chunk_for(int64_t) { return 8; }

// row i: lo[i] x[i-1] + d[i] x[i] + up[i] x[i+1] = r[i] - [i==0] bl[0]*u0 - [i==m-1] br[0]*u1
struct TriSys {
  const double* lo;
  const double* d;
  const double* up;
  const double* r;
  const double* bl;
  const double* br;
  double u0, u1;
  int64_t m;
};

__device__ __forceinline__ double lo_at(const TriSys& s, int64_t i) { return i == 0 ? 0.0 : s.lo[i]; }
__device__ __forceinline__ double up_at(const TriSys& s, int64_t i) {
  return i == s.m - 1 ? 0.0 : s.up[i];
}
__device__ __forceinline__ double r_at(const TriSys& s, int64_t i) {
  double v = s.r[i];
  if (i == 0 && s.bl) v -= s.bl[0] * s.u0;
  if (i == s.m - 1 && s.br) v -= s.br[0] * s.u1;
  return v;
}

struct ChunkEnds {
  double yF, vF, wF, yL, vL, wL;
};

// x_interior = y + v * x_{left separator} + w * x_{right separator}; only the values
// at the first and last interior unknown are needed for the reduced system.
// Coefficients of kBatch consecutive rows are loaded back to back into registers before
// they are used: a thread walks its own 256-byte stretch of every array, so its 16 uses of
// a 128-byte line must be adjacent in time or the line is evicted from the 32 KB L1 by the
// other 63 lanes' lines in between (measured: 1.6x less time at 1e7 unknowns than one load
// per step).
constexpr int kBatch = 8;

struct RowBatch {
  double lo[kBatch], d[kBatch], up[kBatch], r[kBatch];
};

// rows i0 .. i0+kBatch-1 (clipped to [b, e)); entries outside are neutral (never used)
__device__ __forceinline__ void load_rows(const TriSys& s, int64_t i0, int64_t b, int64_t e,
                                          RowBatch& rb) {
#pragma unroll
  for (int t = 0; t < kBatch; ++t) {
    const int64_t i = i0 + t;
    const bool in = (i >= b) && (i < e);
    const int64_t ii = in ? i : b;
    rb.lo[t] = lo_at(s, ii);
    rb.d[t] = s.d[ii];
    rb.up[t] = up_at(s, ii);
    rb.r[t] = r_at(s, ii);
  }
}

template <int kLc>
__global__ __launch_bounds__(kBlock) void tri_condense_kernel(TriSys s, int64_t nc,
                                                               ChunkEnds* __restrict__ ends) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= nc) return;
  const int64_t b = j * kLc;
  const int64_t e = (b + kLc - 1 < s.m) ? b + kLc - 1 : s.m;   // interior = [b, e)
  ChunkEnds c;
  {  // downward sweep -> values at the last interior unknown
    double den = 1.0, cp = 0.0, y = 0.0, v = 0.0;
    for (int64_t i0 = b; i0 < e; i0 += kBatch) {
      RowBatch rb;
      load_rows(s, i0, b, e, rb);
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
        const int64_t i = i0 + t;
        if (i < e) {
          // one division per row (1/den), three multiplications
          if (i == b) {
            den = 1.0 / rb.d[t];
            y = rb.r[t] * den;
            v = -rb.lo[t] * den;
          } else {
            const double l = rb.lo[t];
            den = 1.0 / (rb.d[t] - l * cp);
            y = (rb.r[t] - l * y) * den;
            v = (-l * v) * den;
          }
          cp = rb.up[t] * den;
        }
      }
    }
    c.yL = y;
    c.vL = v;
    c.wL = -cp;               // rhs -up[e-1] e_last  ->  -up[e-1]/den_last  (den holds 1/den)
  }
  {  // upward sweep -> values at the first interior unknown
    double den = 1.0, bp = 0.0, y = 0.0, w = 0.0;
    for (int64_t i1 = e; i1 > b; i1 -= kBatch) {       // rows i1-kBatch .. i1-1, descending
      RowBatch rb;
      load_rows(s, i1 - kBatch, b, e, rb);
#pragma unroll
      for (int t = kBatch - 1; t >= 0; --t) {
        const int64_t i = i1 - kBatch + t;
        if (i >= b) {
          if (i == e - 1) {
            den = 1.0 / rb.d[t];
            y = rb.r[t] * den;
            w = -rb.up[t] * den;
          } else {
            const double u = rb.up[t];
            den = 1.0 / (rb.d[t] - u * bp);
            y = (rb.r[t] - u * y) * den;
            w = (-u * w) * den;
          }
          bp = rb.lo[t] * den;
        }
      }
    }
    c.yF = y;
    c.wF = w;
    c.vF = -bp;
  }
  ends[j] = c;
}

// separator j sits at p = j*kLc + kLc-1, between chunk j (left) and chunk j+1 (right)
template <int kLc>
__global__ __launch_bounds__(kBlock) void tri_reduce_kernel(TriSys s, int64_t ns, int64_t nc,
                                                             const ChunkEnds* __restrict__ ends,
                                                             double* __restrict__ LO,
                                                             double* __restrict__ D,
                                                             double* __restrict__ UP,
                                                             double* __restrict__ R) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= ns) return;
  const int64_t p = j * kLc + kLc - 1;
  const double l = s.lo[p];
  const double u = up_at(s, p);
  const ChunkEnds cl = ends[j];
  double dd = s.d[p] + l * cl.wL;
  double rr = r_at(s, p) - l * cl.yL;
  double uu = 0.0;
  if (j + 1 < nc) {
    const ChunkEnds cr = ends[j + 1];
    dd += u * cr.vF;
    rr -= u * cr.yF;
    uu = u * cr.wF;
  }
  LO[j] = l * cl.vL;
  D[j] = dd;
  UP[j] = uu;
  R[j] = rr;
}

// re-solve every chunk with its separator values known; x (length m) receives the
// whole level's solution.  cp: scratch of length m.
template <int kLc>
__global__ __launch_bounds__(kBlock) void tri_expand_kernel(TriSys s, int64_t ns, int64_t nc,
                                                             const double* __restrict__ X,
                                                             double* __restrict__ x,
                                                             double* __restrict__ cp) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= nc) return;
  const int64_t b = j * kLc;
  const int64_t e = (b + kLc - 1 < s.m) ? b + kLc - 1 : s.m;
  const double xl = j > 0 ? X[j - 1] : 0.0;
  const double xr = j < ns ? X[j] : 0.0;
  // forward elimination; the modified coefficients stay in registers (kLc-1 = 7 of each),
  // so the back substitution touches memory only to store the solution
  double cc[kLc], yy[kLc];
  double den = 1.0, c = 0.0, y = 0.0;
#pragma unroll
  for (int g = 0; g < kLc / kBatch; ++g) {
    const int64_t i0 = b + g * kBatch;
    if (i0 < e) {
      RowBatch rb;
      load_rows(s, i0, b, e, rb);
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
        const int64_t i = i0 + t;
        if (i < e) {
          double ri = rb.r[t];
          if (i == b) ri -= rb.lo[t] * xl;
          if (i == e - 1) ri -= rb.up[t] * xr;
          if (i == b) {
            den = 1.0 / rb.d[t];
            y = ri * den;
          } else {
            const double l = rb.lo[t];
            den = 1.0 / (rb.d[t] - l * c);
            y = (ri - l * y) * den;
          }
          c = rb.up[t] * den;
        }
        cc[g * kBatch + t] = c;
        yy[g * kBatch + t] = y;
      }
    }
  }
  (void)cp;
  double xn = 0.0;
#pragma unroll
  for (int k = kLc - 1; k >= 0; --k) {
    const int64_t i = b + k;
    if (i < e) {
      xn = (i == e - 1) ? yy[k] : yy[k] - cc[k] * xn;
      x[i] = xn;
    }
  }
  if (j < ns) x[j * kLc + kLc - 1] = xr;
}

// Base level (m <= kBase unknowns): parallel cyclic reduction in LDS, one workgroup of kBase
// threads, ceil(log2 m) steps -- a serial Thomas sweep by one thread would pay a global-memory
// round trip per unknown (~100 us for 100 unknowns; this takes a few us).  PCR needs no
// pivoting for the SPD / diagonally dominant systems that reach this level.
__global__ __launch_bounds__(kBase) void tri_base_kernel(TriSys s, double* __restrict__ x,
                                                          double* __restrict__ cp) {
  __shared__ double lo[kBase], d[kBase], up[kBase], r[kBase];
  (void)cp;
  const int i = threadIdx.x;
  const int m = (int)s.m;
  const bool in = i < m;
  double li = 0.0, di = 1.0, ui = 0.0, ri = 0.0;
  if (in) {
    li = lo_at(s, i);
    di = s.d[i];
    ui = up_at(s, i);
    ri = r_at(s, i);
  }
  for (int st = 1; st < m; st <<= 1) {
    lo[i] = li;
    d[i] = di;
    up[i] = ui;
    r[i] = ri;
    __syncthreads();
    if (in) {
      double al = 0.0, be = 0.0;
      double nl = 0.0, nu = 0.0;
      if (i - st >= 0) {
        al = -li / d[i - st];
        di += al * up[i - st];
        ri += al * r[i - st];
        nl = al * lo[i - st];
      }
      if (i + st < m) {
        be = -ui / d[i + st];
        di += be * lo[i + st];
        ri += be * r[i + st];
        nu = be * up[i + st];
      }
      li = nl;
      ui = nu;
    }
    __syncthreads();
  }
  if (in) x[i] = ri / di;
}

__global__ void tri_ends_kernel(double* u, int64_t ne, double u0, double u1) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    u[0] = u0;
    u[ne] = u1;
  }
}

// workspace (in doubles): per level cp[m] + ends[6*nc] + reduced LO,D,UP,R,X [5*ns]
static int64_t level_doubles(int64_t m) {
  int64_t tot = 0;
  while (m > kBase) {
    const int kLc = chunk_for(m);
    const int64_t nc = (m + kLc - 1) / kLc, ns = m / kLc;
    tot += m + 6 * nc + 5 * ns + 16;
    m = ns;
  }
  return tot + m + 16;
}

int64_t tridiag_work_bytes(int64_t ne) {
  const int64_t m = ne > 1 ? ne - 1 : 0;
  return 8 * level_doubles(m) + 256;
}

static hipError_t solve_level(const TriSys& s, double* x, double* work, hipStream_t st);

template <int kLc>
static hipError_t solve_level_chunked(const TriSys& s, double* x, double* work, hipStream_t st) {
  const int64_t nc = (s.m + kLc - 1) / kLc, ns = s.m / kLc;
  double* cp = work;
  ChunkEnds* ends = reinterpret_cast<ChunkEnds*>(cp + s.m);
  double* LO = reinterpret_cast<double*>(ends + nc);
  double* D = LO + ns;
  double* UP = D + ns;
  double* R = UP + ns;
  double* X = R + ns;
  double* next = X + ns + 16;
  const unsigned gc = (unsigned)((nc + kBlock - 1) / kBlock);
  const unsigned gs = (unsigned)((ns + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(tri_condense_kernel<kLc>, dim3(gc), dim3(kBlock), 0, st, s, nc, ends);
  hipLaunchKernelGGL(tri_reduce_kernel<kLc>, dim3(gs), dim3(kBlock), 0, st, s, ns, nc, ends, LO, D, UP, R);
  TriSys r{LO, D, UP, R, nullptr, nullptr, 0.0, 0.0, ns};
  hipError_t err = solve_level(r, X, next, st);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(tri_expand_kernel<kLc>, dim3(gc), dim3(kBlock), 0, st, s, ns, nc, X, x, cp);
  return hipGetLastError();
}

static hipError_t solve_level(const TriSys& s, double* x, double* work, hipStream_t st) {
  if (s.m <= 0) return hipSuccess;
  if (s.m <= kBase) {
    hipLaunchKernelGGL(tri_base_kernel, dim3(1), dim3((unsigned)kBase), 0, st, s, x, work);
    return hipGetLastError();
  }
  return solve_level_chunked<8>(s, x, work, st);
}

hipError_t tridiag_dirichlet_solve(const double* diag, const double* off, const double* load,
                                   int64_t ne, double u0, double u1, double* u, void* work,
                                   hipStream_t st) {
  hipLaunchKernelGGL(tri_ends_kernel, dim3(1), dim3(64), 0, st, u, ne, u0, u1);
  const int64_t m = ne - 1;
  if (m <= 0) return hipGetLastError();
  // interior unknown k <-> node k+1: lo = off[k], d = diag[k+1], up = off[k+1], r = load[k+1]
  TriSys s{off, diag + 1, off + 1, load + 1, off, off + (ne - 1), u0, u1, m};
  return solve_level(s, u + 1, reinterpret_cast<double*>(work), st);
}

}  // namespace lssvr
