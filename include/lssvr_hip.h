/*
 * lssvr_hip.h -- C ABI of the MI355X (gfx950) per-element LSSVR enhancement path.
 *
 * The reference (maryambabaei/hybrid-FEM-LSSVR) is pure Python and has no FFI; the
 * "reference interface" each entry point replaces is therefore a Python call
 * site in /root/reference/1D-Possion/Hybrid-FEM-LSSVR-Dual.py ("Dual.py").  The
 * ctypes binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - the caller owns every buffer; the library allocates nothing and keeps no
 *     state besides a thread-local last-error string;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and is safe to capture in a hipGraph;
 *   - return value: 0 = launched, <0 = argument error (see lssvr_last_error());
 *   - all floating point is IEEE binary64; element/node indices are int64.
 *   - element e of a mesh shard has end points x[e], x[e+1] and nodal values
 *     u[e], u[e+1] (Dual.py:143-147: element i <-> nodes (i, i+1)).
 */
#ifndef LSSVR_HIP_H
#define LSSVR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSSVR_ABI_VERSION 5

/* error codes */
#define LSSVR_OK              0
#define LSSVR_ERR_NULL      (-1)  /* a required pointer is NULL                   */
#define LSSVR_ERR_SIZE      (-2)  /* ne / P / n_colloc out of range               */
#define LSSVR_ERR_DEGREE    (-3)  /* M outside the supported range                */
#define LSSVR_ERR_RHS       (-4)  /* unknown rhs_id or missing rhs data           */
#define LSSVR_ERR_SOLVER    (-5)  /* unknown solver_id                            */
#define LSSVR_ERR_LAUNCH    (-6)  /* hipLaunch / runtime error                    */
#define LSSVR_ERR_QUAD      (-7)  /* unsupported quadrature order                 */

/* right-hand side f(x) of -u'' = f (Dual.py:11-12 `poisson_rhs`, passed as the
 * callable `rhs_func` at Dual.py:20,157).  A Python callable cannot cross the
 * ABI, so f is either tabulated by the host facade or named: */
#define LSSVR_RHS_ARRAY  0  /* rhs_values[e*n_colloc + k] = f(x_k of element e)   */
#define LSSVR_RHS_SIN    1  /* f(x) = p[0] * sin(p[1] * x), rounded like numpy's
                               `amp * np.sin(omega * x)`; Poisson: p = {pi^2, pi} */
#define LSSVR_RHS_ARRAY_PM 2 /* the same table POINT-major: rhs_values[k*ne + e] (ne = the launch's
                               element count; lssvr_enhance_subset: nsub, e = position in elem_ids).
                               The layout the lane-per-element kernels (M <= 22) read at full
                               HBM rate -- consecutive lanes = consecutive elements, no staging;
                               element-major rows reach them as 64-byte half-line requests, which
                               the memory system serves at ~3.3 TB/s.  ABI 4 */

/* layout of the tabulated arrays of lssvr_enhance_varcoef_ws (a_values, da_values, rhs_values) */
#define LSSVR_TABLE_ELEMENT_MAJOR 0  /* t[e*n_colloc + k]: what lssvr_colloc_points produces       */
#define LSSVR_TABLE_POINT_MAJOR   1  /* t[k*ne + e]: see LSSVR_RHS_ARRAY_PM                         */

/* per-element solver */
#define LSSVR_SOLVER_PRIMAL 0 /* BC-eliminated primal normal equations, (M-2) SPD, LDL^T
                                 (default; <=1e-15 of the exact minimiser on every BASELINE
                                 config).  When n_colloc < M-2 the primal Gram is rank
                                 deficient and the call is routed to LSSVR_SOLVER_DUAL (its
                                 limits then apply)                                      */
#define LSSVR_SOLVER_DUAL   1 /* north_star's dual Gram form (K + I/gamma) alpha = y: kernel
                                 Gram matrix of the collocation rows (boundary rows eliminated
                                 as a 2x2 block pivot), Jacobi-equilibrated, LU with partial
                                 pivoting, up to 3 safeguarded steps of iterative refinement
                                 (operator-form residual in compensated arithmetic).  n_colloc <= 64,
                                 M <= 33, Poisson and variable-coefficient rows.  Measured against the
                                 exact minimiser: <= 1e-12 on BASELINE configs 1-3, <= 5e-11 at degree
                                 32 / 64 points (config 4), 1e-8 where n_colloc ~ M .. M+6 (DESIGN.md
                                 section 3.2b).  FP64 vector FMAs only, no MFMA; 40x slower than
                                 LSSVR_SOLVER_PRIMAL: the cross-check solver                      */
#define LSSVR_SOLVER_PRIMAL_MOMENT 3 /* same algorithm as PRIMAL as the kernel sequence of
                                 csrc/enhance_large_cheb.hip (Chebyshev-moment Gram, four systems
                                 per wave in the LDL^T) for ANY M, Poisson rows -- what PRIMAL
                                 itself runs above M = 22 with a workspace; for A/B
                                 measurements below.  lssvr_enhance_ws only: it needs the
                                 workspace of lssvr_enhance_work_bytes() (LSSVR_ERR_SOLVER
                                 without one) */
#define LSSVR_SOLVER_PRIMAL_WAVE 2 /* same algorithm as PRIMAL, forced onto the
                                 wave-per-element / f64-MFMA Gram mapping whatever M is
                                 (PRIMAL picks lane-per-element for M <= 22); for A/B
                                 measurements of the two mappings */

/* per-element status written to status[e] */
#define LSSVR_ST_OK        0
#define LSSVR_ST_FALLBACK  1  /* factorisation broke down / non-finite: the element got
                                 the linear interpolant of (g_l, g_r) -- Dual.py:164-169 */

/* ABI version, == LSSVR_ABI_VERSION of the header the library was built from. */
int lssvr_version(void);

/* Message for the last <0 return on this thread ("" if none). */
const char* lssvr_last_error(void);

/*
 * lssvr_enhance -- the hot path.  Replaces the serial loop
 * `solve_lssvr_subproblems` (Dual.py:139-169) and every `lssvr_primal` call in it
 * (Dual.py:20-98): one independent QP per element, solved in closed form.
 *
 *   x[ne+1], u[ne+1]  node coordinates / FEM nodal values of this shard
 *                     (Dual.py:144-147; fem_nodes / fem_values, Dual.py:134-135)
 *   ne                elements in this shard (>= 0; 0 is a no-op)
 *   elem_offset       global index of the shard's first element, ne_global = total
 *                     elements: element is_left/is_right_boundary iff its global
 *                     index is 0 / ne_global-1 (Dual.py:150-151)
 *   gxmin, gxmax      global_domain (Dual.py:101,161); the Dirichlet value bc_left /
 *                     bc_right replaces u on a boundary element only if its end
 *                     point == gxmin / gxmax exactly (Dual.py:65,72)
 *   M                 number of Legendre coefficients (`lssvr_M`, Dual.py:47), 2..33
 *   n_colloc          collocation points per element, end points included, >= 2
 *                     (hard-coded 12 at Dual.py:40)
 *   gamma             `lssvr_gamma` (Dual.py:49)
 *   rhs_id/rhs_params/rhs_values   see LSSVR_RHS_* (rhs_params is a HOST pointer)
 *   W[ne*M]           out: row e = Legendre coefficients of element e on domain
 *                     [x[e], x[e+1]], window [-1,1] (`Legendre(res.x[:M], domain)`,
 *                     Dual.py:95)
 *   status[ne]        out (may be NULL): LSSVR_ST_*
 *   fail_count        in/out (may be NULL): device int32, incremented once per
 *                     fallback element (the reference prints per element instead,
 *                     Dual.py:165)
 */
int lssvr_enhance(const double* x, const double* u, int64_t ne,
                  int64_t elem_offset, int64_t ne_global,
                  double gxmin, double gxmax, double bc_left, double bc_right,
                  int M, int n_colloc, double gamma,
                  int rhs_id, const double* rhs_params_host, const double* rhs_values,
                  int solver_id,
                  double* W, int32_t* status, int32_t* fail_count, void* stream);

/*
 * lssvr_enhance_ws -- lssvr_enhance with a caller-provided device workspace (the library still
 * allocates nothing).  Poisson rows above M = 22 then run as TWO kernels -- Chebyshev moments of
 * the collocation points (96 doubles per element into `work`), then the four-systems-per-wave
 * solve -- twice the speed of the single f64-MFMA kernel lssvr_enhance launches without a
 * workspace.  Where n_colloc - (M-2) <= 14 (about as many equispaced
 * points as bubble coefficients: normal equations lose up to ten digits) the pair is followed by
 * 1-3 refinement steps with the residual taken through the collocation rows (32 more doubles per
 * element; 5e-14 instead of 1e-6 at M = 33, n_colloc = 31 -- DESIGN.md section 2).  Every other
 * case behaves exactly like lssvr_enhance.
 *   lssvr_enhance_work_bytes(ne, M, n_colloc, solver_id)   bytes `work` must hold (0: none needed)
 *   work / work_bytes      device scratch, contents undefined afterwards.  NULL: the workspace-free
 *                          kernels run (above M = 22 the single f64-MFMA kernel: about half the
 *                          speed, no near-square refinement).  Non-NULL but smaller than
 *                          lssvr_enhance_work_bytes(): LSSVR_ERR_SIZE (ABI 4; ABI 3 fell back
 *                          silently)
 *   kernel_ms_host != NULL BLOCKING measurement aid like lssvr_enhance_profiled: the duration of
 *                          the launch (of the PAIR of kernels, gap included, on the split path)
 */
int64_t lssvr_enhance_work_bytes(int64_t ne, int M, int n_colloc, int solver_id);
int lssvr_enhance_ws(const double* x, const double* u, int64_t ne,
                     int64_t elem_offset, int64_t ne_global,
                     double gxmin, double gxmax, double bc_left, double bc_right,
                     int M, int n_colloc, double gamma,
                     int rhs_id, const double* rhs_params_host, const double* rhs_values,
                     int solver_id,
                     double* W, int32_t* status, int32_t* fail_count,
                     void* work, int64_t work_bytes, void* stream, float* kernel_ms_host);

/*
 * lssvr_step -- one whole step of the hot path on one mesh shard in ONE launch:
 * lssvr_p1_assemble (in-kernel rhs, nquad-point Gauss) + lssvr_enhance (primal
 * solver, in-kernel rhs).  For M <= 22 the two run as disjoint block ranges of a single
 * grid; arguments as in the two separate calls.  PRIMAL SOLVER ONLY: n_colloc < M-2 (rank-
 * deficient primal normal equations) returns LSSVR_ERR_SOLVER -- use lssvr_p1_assemble +
 * lssvr_enhance, which routes that regime to the dual solver.
 */
int lssvr_step(const double* x, const double* u, int64_t ne,
               int64_t elem_offset, int64_t ne_global,
               double gxmin, double gxmax, double bc_left, double bc_right,
               int M, int n_colloc, double gamma, const double* rhs_params_host, int nquad,
               double* diag, double* off, double* load,
               double* W, int32_t* status, int32_t* fail_count, void* stream);

/*
 * lssvr_step_plan_* -- lssvr_step bound once, launched many times.  A time-stepping caller issues the
 * same step on the same resident buffers over and over (the reference's loop, Dual.py:139-169, re-run
 * after every FEM solve); validating and marshalling 21 arguments per call costs a host more than the
 * 3 us the launch itself does, and at 7-8 us per step that is what decides whether a stream stays fed.
 *   create : the arguments of lssvr_step (without the stream), checked exactly as lssvr_step checks
 *            them; *plan receives an opaque handle (a small HOST allocation: no device memory, no
 *            HIP call).  The buffers are referenced, not copied: their CONTENTS may change between
 *            launches, their addresses and sizes may not.
 *   launch : what lssvr_step would enqueue, on `stream`; asynchronous.  A plan may be launched on any
 *            stream, and concurrently from several threads (it is read-only after create).
 *   destroy: frees the handle (NULL is allowed); launches already enqueued are not affected.
 */
typedef struct lssvr_step_plan lssvr_step_plan;
int lssvr_step_plan_create(lssvr_step_plan** plan, const double* x, const double* u, int64_t ne,
                           int64_t elem_offset, int64_t ne_global,
                           double gxmin, double gxmax, double bc_left, double bc_right,
                           int M, int n_colloc, double gamma, const double* rhs_params_host, int nquad,
                           double* diag, double* off, double* load,
                           double* W, int32_t* status, int32_t* fail_count);
int lssvr_step_plan_launch(const lssvr_step_plan* plan, void* stream);
int lssvr_step_plan_destroy(lssvr_step_plan* plan);

/*
 * lssvr_enhance_varcoef -- BASELINE config 5, -(a u')' = f (no reference
 * counterpart: Dual.py:44,119 hard-code -u'').  PDE row k of element e is
 *   -a_k (2/h)^2 L_p''(t_k) - da_k (2/h) L_p'(t_k),
 * with a_values/da_values/rhs_values tabulated at the collocation points
 * ([ne*n_colloc], row-major per element).  Other arguments as lssvr_enhance.
 * n_colloc < M-2 is routed to the dual solver (n_colloc <= 64), like lssvr_enhance.
 */
int lssvr_enhance_varcoef(const double* x, const double* u, int64_t ne,
                          int64_t elem_offset, int64_t ne_global,
                          double gxmin, double gxmax, double bc_left, double bc_right,
                          int M, int n_colloc, double gamma,
                          const double* a_values, const double* da_values,
                          const double* rhs_values,
                          double* W, int32_t* status, int32_t* fail_count, void* stream);

/*
 * lssvr_enhance_varcoef_ws -- lssvr_enhance_varcoef with a caller workspace and the measurement aid
 * of lssvr_enhance_ws (ABI 4).  lssvr_enhance_varcoef_work_bytes(ne, M, n_colloc) bytes (0: none
 * needed); work == NULL runs the workspace-free kernels; too small: LSSVR_ERR_SIZE.
 * table_layout: LSSVR_TABLE_* of a_values / da_values / rhs_values (all three alike).
 * kernel_ms_host != NULL: BLOCKING, the duration of the launch (bench.py's roofline).
 */
int64_t lssvr_enhance_varcoef_work_bytes(int64_t ne, int M, int n_colloc);
int lssvr_enhance_varcoef_ws(const double* x, const double* u, int64_t ne,
                             int64_t elem_offset, int64_t ne_global,
                             double gxmin, double gxmax, double bc_left, double bc_right,
                             int M, int n_colloc, double gamma,
                             const double* a_values, const double* da_values,
                             const double* rhs_values, int table_layout,
                             double* W, int32_t* status, int32_t* fail_count,
                             void* work, int64_t work_bytes, void* stream, float* kernel_ms_host);

/*
 * lssvr_step_varcoef -- one whole step of BASELINE config 5 on one mesh shard: the a-weighted P1
 * assembly (lssvr_p1_assemble with LSSVR_RHS_ARRAY tables rhs_quad / a_quad at the nquad Gauss points,
 * lssvr_quad_points) + lssvr_enhance_varcoef (a_values / da_values / rhs_values at the collocation
 * points, table_layout = LSSVR_TABLE_*), as disjoint block ranges of ONE grid for M <= 12 (two
 * launches above).  Primal solver only (n_colloc >= M-2).  Arguments as in the two calls.  ABI 4.
 */
int lssvr_step_varcoef(const double* x, const double* u, int64_t ne,
                       int64_t elem_offset, int64_t ne_global,
                       double gxmin, double gxmax, double bc_left, double bc_right,
                       int M, int n_colloc, double gamma,
                       const double* a_values, const double* da_values, const double* rhs_values,
                       int table_layout, int nquad, const double* rhs_quad, const double* a_quad,
                       double* diag, double* off, double* load,
                       double* W, int32_t* status, int32_t* fail_count, void* stream);

/*
 * lssvr_enhance_subset -- heterogeneous meshes (per-element gamma, degree and
 * collocation count; the reference has one lssvr_M / lssvr_gamma for the whole mesh,
 * Dual.py:101).  Enhances the nsub elements elem_ids[0..nsub) of a shard of ne_mesh elements
 * with ONE (M, n_colloc); a p-adaptive mesh is one call per distinct (M, n_colloc) group.
 *   elem_ids[nsub]       device int64 mesh indices, each in [0, ne_mesh) and distinct
 *                        (NULL = all elements in order; nsub must then equal ne_mesh)
 *   gamma_values[ne_mesh] device, indexed by MESH element (NULL = the scalar gamma)
 *   rhs_values[nsub*n_colloc]  (LSSVR_RHS_ARRAY) indexed by position k in elem_ids
 *   W, ldw               row of mesh element id starts at W + id*ldw (ldw >= M; 0 = M): with
 *                        ldw = max M of the mesh and W zeroed beforehand every row is a valid
 *                        Legendre series for lssvr_eval (trailing zeros change nothing)
 *   status[ne_mesh]      indexed by mesh element (may be NULL)
 * Primal solver only (n_colloc >= M-2).  x, u, elem_offset, ne_global, ... as lssvr_enhance.
 */
int lssvr_enhance_subset(const double* x, const double* u, int64_t ne_mesh,
                         const int64_t* elem_ids, int64_t nsub,
                         int64_t elem_offset, int64_t ne_global,
                         double gxmin, double gxmax, double bc_left, double bc_right,
                         int M, int n_colloc, double gamma, const double* gamma_values,
                         int rhs_id, const double* rhs_params_host, const double* rhs_values,
                         double* W, int64_t ldw, int32_t* status, int32_t* fail_count,
                         void* stream);

/*
 * lssvr_enhance_subset_ws -- lssvr_enhance_subset with a caller workspace (ABI 4): above M = 22 the
 * group then runs as the moment / solve kernel pair of lssvr_enhance_ws (twice the speed of the
 * single f64-MFMA kernel, and the near-square refinement) -- rows, status and gamma_values by mesh
 * index, tables and the workspace by position in elem_ids.
 *   work / work_bytes   lssvr_enhance_work_bytes(nsub, M, n_colloc, LSSVR_SOLVER_PRIMAL) bytes of device
 *                       scratch (0 below M = 23); NULL: as lssvr_enhance_subset; too small: LSSVR_ERR_SIZE
 */
int lssvr_enhance_subset_ws(const double* x, const double* u, int64_t ne_mesh,
                            const int64_t* elem_ids, int64_t nsub,
                            int64_t elem_offset, int64_t ne_global,
                            double gxmin, double gxmax, double bc_left, double bc_right,
                            int M, int n_colloc, double gamma, const double* gamma_values,
                            int rhs_id, const double* rhs_params_host, const double* rhs_values,
                            double* W, int64_t ldw, int32_t* status, int32_t* fail_count,
                            void* work, int64_t work_bytes, void* stream);

/*
 * lssvr_enhance_shared -- UNIFORM meshes only; a separate, faster form of the hot path, never
 * chosen implicitly.  On a uniform mesh every element has the same system matrix, so the
 * coefficients are a linear map of the element's data:
 *     W[e,:] = sum_k op[k,:] f(x_k)/scl_e^2 + op[n,:] g_l + op[n+1,:] g_r .
 * op[(n_colloc+2)*M] (device, row-major) is built by the caller with lssvr_enhance itself on a
 * few elements of the mesh's spacing h: rows k < n = response to rhs_values = scl^2 e_k with zero
 * nodal values, row n / n+1 = response to (g_l, g_r) = (1,0) / (0,1) with zero rhs
 * (hybrid_fem_lssvr_amd.ops.build_shared_operator does exactly that; gamma enters only there).
 * Per element the abscissae, f, scl and the boundary rule of lssvr_enhance stay exact; shared
 * is the operator: relative L2 distance from lssvr_enhance ~ (|x|/h) * 2e-16 (1e-11 at 1e5
 * elements of h = 1/12).  M <= 33;
 * LSSVR_RHS_SIN needs |omega x| < 3e9 (beyond: status = LSSVR_ST_FALLBACK).
 * kernel_ms_host != NULL: blocking, returns the dispatch's own duration (measurement aid).
 * The caller is responsible for the mesh being uniform.
 */
int lssvr_enhance_shared(const double* x, const double* u, int64_t ne,
                         int64_t elem_offset, int64_t ne_global,
                         double gxmin, double gxmax, double bc_left, double bc_right,
                         int M, int n_colloc,
                         int rhs_id, const double* rhs_params_host, const double* rhs_values,
                         const double* op,
                         double* W, int32_t* status, int32_t* fail_count, void* stream,
                         float* kernel_ms_host);

/*
 * lssvr_colloc_points -- x_k of every element exactly as `np.linspace(xmin, xmax, n)`
 * produces them (Dual.py:40): xc[e*n + k] = fl(fl(k*step)+x[e]), last = x[e+1].
 * Lets the host tabulate an arbitrary `rhs_func` for LSSVR_RHS_ARRAY.
 */
int lssvr_colloc_points(const double* x, int64_t ne, int n_colloc, double* xc, void* stream);
/* the same abscissae POINT-major, xc[k*ne + e] (ABI 4): tabulate a function on it and the table is
 * in LSSVR_RHS_ARRAY_PM / LSSVR_TABLE_POINT_MAJOR layout */
int lssvr_colloc_points_pm(const double* x, int64_t ne, int n_colloc, double* xc, void* stream);

/*
 * lssvr_p1_assemble -- element-local P1 stiffness and load and their scatter to
 * the global tridiagonal system.  Replaces `laplace.assemble(basis)` /
 * `load.assemble(basis)` (Dual.py:117-128) for ElementLineP1 on a MeshLine:
 *   k_e = abar_e/h_e [[1,-1],[-1,1]],  f_e[j] = sum_q w_q h_e f(x_q) phi_j(xi_q),
 * Gauss-Legendre with `nquad` points per element (scikit-fem's default for P1 is
 * 2).  abar_e = quadrature mean of a (1 when a_quad is NULL).
 *   rhs_id = LSSVR_RHS_SIN: f evaluated in-kernel; LSSVR_RHS_ARRAY:
 *   rhs_quad[e*nquad + q] = f(x_q).   a_quad[e*nquad + q] likewise (may be NULL).
 *   diag[ne+1], off[ne], load[ne+1]  out: assembled bands (off[i] couples i,i+1)
 *   kloc[ne], floc[2*ne]             out, may be NULL: element-local k_e scale and
 *                                    the two load entries
 */
int lssvr_p1_assemble(const double* x, int64_t ne, int nquad,
                      int rhs_id, const double* rhs_params_host, const double* rhs_quad,
                      const double* a_quad,
                      double* diag, double* off, double* load,
                      double* kloc, double* floc, void* stream);

/*
 * lssvr_quad_points -- quadrature abscissae xq[e*nquad + q] used by
 * lssvr_p1_assemble (for host tabulation of rhs / a).
 */
int lssvr_quad_points(const double* x, int64_t ne, int nquad, double* xq, void* stream);

/*
 * lssvr_tridiag_dirichlet_solve -- `enforce(A, b, D=all boundary dofs)` + `solve`
 * (Dual.py:129-130) for the assembled P1 bands: u[0]=u0, u[ne]=u1, interior by a
 * device tridiagonal solve.  work: device scratch of lssvr_tridiag_work_bytes(ne).
 */
int64_t lssvr_tridiag_work_bytes(int64_t ne);
int lssvr_tridiag_dirichlet_solve(const double* diag, const double* off, const double* load,
                                  int64_t ne, double u0, double u1,
                                  double* u, void* work, void* stream);

/*
 * lssvr_p1_flux_solve -- the same `enforce` + `solve` (Dual.py:129-130) for the P1 system
 * that lssvr_p1_assemble produces, from the element stiffnesses kloc[ne] (k_e = abar_e/h_e)
 * and the assembled load[ne+1]: A = D^T K D, so u follows from one prefix scan of the
 * element fluxes (no elimination, no amplification by cond(A) ~ ne^2).  u[0] = u0,
 * u[ne] = u1.  work: device scratch of lssvr_p1_flux_work_bytes(ne).  This is what the
 * Python facade's solve_fem uses; lssvr_tridiag_dirichlet_solve takes arbitrary bands.
 */
int64_t lssvr_p1_flux_work_bytes(int64_t ne);
int lssvr_p1_flux_solve(const double* kloc, const double* load, int64_t ne, double u0, double u1,
                        double* u, void* work, void* stream);

/*
 * Sharded form of lssvr_p1_flux_solve (one process per GPU, contiguous element shards).
 * The scan operator is associative, so a shard is summarised by ONE aggregate:
 *   lssvr_p1_flux_aggregate  -> agg3[3] (device) for the shard's ne elements; `work` (same
 *                               size rule) keeps the block scan for the second call;
 *   (caller: all-gather the 24-byte aggregates, combine those of the lower ranks in rank
 *    order into prefix3 and all of them into grand3 with
 *    (a1,r1,g1) o (a2,r2,g2) = (a1+a2, r1+r2, g1+g2 + r2*a1) )
 *   lssvr_p1_flux_finish     -> u[ne+1] of the shard's nodes.
 * first_global / last_global: the shard holds the mesh's first / last element.  kloc[ne],
 * load[ne+1] are the shard's slices (load[i] must be complete for i < ne: assemble with one
 * halo node on the left).  prefix3 / grand3 are DEVICE double[3]; NULL = empty / own total.
 */
int lssvr_p1_flux_aggregate(const double* kloc, const double* load, int64_t ne, int first_global,
                            void* work, double* agg3, void* stream);
int lssvr_p1_flux_finish(const double* kloc, const double* load, int64_t ne, int first_global,
                         int last_global, const void* work, const double* prefix3,
                         const double* grand3, double u0, double u1, double* u, void* stream);

/*
 * lssvr_eval -- `evaluate_solution` (Dual.py:176-203): for each query point the
 * first element j with x[j] <= xq <= x[j+1] (points on an interior node take the
 * LEFT element; below/above the mesh -> element 0 / ne-1, polynomial
 * extrapolation; NaN -> elem -1, value 0), then Clenshaw evaluation in numpy's
 * operation order (legendre.py `legval`).
 *   uq[P] out; elem[P] out (may be NULL), int64 element indices.
 */
int lssvr_eval(const double* x, const double* W, int64_t ne, int M,
               const double* xq, int64_t P, double* uq, int64_t* elem, void* stream);

/*
 * lssvr_eval_error -- error norms of the hybrid solution against the exact solution
 * ex(x) = p[0] * sin(p[1] * x) on the query points (Dual.py:216-217 evaluates
 * `computed_solution` and `exact_solution = true_solution(test_points)`, Dual.py:8-9:
 * p = {1, pi}); reductions on the device, the query values never leave HBM:
 *   out3[0] += sum (u - ex)^2,  out3[1] += sum ex^2,  out3[2] = max(out3[2], max |u - ex|).
 * out3 is a DEVICE double[3] the caller zeroes first (accumulates across calls / shards);
 * NaN query points are skipped, like in lssvr_eval.
 */
int lssvr_eval_error(const double* x, const double* W, int64_t ne, int M,
                     const double* xq, int64_t P, const double* exact_params_host,
                     double* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSSVR_HIP_H */
