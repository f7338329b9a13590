// extern "C" layer: argument validation, error strings, dispatch.  No torch,
// no Python types -- plain pointers and sizes (include/lssvr_hip.h).
#include <cstdarg>
#include <vector>
#include <cstdio>
#include <cstring>

#include <new>
#include "lssvr_kernels.hpp"

namespace {

thread_local char g_err[256] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(hipError_t e, const char* what) {
  if (e == hipSuccess) return LSSVR_OK;
  return fail(LSSVR_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

int fill_enhance_args(lssvr::EnhanceArgs& a, const double* x, const double* u, int64_t ne,
                      int64_t elem_offset, int64_t ne_global, double gxmin, double gxmax,
                      double bc_left, double bc_right, int M, int n_colloc, double gamma,
                      double* W) {
  if (ne < 0) return fail(LSSVR_ERR_SIZE, "ne = %lld < 0", (long long)ne);
  if (ne > 0 && (!x || !u || !W)) return fail(LSSVR_ERR_NULL, "x, u and W must be non-NULL");
  if (elem_offset < 0 || ne_global < elem_offset + ne)
    return fail(LSSVR_ERR_SIZE, "shard [%lld, %lld) does not fit ne_global = %lld",
                (long long)elem_offset, (long long)(elem_offset + ne), (long long)ne_global);
  if (M < 2 || M > lssvr::kLargeMaxM)
    return fail(LSSVR_ERR_DEGREE, "M = %d outside [2, %d]", M, lssvr::kLargeMaxM);
  if (n_colloc < 2 || n_colloc > 4096)
    return fail(LSSVR_ERR_SIZE, "n_colloc = %d outside [2, 4096]", n_colloc);
  if (!(gamma > 0.0)) return fail(LSSVR_ERR_SIZE, "gamma must be > 0");
  if (ne * (int64_t)M / M != ne) return fail(LSSVR_ERR_SIZE, "ne*M overflows");
  a = lssvr::EnhanceArgs{};
  a.x = x;
  a.u = u;
  a.ne = ne;
  a.ne_mesh = ne;
  a.elem_offset = elem_offset;
  a.ne_global = ne_global;
  a.gxmin = gxmin;
  a.gxmax = gxmax;
  a.bc_left = bc_left;
  a.bc_right = bc_right;
  a.gamma = gamma;
  a.inv_gamma = 1.0 / gamma;
  a.M = M;
  a.n = n_colloc;
  a.refine = lssvr::enhance_small_refine_steps(M, n_colloc);     // (read by the Poisson lane kernel only)
  a.tab_es = n_colloc;      // tabulated arrays: element-major unless set_rhs / the caller says otherwise
  a.tab_ps = 1;
  a.W = W;
  static const lssvr::TrigTables trig = lssvr::make_trig_tables();
  a.trig = trig;
  return LSSVR_OK;
}

// right-hand side of an enhancement call: named (in-kernel) or tabulated
int set_rhs(lssvr::EnhanceArgs& a, int rhs_id, const double* rhs_params_host,
            const double* rhs_values, bool need_values, const char* count_name) {
  a.rhs_id = rhs_id;
  if (rhs_id == LSSVR_RHS_SIN) {
    if (!rhs_params_host) return fail(LSSVR_ERR_RHS, "LSSVR_RHS_SIN needs rhs_params = {amp, omega}");
    a.rhs_amp = rhs_params_host[0];
    a.rhs_omega = rhs_params_host[1];
    return LSSVR_OK;
  }
  if (rhs_id == LSSVR_RHS_ARRAY || rhs_id == LSSVR_RHS_ARRAY_PM) {
    if (need_values && !rhs_values)
      return fail(LSSVR_ERR_RHS, "LSSVR_RHS_ARRAY needs rhs_values[%s*n_colloc]", count_name);
    a.rhs_id = LSSVR_RHS_ARRAY;
    a.rhs_values = rhs_values;
    if (rhs_id == LSSVR_RHS_ARRAY_PM) {     // rhs_values[k * count + e]
      a.tab_es = 1;
      a.tab_ps = a.ne > 0 ? a.ne : 1;
    }
    return LSSVR_OK;
  }
  return fail(LSSVR_ERR_RHS, "unknown rhs_id %d", rhs_id);
}

}  // namespace

namespace {
// shared tail of lssvr_enhance / lssvr_enhance_profiled
int enhance_dispatch(const lssvr::EnhanceArgs& a, int solver_id, hipStream_t s,
                     const lssvr::LaunchOpts* o, void* work = nullptr, int64_t work_bytes = 0) {
  // Fewer collocation points than bubble coefficients: the primal normal equations are rank
  // deficient (float64 returns O(1) errors there), the dual Gram system is well conditioned.
  if (solver_id != LSSVR_SOLVER_DUAL && a.n < a.M - 2) solver_id = LSSVR_SOLVER_DUAL;
  if (solver_id == LSSVR_SOLVER_DUAL) {
    if (a.n > 64) return fail(LSSVR_ERR_SIZE, "dual solver: n_colloc = %d > 64", a.n);
    if (a.elem_ids) return fail(LSSVR_ERR_SOLVER, "dual solver: no subset form");
    return check_launch(lssvr::enhance_dual(a, s, o), "enhance_dual");
  }
  if (a.M <= lssvr::kSmallMaxM && solver_id == LSSVR_SOLVER_PRIMAL)
    return check_launch(lssvr::enhance_small(a, s, o), "enhance_small");
  if (a.elem_ids && solver_id != LSSVR_SOLVER_PRIMAL)
    return fail(LSSVR_ERR_SOLVER, "subset launches take LSSVR_SOLVER_PRIMAL");
  // large degree, Poisson rows, workspace given: Chebyshev moments + four-systems-per-wave solve as
  // two kernels (twice the speed of the MFMA kernel)
  if (solver_id == LSSVR_SOLVER_PRIMAL && !a.a_values && work &&
      work_bytes >= lssvr::enhance_moment_ws_bytes(a.ne, a.M, a.n))
    return check_launch(lssvr::enhance_large_split(a, work, s, o), "enhance_large_split");
  // LSSVR_SOLVER_PRIMAL_MOMENT forces that sequence for any M (A/B against the lane kernel below M = 23)
  if (solver_id == LSSVR_SOLVER_PRIMAL_MOMENT) {
    if (a.a_values) return fail(LSSVR_ERR_SOLVER, "LSSVR_SOLVER_PRIMAL_MOMENT: Poisson rows only");
    if (a.elem_ids) return fail(LSSVR_ERR_SOLVER, "LSSVR_SOLVER_PRIMAL_MOMENT: no subset form");
    if (!work || work_bytes < lssvr::enhance_moment_ws_bytes(a.ne, a.M, a.n))
      return fail(LSSVR_ERR_SOLVER, "LSSVR_SOLVER_PRIMAL_MOMENT needs a workspace of "
                  "lssvr_enhance_work_bytes() bytes (lssvr_enhance_ws)");
    return check_launch(lssvr::enhance_large_split(a, work, s, o), "enhance_large_split");
  }
  // otherwise the direct Gram on the f64 matrix cores
  return check_launch(lssvr::enhance_large(a, s, o), "enhance_large");
}
}  // namespace

namespace {
// enhance_dispatch, optionally BLOCKING and stamped with the dispatch's own begin / end times
int dispatch_timed(const lssvr::EnhanceArgs& a, int solver_id, hipStream_t s, void* work,
                   int64_t work_bytes, float* kernel_ms_host) {
  if (!kernel_ms_host) return enhance_dispatch(a, solver_id, s, nullptr, work, work_bytes);
  lssvr::LaunchOpts o;
  if (hipEventCreate(&o.start) != hipSuccess || hipEventCreate(&o.stop) != hipSuccess)
    return fail(LSSVR_ERR_LAUNCH, "hipEventCreate failed");
  int rc = enhance_dispatch(a, solver_id, s, &o, work, work_bytes);
  if (rc == LSSVR_OK) {
    hipError_t e = hipEventSynchronize(o.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(kernel_ms_host, o.start, o.stop);
    if (e != hipSuccess) rc = fail(LSSVR_ERR_LAUNCH, "profiled launch: %s", hipGetErrorString(e));
  }
  (void)hipEventDestroy(o.start);
  (void)hipEventDestroy(o.stop);
  return rc;
}
}  // namespace

namespace {
// `repeats` launches of enhance_dispatch back to back, each with its own begin / end stamps, ONE
// synchronisation at the end (lssvr_enhance_ws_sequence, lssvr_enhance_varcoef_ws_sequence)
int dispatch_sequence(const lssvr::EnhanceArgs& a, int solver_id, hipStream_t s, void* work, int64_t work_bytes,
                      int repeats, float* kernel_ms_host) {
  std::vector<lssvr::LaunchOpts> ev((size_t)repeats);
  int made = 0;
  for (; made < repeats; ++made)
    if (hipEventCreate(&ev[made].start) != hipSuccess || hipEventCreate(&ev[made].stop) != hipSuccess) break;
  int rc = made == repeats ? LSSVR_OK : fail(LSSVR_ERR_LAUNCH, "hipEventCreate failed");
  for (int r = 0; r < repeats && rc == LSSVR_OK; ++r) rc = enhance_dispatch(a, solver_id, s, &ev[r], work, work_bytes);
  hipError_t e = hipStreamSynchronize(s);            // (also after a failed launch: earlier ones are in flight)
  if (rc == LSSVR_OK && e != hipSuccess) rc = fail(LSSVR_ERR_LAUNCH, "profiled sequence: %s", hipGetErrorString(e));
  for (int r = 0; r < repeats && rc == LSSVR_OK; ++r) {
    e = hipEventElapsedTime(&kernel_ms_host[r], ev[r].start, ev[r].stop);
    if (e != hipSuccess) rc = fail(LSSVR_ERR_LAUNCH, "profiled sequence: %s", hipGetErrorString(e));
  }
  for (int r = 0; r < repeats; ++r) {
    if (ev[r].start) (void)hipEventDestroy(ev[r].start);
    if (ev[r].stop) (void)hipEventDestroy(ev[r].stop);
  }
  return rc;
}
}  // namespace

extern "C" {

int lssvr_version(void) { return LSSVR_ABI_VERSION; }

const char* lssvr_last_error(void) { return g_err; }

int lssvr_enhance(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                  int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
                  int M, int n_colloc, double gamma, int rhs_id, const double* rhs_params_host,
                  const double* rhs_values, int solver_id, double* W, int32_t* status,
                  int32_t* fail_count, void* stream) {
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, ne > 0, "ne");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  a.fail_count = fail_count;
  if (solver_id != LSSVR_SOLVER_PRIMAL && solver_id != LSSVR_SOLVER_DUAL &&
      solver_id != LSSVR_SOLVER_PRIMAL_WAVE && solver_id != LSSVR_SOLVER_PRIMAL_MOMENT)
    return fail(LSSVR_ERR_SOLVER, "unknown solver_id %d", solver_id);
  if (ne == 0) return LSSVR_OK;
  return enhance_dispatch(a, solver_id, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int64_t lssvr_enhance_work_bytes(int64_t ne, int M, int n_colloc, int solver_id) {
  if (ne <= 0) return 0;
  if (solver_id == LSSVR_SOLVER_PRIMAL_MOMENT) return lssvr::enhance_moment_ws_bytes(ne, M, n_colloc);
  if (solver_id != LSSVR_SOLVER_PRIMAL || M <= lssvr::kSmallMaxM) return 0;
  return lssvr::enhance_moment_ws_bytes(ne, M, n_colloc);
}

int lssvr_enhance_ws(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                     int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
                     int M, int n_colloc, double gamma, int rhs_id, const double* rhs_params_host,
                     const double* rhs_values, int solver_id, double* W, int32_t* status,
                     int32_t* fail_count, void* work, int64_t work_bytes, void* stream,
                     float* kernel_ms_host) {
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, ne > 0, "ne");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  a.fail_count = fail_count;
  if (solver_id != LSSVR_SOLVER_PRIMAL && solver_id != LSSVR_SOLVER_DUAL &&
      solver_id != LSSVR_SOLVER_PRIMAL_WAVE && solver_id != LSSVR_SOLVER_PRIMAL_MOMENT)
    return fail(LSSVR_ERR_SOLVER, "unknown solver_id %d", solver_id);
  if (work_bytes < 0 || (work_bytes > 0 && !work)) return fail(LSSVR_ERR_NULL, "work / work_bytes inconsistent");
  // a workspace that is given but too small is an error, not a silent change of kernel (and, in
  // the near-square regime, of accuracy: the refinement needs its 32 extra doubles per element)
  const int64_t need = lssvr_enhance_work_bytes(ne, M, n_colloc, solver_id);
  if (work && work_bytes < need)
    return fail(LSSVR_ERR_SIZE, "work holds %lld bytes, lssvr_enhance_work_bytes(%lld, %d, %d, %d) = %lld "
                "(pass work = NULL for the workspace-free kernels)", (long long)work_bytes, (long long)ne, M,
                n_colloc, solver_id, (long long)need);
  if (ne == 0) return LSSVR_OK;
  return dispatch_timed(a, solver_id, reinterpret_cast<hipStream_t>(stream), work, work_bytes, kernel_ms_host);
}

int lssvr_enhance_ws_sequence(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                              int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
                              int M, int n_colloc, double gamma, int rhs_id, const double* rhs_params_host,
                              const double* rhs_values, int solver_id, double* W, int32_t* status,
                              int32_t* fail_count, void* work, int64_t work_bytes, void* stream,
                              int repeats, float* kernel_ms_host) {
  if (!kernel_ms_host) return fail(LSSVR_ERR_NULL, "kernel_ms_host must be non-NULL (float[repeats])");
  if (repeats < 1 || repeats > 100000) return fail(LSSVR_ERR_SIZE, "repeats = %d outside [1, 100000]", repeats);
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "nothing to profile: ne = %lld", (long long)ne);
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, true, "ne");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  a.fail_count = fail_count;
  if (solver_id != LSSVR_SOLVER_PRIMAL && solver_id != LSSVR_SOLVER_DUAL &&
      solver_id != LSSVR_SOLVER_PRIMAL_WAVE && solver_id != LSSVR_SOLVER_PRIMAL_MOMENT)
    return fail(LSSVR_ERR_SOLVER, "unknown solver_id %d", solver_id);
  if (work_bytes < 0 || (work_bytes > 0 && !work)) return fail(LSSVR_ERR_NULL, "work / work_bytes inconsistent");
  const int64_t need = lssvr_enhance_work_bytes(ne, M, n_colloc, solver_id);
  if (work && work_bytes < need)
    return fail(LSSVR_ERR_SIZE, "work holds %lld bytes, lssvr_enhance_work_bytes() = %lld", (long long)work_bytes,
                (long long)need);
  return dispatch_sequence(a, solver_id, reinterpret_cast<hipStream_t>(stream), work, work_bytes, repeats,
                           kernel_ms_host);
}

int lssvr_enhance_profiled(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                           int64_t ne_global, double gxmin, double gxmax, double bc_left,
                           double bc_right, int M, int n_colloc, double gamma, int rhs_id,
                           const double* rhs_params_host, const double* rhs_values, int solver_id,
                           double* W, int32_t* status, void* stream, float* kernel_ms_host) {
  if (!kernel_ms_host) return fail(LSSVR_ERR_NULL, "kernel_ms_host must be non-NULL");
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (ne == 0) return fail(LSSVR_ERR_SIZE, "nothing to profile: ne = 0");
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, true, "ne");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  if (solver_id != LSSVR_SOLVER_PRIMAL && solver_id != LSSVR_SOLVER_DUAL &&
      solver_id != LSSVR_SOLVER_PRIMAL_WAVE && solver_id != LSSVR_SOLVER_PRIMAL_MOMENT)
    return fail(LSSVR_ERR_SOLVER, "unknown solver_id %d", solver_id);
  return dispatch_timed(a, solver_id, reinterpret_cast<hipStream_t>(stream), nullptr, 0, kernel_ms_host);
}

// lssvr_step's arguments, validated and turned into the kernels' argument blocks: what a plan keeps
struct lssvr_step_plan {
  lssvr::EnhanceArgs a;
  lssvr::P1Args p;
};

static int bind_step(lssvr_step_plan& b, const double* x, const double* u, int64_t ne, int64_t elem_offset,
                     int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
                     int M, int n_colloc, double gamma, const double* rhs_params_host, int nquad,
                     double* diag, double* off, double* load, double* W, int32_t* status,
                     int32_t* fail_count) {
  lssvr::EnhanceArgs& a = b.a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!diag || !off || !load) return fail(LSSVR_ERR_NULL, "diag, off, load must be non-NULL");
  if (!rhs_params_host) return fail(LSSVR_ERR_RHS, "rhs_params = {amp, omega} required");
  if (nquad < 1 || nquad > 5) return fail(LSSVR_ERR_QUAD, "nquad = %d outside [1,5]", nquad);
  if (n_colloc < M - 2)
    return fail(LSSVR_ERR_SOLVER, "lssvr_step: n_colloc = %d < M-2 = %d: the primal normal equations are "
                                  "rank deficient; use lssvr_p1_assemble + lssvr_enhance (dual solver)",
                n_colloc, M - 2);
  a.rhs_id = LSSVR_RHS_SIN;
  a.rhs_amp = rhs_params_host[0];
  a.rhs_omega = rhs_params_host[1];
  a.status = status;
  a.fail_count = fail_count;
  lssvr::P1Args& p = b.p;
  p = lssvr::P1Args{};
  p.x = x;
  p.ne = ne;
  p.nquad = nquad;
  p.rhs_id = LSSVR_RHS_SIN;
  p.rhs_amp = a.rhs_amp;
  p.rhs_omega = a.rhs_omega;
  p.diag = diag;
  p.off = off;
  p.load = load;
  return LSSVR_OK;
}

static int run_step(const lssvr_step_plan& b, hipStream_t s) {
  const lssvr::EnhanceArgs& a = b.a;
  const lssvr::P1Args& p = b.p;
  // (near-square regime, a.refine > 0: the refinement lives in a kernel of its own -- two launches)
  if (a.M <= lssvr::kSmallMaxM && a.refine == 0) return check_launch(lssvr::step_small(a, p, s), "step_small");
  int rc = check_launch(lssvr::p1_assemble(p, s), "p1_assemble");
  if (rc != LSSVR_OK) return rc;
  if (a.M <= lssvr::kSmallMaxM) return check_launch(lssvr::enhance_small(a, s), "enhance_small(refine)");
  // large degree: the enhancement is long enough that a fused launch buys nothing
  return check_launch(lssvr::enhance_large(a, s), "enhance_large");
}

int lssvr_step(const double* x, const double* u, int64_t ne, int64_t elem_offset,
               int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
               int M, int n_colloc, double gamma, const double* rhs_params_host, int nquad,
               double* diag, double* off, double* load, double* W, int32_t* status,
               int32_t* fail_count, void* stream) {
  lssvr_step_plan b;
  const int rc = bind_step(b, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right, M, n_colloc,
                           gamma, rhs_params_host, nquad, diag, off, load, W, status, fail_count);
  if (rc != LSSVR_OK) return rc;
  return run_step(b, reinterpret_cast<hipStream_t>(stream));
}

int lssvr_step_plan_create(lssvr_step_plan** plan, const double* x, const double* u, int64_t ne,
                           int64_t elem_offset, int64_t ne_global, double gxmin, double gxmax,
                           double bc_left, double bc_right, int M, int n_colloc, double gamma,
                           const double* rhs_params_host, int nquad, double* diag, double* off,
                           double* load, double* W, int32_t* status, int32_t* fail_count) {
  if (!plan) return fail(LSSVR_ERR_NULL, "plan must be non-NULL");
  *plan = nullptr;
  lssvr_step_plan* b = new (std::nothrow) lssvr_step_plan;
  if (!b) return fail(LSSVR_ERR_LAUNCH, "out of host memory");
  const int rc = bind_step(*b, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right, M, n_colloc,
                           gamma, rhs_params_host, nquad, diag, off, load, W, status, fail_count);
  if (rc != LSSVR_OK) {
    delete b;
    return rc;
  }
  *plan = b;
  return LSSVR_OK;
}

int lssvr_step_plan_launch(const lssvr_step_plan* plan, void* stream) {
  if (!plan) return fail(LSSVR_ERR_NULL, "plan must be non-NULL");
  return run_step(*plan, reinterpret_cast<hipStream_t>(stream));
}

int lssvr_step_plan_destroy(lssvr_step_plan* plan) {
  delete plan;
  return LSSVR_OK;
}

int lssvr_enhance_varcoef(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                          int64_t ne_global, double gxmin, double gxmax, double bc_left,
                          double bc_right, int M, int n_colloc, double gamma,
                          const double* a_values, const double* da_values,
                          const double* rhs_values, double* W, int32_t* status,
                          int32_t* fail_count, void* stream) {
  return lssvr_enhance_varcoef_ws(x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right, M,
                                  n_colloc, gamma, a_values, da_values, rhs_values, LSSVR_TABLE_ELEMENT_MAJOR,
                                  W, status, fail_count, nullptr, 0, stream, nullptr);
}

int lssvr_step_varcoef(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                       int64_t ne_global, double gxmin, double gxmax, double bc_left, double bc_right,
                       int M, int n_colloc, double gamma, const double* a_values,
                       const double* da_values, const double* rhs_values, int table_layout, int nquad,
                       const double* rhs_quad, const double* a_quad, double* diag, double* off,
                       double* load, double* W, int32_t* status, int32_t* fail_count, void* stream) {
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!a_values || !da_values || !rhs_values)
    return fail(LSSVR_ERR_NULL, "a_values, da_values and rhs_values must be non-NULL");
  if (!rhs_quad || !a_quad) return fail(LSSVR_ERR_NULL, "rhs_quad and a_quad must be non-NULL");
  if (!diag || !off || !load) return fail(LSSVR_ERR_NULL, "diag, off, load must be non-NULL");
  if (nquad < 1 || nquad > 5) return fail(LSSVR_ERR_QUAD, "nquad = %d outside [1,5]", nquad);
  if (table_layout != LSSVR_TABLE_ELEMENT_MAJOR && table_layout != LSSVR_TABLE_POINT_MAJOR)
    return fail(LSSVR_ERR_SIZE, "unknown table_layout %d", table_layout);
  if (n_colloc < M - 2)
    return fail(LSSVR_ERR_SOLVER, "lssvr_step_varcoef: n_colloc = %d < M-2 = %d: use lssvr_p1_assemble + "
                                  "lssvr_enhance_varcoef (dual solver)", n_colloc, M - 2);
  a.rhs_id = LSSVR_RHS_ARRAY;
  a.rhs_values = rhs_values;
  a.a_values = a_values;
  a.da_values = da_values;
  if (table_layout == LSSVR_TABLE_POINT_MAJOR) {
    a.tab_es = 1;
    a.tab_ps = ne;
  }
  a.status = status;
  a.fail_count = fail_count;
  lssvr::P1Args p{};
  p.x = x;
  p.ne = ne;
  p.nquad = nquad;
  p.rhs_id = LSSVR_RHS_ARRAY;
  p.rhs_quad = rhs_quad;
  p.a_quad = a_quad;
  p.diag = diag;
  p.off = off;
  p.load = load;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (M <= lssvr::kStepVarcoefFusedMaxM) return check_launch(lssvr::step_small_vc(a, p, s), "step_small_vc");
  rc = check_launch(lssvr::p1_assemble(p, s), "p1_assemble");
  if (rc != LSSVR_OK) return rc;
  return enhance_dispatch(a, LSSVR_SOLVER_PRIMAL, s, nullptr);
}

int64_t lssvr_enhance_varcoef_work_bytes(int64_t ne, int M, int n_colloc) {
  (void)ne; (void)M; (void)n_colloc;
  return 0;
}

int lssvr_enhance_varcoef_ws(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                             int64_t ne_global, double gxmin, double gxmax, double bc_left,
                             double bc_right, int M, int n_colloc, double gamma,
                             const double* a_values, const double* da_values,
                             const double* rhs_values, int table_layout, double* W, int32_t* status,
                             int32_t* fail_count, void* work, int64_t work_bytes, void* stream,
                             float* kernel_ms_host) {
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (ne > 0 && (!a_values || !da_values || !rhs_values))
    return fail(LSSVR_ERR_NULL, "a_values, da_values and rhs_values must be non-NULL");
  if (table_layout != LSSVR_TABLE_ELEMENT_MAJOR && table_layout != LSSVR_TABLE_POINT_MAJOR)
    return fail(LSSVR_ERR_SIZE, "unknown table_layout %d", table_layout);
  a.rhs_id = LSSVR_RHS_ARRAY;
  a.rhs_values = rhs_values;
  a.a_values = a_values;
  a.da_values = da_values;
  if (table_layout == LSSVR_TABLE_POINT_MAJOR) {
    a.tab_es = 1;
    a.tab_ps = ne > 0 ? ne : 1;
  }
  a.status = status;
  a.fail_count = fail_count;
  if (work_bytes < 0 || (work_bytes > 0 && !work)) return fail(LSSVR_ERR_NULL, "work / work_bytes inconsistent");
  const int64_t need = lssvr_enhance_varcoef_work_bytes(ne, M, n_colloc);
  if (work && work_bytes < need)
    return fail(LSSVR_ERR_SIZE, "work holds %lld bytes, lssvr_enhance_varcoef_work_bytes() = %lld",
                (long long)work_bytes, (long long)need);
  if (ne == 0) return LSSVR_OK;
  // (n_colloc < M-2: rank-deficient primal normal equations -> the dual Gram solver)
  return dispatch_timed(a, LSSVR_SOLVER_PRIMAL, reinterpret_cast<hipStream_t>(stream), work, work_bytes,
                        kernel_ms_host);
}

int lssvr_enhance_varcoef_ws_sequence(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                                      int64_t ne_global, double gxmin, double gxmax, double bc_left,
                                      double bc_right, int M, int n_colloc, double gamma,
                                      const double* a_values, const double* da_values,
                                      const double* rhs_values, int table_layout, double* W, int32_t* status,
                                      int32_t* fail_count, void* work, int64_t work_bytes, void* stream,
                                      int repeats, float* kernel_ms_host) {
  if (!kernel_ms_host) return fail(LSSVR_ERR_NULL, "kernel_ms_host must be non-NULL (float[repeats])");
  if (repeats < 1 || repeats > 100000) return fail(LSSVR_ERR_SIZE, "repeats = %d outside [1, 100000]", repeats);
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "nothing to profile: ne = %lld", (long long)ne);
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (!a_values || !da_values || !rhs_values)
    return fail(LSSVR_ERR_NULL, "a_values, da_values and rhs_values must be non-NULL");
  if (table_layout != LSSVR_TABLE_ELEMENT_MAJOR && table_layout != LSSVR_TABLE_POINT_MAJOR)
    return fail(LSSVR_ERR_SIZE, "unknown table_layout %d", table_layout);
  a.rhs_id = LSSVR_RHS_ARRAY;
  a.rhs_values = rhs_values;
  a.a_values = a_values;
  a.da_values = da_values;
  if (table_layout == LSSVR_TABLE_POINT_MAJOR) {
    a.tab_es = 1;
    a.tab_ps = ne;
  }
  a.status = status;
  a.fail_count = fail_count;
  if (work_bytes < 0 || (work_bytes > 0 && !work)) return fail(LSSVR_ERR_NULL, "work / work_bytes inconsistent");
  const int64_t need = lssvr_enhance_varcoef_work_bytes(ne, M, n_colloc);
  if (work && work_bytes < need)
    return fail(LSSVR_ERR_SIZE, "work holds %lld bytes, lssvr_enhance_varcoef_work_bytes() = %lld",
                (long long)work_bytes, (long long)need);
  return dispatch_sequence(a, LSSVR_SOLVER_PRIMAL, reinterpret_cast<hipStream_t>(stream), work, work_bytes,
                           repeats, kernel_ms_host);
}

int lssvr_enhance_subset(const double* x, const double* u, int64_t ne_mesh,
                         const int64_t* elem_ids, int64_t nsub, int64_t elem_offset,
                         int64_t ne_global, double gxmin, double gxmax, double bc_left,
                         double bc_right, int M, int n_colloc, double gamma,
                         const double* gamma_values, int rhs_id, const double* rhs_params_host,
                         const double* rhs_values, double* W, int64_t ldw, int32_t* status,
                         int32_t* fail_count, void* stream) {
  return lssvr_enhance_subset_ws(x, u, ne_mesh, elem_ids, nsub, elem_offset, ne_global, gxmin, gxmax, bc_left,
                                 bc_right, M, n_colloc, gamma, gamma_values, rhs_id, rhs_params_host, rhs_values,
                                 W, ldw, status, fail_count, nullptr, 0, stream);
}

int lssvr_enhance_subset_ws(const double* x, const double* u, int64_t ne_mesh,
                            const int64_t* elem_ids, int64_t nsub, int64_t elem_offset,
                            int64_t ne_global, double gxmin, double gxmax, double bc_left,
                            double bc_right, int M, int n_colloc, double gamma,
                            const double* gamma_values, int rhs_id, const double* rhs_params_host,
                            const double* rhs_values, double* W, int64_t ldw, int32_t* status,
                            int32_t* fail_count, void* work, int64_t work_bytes, void* stream) {
  if (ne_mesh < 0 || nsub < 0) return fail(LSSVR_ERR_SIZE, "ne_mesh / nsub < 0");
  if (!elem_ids && nsub != ne_mesh)
    return fail(LSSVR_ERR_SIZE, "elem_ids == NULL means every element: nsub must equal ne_mesh");
  if (elem_ids && nsub > ne_mesh) return fail(LSSVR_ERR_SIZE, "nsub = %lld > ne_mesh = %lld",
                                              (long long)nsub, (long long)ne_mesh);
  if (ldw != 0 && ldw < M) return fail(LSSVR_ERR_SIZE, "ldw = %lld < M = %d", (long long)ldw, M);
  if (ne_global < elem_offset + ne_mesh)
    return fail(LSSVR_ERR_SIZE, "shard [%lld, %lld) does not fit ne_global = %lld",
                (long long)elem_offset, (long long)(elem_offset + ne_mesh), (long long)ne_global);
  lssvr::EnhanceArgs a;
  // (the shard check of fill_enhance_args is on the subset size here: done above for the mesh)
  int rc = fill_enhance_args(a, x, u, nsub, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, gamma_values ? 1.0 : gamma, W);
  if (rc != LSSVR_OK) return rc;
  if (n_colloc < M - 2)
    return fail(LSSVR_ERR_SOLVER, "lssvr_enhance_subset: n_colloc < M-2 needs the dual solver, "
                                  "which has no subset form");
  a.gamma = gamma;
  a.inv_gamma = 1.0 / gamma;
  a.elem_ids = elem_ids;
  a.ne_mesh = ne_mesh;
  a.gamma_values = gamma_values;
  a.ldw = ldw;
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, nsub > 0, "nsub");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  a.fail_count = fail_count;
  if (work_bytes < 0 || (work_bytes > 0 && !work)) return fail(LSSVR_ERR_NULL, "work / work_bytes inconsistent");
  const int64_t need = lssvr_enhance_work_bytes(nsub, M, n_colloc, LSSVR_SOLVER_PRIMAL);
  if (work && work_bytes < need)
    return fail(LSSVR_ERR_SIZE, "work holds %lld bytes, lssvr_enhance_work_bytes(nsub = %lld, %d, %d, 0) = %lld",
                (long long)work_bytes, (long long)nsub, M, n_colloc, (long long)need);
  if (nsub == 0) return LSSVR_OK;
  // (M <= 22: the lane kernel; above, with a workspace: moments + solve kernels, without: the MFMA kernel)
  return enhance_dispatch(a, LSSVR_SOLVER_PRIMAL, reinterpret_cast<hipStream_t>(stream), nullptr, work, work_bytes);
}

int lssvr_enhance_shared(const double* x, const double* u, int64_t ne, int64_t elem_offset,
                         int64_t ne_global, double gxmin, double gxmax, double bc_left,
                         double bc_right, int M, int n_colloc, int rhs_id,
                         const double* rhs_params_host, const double* rhs_values, const double* op,
                         double* W, int32_t* status, int32_t* fail_count, void* stream,
                         float* kernel_ms_host) {
  lssvr::EnhanceArgs a;
  int rc = fill_enhance_args(a, x, u, ne, elem_offset, ne_global, gxmin, gxmax, bc_left, bc_right,
                             M, n_colloc, 1.0, W);
  if (rc != LSSVR_OK) return rc;
  if (M > lssvr::kSharedMaxM)
    return fail(LSSVR_ERR_DEGREE, "shared-operator path: M = %d > %d", M, lssvr::kSharedMaxM);
  if (ne > 0 && !op) return fail(LSSVR_ERR_NULL, "op[(n_colloc+2)*M] must be non-NULL");
  rc = set_rhs(a, rhs_id, rhs_params_host, rhs_values, ne > 0, "ne");
  if (rc != LSSVR_OK) return rc;
  a.status = status;
  a.fail_count = fail_count;
  if (ne == 0) return LSSVR_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!kernel_ms_host) return check_launch(lssvr::enhance_shared(a, op, s), "enhance_shared");
  lssvr::LaunchOpts o;
  if (hipEventCreate(&o.start) != hipSuccess || hipEventCreate(&o.stop) != hipSuccess)
    return fail(LSSVR_ERR_LAUNCH, "hipEventCreate failed");
  rc = check_launch(lssvr::enhance_shared(a, op, s, &o), "enhance_shared(profiled)");
  if (rc == LSSVR_OK) {
    if (hipEventSynchronize(o.stop) != hipSuccess ||
        hipEventElapsedTime(kernel_ms_host, o.start, o.stop) != hipSuccess)
      rc = fail(LSSVR_ERR_LAUNCH, "event timing failed");
  }
  (void)hipEventDestroy(o.start);
  (void)hipEventDestroy(o.stop);
  return rc;
}

int lssvr_colloc_points(const double* x, int64_t ne, int n_colloc, double* xc, void* stream) {
  if (ne < 0) return fail(LSSVR_ERR_SIZE, "ne < 0");
  if (n_colloc < 2) return fail(LSSVR_ERR_SIZE, "n_colloc < 2");
  if (ne > 0 && (!x || !xc)) return fail(LSSVR_ERR_NULL, "x and xc must be non-NULL");
  return check_launch(lssvr::colloc_points(x, ne, n_colloc, xc, reinterpret_cast<hipStream_t>(stream)),
                      "colloc_points");
}

int lssvr_colloc_points_pm(const double* x, int64_t ne, int n_colloc, double* xc, void* stream) {
  if (ne < 0) return fail(LSSVR_ERR_SIZE, "ne < 0");
  if (n_colloc < 2) return fail(LSSVR_ERR_SIZE, "n_colloc < 2");
  if (ne > 0 && (!x || !xc)) return fail(LSSVR_ERR_NULL, "x and xc must be non-NULL");
  return check_launch(lssvr::colloc_points(x, ne, n_colloc, xc, reinterpret_cast<hipStream_t>(stream), true),
                      "colloc_points(point-major)");
}

int lssvr_p1_assemble(const double* x, int64_t ne, int nquad, int rhs_id,
                      const double* rhs_params_host, const double* rhs_quad, const double* a_quad,
                      double* diag, double* off, double* load, double* kloc, double* floc,
                      void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!x || !diag || !off || !load) return fail(LSSVR_ERR_NULL, "x, diag, off, load must be non-NULL");
  if (nquad < 1 || nquad > 5) return fail(LSSVR_ERR_QUAD, "nquad = %d outside [1,5]", nquad);
  lssvr::P1Args a{};
  a.x = x;
  a.ne = ne;
  a.nquad = nquad;
  a.rhs_id = rhs_id;
  if (rhs_id == LSSVR_RHS_SIN) {
    if (!rhs_params_host) return fail(LSSVR_ERR_RHS, "LSSVR_RHS_SIN needs rhs_params = {amp, omega}");
    a.rhs_amp = rhs_params_host[0];
    a.rhs_omega = rhs_params_host[1];
  } else if (rhs_id == LSSVR_RHS_ARRAY) {
    if (!rhs_quad) return fail(LSSVR_ERR_RHS, "LSSVR_RHS_ARRAY needs rhs_quad[ne*nquad]");
    a.rhs_quad = rhs_quad;
  } else {
    return fail(LSSVR_ERR_RHS, "unknown rhs_id %d", rhs_id);
  }
  a.a_quad = a_quad;
  a.diag = diag;
  a.off = off;
  a.load = load;
  a.kloc = kloc;
  a.floc = floc;
  return check_launch(lssvr::p1_assemble(a, reinterpret_cast<hipStream_t>(stream)), "p1_assemble");
}

int lssvr_quad_points(const double* x, int64_t ne, int nquad, double* xq, void* stream) {
  if (ne < 0) return fail(LSSVR_ERR_SIZE, "ne < 0");
  if (nquad < 1 || nquad > 5) return fail(LSSVR_ERR_QUAD, "nquad = %d outside [1,5]", nquad);
  if (ne > 0 && (!x || !xq)) return fail(LSSVR_ERR_NULL, "x and xq must be non-NULL");
  return check_launch(lssvr::quad_points(x, ne, nquad, xq, reinterpret_cast<hipStream_t>(stream)),
                      "quad_points");
}

int64_t lssvr_tridiag_work_bytes(int64_t ne) { return lssvr::tridiag_work_bytes(ne); }

int lssvr_tridiag_dirichlet_solve(const double* diag, const double* off, const double* load,
                                  int64_t ne, double u0, double u1, double* u, void* work,
                                  void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!diag || !off || !load || !u || !work)
    return fail(LSSVR_ERR_NULL, "diag, off, load, u, work must be non-NULL");
  return check_launch(lssvr::tridiag_dirichlet_solve(diag, off, load, ne, u0, u1, u, work,
                                                     reinterpret_cast<hipStream_t>(stream)),
                      "tridiag_dirichlet_solve");
}

int64_t lssvr_p1_flux_work_bytes(int64_t ne) { return lssvr::flux_work_bytes(ne); }

int lssvr_p1_flux_solve(const double* kloc, const double* load, int64_t ne, double u0, double u1,
                        double* u, void* work, void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!kloc || !load || !u || !work) return fail(LSSVR_ERR_NULL, "kloc, load, u, work must be non-NULL");
  return check_launch(lssvr::flux_dirichlet_solve(kloc, load, ne, u0, u1, u, work,
                                                  reinterpret_cast<hipStream_t>(stream)),
                      "flux_dirichlet_solve");
}

int lssvr_p1_flux_aggregate(const double* kloc, const double* load, int64_t ne, int first_global,
                            void* work, double* agg3, void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!kloc || !load || !work || !agg3) return fail(LSSVR_ERR_NULL, "kloc, load, work, agg3 must be non-NULL");
  return check_launch(lssvr::flux_aggregate(kloc, load, ne, first_global != 0, work, agg3,
                                            reinterpret_cast<hipStream_t>(stream)),
                      "flux_aggregate");
}

int lssvr_p1_flux_finish(const double* kloc, const double* load, int64_t ne, int first_global,
                         int last_global, const void* work, const double* prefix3,
                         const double* grand3, double u0, double u1, double* u, void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (!kloc || !load || !work || !u) return fail(LSSVR_ERR_NULL, "kloc, load, work, u must be non-NULL");
  return check_launch(lssvr::flux_finish(kloc, load, ne, first_global != 0, last_global != 0, work,
                                         prefix3, grand3, u0, u1, u,
                                         reinterpret_cast<hipStream_t>(stream)),
                      "flux_finish");
}

int lssvr_eval(const double* x, const double* W, int64_t ne, int M, const double* xq, int64_t P,
               double* uq, int64_t* elem, void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (P < 0) return fail(LSSVR_ERR_SIZE, "P < 0");
  if (M < 1) return fail(LSSVR_ERR_DEGREE, "M = %d < 1", M);
  if (!x || !W || (P > 0 && (!xq || !uq))) return fail(LSSVR_ERR_NULL, "x, W, xq, uq must be non-NULL");
  return check_launch(lssvr::eval_points(x, W, ne, M, xq, P, uq, elem,
                                         reinterpret_cast<hipStream_t>(stream)),
                      "eval_points");
}

int lssvr_eval_error(const double* x, const double* W, int64_t ne, int M, const double* xq,
                     int64_t P, const double* exact_params_host, double* out3, void* stream) {
  if (ne < 1) return fail(LSSVR_ERR_SIZE, "ne = %lld < 1", (long long)ne);
  if (P < 0) return fail(LSSVR_ERR_SIZE, "P < 0");
  if (M < 1) return fail(LSSVR_ERR_DEGREE, "M = %d < 1", M);
  if (!x || !W || !out3 || !exact_params_host || (P > 0 && !xq))
    return fail(LSSVR_ERR_NULL, "x, W, xq, exact_params, out3 must be non-NULL");
  return check_launch(lssvr::eval_error(x, W, ne, M, xq, P, exact_params_host[0],
                                        exact_params_host[1], out3,
                                        reinterpret_cast<hipStream_t>(stream)),
                      "eval_error");
}

int lssvr_stream_probe(const double* src, double* dst, int64_t n, void* stream) {
  if (!src || !dst) return fail(LSSVR_ERR_NULL, "src and dst must be non-NULL");
  if (n < 1) return fail(LSSVR_ERR_SIZE, "n must be >= 1");
  return check_launch(lssvr::stream_probe(src, dst, n, reinterpret_cast<hipStream_t>(stream)),
                      "stream_probe");
}

int lssvr_row_chunk_probe(const double* src, double* dst, int64_t nrows, int rowlen, int chunk, void* stream) {
  if (!src || !dst) return fail(LSSVR_ERR_NULL, "src and dst must be non-NULL");
  if (nrows < 1 || rowlen < 1) return fail(LSSVR_ERR_SIZE, "nrows and rowlen must be >= 1");
  if (chunk != 8 && chunk != 16) return fail(LSSVR_ERR_SIZE, "chunk must be 8 or 16");
  return check_launch(lssvr::row_chunk_probe(src, dst, nrows, rowlen, chunk, reinterpret_cast<hipStream_t>(stream)),
                      "row_chunk_probe");
}

int lssvr_fp64_probe(double* out, int blocks, int iters, int use_mfma, void* stream) {
  if (!out) return fail(LSSVR_ERR_NULL, "out must be non-NULL");
  if (blocks < 1 || iters < 1) return fail(LSSVR_ERR_SIZE, "blocks and iters must be >= 1");
  return check_launch(lssvr::fp64_probe(out, blocks, iters, use_mfma,
                                        reinterpret_cast<hipStream_t>(stream)),
                      "fp64_probe");
}

}  // extern "C"
