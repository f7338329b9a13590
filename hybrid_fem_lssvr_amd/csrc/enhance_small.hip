// Dispatch of the lane-per-element kernels (enhance_small_impl.hpp) over M = 2 .. 22; the
// instantiations live in enhance_small_{a,b,c,d}.hip.
#include "lssvr_kernels.hpp"
#include "lssvr_p1.hpp"

namespace lssvr {

#define LSSVR_DECLARE_SMALL_RANGE(NAME)                                                          \
  hipError_t enhance_small_##NAME(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o);     \
  hipError_t step_small_##NAME(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,         \
                               hipStream_t s, const LaunchOpts* o);                              \
  hipError_t step_small_vc_##NAME(const EnhanceArgs& e, const P1Args& a, const QuadRule& q,      \
                                  hipStream_t s, const LaunchOpts* o);
LSSVR_DECLARE_SMALL_RANGE(a)   // M = 2..10
LSSVR_DECLARE_SMALL_RANGE(b)   // M = 11..15
LSSVR_DECLARE_SMALL_RANGE(c)   // M = 16..19
LSSVR_DECLARE_SMALL_RANGE(d)   // M = 20..22

hipError_t enhance_small(const EnhanceArgs& a, hipStream_t s, const LaunchOpts* o) {
  if (a.M <= 10) return enhance_small_a(a, s, o);
  if (a.M <= 15) return enhance_small_b(a, s, o);
  if (a.M <= 19) return enhance_small_c(a, s, o);
  return enhance_small_d(a, s, o);
}

hipError_t step_small(const EnhanceArgs& e, const P1Args& a, hipStream_t s, const LaunchOpts* o) {
  QuadRule q;
  if (!quad_rule(a.nquad, q)) return hipErrorInvalidValue;
  if (e.M <= 10) return step_small_a(e, a, q, s, o);
  if (e.M <= 15) return step_small_b(e, a, q, s, o);
  if (e.M <= 19) return step_small_c(e, a, q, s, o);
  return step_small_d(e, a, q, s, o);
}

// variable-coefficient step in one launch (M <= 12; hipErrorInvalidValue above: the caller issues two)
hipError_t step_small_vc(const EnhanceArgs& e, const P1Args& a, hipStream_t s, const LaunchOpts* o) {
  QuadRule q;
  if (!quad_rule(a.nquad, q)) return hipErrorInvalidValue;
  if (e.M <= 10) return step_small_vc_a(e, a, q, s, o);
  if (e.M <= 15) return step_small_vc_b(e, a, q, s, o);
  return hipErrorInvalidValue;
}

}  // namespace lssvr
