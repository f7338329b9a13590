// Lane-per-element kernels, M = 16 .. 19 (see enhance_small_impl.hpp).
#include "enhance_small_impl.hpp"

namespace lssvr {
#define LSSVR_RANGE_C(X) X(16) X(17) X(18) X(19)
LSSVR_DEFINE_SMALL_RANGE(c, LSSVR_RANGE_C)
}  // namespace lssvr
