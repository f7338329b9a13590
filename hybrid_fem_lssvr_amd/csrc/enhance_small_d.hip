// Lane-per-element kernels, M = 20 .. 22 (see enhance_small_impl.hpp).
#include "enhance_small_impl.hpp"

namespace lssvr {
#define LSSVR_RANGE_D(X) X(20) X(21) X(22)
LSSVR_DEFINE_SMALL_RANGE(d, LSSVR_RANGE_D)
}  // namespace lssvr
