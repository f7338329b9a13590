// Lane-per-element kernels, M = 11 .. 15 (see enhance_small_impl.hpp).
#include "enhance_small_impl.hpp"

namespace lssvr {
#define LSSVR_RANGE_B(X) X(11) X(12) X(13) X(14) X(15)
LSSVR_DEFINE_SMALL_RANGE(b, LSSVR_RANGE_B)
}  // namespace lssvr
