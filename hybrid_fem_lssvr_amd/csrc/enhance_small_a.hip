// Lane-per-element kernels, M = 2 .. 10 (see enhance_small_impl.hpp).
#include "enhance_small_impl.hpp"

namespace lssvr {
#define LSSVR_RANGE_A(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
LSSVR_DEFINE_SMALL_RANGE(a, LSSVR_RANGE_A)
}  // namespace lssvr
