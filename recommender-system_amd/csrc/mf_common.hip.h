// mf_common.hip.h -- shared constants of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mf {

constexpr int kWave = 64;

}  // namespace mf
