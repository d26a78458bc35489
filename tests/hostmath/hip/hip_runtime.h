// TEST INFRASTRUCTURE (tests/hostmath): a stand-in for <hip/hip_runtime.h> that lets g++ compile the per-pair DEVICE math
// of csrc/gd3d_device.h for the host, so that the CPU test suite can run the kernel's own arithmetic (NaN / inf rules,
// clamp branches, tie rules) against the golden vectors without a GPU.  The hardware approximations (v_rcp / v_rsq /
// v_sqrt / v_log / v_exp, 1 ulp) are replaced by their IEEE limits with the hardware's special-value behaviour.
// Nothing in the product includes this file.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))

static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
static inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
  // v_med3_f32: with a NaN operand the hardware returns min3 (ISA: "if any input is NaN, return min3")
  if (a != a || b != b || c != c) {
    auto mn = [](float x, float y) { return (x != x) ? y : ((y != y) ? x : (x < y ? x : y)); };
    return mn(mn(a, b), c);
  }
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
}
