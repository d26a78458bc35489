// gd3d_host_math.h — what csrc/gd3d_device.h needs from the compiler when the `_cpu` twins (csrc/gd3d_cpu.cpp) evaluate
// the per-pair math on the HOST: the function-space keywords become nothing, and the five gfx950 hardware operations the
// header names (v_rcp / v_sqrt / v_rsq / v_exp / v_log, 1 ulp each, and v_med3) become their IEEE-exact counterparts with the
// hardware's special-value behaviour.  Everything else in gd3d_device.h is the same source for both targets.
// Compiled by the ROCm toolchain's clang++ only (gd3d_device.h uses clang's ext_vector_type for its packed-fp32 pieces).
#pragma once
#include <cmath>
#include <cstdint>

#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))

namespace gd3d_host {
inline float min_skip_nan(float x, float y) { return (x != x) ? y : ((y != y) ? x : (x < y ? x : y)); }
}  // namespace gd3d_host

static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
static inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }
// v_med3_f32: the median of three; with a NaN operand the instruction returns min3 of the others (CDNA ISA), which is what
// makes half_clamp() send a NaN dim to the lower bound with a zero pass-through mask
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
  if (a != a || b != b || c != c) return gd3d_host::min_skip_nan(gd3d_host::min_skip_nan(a, b), c);
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
}
