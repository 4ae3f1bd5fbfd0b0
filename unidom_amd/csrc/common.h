// Shared host/device helpers of libunidom_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdio>

#include "../../include/unidom_hip.h"

namespace ud {

void set_error(const char* fmt, ...);

#define UD_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ud::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return UD_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// gradient factor of jnp.clip = minimum(hi, maximum(lo, x)); lax.max / lax.min split ties 0.5 / 0.5
__device__ __forceinline__ float clip_grad(float x, float lo, float hi) {
  float m = fmaxf(lo, x);
  float f1 = (x == m) ? ((lo == m) ? 0.5f : 1.0f) : 0.0f;
  float a = fminf(hi, m);
  float f2 = (m == a) ? ((hi == a) ? 0.5f : 1.0f) : 0.0f;
  return f1 * f2;
}

__device__ __forceinline__ float nan_to_num(float x) {  // jnp.nan_to_num defaults
  if (x != x) return 0.0f;
  if (fabsf(x) == INFINITY) return copysignf(FLT_MAX, x);
  return x;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Sum over the 64 lanes of a wave; every lane returns the total. DPP butterflies inside each row of 16
// (quad_perm xor-1, xor-2, row_half_mirror, row_mirror), then the four row totals through readlane.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  v += dpp_f<0x140>(v);  // row_mirror
  int iv = __builtin_bit_cast(int, v);
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

}  // namespace ud
