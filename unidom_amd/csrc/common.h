// Shared host/device helpers of libunidom_hip (gfx950 only).
#pragma once
#ifdef UD_HOST_BUILD
// oracle/csrc/mpm_det_host.cpp: the plain-arithmetic device headers (mpm_device.h, mpm_det.h) compiled by the host compiler, so that
// the deterministic mode's CPU restatement IS the device source (no HIP runtime, no device intrinsics on that side)
#include <algorithm>
#include <cfloat>
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
using std::max;
using std::min;
#else
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdio>

#include "../../include/unidom_hip.h"
#endif

namespace ud {

#ifndef UD_HOST_BUILD
void set_error(const char* fmt, ...);
#endif

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

// same factor for lo < hi known at the call site: 1 inside, 0.5 on either bound, 0 outside (and for NaN)
__device__ __forceinline__ float clip_grad_lt(float x, float lo, float hi) {
  const bool in_closed = (x >= lo) && (x <= hi), in_open = (x > lo) && (x < hi);
  return in_open ? 1.0f : (in_closed ? 0.5f : 0.0f);
}

__device__ __forceinline__ float nan_to_num(float x) {  // jnp.nan_to_num defaults
  if (x != x) return 0.0f;
  if (fabsf(x) == INFINITY) return copysignf(FLT_MAX, x);
  return x;
}

#ifndef UD_HOST_BUILD   // ---- wave-level helpers: device only from here on ----
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

// Eight wave-wide sums at once (transposed butterfly): each exchange halves the number of live values, so the
// whole reduction is 18 + 4 + 5 VALU instead of 8 x 6 dependent DPP steps with hazard nops between them.
// Lane l returns the wave total of value j = 4*bit4(l) + 2*bit1(l) + bit0(l) (lanes 0-3 and 16-19 cover 0..7).
__device__ __forceinline__ float wave_sum8_t(const float (&v)[8], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  float w[4], u[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) w[p] = (b0 ? v[2 * p + 1] : v[2 * p]) + dpp_f<0xB1>(b0 ? v[2 * p] : v[2 * p + 1]);   // lane ^ 1
#pragma unroll
  for (int p = 0; p < 2; ++p) u[p] = (b1 ? w[2 * p + 1] : w[2 * p]) + dpp_f<0x4E>(b1 ? w[2 * p] : w[2 * p + 1]);   // lane ^ 2
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    u[p] += dpp_f<0x124>(u[p]);   // row_ror:4
    u[p] += dpp_f<0x128>(u[p]);   // row_ror:8 -> row totals
  }
  // gfx950 lane swaps: odd rows of u0 <-> even rows of u1, then upper half <-> lower half.  Inline asm (with the
  // VALU->permlane wait states written out) because hipcc 7.2 folds the builtin's two results into one register.
  float a0 = u[0], a1 = u[1];
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
  float s0 = a0 + a1, s1 = s0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s0), "+v"(s1));
  return s0 + s1;
}

#endif  // !UD_HOST_BUILD

}  // namespace ud
