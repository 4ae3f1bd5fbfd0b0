// Correctly rounded f32 sqrt and reciprocal for arguments whose range is known, used by the bit-exact "v2" cloth
// forward.  They are hipcc's own IEEE expansions of sqrtf(x) and 1.0f / x (v_sqrt_f32 / v_rcp_f32 seeds plus FMA
// residual corrections) with the parts removed that only matter outside the stated range: the 2^32 pre-scaling
// of tiny sqrt arguments, v_div_scale / v_div_fmas exponent scaling, and the v_cmp_class / v_div_fixup special
// cases.  Inside the range the results are bit-identical to sqrtf / division (checked exhaustively over every
// float in range on an MI355X by tools/check_exact_math.hip) at roughly half the instructions.
#pragma once
#include <hip/hip_runtime.h>

namespace ud {

// sqrtf(x) for 2^-96 <= x <= FLT_MAX  (also right for +inf and NaN)
__device__ __forceinline__ float sqrt_rn_inrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);                       // <= 1 ulp
  const int sb = __builtin_bit_cast(int, s);
  const float sd = __builtin_bit_cast(float, sb - 1), su = __builtin_bit_cast(float, sb + 1);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  return r;
}

// 1.0f / d for 2^-64 <= d <= 2^64
__device__ __forceinline__ float rcp_rn_inrange(float d) {
  float r = __builtin_amdgcn_rcpf(d);                               // <= 1 ulp
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  const float e2 = __builtin_fmaf(-d, r, 1.0f);
  const float q = __builtin_fmaf(e2, r, r);
  const float e3 = __builtin_fmaf(-d, q, 1.0f);
  return __builtin_fmaf(e3, r, q);
}

}  // namespace ud
