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

// a / d for 2^-40 <= |d| <= 2^40 and a == 0 or 2^-60 <= |a| <= 2^60: hipcc's own IEEE expansion of f32 division (AMDGPU LowerFDIV32:
// rcp, two FMAs refining it, the quotient, residual / correction twice) without v_div_scale / v_div_fmas exponent scaling and without
// v_div_fixup -- inside the range none of them changes a bit (tools/check_exact_div.hip: random, near-halfway and edge operands against
// the compiler's division on an MI355X).  The refined reciprocal depends on d alone: several numerators over one denominator share it
// (div_prep once, div_rn_prepped per numerator: 5 instructions + the zero select instead of ~11 each).  +-0 / d keeps IEEE's sign.
__device__ __forceinline__ float div_prep(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_rn_prepped(float a, float d, float rd) {
  const float q0 = a * rd;
  float e = __builtin_fmaf(-d, q0, a);
  float q = __builtin_fmaf(e, rd, q0);
  e = __builtin_fmaf(-d, q, a);
  q = __builtin_fmaf(e, rd, q);
  return (a == 0.0f) ? q0 : q;          // (+-0) * rd is the correctly signed zero; the FMA chain would return +0
}
// the same without the select: a zero numerator returns +0 whatever its sign (callers that only add the quotient to a running sum or
// subtract it from one do not see the difference).  Also checked for |a| in [2^-100, 2^100), |d| in [2^-24, 2^24) (check_exact_div 33 100 24):
// what the sequence needs is a normal quotient and residuals above the denormal floor, not these particular windows.
__device__ __forceinline__ float div_rn_prepped_nz(float a, float d, float rd) {
  float q = a * rd;
  float e = __builtin_fmaf(-d, q, a);
  q = __builtin_fmaf(e, rd, q);
  e = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(e, rd, q);
}
__device__ __forceinline__ bool div_den_inrange(float d) { return __builtin_fabsf(d) >= 0x1p-40f && __builtin_fabsf(d) <= 0x1p40f; }   // false for NaN
__device__ __forceinline__ bool div_num_inrange(float a) { return __builtin_fabsf(a) <= 0x1p60f && (__builtin_fabsf(a) >= 0x1p-60f || a == 0.0f); }
// correctly rounded a / d for ANY operands: the in-range sequence where it applies, the f64 route otherwise (53 >= 2 * 24 + 2 bits: the
// second rounding is innocuous) -- a rarely taken branch.  `rd` / `dok` = div_prep(d) / div_den_inrange(d), shared by the numerators of one d.
__device__ __forceinline__ float div_rn_shared(float a, float d, float rd, bool dok) {
  if (__builtin_expect(dok && div_num_inrange(a), 1)) return div_rn_prepped(a, d, rd);
  return (float)((double)a / (double)d);
}
__device__ __forceinline__ float div_rn(float a, float d) { return div_rn_shared(a, d, div_prep(d), div_den_inrange(d)); }
// correctly rounded sqrt for ANY argument
__device__ __forceinline__ float sqrt_rn(float x) {
  if (__builtin_expect(x >= 0x1p-96f, 1)) return sqrt_rn_inrange(x);      // (false for NaN, zero, negatives, tiny arguments)
  return (float)__builtin_sqrt((double)x);
}

// The same two functions on a pair of values (v_pk_fma_f32 for the residual corrections; IEEE per component).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 sqrt_rn_inrange2(f2 x) {
  const f2 s = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
  const i2 sb = __builtin_bit_cast(i2, s);
  const f2 sd = __builtin_bit_cast(f2, sb - 1), su = __builtin_bit_cast(f2, sb + 1);
  const f2 rd = __builtin_elementwise_fma(-sd, s, x), ru = __builtin_elementwise_fma(-su, s, x);
  f2 r;
  r.x = (rd.x <= 0.0f) ? sd.x : s.x; r.x = (ru.x > 0.0f) ? su.x : r.x;
  r.y = (rd.y <= 0.0f) ? sd.y : s.y; r.y = (ru.y > 0.0f) ? su.y : r.y;
  return r;
}

__device__ __forceinline__ f2 rcp_rn_inrange2(f2 d) {
  const f2 one = {1.0f, 1.0f};
  f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 e = __builtin_elementwise_fma(-d, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  const f2 e2 = __builtin_elementwise_fma(-d, r, one);
  const f2 q = __builtin_elementwise_fma(e2, r, r);
  const f2 e3 = __builtin_elementwise_fma(-d, q, one);
  return __builtin_elementwise_fma(e3, r, q);
}

// Four pairs at once, written stage by stage: a dependent v_pk_fma_f32 -> v_pk_fma_f32 costs a wait state, and the
// four chains are independent, so breadth-first order lets each stage's instructions fill the others' latency.
__device__ __forceinline__ void rcp_sqrt_rn_inrange2x4(const f2 (&x)[4], f2 (&out)[4]) {
  const f2 one = {1.0f, 1.0f};
  f2 s[4], sd[4], su[4], rd[4], ru[4], d[4], r[4], e[4], q[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) s[p] = f2{__builtin_amdgcn_sqrtf(x[p].x), __builtin_amdgcn_sqrtf(x[p].y)};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const i2 sb = __builtin_bit_cast(i2, s[p]);
    sd[p] = __builtin_bit_cast(f2, sb - 1); su[p] = __builtin_bit_cast(f2, sb + 1);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) rd[p] = __builtin_elementwise_fma(-sd[p], s[p], x[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) ru[p] = __builtin_elementwise_fma(-su[p], s[p], x[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    d[p].x = (rd[p].x <= 0.0f) ? sd[p].x : s[p].x; d[p].x = (ru[p].x > 0.0f) ? su[p].x : d[p].x;
    d[p].y = (rd[p].y <= 0.0f) ? sd[p].y : s[p].y; d[p].y = (ru[p].y > 0.0f) ? su[p].y : d[p].y;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) r[p] = f2{__builtin_amdgcn_rcpf(d[p].x), __builtin_amdgcn_rcpf(d[p].y)};
#pragma unroll
  for (int p = 0; p < 4; ++p) e[p] = __builtin_elementwise_fma(-d[p], r[p], one);
#pragma unroll
  for (int p = 0; p < 4; ++p) r[p] = __builtin_elementwise_fma(e[p], r[p], r[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) e[p] = __builtin_elementwise_fma(-d[p], r[p], one);
#pragma unroll
  for (int p = 0; p < 4; ++p) q[p] = __builtin_elementwise_fma(e[p], r[p], r[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) e[p] = __builtin_elementwise_fma(-d[p], q[p], one);
#pragma unroll
  for (int p = 0; p < 4; ++p) out[p] = __builtin_elementwise_fma(e[p], r[p], q[p]);
}

}  // namespace ud
