// Pieces of the restructured cloth adjoint (see cloth_fast.hip), shared by the one-workgroup kernel (cloth_fast.hip) and
// the several-workgroups-per-env kernel (cloth_cluster_bwd.hip).  Including files are compiled with -ffp-contract=fast,
// except inside grip_own, whose squared distances must carry the forward's bits.
#pragma once
#include "cloth_common.h"

namespace ud {

__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ void macro_action_f(const float* a8, float* act) {  // cloth_simulator.py:168-169
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int c = 0; c < 3; ++c) act[g * 4 + c] = clipf(a8[g * 4 + c], -2.0f, 2.0f) * (1.0f / 50.0f);   // "/ 50." under jit = * (1 / 50) (pinned by the demos)
    act[g * 4 + 3] = a8[g * 4 + 3];
  }
}

// grippers, own-particle part only (:198-226): masks and displaced positions.  No FMA contraction in here: the
// squared distances must carry the same bits as the forward that wrote the checkpoints (cloth_v2.hip) so that the
// discrete grasp sets of the adjoint are the forward's.  thr0/thr1 = grasp_thr(radius), see cloth_common.h.
__device__ __forceinline__ void grip_own(const float* x, const float* ps, const float* act, float thr0, float thr1, bool& m0,
                                         bool& m1, float* x2) {
#pragma clang fp contract(off)
  float d0 = x[0] - ps[0], d1 = x[1] - ps[1], d2 = x[2] - ps[2];
  m0 = (d0 * d0 + d1 * d1 + d2 * d2) <= thr0;
  float x1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) x1[a] = m0 ? x[a] + act[a] * (1.f - act[3]) : x[a];
  d0 = x1[0] - ps[4]; d1 = x1[1] - ps[5]; d2 = x1[2] - ps[6];
  m1 = (d0 * d0 + d1 * d1 + d2 * d2) <= thr1;
#pragma unroll
  for (int a = 0; a < 3; ++a) x2[a] = m1 ? x1[a] + act[4 + a] * (1.f - act[7]) : x1[a];
}

// The adjoint works on link PAIRS: pair p = (straight link p, diagonal link p+4), one float2 per quantity, so that
// the per-link arithmetic is v_pk_{add,mul,fma}_f32 (two links per instruction).  Positions and force cotangents
// are staged in LDS as SoA planes with a compile-time stride, so a pair is two ds_read_b32 with immediate plane
// offsets landing in adjacent registers -- no register shuffling to build the operands.
typedef float f2 __attribute__((ext_vector_type(2)));

struct PairInter {
  float F1, cF, muF, xV, yV, isV, tf;   // friction block
  float S0, S1, S2;                     // sum_l w_l r_l = spring force / k (the stiffness gradient is gF . S)
  f2 r0[4], r1[4], r2[4];               // link vectors
  f2 w[4];                              // 1/L0 - 1/|r|
  f2 c2k[4];                            // k / |r|^3, or 0 where clip(|r|^2, 1e-12) is active
};

template <int STRIDE>   // LDS plane stride in floats; odd on purpose (see STRIDE in cloth_fast.hip)
__device__ __forceinline__ void force_pairs(const ClothConst& c, const int* nbs, const float* Xs, float k, f2 iL2, float mu,
                                            const float* x, const float* v, float* v3, PairInter* in) {
  f2 F0 = {0.f, 0.f}, F1 = {0.f, 0.f}, F2 = {0.f, 0.f};
  f2 q0[4], q1[4], q2[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {   // all 24 LDS reads in flight before the first use
    const int ja = nbs[p], jb = nbs[p + 4];
    q0[p] = f2{Xs[ja], Xs[jb]};
    q1[p] = f2{Xs[STRIDE + ja], Xs[STRIDE + jb]};
    q2[p] = f2{Xs[2 * STRIDE + ja], Xs[2 * STRIDE + jb]};
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f2 r0 = q0[p] - x[0], r1 = q1[p] - x[1], r2 = q2[p] - x[2];
    const f2 s2 = r0 * r0 + r1 * r1 + r2 * r2;
    const f2 inv = {rsq(fmaxf(s2.x, 1e-12f)), rsq(fmaxf(s2.y, 1e-12f))};
    const f2 w = iL2 - inv;
    F0 += w * r0; F1 += w * r1; F2 += w * r2;
    const f2 c3 = (k * inv) * (inv * inv);
    in->r0[p] = r0; in->r1[p] = r1; in->r2[p] = r2; in->w[p] = w;
    in->c2k[p] = f2{s2.x > 1e-12f ? c3.x : 0.f, s2.y > 1e-12f ? c3.y : 0.f};
  }
  const float S0 = F0.x + F0.y, S1 = F1.x + F1.y, S2 = F2.x + F2.y;
  const float Fx = k * S0, Fz = k * S2;
  const float Fy = k * S1 - c.g;                    // :278
  const float v1y = v[1] - c.gdt;                   // :259
  const bool fm = x[1] <= c.eps;                    // :281
  const float cF = fminf(Fy, 0.f);
  const float muF = -(mu * cF);                     // :282
  const float xV = v[0], yV = v[2];
  const float isV = rsq(xV * xV + yV * yV + c.eps); // :285
  const float tf = fm ? muF * isV : 0.f;            // :288-290 (sV > small_num always holds)
  const float Ax = Fx - tf * xV, Az = Fz - tf * yV;
  v3[0] = (xV + Ax * c.dt) * c.damp;                // :308-309
  v3[1] = (v1y + Fy * c.dt) * c.damp;
  v3[2] = (yV + Az * c.dt) * c.damp;
  in->S0 = S0; in->S1 = S1; in->S2 = S2;
  in->F1 = Fy; in->cF = cF; in->muF = muF; in->xV = xV; in->yV = yV; in->isV = isV; in->tf = tf;
}

// wave-wide sum that leaves the total in lane 63 (row butterflies + row_bcast15 / row_bcast31)
__device__ __forceinline__ float wave_sum_l63(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return v;
}

// 1 / (n_mask * sqrt(n2)) with norm_grad's nan_to_num semantics: a zero (or non-finite) norm zeroes the cotangent
__device__ __forceinline__ float inv_norm(float n2, float inv_n_mask) {
  const bool okv = (n2 > 0.f) && (n2 < INFINITY);
  return okv ? rsq(n2) * inv_n_mask : 0.f;
}

#define UD_NSUM 9
#define UD_RSTR 16   // floats per wave in the partial-sum buffer (9 used)

// sum over the four 16-lane rows of a wave, position by position; every row returns the total (gfx950 lane swaps)
__device__ __forceinline__ float rows_sum4(float v) {
  float a0 = v, a1 = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
  float s0 = a0 + a1, s1 = s0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s0), "+v"(s1));
  return s0 + s1;
}

}  // namespace ud
