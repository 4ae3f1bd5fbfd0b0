// The "v2" cloth forward arithmetic (see cloth_v2.hip), shared by the one-workgroup kernel (cloth_v2.hip) and the
// several-workgroups-per-env kernel (cloth_cluster_fwd.hip).  Every including file is compiled with -ffp-contract=off:
// only +, -, *, /, sqrt, one rounding each -- bit-identical to oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd_v2.
#pragma once
#include "cloth_common.h"
#include "exact_math.h"

namespace ud {

__device__ __forceinline__ void macro_action_f(const float* a8, float* act) {  // cloth_simulator.py:168-169
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int c = 0; c < 3; ++c) act[g * 4 + c] = clipf(a8[g * 4 + c], -2.0f, 2.0f) * (1.0f / 50.0f);   // "/ 50." under jit = * (1 / 50) (pinned by the demos)
    act[g * 4 + 3] = a8[g * 4 + 3];
  }
}

// grippers, own-particle part only (:198-226): masks and displaced positions
// thr0/thr1 = grasp_thr(radius): s <= thr is the same boolean as sqrtf(s) <= radius (cloth_common.h)
__device__ __forceinline__ void grip_own(const float* x, const float* ps, const float* act, float thr0, float thr1, bool& m0,
                                         bool& m1, float* x2) {
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

// spring + gravity + ground friction + damping in the re-associated IEEE order "v2"
// (oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd_v2): only +,-,*,/,sqrt, no FMA contraction in this file.
// Links are processed as PAIRS (straight link p, diagonal link p+4) in float2 = v_pk_{add,mul,fma}_f32, which are
// IEEE per component; the straight and the diagonal forces are summed separately (each in link order) and added
// at the end -- the order the oracle's v2 uses.  Positions sit in LDS as SoA planes with a compile-time stride so
// that a pair is two ds_read_b32 with immediate offsets into adjacent registers.
// nbs[l] = neighbour index, or the particle itself where the lattice has no neighbour.  Then r == 0 exactly and
// coef is finite, so coef * r == +-0, and F (which starts at +0 and therefore is never -0) takes it without
// changing a bit: the same result as the oracle's "skip the link" without three selects per link.
template <int STRIDE>   // LDS plane stride in floats
__device__ __forceinline__ void force_v2(const ClothConst& c, const int* nbs, const float* Xs, float k, f2 kL2, float mu,
                                         const float* x, const float* v, float isV, float* v3) {
  f2 r0[4], r1[4], r2[4], cl[4], inv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ja = nbs[p], jb = nbs[p + 4];
    r0[p] = f2{Xs[ja], Xs[jb]} - x[0];
    r1[p] = f2{Xs[STRIDE + ja], Xs[STRIDE + jb]} - x[1];
    r2[p] = f2{Xs[2 * STRIDE + ja], Xs[2 * STRIDE + jb]} - x[2];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f2 s2 = r0[p] * r0[p] + r1[p] * r1[p] + r2[p] * r2[p];
    // the clip keeps the sqrt argument in [1e-12, FLT_MAX] (an overflowed |r|^2 gives 1/len = 5e-20 instead of 0,
    // which k/L0 - k/len rounds to the same float) and len in [1e-6, 2^64]: the exact_math.h ranges
    cl[p] = f2{fminf(fmaxf(s2.x, 1e-12f), FLT_MAX), fminf(fmaxf(s2.y, 1e-12f), FLT_MAX)};
  }
  rcp_sqrt_rn_inrange2x4(cl, inv);   // 1 / sqrt, both correctly rounded
  f2 F0 = {0.f, 0.f}, F1 = {0.f, 0.f}, F2 = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f2 coef = kL2 - k * inv[p];
    F0 += coef * r0[p]; F1 += coef * r1[p]; F2 += coef * r2[p];
  }
  const float Fx = F0.x + F0.y, Fz = F2.x + F2.y;
  float Fy = F1.x + F1.y;
  Fy += -c.g;
  const float v1y = v[1] - c.gdt;
  const bool fm = x[1] <= c.eps;
  const float cF = fminf(Fy, 0.f);
  const float muF = mu * cF * -1.0f;
  const float xV = v[0], yV = v[2];
  const float tf = fm ? muF * isV : 0.f;   // isV = 1 / sqrt(xV^2 + yV^2 + eps), computed by the caller ahead of the barrier
  const float Ax = Fx - tf * xV, Az = Fz - tf * yV;
  v3[0] = (xV + Ax * c.dt) * c.damp;
  v3[1] = (v1y + Fy * c.dt) * c.damp;
  v3[2] = (yV + Az * c.dt) * c.damp;
}

}  // namespace ud
