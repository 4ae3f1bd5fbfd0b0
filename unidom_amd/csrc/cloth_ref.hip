// Mode-3 cloth forward for gfx950: the reference's LITERAL operation order (cloth_simulator.py:257-337 as written -- spring force
// k * r / len * (len - L0) / L0 per component, the whole friction block :281-306, gravity twice) with the SAME bits as
// cloth_ref_order.h::substep_fwd and the CPU restatement of that order (oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd), at about
// two thirds of the instructions.
//
// What the literal kernel spends its time on is 51 IEEE divisions and 12 IEEE square roots per particle-substep: hipcc expands each
// division into v_div_scale x2, v_rcp, five FMAs, v_div_fmas, v_div_fixup (11 instructions) and each sqrt into ~15.  Here
//   * a division is exact_math.h's div_rn_prepped_nz: the compiler's own sequence minus its exponent scaling and fix-up (5 instructions),
//     with the refined reciprocal shared by the numerators of one denominator -- the three components of a link share 1 / len, and
//     1 / L0 is one of two per-launch constants (straight / diagonal rest length);
//   * a sqrt is sqrt_rn_inrange (v_sqrt + two residual FMAs + two selects);
//   * the two grasp tests compare squared distances with the exact thresholds of cloth_common.h (same booleans as the sqrt form);
//   * a missing neighbour points at the particle itself: r = 0 exactly, the force component is -0 and adding it changes no bit
//     (F starts at +0 and can never become -0), so the per-link selects go.
// Those sequences return the correctly rounded result only inside operand windows (tools/check_exact_math.hip,
// tools/check_exact_div.hip: |a| in [2^-100, 2^100), |d| in [2^-24, 2^24), sqrt arguments >= 2^-96).  The windows are not assumed:
// every lane tracks the frexp exponents of its 24 link components (0, or 2^-36 <= |r| < 2^8 -- with the per-launch checks on k, L0 and
// small_num that puts every numerator and denominator of the spring block inside) and of the three friction numerators, and a wave
// in which ANY lane left a window repeats the substep for its 64 particles with the literal code (wave-uniform branch; a lane's
// result depends on its own arithmetic only).  A launch whose constants are outside the per-launch checks runs the literal kernel.
// So: bit-identical to the literal order for any input, fast for every input a cloth produces (positions in [0, 1], |v| <= max_v).
#include <cmath>

#include "cloth_ref_order.h"
#include "cloth_v2_force.h"

namespace ud {

constexpr int UD_REF_MAXP = 512;   // LDS plane stride (floats); this kernel serves Pp <= 512

// frexp exponents (0 for +-0; inf / NaN also give 0 and are caught by `amax`) of the values seen so far
struct RefTrack {
  int rmin, rmax;     // link components
  int amin, amax_e;   // friction numerators
  float amax, dmax;   // |numerators| and |denominators| as floats: inf shows here
  __device__ __forceinline__ void init() { rmin = 0; rmax = 0; amin = 0; amax_e = 0; amax = 0.f; dmax = 0.f; }
  __device__ __forceinline__ void link(float r0, float r1, float r2) {
    const int e0 = __builtin_amdgcn_frexp_expf(r0), e1 = __builtin_amdgcn_frexp_expf(r1), e2 = __builtin_amdgcn_frexp_expf(r2);
    rmin = min(rmin, min(e0, min(e1, e2)));
    rmax = max(rmax, max(e0, max(e1, e2)));
    amax = fmaxf(amax, fmaxf(__builtin_fabsf(r0), fmaxf(__builtin_fabsf(r1), __builtin_fabsf(r2))));
  }
  __device__ __forceinline__ void num(float a) {
    const int e = __builtin_amdgcn_frexp_expf(a);
    amin = min(amin, e); amax_e = max(amax_e, e); amax = fmaxf(amax, __builtin_fabsf(a));
  }
  __device__ __forceinline__ void den(float d) { dmax = fmaxf(dmax, __builtin_fabsf(d)); }
  // link components: 0 or 2^-36 <= |r| < 2^8 (frexp exponent -35 .. 8); friction numerators: 0 or 2^-100 <= |a| < 2^100;
  // friction denominators sqrt(. + small_num): >= sqrt(small_num) >= 2^-24 by the per-launch check, here only < 2^24
  __device__ __forceinline__ bool bad() const {
    return rmin < -35 || rmax > 8 || amin < -99 || amax_e > 100 || !(amax < 0x1p100f) || !(dmax < 0x1p24f);
  }
};

// per-launch constants of the fast path
struct RefConst { float rLs, rLd; };   // div_prep(Ls), div_prep(Ld)

// One forward substep of particle i from own x, v and the neighbours' x in the LDS planes X (force, friction, damping: v -> v3);
// the grippers and the clip / advect are the caller's (they need no neighbour).  Same operations in the same order as
// substep_fwd (cloth_ref_order.h), each with the same correctly rounded result.  Returns false when a tracked operand left its window.
__device__ __forceinline__ bool force_ref(const ClothConst& c, const RefConst& rc, const int* nbs, const float* X, float k, float mu,
                                          const float* x, const float* v, float* v3) {
  RefTrack t;
  t.init();
  const float INF = INFINITY;
  float F[3] = {0.f, 0.f, 0.f};
  float r0[8], r1[8], r2[8], len[8], rl[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const int j = nbs[l];
    r0[l] = X[j] - x[0]; r1[l] = X[UD_REF_MAXP + j] - x[1]; r2[l] = X[2 * UD_REF_MAXP + j] - x[2];
    t.link(r0[l], r1[l], r2[l]);
  }
#pragma unroll
  for (int l = 0; l < 8; ++l) {     // all eight roots and reciprocals before the first quotient: independent chains fill each other's latency
    const float s = r0[l] * r0[l] + r1[l] * r1[l] + r2[l] * r2[l];
    len[l] = sqrt_rn_inrange(clipf(s, 1e-12f, INF));
    rl[l] = div_prep(len[l]);
  }
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const float L = (l < 4) ? c.Ls : c.Ld, rL = (l < 4) ? rc.rLs : rc.rLd;
    const float dl = len[l] - L;
    // k * r / len * (len - L) / L, left to right (:267-268)
    const float f0 = div_rn_prepped_nz(div_rn_prepped_nz(k * r0[l], len[l], rl[l]) * dl, L, rL);
    const float f1 = div_rn_prepped_nz(div_rn_prepped_nz(k * r1[l], len[l], rl[l]) * dl, L, rL);
    const float f2 = div_rn_prepped_nz(div_rn_prepped_nz(k * r2[l], len[l], rl[l]) * dl, L, rL);
    F[0] += f0; F[1] += f1; F[2] += f2;
  }
  const float v1[3] = {v[0], v[1] - c.gdt, v[2]};     // :259
  F[1] += -c.g;                                       // :278
  const bool fm = x[1] <= c.eps;                      // :281
  const float cF = clipf(F[1], -INF, 0.f);
  const float muF = mu * cF * -1.0f;                  // :282
  const float xV = v1[0], yV = v1[2];
  const float sV = sqrt_rn_inrange(xV * xV + yV * yV + c.eps);
  const float dm = (fm && sV > c.eps) ? 1.f : 0.f;
  const float rsV = div_prep(sV);
  const float nx = dm * muF * xV, nz_ = dm * muF * yV;
  t.num(nx); t.num(nz_); t.den(sV);
  const float Ax = F[0] - div_rn_prepped_nz(nx, sV, rsV);
  const float Az = F[2] - div_rn_prepped_nz(nz_, sV, rsV);
  const bool st = fm && (sV <= c.eps);
  const float sF = sqrt_rn_inrange(Ax * Ax + Az * Az + c.eps);
  const float zm = (st && muF > sF) ? 1.f : 0.f;
  const float Bx = 0.f + (1.f - zm) * Ax, Bz = 0.f + (1.f - zm) * Az;
  const float nz = (st && muF <= sF) ? 1.f : 0.f;
  t.num(muF); t.den(sF);
  const float R = 1.f - div_rn_prepped_nz(muF, sF, div_prep(sF));
  const float Cx = (R * Ax) * nz + Bx * (1.f - nz);
  const float Cz = (R * Az) * nz + Bz * (1.f - nz);
  const float Ff[3] = {Cx, F[1], Cz};
#pragma unroll
  for (int a = 0; a < 3; ++a) v3[a] = (v1[a] + Ff[a] * c.dt) * c.damp;
  return !t.bad();
}

__global__ void __launch_bounds__(512) cloth_rollout_fwd_ref_kernel(ClothFwdArgs a) {
  extern __shared__ float ldsf[];  // Xs[2][3][UD_REF_MAXP], double-buffered by substep parity
  const ClothConst c = a.c;
  const int i = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const bool live = i < P;
  int nbs[8], nb[8];
  float L0[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { nb[l] = a.nbr[l * Pp + i]; nbs[l] = nb[l] >= 0 ? nb[l] : i; L0[l] = (l < 4) ? c.Ls : c.Ld; }
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = a.x[((size_t)b * P + i) * 3 + d]; v[d] = a.v[((size_t)b * P + i) * 3 + d]; }
  }
  float ps[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ps[d] = a.prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  const RefConst rc = {div_prep(c.Ls), div_prep(c.Ld)};
  // per-env constants of the fast path: 2^-8 <= |k| < 2^24 puts k r, k r / len and (k r / len)(len - L) inside the division's windows for
  // every tracked r (cloth_ref.hip header); otherwise this env runs the literal code throughout (block-uniform)
  const bool k_ok = __builtin_fabsf(k) >= 0x1p-8f && __builtin_fabsf(k) < 0x1p24f;
  GraspThr th0, th1;
  th0.init(ps[3]); th1.init(ps[7]);
  const size_t rec = cloth_rec_floats(Pp);
  float* ckb = a.ckpt ? a.ckpt + (size_t)b * cloth_env_records(T, S) * rec : nullptr;
  unsigned step = 0;
  for (int t = 0; t < T; ++t) {
    float act[8];
    macro_action_f(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      float* Xs = ldsf + (step & 1u) * (3 * UD_REF_MAXP);
      Xs[i] = x[0]; Xs[UD_REF_MAXP + i] = x[1]; Xs[2 * UD_REF_MAXP + i] = x[2];
      if (ckb) {
        float* r = ckb + (size_t)step * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
        if (i == 0) {
#pragma unroll
          for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
        }
      }
      // no neighbour needed: grasp tests, displaced position, primitive update -- between the LDS write and the barrier
      float vv[3], x2[3];
      bool m0, m1;
      grip_own(x, ps, act, th0.at(step == 0), th1.at(step == 0), m0, m1, x2);
      float po[8];
      prim_update(ps, act, po);
      __syncthreads();
      bool ok = k_ok && force_ref(c, rc, nbs, Xs, k, mu, x, v, vv);
      if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
        // a lane of this wave left an operand window (or the env's k is outside its check): the literal substep, grippers and clip
        // included, for the whole wave -- its outputs replace what the fast path computed
        float xo[3], vo[3];
        Inter in;
        substep_fwd<false>(c, i, nb, L0, Xs, k, mu, x, v, ps, act, xo, vo, &in, UD_REF_MAXP);
        if (a.grasp && live) {
          uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
          g[i] = in.m0; g[P + i] = in.m1;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) { x[d] = xo[d]; v[d] = vo[d]; }
      } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {           // grippers 0 then 1 (:313-314): v <- suction * v where grasped
          vv[d] = m0 ? act[3] * vv[d] : vv[d];
          vv[d] = m1 ? act[7] * vv[d] : vv[d];
        }
        if (a.grasp && live) {
          uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
          g[i] = m0; g[P + i] = m1;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {           // :326-329
          const float vc = clipf(vv[d], -c.max_v, c.max_v);
          x[d] = clipf(x2[d], 0.f, 1.f) + c.dt * vc;
          v[d] = vc;
        }
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = po[d];
    }
    if (live) {
      const size_t o = (((size_t)t * B + b) * P + i) * 3;
      if (a.x_list) { a.x_list[o] = x[0]; a.x_list[o + 1] = x[1]; a.x_list[o + 2] = x[2]; }
      if (a.v_list) { a.v_list[o] = v[0]; a.v_list[o + 1] = v[1]; a.v_list[o + 2] = v[2]; }
    }
    if (a.prim_list && i == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) a.prim_list[((size_t)t * B + b) * 8 + d] = ps[d];
    }
  }
  if (live) {
    const size_t o = ((size_t)b * P + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.x_out[o + d] = x[d]; a.v_out[o + d] = v[d]; }
  }
  if (i == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) a.prim_out[b * 8 + d] = ps[d];
  }
  if (ckb) {
    float* r = ckb + (size_t)T * S * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
    if (i == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
    }
  }
}

// The per-launch constants the fast path relies on (cloth_ref.hip header): rest lengths in [2^-20, 2^4], small_num >= 2^-48 (so that
// sqrt(. + small_num) >= 2^-24), dt / damp / gravity finite.  false -> the caller launches the literal kernel instead.
bool cloth_ref_fast_ok(const ClothConst& c) {
  auto in = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
  return c.Pp <= UD_REF_MAXP && in(c.Ls, 0x1p-20f, 0x1p4f) && in(c.Ld, 0x1p-20f, 0x1p4f) && in(c.eps, 0x1p-48f, 0x1p20f) &&
         std::isfinite(c.dt) && std::isfinite(c.damp) && std::isfinite(c.g) && std::isfinite(c.gdt) && std::isfinite(c.max_v);
}

void cloth_launch_fwd_ref(const ClothFwdArgs& a, hipStream_t stream) {
  const size_t shmem = (size_t)2 * 3 * UD_REF_MAXP * sizeof(float);
  hipLaunchKernelGGL(cloth_rollout_fwd_ref_kernel, dim3(a.B), dim3(a.c.Pp), shmem, stream, a);
}

}  // namespace ud
