// Default cloth forward for gfx950: operation order "v2" -- the reference's formulas re-associated into far fewer
// IEEE operations (1 division + 1 sqrt per link instead of 6 + 1) and compiled WITHOUT FMA contraction, so that
// it is bit-identical to the CPU restatement of the same order (oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd_v2)
// over a whole 2000-substep step_diff, including the discrete grasp sets.  Against the reference's
// literal order (cloth.hip, mode 1) it differs by f32 round-off per substep -- see DESIGN.md "Numerical sensitivity".
// Same mapping as the other cloth kernels: one workgroup per env, one particle per lane, float4 positions
// double-buffered in LDS, one barrier per substep, per-substep checkpoints to HBM.
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
constexpr int UD_V2_MAXP = 1024;   // LDS plane stride (floats); the kernels refuse Pp > 1024

__device__ __forceinline__ void force_v2(const ClothConst& c, const int* nbs, const float* Xs, float k, f2 kL2, float mu,
                                         const float* x, const float* v, float isV, float* v3) {
  f2 r0[4], r1[4], r2[4], cl[4], inv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ja = nbs[p], jb = nbs[p + 4];
    r0[p] = f2{Xs[ja], Xs[jb]} - x[0];
    r1[p] = f2{Xs[UD_V2_MAXP + ja], Xs[UD_V2_MAXP + jb]} - x[1];
    r2[p] = f2{Xs[2 * UD_V2_MAXP + ja], Xs[2 * UD_V2_MAXP + jb]} - x[2];
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

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) cloth_rollout_fwd_v2_kernel(ClothFwdArgs a) {
  extern __shared__ float ldsf[];  // Xs[2][3][UD_V2_MAXP], double-buffered by substep parity
  const ClothConst c = a.c;
  const int i = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const bool live = i < P;
  int nbs[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { const int j = a.nbr[l * Pp + i]; nbs[l] = j >= 0 ? j : i; }
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = a.x[((size_t)b * P + i) * 3 + d]; v[d] = a.v[((size_t)b * P + i) * 3 + d]; }
  }
  float ps[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ps[d] = a.prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  const f2 kL2 = {k / c.Ls, k / c.Ld};   // k / L0 with the rest lengths of cloth_simulator.py:61-63
  GraspThr th0, th1;
  th0.init(ps[3]); th1.init(ps[7]);
  const size_t rec = cloth_rec_floats(Pp);
  float* ckb = a.ckpt ? a.ckpt + (size_t)b * cloth_env_records(T, S) * rec : nullptr;
  unsigned step = 0;
  for (int t = 0; t < T; ++t) {
    float act[8];
    macro_action_f(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      float* Xs = ldsf + (step & 1u) * (3 * UD_V2_MAXP);
      Xs[i] = x[0]; Xs[UD_V2_MAXP + i] = x[1]; Xs[2 * UD_V2_MAXP + i] = x[2];
      if (ckb) {
        float* r = ckb + (size_t)step * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
        if (i == 0) {
#pragma unroll
          for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
        }
      }
      // everything that needs no neighbour goes between the LDS write and the barrier, where it hides the write
      // latency and the arrival skew of the other waves
      float vv[3], x2[3];
      bool m0, m1;
      grip_own(x, ps, act, th0.at(step == 0), th1.at(step == 0), m0, m1, x2);
      const float isV = 1.0f / sqrtf(v[0] * v[0] + v[2] * v[2] + c.eps);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) ps[g * 4 + d] = clipf(ps[g * 4 + d] + (d < 3 ? act[g * 4 + d] : 0.f), 0.f, 1.f);  // :322-323
      __syncthreads();
      force_v2(c, nbs, Xs, k, kL2, mu, x, v, isV, vv);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        vv[d] = m0 ? act[3] * vv[d] : vv[d];
        vv[d] = m1 ? act[7] * vv[d] : vv[d];
      }
      if (a.grasp && live) {
        uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
        g[i] = m0; g[P + i] = m1;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {   // :326-329
        const float vc = clipf(vv[d], -c.max_v, c.max_v);
        x[d] = clipf(x2[d], 0.f, 1.f) + c.dt * vc;
        v[d] = vc;
      }
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


void cloth_launch_fwd_v2(const ClothFwdArgs& a, hipStream_t stream) {
  const size_t shmem = (size_t)2 * 3 * UD_V2_MAXP * sizeof(float);
  hipLaunchKernelGGL(cloth_rollout_fwd_v2_kernel, dim3(a.B), dim3(a.c.Pp), shmem, stream, a);
}

}  // namespace ud
