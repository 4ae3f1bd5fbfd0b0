// Default cloth forward for gfx950: operation order "v2" -- the reference's formulas re-associated into far fewer
// IEEE operations (1 division + 1 sqrt per link instead of 6 + 1) and compiled WITHOUT FMA contraction, so that
// it is bit-identical to the CPU restatement of the same order (oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd_v2)
// over a whole 2000-substep step_diff, including the discrete grasp sets.  Against the reference's
// literal order (cloth.hip, mode 1) it differs by f32 round-off per substep -- see DESIGN.md "Numerical sensitivity".
// Same mapping as the other cloth kernels: one workgroup per env, one particle per lane, float4 positions
// double-buffered in LDS, one barrier per substep, per-substep checkpoints to HBM.
#include "cloth_v2_force.h"

namespace ud {

constexpr int UD_V2_MAXP = 1024;   // LDS plane stride (floats); the kernels refuse Pp > 1024

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
      force_v2<UD_V2_MAXP>(c, nbs, Xs, k, kL2, mu, x, v, isV, vv);
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
