// Cloth forward for bodies of more than 1024 particles, several workgroups per env (protocol and rationale: cloth_cluster.h).
// Per particle this is cloth_v2.hip -- the same force_v2 / grip_own, operation order "v2", no FMA contraction -- so the
// result is bit-identical to the CPU restatement of that order (oracle cloth_substep_fwd_v2) whatever the number of parts.
// Per substep a part publishes its 512 positions (3 granule stores per lane), does everything that needs no neighbour,
// then polls the <= 2H halo positions it needs from the parts below and above into its LDS window.
#include "cloth_cluster.h"
#include "cloth_v2_force.h"

namespace ud {

__global__ void __launch_bounds__(CL_T) cloth_cluster_fwd_kernel(ClothFwdArgs a, ClusterArgs q) {
  extern __shared__ float ldsf[];  // Xs[2][3][CL_STRIDE], double-buffered by substep parity | bail[2]
  int bl, w;
  cl_decode(q.W, bl, w);
  if (bl >= q.Bl) return;
  const int b = q.b0 + bl;
  const ClothConst c = a.c;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const int i = threadIdx.x, base = w * CL_T, gi = base + i;
  const bool inp = gi < Pp, live = gi < P;
  const int lo = max(0, base - q.H), hi = min(Pp, base + CL_T + q.H);   // LDS window = particles [lo, hi)
  const int nlo = base - lo, nhi = max(0, hi - (base + CL_T));            // halo entries below / above the part
  const int li = gi - lo;
  const bool hl = i < nlo + nhi;                                          // this lane fetches one halo particle
  const int hidx = i < nlo ? lo + i : base + CL_T + (i - nlo);
  int* bail = (int*)(ldsf + 6 * CL_STRIDE);
  if (i < 2) bail[i] = 0;
  int nbs[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { const int j = inp ? a.nbr[l * Pp + gi] : -1; nbs[l] = (j >= 0 ? j : gi) - lo; }
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = a.x[((size_t)b * P + gi) * 3 + d]; v[d] = a.v[((size_t)b * P + gi) * 3 + d]; }
  }
  float ps[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ps[d] = a.prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  const f2 kL2 = {k / c.Ls, k / c.Ld};
  GraspThr th0, th1;
  th0.init(ps[3]); th1.init(ps[7]);
  const size_t rec = cloth_rec_floats(Pp);
  float* ckb = a.ckpt ? a.ckpt + (size_t)b * cloth_env_records(T, S) * rec : nullptr;
  cl_granule* ar = q.arena + (size_t)bl * cl_env_granules(Pp, q.W);   // XE[2][3][Pp] first
  __syncthreads();
  unsigned step = 0;
  bool dead = false;
  for (int t = 0; t < T && !dead; ++t) {
    float act[8];
    macro_action_f(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      const unsigned tag = step + 1u;
      float* Xs = ldsf + (step & 1u) * (3 * CL_STRIDE);
      Xs[li] = x[0]; Xs[CL_STRIDE + li] = x[1]; Xs[2 * CL_STRIDE + li] = x[2];
      // XE is double-buffered by step parity: a part writes positions n+2 only after it has consumed its neighbours'
      // positions n+1, which they published after reading positions n -- so nobody still reads what is overwritten
      cl_granule* xe = ar + (size_t)(step & 1u) * 3 * Pp;
      if (inp) { cl_put(xe + gi, x[0], tag); cl_put(xe + Pp + gi, x[1], tag); cl_put(xe + 2 * (size_t)Pp + gi, x[2], tag); }
      if (ckb && inp) {
        float* r = ckb + (size_t)step * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * Pp + gi] = x[d]; r[(3 + d) * Pp + gi] = v[d]; }
        if (gi == 0) {
#pragma unroll
          for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
        }
      }
      float vv[3], x2[3];
      bool m0, m1;
      grip_own(x, ps, act, th0.at(step == 0), th1.at(step == 0), m0, m1, x2);
      const float isV = 1.0f / sqrtf(v[0] * v[0] + v[2] * v[2] + c.eps);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) ps[g * 4 + d] = clipf(ps[g * 4 + d] + (d < 3 ? act[g * 4 + d] : 0.f), 0.f, 1.f);  // :322-323
      if (__builtin_amdgcn_ballot_w64(hl) != 0) {   // waves that hold halo lanes (the first (nlo + nhi) / 64 of the part)
        float h[3];
        const bool ok = cl_poll3(xe + hidx, (size_t)Pp, tag, hl, h);
        if (hl) { Xs[hidx - lo] = h[0]; Xs[CL_STRIDE + hidx - lo] = h[1]; Xs[2 * CL_STRIDE + hidx - lo] = h[2]; }
        if (!ok) bail[step & 1u] = 1;
      }
      __syncthreads();
      if (bail[step & 1u]) { dead = true; break; }
      force_v2<CL_STRIDE>(c, nbs, Xs, k, kL2, mu, x, v, isV, vv);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        vv[d] = m0 ? act[3] * vv[d] : vv[d];
        vv[d] = m1 ? act[7] * vv[d] : vv[d];
      }
      if (a.grasp && live) {
        uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
        g[gi] = m0; g[P + gi] = m1;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {   // :326-329
        const float vc = clipf(vv[d], -c.max_v, c.max_v);
        x[d] = clipf(x2[d], 0.f, 1.f) + c.dt * vc;
        v[d] = vc;
      }
    }
    if (dead) break;
    if (live) {
      const size_t o = (((size_t)t * B + b) * P + gi) * 3;
      if (a.x_list) { a.x_list[o] = x[0]; a.x_list[o + 1] = x[1]; a.x_list[o + 2] = x[2]; }
      if (a.v_list) { a.v_list[o] = v[0]; a.v_list[o + 1] = v[1]; a.v_list[o + 2] = v[2]; }
    }
    if (a.prim_list && gi == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) a.prim_list[((size_t)t * B + b) * 8 + d] = ps[d];
    }
  }
  if (dead) {   // a part of this env never showed up: make it loud
    if (i == 0 && q.timeouts) atomicAdd(q.timeouts, 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = NAN; v[d] = NAN; }
#pragma unroll
    for (int d = 0; d < 8; ++d) ps[d] = NAN;
  }
  if (live) {
    const size_t o = ((size_t)b * P + gi) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.x_out[o + d] = x[d]; a.v_out[o + d] = v[d]; }
  }
  if (gi == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) a.prim_out[b * 8 + d] = ps[d];
  }
  if (ckb && inp) {
    float* r = ckb + (size_t)T * S * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) { r[d * Pp + gi] = x[d]; r[(3 + d) * Pp + gi] = v[d]; }
    if (gi == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
    }
  }
}

void cloth_launch_fwd_cluster(const ClothFwdArgs& a, const ClusterArgs& q, hipStream_t stream) {
  const size_t shmem = (size_t)(6 * CL_STRIDE + 2) * sizeof(float);
  hipLaunchKernelGGL(cloth_cluster_fwd_kernel, dim3(cl_grid(q.Bl, q.W)), dim3(CL_T), shmem, stream, a, q);
}

}  // namespace ud
