// Mass-spring cloth rollout for gfx950: one workgroup per environment, one particle per lane.
//
// What it replaces (reference, /root/reference/DaXBench/daxbench/core/engine/cloth_simulator.py):
//   forward  = lax.scan over macro actions of robot_step (:163-180) = fori_loop(50) of step (:257-337)
//   backward = jax.grad through robot_step_wrapper / step_wrapper (:107-145, :228-255) with the live
//              norm_grad rescaling (:182-196).  The reference rematerialises (3 forwards + 1 adjoint per
//              substep at mem_saving_level 2); here the forward writes one (x,v,primitives) checkpoint per
//              substep into HBM (6*Ppad+8 floats, SoA) and the backward streams them back in reverse, so
//              fwd+bwd costs one forward and one adjoint.
//
// Mapping: the whole T x substeps rollout of one env runs inside ONE launch.  x lives double-buffered in LDS
// (2 x 3 x Ppad floats) for the 8-neighbour stencil, v / cotangents / link tables live in registers, there
// is one s_barrier per forward substep.  The kernel is latency/occupancy bound:
// at num_envs=4 only 4 of 256 CUs have work.
//
// Forward arithmetic keeps the reference's operation order and is compiled with -ffp-contract=off and
// correctly rounded f32 divide/sqrt, so it is bit-identical to the CPU restatement (the grasp test
// |x - pos| <= radius is a discrete event).
#include "cloth_ref_order.h"

namespace ud {

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int MAXT>
__global__ void __launch_bounds__(MAXT) cloth_rollout_fwd_kernel(ClothFwdArgs a) {
  extern __shared__ float lds[];  // [2][3][Pp]
  const ClothConst c = a.c;
  const int i = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const bool live = i < P;
  int nb[8];
  float L0[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { nb[l] = a.nbr[l * Pp + i]; L0[l] = a.L0[l * Pp + i]; }
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = a.x[((size_t)b * P + i) * 3 + d]; v[d] = a.v[((size_t)b * P + i) * 3 + d]; }
  }
  float ps[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ps[d] = a.prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  const size_t rec = (size_t)6 * Pp + 8;
  unsigned step = 0;
  for (int t = 0; t < T; ++t) {
    float act[8];
    macro_action(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      float* X = lds + (step & 1u) * 3 * Pp;
      X[i] = x[0]; X[Pp + i] = x[1]; X[2 * Pp + i] = x[2];
      if (a.ckpt) {
        float* r = a.ckpt + ((size_t)b * ((size_t)T * S + 1) + (size_t)t * S + s) * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
        if (i == 0) {
#pragma unroll
          for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
        }
      }
      __syncthreads();
      float xo[3], vo[3], po[8];
      Inter in;
      substep_fwd<false>(c, i, nb, L0, X, k, mu, x, v, ps, act, xo, vo, &in);
      if (a.grasp && live) {
        uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
        g[i] = in.m0; g[P + i] = in.m1;
      }
      prim_update(ps, act, po);
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = xo[d]; v[d] = vo[d]; }
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
  if (a.ckpt) {  // final record (the fast backward derives the velocity-clip mask of the last substep from it)
    float* r = a.ckpt + ((size_t)b * ((size_t)T * S + 1) + (size_t)T * S) * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
    if (i == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// block-wide sums of two values; red points at a private [2*16] LDS slot for this round
__device__ __forceinline__ void block_sum2(float& s0, float& s1, float* red, int nw) {
  float a = wave_sum(s0), b = wave_sum(s1);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
  __syncthreads();
  float t0 = 0.f, t1 = 0.f;
  for (int q = 0; q < nw; ++q) { t0 += red[2 * q]; t1 += red[2 * q + 1]; }
  s0 = t0; s1 = t1;
}

__device__ __forceinline__ void norm3(float* g, float nrm2, float n_mask) {  // :189-194
  float nrm = sqrtf(nrm2);
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = nan_to_num(g[a] / nrm) / n_mask;
}

__device__ __forceinline__ void norm4(float* g, float n_mask) {
  float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3]);
#pragma unroll
  for (int a = 0; a < 4; ++a) g[a] = nan_to_num(g[a] / nrm) / n_mask;
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) cloth_rollout_bwd_kernel(ClothBwdArgs a) {
  extern __shared__ float lds[];  // X [3][Pp] | G [3][Pp] | red [6][32]
  const ClothConst c = a.c;
  const int i = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const int nw = Pp >> 6;
  const bool live = i < P;
  const bool norm = a.normalize != 0;
  float* X = lds;
  float* G = lds + 3 * Pp;
  float* red = lds + 6 * Pp;
  int nb[8];
  float L0[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { nb[l] = a.nbr[l * Pp + i]; L0[l] = a.L0[l * Pp + i]; }
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = a.g_x[((size_t)b * P + i) * 3 + d]; gv[d] = a.g_v[((size_t)b * P + i) * 3 + d]; }
  }
  float gp[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) gp[d] = a.g_prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  float gk = 0.f, gmu = 0.f;
  const size_t rec = (size_t)6 * Pp + 8;
  const float* ck = a.ckpt + (size_t)b * ((size_t)T * S + 1) * rec;
  // prefetch the last record
  float nx[3], nv[3], nps[8];
  {
    const float* r = ck + ((size_t)T * S - 1) * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) { nx[d] = r[d * Pp + i]; nv[d] = r[(3 + d) * Pp + i]; }
#pragma unroll
    for (int d = 0; d < 8; ++d) nps[d] = r[6 * Pp + d];
  }
  unsigned step = 0;
  for (int t = T - 1; t >= 0; --t) {
    if (live) {
      const size_t o = (((size_t)t * B + b) * P + i) * 3;
      if (a.g_x_list) { gx[0] += a.g_x_list[o]; gx[1] += a.g_x_list[o + 1]; gx[2] += a.g_x_list[o + 2]; }
      if (a.g_v_list) { gv[0] += a.g_v_list[o]; gv[1] += a.g_v_list[o + 1]; gv[2] += a.g_v_list[o + 2]; }
    }
    if (a.g_prim_list) {
#pragma unroll
      for (int d = 0; d < 8; ++d) gp[d] += a.g_prim_list[((size_t)t * B + b) * 8 + d];
    }
    const float* a8 = a.actions + ((size_t)t * B + b) * 8;
    float act[8], ga[8];
    macro_action(a8, act);
#pragma unroll
    for (int d = 0; d < 8; ++d) ga[d] = 0.f;
    for (int s = S - 1; s >= 0; --s, ++step) {
      float x[3], v[3], ps[8];
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = nx[d]; v[d] = nv[d]; }
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = nps[d];
      {  // prefetch the previous substep's record (the next one this loop consumes)
        const long q = (long)t * S + s - 1;
        const float* r = ck + (size_t)(q < 0 ? 0 : q) * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { nx[d] = r[d * Pp + i]; nv[d] = r[(3 + d) * Pp + i]; }
#pragma unroll
        for (int d = 0; d < 8; ++d) nps[d] = r[6 * Pp + d];
      }
      float* rd = red + (step & 1u) * 96;
      // -- stage x for the stencil; (G of the previous substep has been consumed before its last barrier)
      X[i] = x[0]; X[Pp + i] = x[1]; X[2 * Pp + i] = x[2];
      // -- reduction round 1: |g_x|, |g_v| (:331-332) -- its barrier also publishes X
      float n0 = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
      float n1 = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2];
      block_sum2(n0, n1, rd, nw);
      if (norm) {
        norm3(gx, n0, c.n_mask);
        norm3(gv, n1, c.n_mask);
        norm4(gp, c.n_mask);      // :333-334
        norm4(gp + 4, c.n_mask);
      }
      float xo[3], vo[3];
      Inter in;
      substep_fwd<true>(c, i, nb, L0, X, k, mu, x, v, ps, act, xo, vo, &in);
      // x_out = clip(x2) + dt*clip(v5)   (:326-329)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float gxc = gx[d];
        float gvc = gv[d] + c.dt * gx[d];
        gx[d] = gxc * clip_grad(in.x2[d], 0.f, 1.f);
        gv[d] = gvc * clip_grad(in.v5[d], -c.max_v, c.max_v);
      }
      // primitives (:322-323); uniform across lanes, counted once (lane 0) in the action accumulators
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float add = d < 3 ? act[g * 4 + d] : 0.f;
          float tt = gp[g * 4 + d] * clip_grad(ps[g * 4 + d] + add, 0.f, 1.f);
          gp[g * 4 + d] = tt;
          if (d < 3) ga[g * 4 + d] += (i == 0) ? tt : 0.f;
        }
      // grippers in reverse order (:313-314, :198-226)
#pragma unroll
      for (int g = 1; g >= 0; --g) {
        n0 = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
        n1 = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2];
        block_sum2(n0, n1, rd + 32 * (2 - g), nw);
        if (norm) { norm3(gx, n0, c.n_mask); norm3(gv, n1, c.n_mask); }  // :223-224
        const bool m = (g ? in.m1 : in.m0) && live;
        const float* vin = g ? in.v4 : in.v3;
        const float suction = act[g * 4 + 3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float gvo = gv[d], gxo = gx[d];
          ga[g * 4 + 3] += m ? (vin[d] * gvo - gxo * act[g * 4 + d]) : 0.f;
          ga[g * 4 + d] += m ? gxo * (1.f - suction) : 0.f;
          gv[d] = m ? suction * gvo : gvo;
        }
      }
      // v3 = (v1 + F*dt)*damp (:308-309); friction block (:281-306)
      float gF[3];
      {
        float gv2[3], gFf[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { gv2[d] = gv[d] * c.damp; gFf[d] = gv2[d] * c.dt; }
        float gCx = gFf[0], gCz = gFf[2];
        float gBx = gCx * (1.f - in.nz), gBz = gCz * (1.f - in.nz);
        float gR = gCx * in.nz * in.Ax + gCz * in.nz * in.Az;
        float gAx = gCx * in.nz * in.R + gBx * (1.f - in.zm);
        float gAz = gCz * in.nz * in.R + gBz * (1.f - in.zm);
        float gmuF = -gR / in.sF;
        float gsF = gR * in.muF / (in.sF * in.sF);
        gAx += gsF * in.Ax / in.sF;
        gAz += gsF * in.Az / in.sF;
        gmuF += -(gAx * in.dm * in.xV / in.sV + gAz * in.dm * in.yV / in.sV);
        float gxV = -gAx * in.dm * in.muF / in.sV, gyV = -gAz * in.dm * in.muF / in.sV;
        float gsV = (gAx * in.dm * in.muF * in.xV + gAz * in.dm * in.muF * in.yV) / (in.sV * in.sV);
        gxV += gsV * in.xV / in.sV;
        gyV += gsV * in.yV / in.sV;
        gmu += live ? -gmuF * in.cF : 0.f;
        float gcF = -gmuF * mu;
        gF[0] = gAx;
        gF[1] = gFf[1] + gcF * clip_grad(in.F1, -INFINITY, 0.f);
        gF[2] = gAz;
        gv[0] = gv2[0] + gxV;
        gv[1] = gv2[1];
        gv[2] = gv2[2] + gyV;
        if (!live) { gF[0] = gF[1] = gF[2] = 0.f; }
      }
      // spring forces (:262-277), gather form: g_x_i += sum_l J_il (gF_j - gF_i)
      G[i] = gF[0]; G[Pp + i] = gF[1]; G[2 * Pp + i] = gF[2];
      __syncthreads();
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const int j = nb[l];
        const bool ok = j >= 0;
        const int jj = ok ? j : i;
        float r0 = X[jj] - x[0], r1 = X[Pp + jj] - x[1], r2 = X[2 * Pp + jj] - x[2];
        float s2 = r0 * r0 + r1 * r1 + r2 * r2;
        float cf = clip_grad(s2, 1e-12f, INFINITY);
        float len = sqrtf(clipf(s2, 1e-12f, INFINITY));
        float L = L0[l];
        float d0 = G[jj] - gF[0], d1 = G[Pp + jj] - gF[1], d2 = G[2 * Pp + jj] - gF[2];
        float rd_ = r0 * d0 + r1 * d1 + r2 * d2;
        float rg = r0 * gF[0] + r1 * gF[1] + r2 * gF[2];
        float c1 = (k / L) * (1.f - L / len);
        float c2 = (k / L) * cf * L / (len * len * len) * rd_;
        gk += ok ? rg / len * (len - L) / L : 0.f;
        gx[0] += ok ? c1 * d0 + c2 * r0 : 0.f;
        gx[1] += ok ? c1 * d1 + c2 * r1 : 0.f;
        gx[2] += ok ? c1 * d2 + c2 * r2 : 0.f;
      }
      // the next substep overwrites X before its first barrier and G after it; a trailing barrier keeps
      // slow lanes' stencil reads of X/G ahead of those writes
      __syncthreads();
    }
    // macro-step boundary: robot_step's action transform (:168-169)
    {
      float* rd = red + 192;
      float part[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) part[d] = wave_sum(ga[d]);
      const int lane = i & 63, w = i >> 6;
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) rd[w * 8 + d] = part[d];
      }
      __syncthreads();
      if (i < 8) {
        float tot = 0.f;
        for (int q = 0; q < nw; ++q) tot += rd[q * 8 + i];
        const int d = i & 3;
        float out = (d < 3) ? tot * (1.0f / 50.0f) * clip_grad(a8[i], -2.0f, 2.0f) : tot;
        a.g_actions[((size_t)t * B + b) * 8 + i] = out;
      }
      __syncthreads();
    }
  }
  if (live) {
    const size_t o = ((size_t)b * P + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.g_x0[o + d] = gx[d]; a.g_v0[o + d] = gv[d]; }
  }
  float rk = gk, rmu = gmu;
  block_sum2(rk, rmu, red + 192, nw);
  if (i == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) a.g_prim0[b * 8 + d] = gp[d];
    a.g_k[b] = rk;
    a.g_mu[b] = rmu;
  }
}


// ------------------------------------------------------------------------------------------------
// Bodies of more than 1024 particles (fold_tshirt: 3573, fold_cloth_tshirt_env.py:19-50): still one workgroup per env,
// 1024 lanes, each lane owns particles tid, tid + 1024, ... (at most UD_BIG_PPT of them).  Same device functions, same
// operation order per particle as the kernels above; neighbour tables are re-read from L2 every substep instead of living
// in registers, and the adjoint parks the per-particle intermediates of the recomputed forward in a scratch arena (global,
// SoA) between its barrier-separated phases.  x stays in LDS (2 x 3 x Pp floats forward, X + G planes in the adjoint).
// ------------------------------------------------------------------------------------------------
#define UD_BIG_T 1024
#define UD_BIG_PPT 4
#define UD_BIG_PARK 27   // floats per particle in the adjoint's scratch: F1 cF muF xV yV sV dm Ax Az sF zm nz R v3[3] v4[3] m0 m1 | gx[3] gv[3]

// The per-particle loops below are kept rolled (#pragma unroll 1) and the per-particle state lives in LDS (forward: x
// double-buffered + v) or in the scratch arena (adjoint: cotangents): unrolled four-wide, the same code needed 0.7 KB
// (forward) / 2.4 KB (adjoint) of scratch per lane under the 128-VGPR budget of a 1024-lane workgroup.
__device__ __forceinline__ void big_tables(const ClothConst& c, const int* nbr, int Pp, int i, int* nb, float* L0) {
#pragma unroll
  for (int l = 0; l < 8; ++l) { nb[l] = nbr[l * Pp + i]; L0[l] = (l < 4) ? c.Ls : c.Ld; }   // = the L0 table: :61-63 depend on the link only
}

__global__ void __launch_bounds__(UD_BIG_T) cloth_big_fwd_kernel(ClothFwdArgs a) {
  extern __shared__ float lds[];  // X [2][3][Pp] | V [3][Pp]
  const ClothConst c = a.c;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  float* V = lds + 6 * Pp;
#pragma unroll 1
  for (int i = tid; i < Pp; i += UD_BIG_T) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lds[d * Pp + i] = (i < P) ? a.x[((size_t)b * P + i) * 3 + d] : 0.f;
      V[d * Pp + i] = (i < P) ? a.v[((size_t)b * P + i) * 3 + d] : 0.f;
    }
  }
  float ps[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ps[d] = a.prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  const size_t rec = (size_t)6 * Pp + 8;
  unsigned step = 0;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    float act[8];
    macro_action(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      const float* X = lds + (step & 1u) * 3 * Pp;
      float* Xn = lds + ((step + 1) & 1u) * 3 * Pp;
      float* r = a.ckpt ? a.ckpt + ((size_t)b * ((size_t)T * S + 1) + (size_t)t * S + s) * rec : nullptr;
      if (r && tid == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
      }
#pragma unroll 1
      for (int i = tid; i < Pp; i += UD_BIG_T) {
        const float x[3] = {X[i], X[Pp + i], X[2 * Pp + i]};
        const float v[3] = {V[i], V[Pp + i], V[2 * Pp + i]};
        if (r) {
#pragma unroll
          for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
        }
        int nb[8];
        float L0[8];
        big_tables(c, a.nbr, Pp, i, nb, L0);
        float xo[3], vo[3];
        Inter in;
        substep_fwd<false>(c, i, nb, L0, X, k, mu, x, v, ps, act, xo, vo, &in);
        if (a.grasp && i < P) {
          uint8_t* g = a.grasp + ((((size_t)t * S + s) * B + b) * 2) * P;
          g[i] = in.m0; g[P + i] = in.m1;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) { Xn[d * Pp + i] = xo[d]; V[d * Pp + i] = vo[d]; }
      }
      float po[8];
      prim_update(ps, act, po);
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = po[d];
      __syncthreads();   // Xn complete; nobody reads X any more
    }
    const float* X = lds + (step & 1u) * 3 * Pp;
#pragma unroll 1
    for (int i = tid; i < P; i += UD_BIG_T) {
      const size_t o = (((size_t)t * B + b) * P + i) * 3;
      if (a.x_list) { a.x_list[o] = X[i]; a.x_list[o + 1] = X[Pp + i]; a.x_list[o + 2] = X[2 * Pp + i]; }
      if (a.v_list) { a.v_list[o] = V[i]; a.v_list[o + 1] = V[Pp + i]; a.v_list[o + 2] = V[2 * Pp + i]; }
    }
    if (a.prim_list && tid == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) a.prim_list[((size_t)t * B + b) * 8 + d] = ps[d];
    }
  }
  const float* X = lds + (step & 1u) * 3 * Pp;
  float* r = a.ckpt ? a.ckpt + ((size_t)b * ((size_t)T * S + 1) + (size_t)T * S) * rec : nullptr;   // final record
#pragma unroll 1
  for (int i = tid; i < Pp; i += UD_BIG_T) {
    if (i < P) {
      const size_t o = ((size_t)b * P + i) * 3;
#pragma unroll
      for (int d = 0; d < 3; ++d) { a.x_out[o + d] = X[d * Pp + i]; a.v_out[o + d] = V[d * Pp + i]; }
    }
    if (r) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { r[d * Pp + i] = X[d * Pp + i]; r[(3 + d) * Pp + i] = V[d * Pp + i]; }
    }
  }
  if (tid == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) a.prim_out[b * 8 + d] = ps[d];
    if (r) {
#pragma unroll
      for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
    }
  }
}

// adjoint; `park` = scratch [B][UD_BIG_PARK][Pp]
__global__ void __launch_bounds__(UD_BIG_T) cloth_big_bwd_kernel(ClothBwdArgs a, float* park) {
  extern __shared__ float lds[];  // X [3][Pp] | G [3][Pp] | red [6][32] + [16][8]
  const ClothConst c = a.c;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const int nw = UD_BIG_T >> 6;
  const bool norm = a.normalize != 0;
  float* X = lds;
  float* G = lds + 3 * Pp;
  float* red = lds + 6 * Pp;
  float* pk = park + (size_t)b * UD_BIG_PARK * Pp;
  float* GX = pk + (size_t)21 * Pp;   // cotangent planes gx[3][Pp], gv[3][Pp]: each lane touches only its own particles
  float* GV = pk + (size_t)24 * Pp;
#pragma unroll 1
  for (int i = tid; i < Pp; i += UD_BIG_T) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      GX[(size_t)d * Pp + i] = (i < P) ? a.g_x[((size_t)b * P + i) * 3 + d] : 0.f;
      GV[(size_t)d * Pp + i] = (i < P) ? a.g_v[((size_t)b * P + i) * 3 + d] : 0.f;
    }
  }
  float gp[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) gp[d] = a.g_prim[b * 8 + d];
  const float k = a.k[b], mu = a.mu[b];
  float gk = 0.f, gmu = 0.f;
  const size_t rec = (size_t)6 * Pp + 8;
  const float* ck = a.ckpt + (size_t)b * ((size_t)T * S + 1) * rec;
  unsigned step = 0;
  for (int t = T - 1; t >= 0; --t) {
    if (a.g_x_list || a.g_v_list) {
#pragma unroll 1
      for (int i = tid; i < P; i += UD_BIG_T) {
        const size_t o = (((size_t)t * B + b) * P + i) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (a.g_x_list) GX[(size_t)d * Pp + i] += a.g_x_list[o + d];
          if (a.g_v_list) GV[(size_t)d * Pp + i] += a.g_v_list[o + d];
        }
      }
    }
    if (a.g_prim_list) {
#pragma unroll
      for (int d = 0; d < 8; ++d) gp[d] += a.g_prim_list[((size_t)t * B + b) * 8 + d];
    }
    const float* a8 = a.actions + ((size_t)t * B + b) * 8;
    float act[8], ga[8];
    macro_action(a8, act);
#pragma unroll
    for (int d = 0; d < 8; ++d) ga[d] = 0.f;
    for (int s = S - 1; s >= 0; --s, ++step) {
      const float* r = ck + ((size_t)t * S + s) * rec;
      float ps[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = r[6 * Pp + d];
      float* rd = red + (step & 1u) * 96;
      // -- stage x; round 1: |g_x|, |g_v| (:331-332) -- its barrier also publishes X
      float n0 = 0.f, n1 = 0.f;
#pragma unroll 1
      for (int i = tid; i < Pp; i += UD_BIG_T) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          X[d * Pp + i] = r[d * Pp + i];
          const float gxd = GX[(size_t)d * Pp + i], gvd = GV[(size_t)d * Pp + i];
          n0 += gxd * gxd; n1 += gvd * gvd;
        }
      }
      block_sum2(n0, n1, rd, nw);
      if (norm) {
        norm4(gp, c.n_mask);      // :333-334
        norm4(gp + 4, c.n_mask);
      }
      // -- recomputed forward of every particle; x_out = clip(x2) + dt*clip(v5) (:326-329); park what the later phases need.
      //    The same pass accumulates the norms gripper 1 needs (round 2).
      float m0s = 0.f, m1s = 0.f;
#pragma unroll 1
      for (int i = tid; i < Pp; i += UD_BIG_T) {
        float gx[3] = {GX[i], GX[(size_t)Pp + i], GX[(size_t)2 * Pp + i]}, gv[3] = {GV[i], GV[(size_t)Pp + i], GV[(size_t)2 * Pp + i]};
        if (norm) { norm3(gx, n0, c.n_mask); norm3(gv, n1, c.n_mask); }
        int nb[8];
        float L0[8];
        big_tables(c, a.nbr, Pp, i, nb, L0);
        const float x[3] = {X[i], X[Pp + i], X[2 * Pp + i]};
        const float v[3] = {r[3 * Pp + i], r[4 * Pp + i], r[5 * Pp + i]};
        float xo[3], vo[3];
        Inter in;
        substep_fwd<true>(c, i, nb, L0, X, k, mu, x, v, ps, act, xo, vo, &in);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float gxc = gx[d];
          const float gvc = gv[d] + c.dt * gx[d];
          gx[d] = gxc * clip_grad(in.x2[d], 0.f, 1.f);
          gv[d] = gvc * clip_grad(in.v5[d], -c.max_v, c.max_v);
          GX[(size_t)d * Pp + i] = gx[d]; GV[(size_t)d * Pp + i] = gv[d];
          m0s += gx[d] * gx[d]; m1s += gv[d] * gv[d];
        }
        const float vals[21] = {in.F1, in.cF, in.muF, in.xV, in.yV, in.sV, in.dm, in.Ax, in.Az, in.sF, in.zm, in.nz, in.R,
                                in.v3[0], in.v3[1], in.v3[2], in.v4[0], in.v4[1], in.v4[2], in.m0 ? 1.f : 0.f, in.m1 ? 1.f : 0.f};
#pragma unroll
        for (int e = 0; e < 21; ++e) pk[(size_t)e * Pp + i] = vals[e];
      }
      // primitives (:322-323); uniform across lanes, counted once (lane 0) in the action accumulators
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float add = d < 3 ? act[g * 4 + d] : 0.f;
          const float tt = gp[g * 4 + d] * clip_grad(ps[g * 4 + d] + add, 0.f, 1.f);
          gp[g * 4 + d] = tt;
          if (d < 3) ga[g * 4 + d] += (tid == 0) ? tt : 0.f;
        }
      // grippers in reverse order (:313-314, :198-226); the pass of gripper g accumulates the norms of the next round
      n0 = m0s; n1 = m1s;
#pragma unroll
      for (int g = 1; g >= 0; --g) {
        block_sum2(n0, n1, rd + 32 * (2 - g), nw);
        const float suction = act[g * 4 + 3];
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 1
        for (int i = tid; i < Pp; i += UD_BIG_T) {
          float gx[3] = {GX[i], GX[(size_t)Pp + i], GX[(size_t)2 * Pp + i]}, gv[3] = {GV[i], GV[(size_t)Pp + i], GV[(size_t)2 * Pp + i]};
          if (norm) { norm3(gx, n0, c.n_mask); norm3(gv, n1, c.n_mask); }  // :223-224
          const bool m = (pk[(size_t)(19 + g) * Pp + i] != 0.f) && i < P;
          const float* vinp = pk + (size_t)(g ? 16 : 13) * Pp + i;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const float gvo = gv[d], gxo = gx[d];
            const float vin = vinp[(size_t)d * Pp];
            ga[g * 4 + 3] += m ? (vin * gvo - gxo * act[g * 4 + d]) : 0.f;
            ga[g * 4 + d] += m ? gxo * (1.f - suction) : 0.f;
            gv[d] = m ? suction * gvo : gvo;
            a0 += gx[d] * gx[d]; a1 += gv[d] * gv[d];
          }
          if (g == 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { GX[(size_t)d * Pp + i] = gx[d]; GV[(size_t)d * Pp + i] = gv[d]; }
          } else {
            // v3 = (v1 + F*dt)*damp (:308-309); friction block (:281-306) -> gF into the G planes
            const bool live = i < P;
            const float F1 = pk[i], cF = pk[(size_t)Pp + i], muF = pk[(size_t)2 * Pp + i], xV = pk[(size_t)3 * Pp + i], yV = pk[(size_t)4 * Pp + i],
                        sV = pk[(size_t)5 * Pp + i], dm = pk[(size_t)6 * Pp + i], Ax = pk[(size_t)7 * Pp + i], Az = pk[(size_t)8 * Pp + i],
                        sF = pk[(size_t)9 * Pp + i], zm = pk[(size_t)10 * Pp + i], nz = pk[(size_t)11 * Pp + i], R = pk[(size_t)12 * Pp + i];
            float gv2[3], gFf[3], gF[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { gv2[d] = gv[d] * c.damp; gFf[d] = gv2[d] * c.dt; }
            const float gCx = gFf[0], gCz = gFf[2];
            const float gBx = gCx * (1.f - nz), gBz = gCz * (1.f - nz);
            const float gR = gCx * nz * Ax + gCz * nz * Az;
            float gAx = gCx * nz * R + gBx * (1.f - zm);
            float gAz = gCz * nz * R + gBz * (1.f - zm);
            float gmuF = -gR / sF;
            const float gsF = gR * muF / (sF * sF);
            gAx += gsF * Ax / sF;
            gAz += gsF * Az / sF;
            gmuF += -(gAx * dm * xV / sV + gAz * dm * yV / sV);
            float gxV = -gAx * dm * muF / sV, gyV = -gAz * dm * muF / sV;
            const float gsV = (gAx * dm * muF * xV + gAz * dm * muF * yV) / (sV * sV);
            gxV += gsV * xV / sV;
            gyV += gsV * yV / sV;
            gmu += live ? -gmuF * cF : 0.f;
            const float gcF = -gmuF * mu;
            gF[0] = gAx;
            gF[1] = gFf[1] + gcF * clip_grad(F1, -INFINITY, 0.f);
            gF[2] = gAz;
            if (!live) { gF[0] = gF[1] = gF[2] = 0.f; }
            G[i] = gF[0]; G[Pp + i] = gF[1]; G[2 * Pp + i] = gF[2];
            GX[i] = gx[0]; GX[(size_t)Pp + i] = gx[1]; GX[(size_t)2 * Pp + i] = gx[2];
            GV[i] = gv2[0] + gxV; GV[(size_t)Pp + i] = gv2[1]; GV[(size_t)2 * Pp + i] = gv2[2] + gyV;
          }
        }
        n0 = a0; n1 = a1;
      }
      __syncthreads();
      // spring forces (:262-277), gather form: g_x_i += sum_l J_il (gF_j - gF_i)
#pragma unroll 1
      for (int i = tid; i < Pp; i += UD_BIG_T) {
        const float x[3] = {X[i], X[Pp + i], X[2 * Pp + i]};
        const float gF[3] = {G[i], G[Pp + i], G[2 * Pp + i]};
        float gx[3] = {GX[i], GX[(size_t)Pp + i], GX[(size_t)2 * Pp + i]};
#pragma unroll
        for (int l = 0; l < 8; ++l) {
          const int j = a.nbr[l * Pp + i];
          const bool ok = j >= 0;
          const int jj = ok ? j : i;
          const float r0 = X[jj] - x[0], r1 = X[Pp + jj] - x[1], r2 = X[2 * Pp + jj] - x[2];
          const float s2 = r0 * r0 + r1 * r1 + r2 * r2;
          const float cf = clip_grad(s2, 1e-12f, INFINITY);
          const float len = sqrtf(clipf(s2, 1e-12f, INFINITY));
          const float L = (l < 4) ? c.Ls : c.Ld;
          const float d0 = G[jj] - gF[0], d1 = G[Pp + jj] - gF[1], d2 = G[2 * Pp + jj] - gF[2];
          const float rd_ = r0 * d0 + r1 * d1 + r2 * d2;
          const float rg = r0 * gF[0] + r1 * gF[1] + r2 * gF[2];
          const float c1 = (k / L) * (1.f - L / len);
          const float c2 = (k / L) * cf * L / (len * len * len) * rd_;
          gk += ok ? rg / len * (len - L) / L : 0.f;
          gx[0] += ok ? c1 * d0 + c2 * r0 : 0.f;
          gx[1] += ok ? c1 * d1 + c2 * r1 : 0.f;
          gx[2] += ok ? c1 * d2 + c2 * r2 : 0.f;
        }
        GX[i] = gx[0]; GX[(size_t)Pp + i] = gx[1]; GX[(size_t)2 * Pp + i] = gx[2];
      }
      __syncthreads();   // X / G are rewritten by the next substep
    }
    // macro-step boundary: robot_step's action transform (:168-169)
    {
      float* rd = red + 192;
      float part[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) part[d] = wave_sum(ga[d]);
      const int lane = tid & 63, w = tid >> 6;
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) rd[w * 8 + d] = part[d];
      }
      __syncthreads();
      if (tid < 8) {
        float tot = 0.f;
        for (int qq = 0; qq < nw; ++qq) tot += rd[qq * 8 + tid];
        const int d = tid & 3;
        a.g_actions[((size_t)t * B + b) * 8 + tid] = (d < 3) ? tot * (1.0f / 50.0f) * clip_grad(a8[tid], -2.0f, 2.0f) : tot;
      }
      __syncthreads();
    }
  }
#pragma unroll 1
  for (int i = tid; i < P; i += UD_BIG_T) {
    const size_t o = ((size_t)b * P + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.g_x0[o + d] = GX[(size_t)d * Pp + i]; a.g_v0[o + d] = GV[(size_t)d * Pp + i]; }
  }
  float rk = gk, rmu = gmu;
  block_sum2(rk, rmu, red + 192, nw);
  if (tid == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) a.g_prim0[b * 8 + d] = gp[d];
    a.g_k[b] = rk;
    a.g_mu[b] = rmu;
  }
}

}  // namespace ud

// ------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ------------------------------------------------------------------------------------------------
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cloth_cluster.h"

struct ud_cloth {
  ud::ClothConst c;
  int mode = 0;   // ud_cloth_conf.mode
  int device = 0;
  int* d_nbr = nullptr;
  float* d_L0 = nullptr;
  float* d_park = nullptr;   // P > 1024 only: adjoint scratch [B][UD_BIG_PARK][Pp]
  int park_B = 0;
  // P > 1024, several workgroups per env (cloth_cluster.h): parts per env, halo width (0 = the body does not qualify),
  // CUs of the device, and the hand-off arena [arena_B][cl_env_granules]
  int cl_W = 0, cl_H = 0, n_cu = 0;
  ud::cl_granule* d_arena = nullptr;
  int arena_B = 0;
  int* d_timeouts = nullptr;   // parts of the several-workgroup kernels that gave up a poll (ud_cloth_poll_timeouts)
  int max_envs = 0;            // ud_cloth_conf.max_envs: what d_arena / d_park were sized for at create
  bool one_wg = false;         // ud_cloth_conf.one_workgroup_per_env
};

// Several workgroups per env when the body qualifies (halo <= CL_HMAX, the parts of one env fit on the chip) and the caller
// did not ask for the reference-order kernels (mode 1).  A call is cut into launches of cloth_cluster_envs() envs so that
// every workgroup of a launch is resident at once (cloth_cluster.h, "Progress").
// ud_cloth_conf.one_workgroup_per_env != 0 (fixed at create; diagnostics and tests) keeps the one-workgroup kernels.
static bool cloth_use_cluster(const ud_cloth* h, int B) {
  if (h->c.Pp <= 1024 || h->cl_H == 0 || h->mode == 1 || h->cl_W > h->n_cu / 8) return false;   // the parts of an env share an XCD (cl_decode)
  return !h->one_wg;
}
// Envs per launch.  cl_decode deals the envs of a launch round-robin over the 8 XCDs and keeps the W parts of an env on one, and
// the adjoint kernel (256 VGPRs, 512 lanes) fits once per CU: an XCD (n_cu / 8 CUs) holds floor((n_cu / 8) / W) whole envs.
// (floor(n_cu / W) -- the chip-wide count -- gave 36 for the T-shirt's W = 7: in launches of 33-36 envs XCDs 0-3 were dealt
// 5 x 7 = 35 workgroups for 32 CUs, three parts waited for a CU while their siblings spun: ~2x the time, no deadlock.)
static int cloth_cluster_envs(const ud_cloth* h, int B) { return std::min(B, std::max(1, 8 * ((h->n_cu / 8) / h->cl_W))); }

// zero the hand-off arena (allocated at create for the largest launch: cloth_cluster_envs(max_envs)): tags start at 1, so a zeroed arena
// matches nothing
static int cloth_cluster_arena(ud_cloth* h, int B, hipStream_t stream, ud::ClusterArgs* q) {   // B: envs per launch
  const size_t per = ud::cl_env_granules(h->c.Pp, h->cl_W) * sizeof(ud::cl_granule);
  if (h->arena_B < B) { ud::set_error("ud_cloth: hand-off arena holds %d envs per launch, %d asked", h->arena_B, B); return UD_ERR_INVALID; }
  UD_HIP_CHECK(hipMemsetAsync(h->d_arena, 0, per * B, stream));
  q->W = h->cl_W; q->H = h->cl_H; q->arena = h->d_arena; q->b0 = 0; q->Bl = B; q->timeouts = h->d_timeouts;
  return UD_OK;
}

extern "C" {

int ud_cloth_create(const ud_cloth_conf* conf, const uint8_t* mask, ud_cloth** out) {
  if (!conf || !mask || !out) { ud::set_error("ud_cloth_create: null argument"); return UD_ERR_INVALID; }
  const int N = conf->N;
  if (N < 3 || conf->substeps < 1) { ud::set_error("ud_cloth_create: bad N/substeps"); return UD_ERR_INVALID; }
  static const int links[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};  // :48
  std::vector<int> pid((size_t)N * N, -1), gi, gj;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (mask[i * N + j]) {
        if (i == 0 || j == 0 || i == N - 1 || j == N - 1) {
          ud::set_error("ud_cloth_create: cloth mask touches the lattice border (clipped links, cloth_simulator.py:58) -- unsupported");
          return UD_ERR_UNSUPPORTED;
        }
        pid[i * N + j] = (int)gi.size(); gi.push_back(i); gj.push_back(j);
      }
  const int P = (int)gi.size();
  if (P < 1 || P > UD_BIG_T * UD_BIG_PPT) { ud::set_error("ud_cloth_create: P=%d outside 1..%d (one workgroup per env)", P, UD_BIG_T * UD_BIG_PPT); return UD_ERR_UNSUPPORTED; }
  const int Pp = (P + 63) / 64 * 64;
  std::vector<int> nbr((size_t)8 * Pp, -1);
  std::vector<float> L0((size_t)8 * Pp, 1.0f);
  for (int p = 0; p < P; ++p)
    for (int l = 0; l < 8; ++l) {
      const int ji = gi[p] + links[l][0], jj = gj[p] + links[l][1];
      const int di = links[l][0], dj = links[l][1];
      const float ol = (float)(1.0 / N) * sqrtf((float)(di * di + dj * dj));  // :61 (f32)
      L0[(size_t)l * Pp + p] = fmaxf(ol, 1e-12f);                               // :63
      nbr[(size_t)l * Pp + p] = mask[ji * N + jj] ? pid[ji * N + jj] : -1;      // :276
    }
  auto* h = new ud_cloth();
  h->c.gdt = (float)((double)conf->gravity * (double)conf->dt);
  h->c.g = conf->gravity;
  h->c.dt = conf->dt;
  h->c.damp = expf(-(float)((double)conf->damping * (double)conf->dt));
  h->c.max_v = conf->max_v;
  h->c.eps = conf->small_num;
  h->c.n_mask = (float)P;
  h->c.P = P; h->c.Pp = Pp; h->c.S = conf->substeps;
  h->c.cell = (float)(1.0 / N);
  h->c.Ls = fmaxf((float)(1.0 / N) * sqrtf(1.0f), 1e-12f);
  h->c.Ld = fmaxf((float)(1.0 / N) * sqrtf(2.0f), 1e-12f);
  if (Pp > 1024) {   // widest index distance of a spring -> halo of the several-workgroups-per-env kernels
    int far = 0;
    for (int p = 0; p < P; ++p)
      for (int l = 0; l < 8; ++l) { const int j = nbr[(size_t)l * Pp + p]; if (j >= 0) far = std::max(far, std::abs(j - p)); }
    const int H = (far + 63) / 64 * 64;
    h->cl_W = (P + ud::CL_T - 1) / ud::CL_T;
    h->cl_H = (H >= 64 && H <= ud::CL_HMAX && h->cl_W <= 8) ? H : 0;
  }
  h->mode = conf->mode;
  if (h->mode < 0 || h->mode > 3) { ud::set_error("ud_cloth_create: mode must be 0, 1, 2 or 3"); delete h; return UD_ERR_INVALID; }
  if (h->mode == 3 && Pp > 512) {   // the restructured adjoint of bodies above 512 particles belongs to the several-workgroup (v2-order) kernels
    ud::set_error("ud_cloth_create: mode 3 (reference-order forward + restructured adjoint) covers bodies of at most 512 particles, P=%d", P);
    delete h; return UD_ERR_UNSUPPORTED;
  }
  if (conf->max_envs < 1) { ud::set_error("ud_cloth_create: max_envs = %d (handle-owned scratch is sized at create: give the largest B any call will pass)", conf->max_envs); delete h; return UD_ERR_INVALID; }
  h->max_envs = conf->max_envs;
  h->one_wg = conf->one_workgroup_per_env != 0;
  hipError_t e = hipGetDevice(&h->device);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_nbr, nbr.size() * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_L0, L0.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->d_nbr, nbr.data(), nbr.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_L0, L0.data(), L0.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && Pp > 1024) e = hipMalloc((void**)&h->d_timeouts, sizeof(int));
  if (e == hipSuccess && Pp > 1024) e = hipMemset(h->d_timeouts, 0, sizeof(int));
  if (e == hipSuccess && Pp > 1024) {   // the kernels for big bodies need more than the default 64 KB of dynamic LDS
    e = hipFuncSetAttribute((const void*)ud::cloth_big_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)9 * Pp * sizeof(float)));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ud::cloth_big_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)6 * Pp + 192 + 128) * sizeof(float)));
  }
  // handle-owned scratch of bodies above 1024 particles, for max_envs envs: the several-workgroup kernels' hand-off arena (per launch) or the
  // one-workgroup adjoint's cotangent parking area (per call) -- nothing is allocated in a rollout call
  if (e == hipSuccess && Pp > 1024) {
    if (cloth_use_cluster(h, h->max_envs)) {
      const int per = cloth_cluster_envs(h, h->max_envs);
      e = hipMalloc((void**)&h->d_arena, ud::cl_env_granules(h->c.Pp, h->cl_W) * sizeof(ud::cl_granule) * per);
      if (e == hipSuccess) h->arena_B = per;
    } else {
      e = hipMalloc((void**)&h->d_park, (size_t)h->max_envs * UD_BIG_PARK * h->c.Pp * sizeof(float));
      if (e == hipSuccess) h->park_B = h->max_envs;
    }
  }
  if (e != hipSuccess) {
    ud::set_error("ud_cloth_create: %s", hipGetErrorString(e));
    if (h->d_nbr) (void)hipFree(h->d_nbr);
    if (h->d_L0) (void)hipFree(h->d_L0);
    if (h->d_timeouts) (void)hipFree(h->d_timeouts);
    if (h->d_arena) (void)hipFree(h->d_arena);
    if (h->d_park) (void)hipFree(h->d_park);
    delete h;
    return UD_ERR_HIP;
  }
  *out = h;
  return UD_OK;
}

void ud_cloth_destroy(ud_cloth* h) {
  if (!h) return;
  (void)hipFree(h->d_nbr);
  if (h->d_park) (void)hipFree(h->d_park);
  if (h->d_arena) (void)hipFree(h->d_arena);
  if (h->d_timeouts) (void)hipFree(h->d_timeouts);
  (void)hipFree(h->d_L0);
  delete h;
}

int ud_cloth_num_particles(const ud_cloth* h) { return h ? h->c.P : UD_ERR_INVALID; }

int ud_cloth_launch_envs(const ud_cloth* h, int B) {
  if (!h || B < 1) return UD_ERR_INVALID;
  return cloth_use_cluster(h, B) ? cloth_cluster_envs(h, B) : B;
}

int ud_cloth_poll_timeouts(ud_cloth* h, void* stream) {
  if (!h) { ud::set_error("ud_cloth_poll_timeouts: null handle"); return UD_ERR_INVALID; }
  if (!h->d_timeouts) return 0;
  int n = 0;
  UD_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  UD_HIP_CHECK(hipMemcpy(&n, h->d_timeouts, sizeof(int), hipMemcpyDeviceToHost));
  if (n) UD_HIP_CHECK(hipMemset(h->d_timeouts, 0, sizeof(int)));
  if (n) ud::set_error("ud_cloth: %d workgroup(s) of the several-workgroup kernels gave up waiting for a sibling part since the last poll; "
                       "the outputs of those envs are NaN", n);
  return n;
}

size_t ud_cloth_ckpt_bytes(const ud_cloth* h, int B, int T) {
  if (!h || B < 0 || T < 0) return 0;
  return (size_t)B * ud::cloth_env_records(T, h->c.S) * ud::cloth_rec_floats(h->c.Pp) * sizeof(float);
}

int ud_cloth_rollout_fwd(ud_cloth* h, int B, int T, const float* x, const float* v, const float* prim,
                         const float* stiffness, const float* mu, const float* actions, float* x_out,
                         float* v_out, float* prim_out, float* x_list, float* v_list, float* prim_list,
                         void* ckpt, uint8_t* grasp, void* stream) {
  if (!h || !x || !v || !prim || !stiffness || !mu || !actions || !x_out || !v_out || !prim_out) {
    ud::set_error("ud_cloth_rollout_fwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1 || T < 1 || B > h->max_envs) { ud::set_error("ud_cloth_rollout_fwd: B=%d T=%d (max_envs=%d)", B, T, h->max_envs); return UD_ERR_INVALID; }
  ud::ClothFwdArgs a{};
  a.c = h->c; a.nbr = h->d_nbr; a.L0 = h->d_L0; a.B = B; a.T = T;
  a.x = x; a.v = v; a.prim = prim; a.k = stiffness; a.mu = mu; a.actions = actions;
  a.x_out = x_out; a.v_out = v_out; a.prim_out = prim_out; a.x_list = x_list; a.v_list = v_list;
  a.prim_list = prim_list; a.ckpt = (float*)ckpt; a.grasp = grasp;
  const size_t shmem = (size_t)2 * 3 * h->c.Pp * sizeof(float);
  if (cloth_use_cluster(h, B)) {
    const int per = cloth_cluster_envs(h, B);
    for (int b0 = 0; b0 < B; b0 += per) {
      ud::ClusterArgs q{};
      const int rc = cloth_cluster_arena(h, std::min(per, B - b0), (hipStream_t)stream, &q);
      if (rc != UD_OK) return rc;
      q.b0 = b0;
      ud::cloth_launch_fwd_cluster(a, q, (hipStream_t)stream);
    }
  } else if (h->c.Pp > 1024)
    hipLaunchKernelGGL(ud::cloth_big_fwd_kernel, dim3(B), dim3(UD_BIG_T), (size_t)9 * h->c.Pp * sizeof(float), (hipStream_t)stream, a);
  else if (h->mode == 2 && h->c.Pp <= 512)
    ud::cloth_launch_fwd_fast(a, (hipStream_t)stream);
  else if (h->mode == 0 && h->c.Pp <= 512)
    ud::cloth_launch_fwd_v2(a, (hipStream_t)stream);
  else if (h->mode == 3 && ud::cloth_ref_fast_ok(h->c))   // the literal order's bits from in-range exact divide / sqrt sequences (cloth_ref.hip)
    ud::cloth_launch_fwd_ref(a, (hipStream_t)stream);
  else if (h->c.Pp <= 512)   // mode 1 (and mode 3 with constants outside cloth_ref.hip's checks): the literal code, compiler's IEEE divide / sqrt
    hipLaunchKernelGGL(ud::cloth_rollout_fwd_kernel<512>, dim3(B), dim3(h->c.Pp), shmem, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ud::cloth_rollout_fwd_kernel<1024>, dim3(B), dim3(h->c.Pp), shmem, (hipStream_t)stream, a);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_cloth_rollout_bwd(ud_cloth* h, int B, int T, const void* ckpt, const float* stiffness, const float* mu,
                         const float* actions, const float* g_x, const float* g_v, const float* g_prim,
                         const float* g_x_list, const float* g_v_list, const float* g_prim_list, int normalize,
                         float* g_x0, float* g_v0, float* g_prim0, float* g_actions, float* g_stiffness,
                         float* g_mu, void* stream) {
  if (!h || !ckpt || !stiffness || !mu || !actions || !g_x || !g_v || !g_prim || !g_x0 || !g_v0 || !g_prim0 ||
      !g_actions || !g_stiffness || !g_mu) {
    ud::set_error("ud_cloth_rollout_bwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1 || T < 1 || B > h->max_envs) { ud::set_error("ud_cloth_rollout_bwd: B=%d T=%d (max_envs=%d)", B, T, h->max_envs); return UD_ERR_INVALID; }
  ud::ClothBwdArgs a{};
  a.c = h->c; a.nbr = h->d_nbr; a.L0 = h->d_L0; a.B = B; a.T = T;
  a.ckpt = (const float*)ckpt; a.k = stiffness; a.mu = mu; a.actions = actions;
  a.g_x = g_x; a.g_v = g_v; a.g_prim = g_prim; a.g_x_list = g_x_list; a.g_v_list = g_v_list;
  a.g_prim_list = g_prim_list; a.normalize = normalize;
  a.g_x0 = g_x0; a.g_v0 = g_v0; a.g_prim0 = g_prim0; a.g_actions = g_actions; a.g_k = g_stiffness; a.g_mu = g_mu;
  const size_t shmem = ((size_t)6 * h->c.Pp + 192 + 128) * sizeof(float);
  if (cloth_use_cluster(h, B)) {
    const int per = cloth_cluster_envs(h, B);
    for (int b0 = 0; b0 < B; b0 += per) {
      ud::ClusterArgs q{};
      const int rc = cloth_cluster_arena(h, std::min(per, B - b0), (hipStream_t)stream, &q);
      if (rc != UD_OK) return rc;
      q.b0 = b0;
      ud::cloth_launch_bwd_cluster(a, q, (hipStream_t)stream);
    }
  } else if (h->c.Pp > 1024) {
    if (h->park_B < B) { ud::set_error("ud_cloth_rollout_bwd: the adjoint's scratch holds %d envs, %d asked", h->park_B, B); return UD_ERR_INVALID; }
    hipLaunchKernelGGL(ud::cloth_big_bwd_kernel, dim3(B), dim3(UD_BIG_T), shmem, (hipStream_t)stream, a, h->d_park);
  } else if (h->mode != 1 && h->c.Pp <= 512)
    ud::cloth_launch_bwd_fast(a, (hipStream_t)stream);
  else if (h->c.Pp <= 512)
    hipLaunchKernelGGL(ud::cloth_rollout_bwd_kernel<512>, dim3(B), dim3(h->c.Pp), shmem, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ud::cloth_rollout_bwd_kernel<1024>, dim3(B), dim3(h->c.Pp), shmem, (hipStream_t)stream, a);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

}  // extern "C"
