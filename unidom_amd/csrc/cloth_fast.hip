// Fast cloth rollout kernels for gfx950 (the default path; the reference-operation-order kernels are in cloth.hip).
//
// Same mapping as cloth.hip (one workgroup per env, one particle per lane, whole T x substeps rollout in one
// launch, per-substep checkpoints in HBM) but the arithmetic is restructured for the CU, not for bit-identity:
//   forward   the spring force is evaluated as  f = r * (k/L0 - k/|r|)  (= k r/|r| (|r|-L0)/L0 of
//             cloth_simulator.py:267-268) with v_rsq_f32, i.e. 1 transcendental + 4 FMAs per link and component
//             triple instead of 6 IEEE divisions + 1 IEEE sqrt; x is staged as float4 in LDS (one ds_read_b128
//             per neighbour); the dead static-friction branch (:293-306, never taken because
//             sqrt(.+small_num) > small_num) is dropped; FMA contraction is on.
//   backward  the reference normalises cotangents six times per substep (norm_grad, :189-194, applied at
//             :223-224 twice and :331-334).  Every map between two normalisations is linear, so all six norms
//             are functions of NINE block-wide sums of the incoming cotangent that can be taken BEFORE the
//             stencil barrier: |gx|^2, |gv|^2, |Dx gx|^2, |Dv gv|^2, <Dv gv, Dv gx>, |Dv gx|^2 and the last three
//             restricted to the particles gripper 1 holds (Dx, Dv = the clip masks of :326-327; Dv is read off
//             the NEXT checkpoint record, whose v is clip(v5)).  That turns 3 dependent reduction rounds + 6
//             barriers per substep into 1 round + 2 barriers.
// Results agree with the reference-order kernels / the CPU oracle to f32 round-off (tests/test_cloth_gpu.py,
// tolerances written there); the discrete grasp test |x - pos| <= radius keeps its exact form.
#include "cloth_fast_adj.h"

namespace ud {

struct FastInter {
  float F1, cF, muF, xV, yV, isV, tf;   // friction block
  float r0[8], r1[8], r2[8];            // link vectors (reused by the spring adjoint)
  float w[8];                           // 1/L0 - 1/|r|            (spring coefficient / k)
  float c2k[8];                         // k / |r|^3, or 0 where clip(|r|^2, 1e-12) is active
};

// spring + gravity + ground friction + damping: (x, v, neighbours in X4) -> v3 ; keeps the adjoint's inputs
template <bool KEEP>
// nbs[l] = neighbour index, or the particle itself where the lattice has no neighbour: then r == 0 exactly and the
// (finite) coefficient multiplies zeros, so neither the force nor its adjoint needs a select.
__device__ __forceinline__ void force_fast(const ClothConst& c, const int* nbs, const float4* X4, float k, float iLs,
                                           float iLd, float mu, const float* x, const float* v, float* v3, FastInter* in) {
  float F0 = 0.f, F1 = 0.f, F2 = 0.f;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const float4 xj = X4[nbs[l]];
    const float r0 = xj.x - x[0], r1 = xj.y - x[1], r2 = xj.z - x[2];
    const float s2 = r0 * r0 + r1 * r1 + r2 * r2;
    const float inv = rsq(fmaxf(s2, 1e-12f));
    const float w = ((l < 4) ? iLs : iLd) - inv;
    const float coef = k * w;
    F0 += coef * r0; F1 += coef * r1; F2 += coef * r2;
    if (KEEP) {
      in->r0[l] = r0; in->r1[l] = r1; in->r2[l] = r2; in->w[l] = w;
      in->c2k[l] = (s2 > 1e-12f) ? k * inv * inv * inv : 0.f;
    }
  }
  F1 -= c.g;                                        // :278
  const float v1y = v[1] - c.gdt;                   // :259
  const bool fm = x[1] <= c.eps;                    // :281
  const float cF = fminf(F1, 0.f);
  const float muF = -(mu * cF);                     // :282
  const float xV = v[0], yV = v[2];
  const float isV = rsq(xV * xV + yV * yV + c.eps); // :285
  const float tf = fm ? muF * isV : 0.f;            // :288-290 (sV > small_num always holds)
  const float Ax = F0 - tf * xV, Az = F2 - tf * yV;
  v3[0] = (xV + Ax * c.dt) * c.damp;                // :308-309
  v3[1] = (v1y + F1 * c.dt) * c.damp;
  v3[2] = (yV + Az * c.dt) * c.damp;
  if (KEEP) { in->F1 = F1; in->cF = cF; in->muF = muF; in->xV = xV; in->yV = yV; in->isV = isV; in->tf = tf; }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) cloth_rollout_fwd_fast_kernel(ClothFwdArgs a) {
  extern __shared__ float4 lds4[];  // [2][Pp]
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
  const float iLs = 1.f / c.Ls, iLd = 1.f / c.Ld;
  GraspThr th0, th1;
  th0.init(ps[3]); th1.init(ps[7]);
  const size_t rec = cloth_rec_floats(Pp);
  float* ckb = a.ckpt ? a.ckpt + (size_t)b * cloth_env_records(T, S) * rec : nullptr;
  unsigned step = 0;
  for (int t = 0; t < T; ++t) {
    float act[8];
    macro_action_f(a.actions + ((size_t)t * B + b) * 8, act);
    for (int s = 0; s < S; ++s, ++step) {
      float4* X4 = lds4 + (step & 1u) * Pp;
      X4[i] = make_float4(x[0], x[1], x[2], 0.f);
      if (ckb) {
        float* r = ckb + (size_t)step * rec;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * Pp + i] = x[d]; r[(3 + d) * Pp + i] = v[d]; }
        if (i == 0) {
#pragma unroll
          for (int d = 0; d < 8; ++d) r[6 * Pp + d] = ps[d];
        }
      }
      __syncthreads();
      float vv[3], x2[3];
      bool m0, m1;
      FastInter dummy;
      force_fast<false>(c, nbs, X4, k, iLs, iLd, mu, x, v, vv, &dummy);
      grip_own(x, ps, act, th0.at(step == 0), th1.at(step == 0), m0, m1, x2);
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
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int d = 0; d < 4; ++d) ps[g * 4 + d] = clipf(ps[g * 4 + d] + (d < 3 ? act[g * 4 + d] : 0.f), 0.f, 1.f);  // :322-323
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

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int UD_CLOTH_MAXP = 1024 + 1;   // LDS plane stride (floats; the kernels refuse Pp > 1024).  Odd on purpose: a stride that is a
                                          // multiple of 64 lets the compiler fuse the x and y reads of one neighbour into ds_read2st64,
                                          // which then needs register moves to regroup them by link pair

__global__ void __launch_bounds__(512) cloth_rollout_bwd_fast_kernel(ClothBwdArgs a) {
  extern __shared__ float ldsf[];  // Xs[3][MAXP] | Gs[3][MAXP] | red[2][16][UD_RSTR] | mac[16*8]
  const ClothConst c = a.c;
  const int i = threadIdx.x, b = blockIdx.x;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T;
  const int nw = Pp >> 6, lane = i & 63, wv = i >> 6;
  const bool live = i < P;
  const bool norm = a.normalize != 0;
  // single-buffered: every X read sits between barrier 1 and barrier 2 and the next X write comes after barrier 2;
  // every G read sits between barrier 2 and the next barrier 1 and the next G write comes after that barrier
  float* Xs = ldsf;
  float* Gs = ldsf + 3 * UD_CLOTH_MAXP;
  float* red = ldsf + 6 * UD_CLOTH_MAXP;
  float* mac = red + 2 * 16 * UD_RSTR;
  int nbs[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { const int j = a.nbr[l * Pp + i]; nbs[l] = j >= 0 ? j : i; }
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = a.g_x[((size_t)b * P + i) * 3 + d]; gv[d] = a.g_v[((size_t)b * P + i) * 3 + d]; }
  }
  // primitive cotangent: component d lives in lane d of wave 0 (lanes 0-3 gripper 0, 4-7 gripper 1)
  float gpl = (i < 8) ? a.g_prim[b * 8 + i] : 0.f;
  const bool pm3 = (i < 8) && ((i & 3) < 3);
  const float inm = 1.f / c.n_mask;
  const float k = a.k[b], mu = a.mu[b];
  const float Ls = c.Ls, Ld = c.Ld;
  const f2 iL2 = {1.f / Ls, 1.f / Ld};
  float gk = 0.f, gmu = 0.f;
  const size_t rec = cloth_rec_floats(Pp);
  const float* ck = a.ckpt + (size_t)b * cloth_env_records(T, S) * rec;
  GraspThr th0, th1;   // from record 0 = the rollout's input primitives, exactly what the forward derived them from
  th0.init(ck[6 * Pp + 3]); th1.init(ck[6 * Pp + 7]);
  // records: `cur` = input of the substep being reversed, `vnext` = v of the record after it (= clip(v5))
  // The primitive rows of the records are read through the constant address space (scalar loads into SGPRs: the
  // checkpoints are read-only for this kernel), straight into `ps` once the previous substep is done with it.
  typedef const __attribute__((address_space(4))) float* cfptr;
  float vnext[3], nx[3], nv[3], ps[8], psl;
  {
    const float* r = ck + (size_t)T * S * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) vnext[d] = r[(3 + d) * Pp + i];
    r = ck + ((size_t)T * S - 1) * rec;
#pragma unroll
    for (int d = 0; d < 3; ++d) { nx[d] = r[d * Pp + i]; nv[d] = r[(3 + d) * Pp + i]; }
#pragma unroll
    for (int d = 0; d < 8; ++d) ps[d] = ((cfptr)r)[6 * Pp + d];
    psl = r[6 * Pp + (i & 7)];
  }
  for (int q = i; q < 2 * 16 * UD_RSTR; q += Pp) red[q] = 0.f;   // slots of waves this launch does not have are read as zeros
  __syncthreads();
  unsigned step = 0;
  const float* rp = ck + ((size_t)T * S - 1) * rec;   // record held in nx/nv/nps
  for (int t = T - 1; t >= 0; --t) {
    if (live) {
      const size_t o = (((size_t)t * B + b) * P + i) * 3;
      if (a.g_x_list) { gx[0] += a.g_x_list[o]; gx[1] += a.g_x_list[o + 1]; gx[2] += a.g_x_list[o + 2]; }
      if (a.g_v_list) { gv[0] += a.g_v_list[o]; gv[1] += a.g_v_list[o + 1]; gv[2] += a.g_v_list[o + 2]; }
    }
    if (a.g_prim_list && i < 8) gpl += a.g_prim_list[((size_t)t * B + b) * 8 + i];
    const float* a8 = a.actions + ((size_t)t * B + b) * 8;
    float act[8], ga[8];
    macro_action_f(a8, act);
#pragma unroll
    for (int d = 0; d < 8; ++d) ga[d] = 0.f;
    const float addl = pm3 ? clipf(a8[i & 7], -2.0f, 2.0f) * (1.0f / 50.0f) : 0.f;   // this lane's component of the primitive move
    float gaP = 0.f;
    for (int s = S - 1; s >= 0; --s, ++step) {
      float x[3], v[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = nx[d]; v[d] = nv[d]; }
      {  // prefetch the record this loop consumes next
        rp = (rp != ck) ? rp - rec : rp;            // uniform; the last iteration re-reads record 0 and ignores it
        const float* r = rp;
#pragma unroll
        for (int d = 0; d < 3; ++d) { nx[d] = r[(unsigned)(d * Pp + i)]; nv[d] = r[(unsigned)((3 + d) * Pp + i)]; }
      }
      const unsigned par = step & 1u;
      float* rd = red + par * 16 * UD_RSTR;
      Xs[i] = x[0]; Xs[UD_CLOTH_MAXP + i] = x[1]; Xs[2 * UD_CLOTH_MAXP + i] = x[2];
      // ---- own-particle forward pieces and the nine sums (no neighbour data needed) ----
      bool m0, m1;
      float x2[3];
      grip_own(x, ps, act, th0.at(t == 0 && s == 0), th1.at(t == 0 && s == 0), m0, m1, x2);
      m0 = m0 && live; m1 = m1 && live;
      float av[3], bv[3], bx[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float Dx = clip_grad_lt(x2[d], 0.f, 1.f);
        const float Dv = (fabsf(vnext[d]) < c.max_v) ? 1.f : 0.f;
        av[d] = Dx * gx[d]; bv[d] = Dv * gv[d]; bx[d] = Dv * gx[d];
      }
      float sm[UD_NSUM];
      sm[0] = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
      sm[1] = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2];
      sm[2] = av[0] * av[0] + av[1] * av[1] + av[2] * av[2];
      sm[3] = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
      sm[4] = bv[0] * bx[0] + bv[1] * bx[1] + bv[2] * bx[2];
      sm[5] = bx[0] * bx[0] + bx[1] * bx[1] + bx[2] * bx[2];
      sm[6] = m1 ? sm[3] : 0.f; sm[7] = m1 ? sm[4] : 0.f; sm[8] = m1 ? sm[5] : 0.f;
      if (norm) {
        const float sm8[8] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5], sm[6], sm[7]};
        const float w8 = wave_sum8_t(sm8, lane);
        if ((lane & 0x2C) == 0) rd[wv * UD_RSTR + (((lane >> 2) & 4) | (lane & 3))] = w8;
        if (__builtin_amdgcn_ballot_w64(m1) != 0) {   // wave-uniform: gripper 1 holds something in this wave
          const float w = wave_sum_l63(sm[8]);
          if (lane == 63) rd[wv * UD_RSTR + 8] = w;
        } else if (lane == 63) {
          rd[wv * UD_RSTR + 8] = 0.f;
        }
      }
      __syncthreads();   // barrier 1: X4 and the wave partials are visible
      float sx = 1.f, sv = 1.f, sA = 1.f, sB = 1.f, s3x = 1.f, s3v = 1.f;   // cumulative scale factors
      if (norm) {
        // row g of the wave adds the partials of waves g, g+4, g+8, g+12 (slots of absent waves stay zero), then the
        // four rows are added position by position: one LDS round trip instead of a dependent read per wave
        float tot = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) tot += rd[((lane >> 4) + 4 * m) * UD_RSTR + (lane & 15)];
        tot = rows_sum4(tot);
        float T_[UD_NSUM];
#pragma unroll
        for (int q = 0; q < UD_NSUM; ++q) T_[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), q));
        sx = inv_norm(T_[0], inm);                                   // :331
        sv = inv_norm(T_[1], inm);                                   // :332
        const float cx = c.dt * sx;
        const float n2x = sx * sx * T_[2];                                // |g_x2|^2
        const float n2v = sv * sv * T_[3] + 2.f * sv * cx * T_[4] + cx * cx * T_[5];
        sA = inv_norm(n2x, inm);                                     // :223 (gripper 1)
        sB = inv_norm(n2v, inm);                                     // :224
        const float n3x = sA * sA * n2x;
        const float s1 = act[7];
        const float nm = sv * sv * T_[6] + 2.f * sv * cx * T_[7] + cx * cx * T_[8];
        const float n3v = sB * sB * (n2v - (1.f - s1 * s1) * nm);
        s3x = inv_norm(n3x, inm);                                    // :223 (gripper 0)
        s3v = inv_norm(fmaxf(n3v, 0.f), inm);                        // :224
        // primitives (:333-334): 4-vector norms, uniform; only wave 0 carries the primitive cotangent
        if (wv == 0) {
          float n2 = gpl * gpl;
          n2 += dpp_f<0xB1>(n2);
          n2 += dpp_f<0x4E>(n2);   // quad total = this gripper's 4-vector norm^2
          gpl *= inv_norm(n2, inm);
        }
      }
      // ---- neighbour-dependent forward recompute ----
      float v3[3], v4[3];
      PairInter in;
      force_pairs<UD_CLOTH_MAXP>(c, nbs, Xs, k, iL2, mu, x, v, v3, &in);
#pragma unroll
      for (int d = 0; d < 3; ++d) v4[d] = m0 ? act[3] * v3[d] : v3[d];
      // ---- reverse: clip (:326-329) and the two grippers (:313-314) with their normalisations folded in ----
      float gx2n[3], gv5n[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        gx2n[d] = sA * (sx * av[d]);
        gv5n[d] = sB * (sv * bv[d] + (c.dt * sx) * bx[d]);
      }
      {  // gripper 1, branch-free: masks as 0/1 factors
        const float s1 = act[7], m1f = m1 ? 1.f : 0.f, sc1 = m1 ? s1 : 1.f, h1 = (1.f - s1) * m1f;
        const float dotv = v4[0] * gv5n[0] + v4[1] * gv5n[1] + v4[2] * gv5n[2];
        const float dotx = gx2n[0] * act[4] + gx2n[1] * act[5] + gx2n[2] * act[6];
        ga[7] += (dotv - dotx) * m1f;
#pragma unroll
        for (int d = 0; d < 3; ++d) { ga[4 + d] += gx2n[d] * h1; gv5n[d] *= sc1; }
      }
      float gxd[3], gv3[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { gxd[d] = s3x * gx2n[d]; gv3[d] = s3v * gv5n[d]; }
      {  // gripper 0
        const float s0 = act[3], m0f = m0 ? 1.f : 0.f, sc0 = m0 ? s0 : 1.f, h0 = (1.f - s0) * m0f;
        const float dotv = v3[0] * gv3[0] + v3[1] * gv3[1] + v3[2] * gv3[2];
        const float dotx = gxd[0] * act[0] + gxd[1] * act[1] + gxd[2] * act[2];
        ga[3] += (dotv - dotx) * m0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) { ga[d] += gxd[d] * h0; gv3[d] *= sc0; }
      }
      // primitives (:322-323), uniform; counted once (lane 0) in the action accumulators
      if (wv == 0) {
        gpl *= clip_grad_lt(psl + addl, 0.f, 1.f);
        gaP += pm3 ? gpl : 0.f;
      }
      // ---- v3 = (v1 + F dt) damp ; ground friction (:281-290) ----
      float gF[3];
      {
        const float g2x = gv3[0] * c.damp, g2y = gv3[1] * c.damp, g2z = gv3[2] * c.damp;
        const float gAx = g2x * c.dt, gFy = g2y * c.dt, gAz = g2z * c.dt;
        const float gt = -(gAx * in.xV + gAz * in.yV);
        float gxV = -gAx * in.tf, gyV = -gAz * in.tf;
        const bool fm = x[1] <= c.eps;
        const float gmuF = fm ? gt * in.isV : 0.f;
        const float gisV = fm ? gt * in.muF : 0.f;
        const float gq = -0.5f * in.isV * in.isV * in.isV * gisV;
        gxV += 2.f * in.xV * gq; gyV += 2.f * in.yV * gq;
        gmu += live ? -gmuF * in.cF : 0.f;
        const float gcF = -gmuF * mu;
        const float cfm = (in.F1 < 0.f) ? 1.f : ((in.F1 == 0.f) ? 0.5f : 0.f);
        gF[0] = live ? gAx : 0.f;
        gF[1] = live ? gFy + gcF * cfm : 0.f;
        gF[2] = live ? gAz : 0.f;
        gv[0] = g2x + gxV; gv[1] = g2y; gv[2] = g2z + gyV;   // v1 = v - (0, g dt, 0)
      }
      Gs[i] = gF[0]; Gs[UD_CLOTH_MAXP + i] = gF[1]; Gs[2 * UD_CLOTH_MAXP + i] = gF[2];
      __syncthreads();   // barrier 2: Gs visible
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = ((cfptr)rp)[6 * Pp + d];   // next substep's primitives (rp already moved)
      psl = rp[(unsigned)(6 * Pp + (i & 7))];
      // ---- spring adjoint, gather form: g_x_i = gxd + sum_l J_il (gF_j - gF_i) ----
      f2 A0 = {gxd[0], 0.f}, A1 = {gxd[1], 0.f}, A2 = {gxd[2], 0.f};
      f2 h0[4], h1[4], h2[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // all 24 LDS reads in flight before the first use
        const int ja = nbs[p], jb = nbs[p + 4];       // a missing neighbour reads gF itself: d = 0 and r = 0
        h0[p] = f2{Gs[ja], Gs[jb]};
        h1[p] = f2{Gs[UD_CLOTH_MAXP + ja], Gs[UD_CLOTH_MAXP + jb]};
        h2[p] = f2{Gs[2 * UD_CLOTH_MAXP + ja], Gs[2 * UD_CLOTH_MAXP + jb]};
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f2 d0 = h0[p] - gF[0], d1 = h1[p] - gF[1], d2 = h2[p] - gF[2];
        const f2 r0 = in.r0[p], r1 = in.r1[p], r2 = in.r2[p];
        const f2 rd_ = r0 * d0 + r1 * d1 + r2 * d2;
        const f2 c1 = k * in.w[p];
        const f2 c2 = in.c2k[p] * rd_;
        A0 += c1 * d0 + c2 * r0; A1 += c1 * d1 + c2 * r1; A2 += c1 * d2 + c2 * r2;
      }
      const float ax0 = A0.x + A0.y, ax1 = A1.x + A1.y, ax2 = A2.x + A2.y;
      gk += gF[0] * in.S0 + gF[1] * in.S1 + gF[2] * in.S2;   // sum_l w_l (r_l . gF) = gF . S
      gx[0] = ax0; gx[1] = ax1; gx[2] = ax2;
#pragma unroll
      for (int d = 0; d < 3; ++d) vnext[d] = v[d];   // this substep's input v is the previous substep's clip(v5)
    }
    // macro-step boundary: robot_step's action transform (:168-169)
    {
      __syncthreads();
      {
        const float w8 = wave_sum8_t(ga, lane);
        if ((lane & 0x2C) == 0) mac[wv * 8 + (((lane >> 2) & 4) | (lane & 3))] = w8;
      }
      __syncthreads();
      if (i < 8) {
        float tot = 0.f;
        for (int q = 0; q < nw; ++q) tot += mac[q * 8 + i];
        tot += gaP;
        const int d = i & 3;
        a.g_actions[((size_t)t * B + b) * 8 + i] = (d < 3) ? tot * (1.0f / 50.0f) * clip_grad(a8[i], -2.0f, 2.0f) : tot;
      }
    }
  }
  if (live) {
    const size_t o = ((size_t)b * P + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.g_x0[o + d] = gx[d]; a.g_v0[o + d] = gv[d]; }
  }
  __syncthreads();
  {
    const float w0 = wave_sum_l63(gk), w1 = wave_sum_l63(gmu);
    if (lane == 63) { mac[wv * 2] = w0; mac[wv * 2 + 1] = w1; }
  }
  __syncthreads();
  if (i < 8) a.g_prim0[b * 8 + i] = gpl;
  if (i == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int q = 0; q < nw; ++q) { t0 += mac[q * 2]; t1 += mac[q * 2 + 1]; }
    a.g_k[b] = t0;
    a.g_mu[b] = t1;
  }
}

void cloth_launch_fwd_fast(const ClothFwdArgs& a, hipStream_t stream) {
  const size_t shmem = (size_t)2 * a.c.Pp * sizeof(float4);
  hipLaunchKernelGGL(cloth_rollout_fwd_fast_kernel, dim3(a.B), dim3(a.c.Pp), shmem, stream, a);
}

void cloth_launch_bwd_fast(const ClothBwdArgs& a, hipStream_t stream) {
  const size_t shmem = (size_t)(6 * UD_CLOTH_MAXP + 2 * 16 * UD_RSTR + 16 * 8) * sizeof(float);
  hipLaunchKernelGGL(cloth_rollout_bwd_fast_kernel, dim3(a.B), dim3(a.c.Pp), shmem, stream, a);
}

}  // namespace ud
