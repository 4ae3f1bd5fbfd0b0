// Cloth adjoint for bodies of more than 1024 particles, several workgroups per env (protocol and rationale: cloth_cluster.h).
// Per particle this is cloth_fast.hip's restructured adjoint -- the nine-block-sum closed form of the six norm_grad
// rescalings, force_pairs, the gather-form spring adjoint -- on a 512-particle part per workgroup.  Per substep:
//   own-particle pieces + the nine sums of the part              (no neighbour data)
//   barrier 1; part totals -> SE slot; force_pairs recompute     (hides the hand-off latency)
//   poll the W slots: cluster totals, identical in every part    -> the six scale factors
//   reverse of clip / grippers / friction -> gF -> LDS + GE granules; poll the halo gF of the parts below and above
//   barrier 2; spring adjoint (gather over the 8 links)
// Checkpoint records are read straight from HBM (own particle + one halo particle per lane of the first waves), one
// substep ahead.  The reductions that only the host sees (action cotangents per macro step, stiffness / mu) go through
// the SA slots; every part polls them (which is what makes their double-buffering safe), part 0 writes the result.
#include "cloth_cluster.h"
#include "cloth_fast_adj.h"

namespace ud {

// Cluster totals of a slot array [W][CL_SLOT] (W <= 8): lane (row = lane >> 4, q = lane & 15) adds the granules of parts
// row and row + 4, then the four rows are added position by position -- a fixed order, the same in every part, own slot
// included, so every part ends up with bit-identical totals.  Wave-uniform bounded poll; false = gave up.
__device__ __forceinline__ bool cl_poll_slots(const cl_granule* slots, int W, unsigned tag, int lane, float& tot) {
  const int m0 = lane >> 4, m1 = m0 + 4, qq = lane & 15;
  bool ok0 = m0 >= W, ok1 = m1 >= W;
  float v0 = 0.f, v1 = 0.f;
  bool done = false;
  for (unsigned spins = 0;; ++spins) {
    if (!ok0) ok0 = cl_get(slots + m0 * CL_SLOT + qq, tag, v0);
    if (!ok1) ok1 = cl_get(slots + m1 * CL_SLOT + qq, tag, v1);
    if (__builtin_amdgcn_ballot_w64(!(ok0 && ok1)) == 0) { done = true; break; }
    if (spins > CL_SPIN_LIMIT) break;
    __builtin_amdgcn_s_sleep(1);
  }
  tot = rows_sum4((m0 < W ? v0 : 0.f) + (m1 < W ? v1 : 0.f));
  return done;
}

__global__ void __launch_bounds__(CL_T) cloth_cluster_bwd_kernel(ClothBwdArgs a, ClusterArgs q) {
  extern __shared__ float ldsf[];  // Xs[3][CL_STRIDE] | Gs[3][CL_STRIDE] | red[2][8][UD_RSTR] | mac[8*8] | bail[2]
  int bl, w;
  cl_decode(q.W, bl, w);
  if (bl >= q.Bl) return;
  const int b = q.b0 + bl;
  const ClothConst c = a.c;
  const int P = c.P, Pp = c.Pp, S = c.S, B = a.B, T = a.T, W = q.W;
  const int i = threadIdx.x, base = w * CL_T, gi = base + i;
  const int lane = i & 63, wv = i >> 6;
  const bool inp = gi < Pp, live = gi < P;
  const int lo = max(0, base - q.H), hi = min(Pp, base + CL_T + q.H);
  const int nlo = base - lo, nhi = max(0, hi - (base + CL_T));
  const int li = gi - lo;
  const bool hl = i < nlo + nhi;
  const int hidx = i < nlo ? lo + i : base + CL_T + (i - nlo);
  const int hli = hidx - lo;
  const bool norm = a.normalize != 0;
  float* Xs = ldsf;
  float* Gs = ldsf + 3 * CL_STRIDE;
  float* red = ldsf + 6 * CL_STRIDE;
  float* mac = red + 2 * 8 * UD_RSTR;
  int* bail = (int*)(mac + 64);
  int nbs[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { const int j = inp ? a.nbr[l * Pp + gi] : -1; nbs[l] = (j >= 0 ? j : gi) - lo; }
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = a.g_x[((size_t)b * P + gi) * 3 + d]; gv[d] = a.g_v[((size_t)b * P + gi) * 3 + d]; }
  }
  // primitive cotangent: component d lives in lane d of wave 0 (lanes 0-3 gripper 0, 4-7 gripper 1); every part carries
  // the same copy (it only depends on the cluster totals), part 0 reports it
  float gpl = (i < 8) ? a.g_prim[b * 8 + i] : 0.f;
  const bool pm3 = (i < 8) && ((i & 3) < 3);
  const float inm = 1.f / c.n_mask;
  const float k = a.k[b], mu = a.mu[b];
  const f2 iL2 = {1.f / c.Ls, 1.f / c.Ld};
  float gk = 0.f, gmu = 0.f;
  const size_t rec = cloth_rec_floats(Pp);
  const float* ck = a.ckpt + (size_t)b * cloth_env_records(T, S) * rec;
  cl_granule* ar = q.arena + (size_t)bl * cl_env_granules(Pp, W);
  cl_granule* ge = ar + (size_t)6 * Pp;                  // GE[2][3][Pp]
  cl_granule* se = ge + (size_t)6 * Pp;                  // SE[2][W][CL_SLOT]
  cl_granule* sa = se + (size_t)2 * W * CL_SLOT;         // SA[2][W][CL_SLOT]
  GraspThr th0, th1;   // from record 0 = the rollout's input primitives, exactly what the forward derived them from
  th0.init(ck[6 * Pp + 3]); th1.init(ck[6 * Pp + 7]);
  typedef const __attribute__((address_space(4))) float* cfptr;
  float vnext[3] = {0.f, 0.f, 0.f}, nx[3] = {0.f, 0.f, 0.f}, nv[3] = {0.f, 0.f, 0.f}, hx[3] = {0.f, 0.f, 0.f}, ps[8], psl;
  {
    const float* r = ck + (size_t)T * S * rec;
    if (inp) {
#pragma unroll
      for (int d = 0; d < 3; ++d) vnext[d] = r[(3 + d) * Pp + gi];
    }
    r = ck + ((size_t)T * S - 1) * rec;
    if (inp) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { nx[d] = r[d * Pp + gi]; nv[d] = r[(3 + d) * Pp + gi]; }
    }
    if (hl) {
#pragma unroll
      for (int d = 0; d < 3; ++d) hx[d] = r[d * Pp + hidx];
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) ps[d] = ((cfptr)r)[6 * Pp + d];
    psl = r[6 * Pp + (i & 7)];
  }
  for (int e = i; e < 2 * 8 * UD_RSTR; e += CL_T) red[e] = 0.f;
  if (i < 2) bail[i] = 0;
  __syncthreads();
  unsigned step = 0;
  bool dead = false;
  const float* rp = ck + ((size_t)T * S - 1) * rec;   // record held in nx / nv / hx
  for (int t = T - 1; t >= 0 && !dead; --t) {
    if (live) {
      const size_t o = (((size_t)t * B + b) * P + gi) * 3;
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
      const unsigned tag = step + 1u, par = step & 1u;
      float x[3], v[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = nx[d]; v[d] = nv[d]; }
      Xs[li] = x[0]; Xs[CL_STRIDE + li] = x[1]; Xs[2 * CL_STRIDE + li] = x[2];
      if (hl) { Xs[hli] = hx[0]; Xs[CL_STRIDE + hli] = hx[1]; Xs[2 * CL_STRIDE + hli] = hx[2]; }
      {  // prefetch the record this loop consumes next (own particle + this lane's halo particle)
        rp = (rp != ck) ? rp - rec : rp;            // uniform; the last iteration re-reads record 0 and ignores it
        const float* r = rp;
        if (inp) {
#pragma unroll
          for (int d = 0; d < 3; ++d) { nx[d] = r[(unsigned)(d * Pp + gi)]; nv[d] = r[(unsigned)((3 + d) * Pp + gi)]; }
        }
        if (hl) {
#pragma unroll
          for (int d = 0; d < 3; ++d) hx[d] = r[(unsigned)(d * Pp + hidx)];
        }
      }
      float* rd = red + par * 8 * UD_RSTR;
      // ---- own-particle forward pieces and the nine sums of this part (no neighbour data needed) ----
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
      if (norm) {
        float sm[UD_NSUM];
        sm[0] = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
        sm[1] = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2];
        sm[2] = av[0] * av[0] + av[1] * av[1] + av[2] * av[2];
        sm[3] = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
        sm[4] = bv[0] * bx[0] + bv[1] * bx[1] + bv[2] * bx[2];
        sm[5] = bx[0] * bx[0] + bx[1] * bx[1] + bx[2] * bx[2];
        sm[6] = m1 ? sm[3] : 0.f; sm[7] = m1 ? sm[4] : 0.f; sm[8] = m1 ? sm[5] : 0.f;
        const float sm8[8] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5], sm[6], sm[7]};
        const float w8 = wave_sum8_t(sm8, lane);
        if ((lane & 0x2C) == 0) rd[wv * UD_RSTR + (((lane >> 2) & 4) | (lane & 3))] = w8;
        if (__builtin_amdgcn_ballot_w64(m1) != 0) {   // wave-uniform: gripper 1 holds something in this wave
          const float ws = wave_sum_l63(sm[8]);
          if (lane == 63) rd[wv * UD_RSTR + 8] = ws;
        } else if (lane == 63) {
          rd[wv * UD_RSTR + 8] = 0.f;
        }
      }
      __syncthreads();   // barrier 1: the X window and the wave partials are visible
      if (norm && wv == 0) {
        // part totals (rows 0-3 of the wave add the partials of waves r and r + 4, then the rows are added), published as
        // one 128-byte line of granules.  SE is double-buffered by step parity: a part reaches the sums of step n + 2
        // only after every part has published those of step n + 1, i.e. after every part has finished reading step n's.
        float tot = rd[(lane >> 4) * UD_RSTR + (lane & 15)] + rd[((lane >> 4) + 4) * UD_RSTR + (lane & 15)];
        tot = rows_sum4(tot);
        if (lane < CL_SLOT) cl_put(se + ((size_t)par * W + w) * CL_SLOT + lane, tot, tag);
      }
      // ---- neighbour-dependent forward recompute (needs X only: it runs while the sums travel) ----
      float v3[3], v4[3];
      PairInter in;
      force_pairs<CL_STRIDE>(c, nbs, Xs, k, iL2, mu, x, v, v3, &in);
#pragma unroll
      for (int d = 0; d < 3; ++d) v4[d] = m0 ? act[3] * v3[d] : v3[d];
      float sx = 1.f, sv = 1.f, sA = 1.f, sB = 1.f, s3x = 1.f, s3v = 1.f;   // cumulative scale factors
      bool okp = true;
      if (norm) {
        float tot;
        okp = cl_poll_slots(se + (size_t)par * W * CL_SLOT, W, tag, lane, tot);
        float T_[UD_NSUM];
#pragma unroll
        for (int e = 0; e < UD_NSUM; ++e) T_[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), e));
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
        if (wv == 0) {   // primitives (:333-334): 4-vector norms
          float n2 = gpl * gpl;
          n2 += dpp_f<0xB1>(n2);
          n2 += dpp_f<0x4E>(n2);   // quad total = this gripper's 4-vector norm^2
          gpl *= inv_norm(n2, inm);
        }
      }
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
      if (wv == 0) {   // primitives (:322-323); part 0 counts them in the action accumulators
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
      Gs[li] = gF[0]; Gs[CL_STRIDE + li] = gF[1]; Gs[2 * CL_STRIDE + li] = gF[2];
      // GE is double-buffered by step parity: a part writes gF(n + 2) only after it has consumed its neighbours'
      // gF(n + 1), which they published after polling gF(n)
      cl_granule* gep = ge + (size_t)par * 3 * Pp;
      if (inp) { cl_put(gep + gi, gF[0], tag); cl_put(gep + Pp + gi, gF[1], tag); cl_put(gep + 2 * (size_t)Pp + gi, gF[2], tag); }
#pragma unroll
      for (int d = 0; d < 8; ++d) ps[d] = ((cfptr)rp)[6 * Pp + d];   // next substep's primitives (rp already moved)
      psl = rp[(unsigned)(6 * Pp + (i & 7))];
      if (__builtin_amdgcn_ballot_w64(hl) != 0) {   // the waves that hold halo lanes fetch the neighbours' gF
        float h[3];
        const bool ok = cl_poll3(gep + hidx, (size_t)Pp, tag, hl, h);
        if (hl) { Gs[hli] = h[0]; Gs[CL_STRIDE + hli] = h[1]; Gs[2 * CL_STRIDE + hli] = h[2]; }
        okp = okp && ok;
      }
      if (!okp) bail[par] = 1;
      __syncthreads();   // barrier 2: the G window is complete
      if (bail[par]) { dead = true; break; }
      // ---- spring adjoint, gather form: g_x_i = gxd + sum_l J_il (gF_j - gF_i) ----
      f2 A0 = {gxd[0], 0.f}, A1 = {gxd[1], 0.f}, A2 = {gxd[2], 0.f};
      f2 h0[4], h1[4], h2[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // all 24 LDS reads in flight before the first use
        const int ja = nbs[p], jb = nbs[p + 4];       // a missing neighbour reads gF itself: d = 0 and r = 0
        h0[p] = f2{Gs[ja], Gs[jb]};
        h1[p] = f2{Gs[CL_STRIDE + ja], Gs[CL_STRIDE + jb]};
        h2[p] = f2{Gs[2 * CL_STRIDE + ja], Gs[2 * CL_STRIDE + jb]};
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
      gk += gF[0] * in.S0 + gF[1] * in.S1 + gF[2] * in.S2;   // sum_l w_l (r_l . gF) = gF . S
      gx[0] = A0.x + A0.y; gx[1] = A1.x + A1.y; gx[2] = A2.x + A2.y;
#pragma unroll
      for (int d = 0; d < 3; ++d) vnext[d] = v[d];   // this substep's input v is the previous substep's clip(v5)
    }
    if (dead) break;
    // macro-step boundary: robot_step's action transform (:168-169); part sums -> SA slot -> cluster sums
    {
      const unsigned mtag = (unsigned)(T - t), mpar = mtag & 1u;
      __syncthreads();
      {
        const float w8 = wave_sum8_t(ga, lane);
        if ((lane & 0x2C) == 0) mac[wv * 8 + (((lane >> 2) & 4) | (lane & 3))] = w8;
      }
      __syncthreads();
      if (wv == 0) {
        float tot = 0.f;
        if (lane < 8) {
          for (int e = 0; e < 8; ++e) tot += mac[e * 8 + lane];
          tot += (w == 0) ? gaP : 0.f;
        }
        if (lane < CL_SLOT) cl_put(sa + ((size_t)mpar * W + w) * CL_SLOT + lane, tot, mtag);
        float ctot;
        const bool ok = cl_poll_slots(sa + (size_t)mpar * W * CL_SLOT, W, mtag, lane, ctot);
        if (!ok) bail[0] = bail[1] = 1;
        if (w == 0 && lane < 8) {
          const int d = lane & 3;
          a.g_actions[((size_t)t * B + b) * 8 + lane] = (d < 3) ? ctot * (1.0f / 50.0f) * clip_grad(a8[lane], -2.0f, 2.0f) : ctot;
        }
      }
      __syncthreads();
      if (bail[0]) { dead = true; break; }
    }
  }
  if (!dead) {   // stiffness / mu cotangents: part sums -> SA slot (entries 8, 9) -> cluster sums
    const unsigned mtag = (unsigned)(T + 1), mpar = mtag & 1u;
    __syncthreads();
    {
      const float w0 = wave_sum_l63(gk), w1 = wave_sum_l63(gmu);
      if (lane == 63) { mac[wv * 2] = w0; mac[wv * 2 + 1] = w1; }
    }
    __syncthreads();
    if (wv == 0) {
      float tot = 0.f;
      if (lane == 8 || lane == 9) {
        for (int e = 0; e < 8; ++e) tot += mac[e * 2 + (lane - 8)];
      }
      if (lane < CL_SLOT) cl_put(sa + ((size_t)mpar * W + w) * CL_SLOT + lane, tot, mtag);
      float ctot;
      const bool ok = cl_poll_slots(sa + (size_t)mpar * W * CL_SLOT, W, mtag, lane, ctot);
      if (!ok) bail[0] = 1;
      if (w == 0 && lane == 8) a.g_k[b] = ok ? ctot : NAN;
      if (w == 0 && lane == 9) a.g_mu[b] = ok ? ctot : NAN;
    }
    __syncthreads();
    if (bail[0]) dead = true;
  }
  if (dead) {   // a part of this env never showed up: make it loud
    if (i == 0 && q.timeouts) atomicAdd(q.timeouts, 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = NAN; gv[d] = NAN; }
    gpl = NAN;
    if (w == 0) {
      for (int e = i; e < T * 8; e += CL_T) a.g_actions[((size_t)(e >> 3) * B + b) * 8 + (e & 7)] = NAN;
      if (i == 0) { a.g_k[b] = NAN; a.g_mu[b] = NAN; }
    }
  }
  if (live) {
    const size_t o = ((size_t)b * P + gi) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.g_x0[o + d] = gx[d]; a.g_v0[o + d] = gv[d]; }
  }
  if (w == 0 && i < 8) a.g_prim0[b * 8 + i] = gpl;
}

void cloth_launch_bwd_cluster(const ClothBwdArgs& a, const ClusterArgs& q, hipStream_t stream) {
  const size_t shmem = (size_t)(6 * CL_STRIDE + 2 * 8 * UD_RSTR + 64 + 2) * sizeof(float);
  hipLaunchKernelGGL(cloth_cluster_bwd_kernel, dim3(cl_grid(q.Bl, q.W)), dim3(CL_T), shmem, stream, a, q);
}

}  // namespace ud
