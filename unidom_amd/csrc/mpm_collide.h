// Soft contact between the grid and one box primitive -- collide_batch and what it calls
//   /root/reference/DaXBench/daxbench/core/engine/primitives/primitives.py
//     collide_batch :154-182   sdf_batch :112-114   inv_trans_batch :105-109   normal_batch / _normal_batch :117-141
//     collider_v_batch :144-151   qrot_batch :95-102   length :68-70
//   /root/reference/DaXBench/daxbench/core/engine/primitives/box.py  _sdf_batch :6-18
//   /root/reference/DaXBench/daxbench/core/engine/primitives/container.py  _sdf_batch :8-16 (cut hollow sphere)
// and its hand-derived adjoint (the reference differentiates it with jax.grad, mpm_simulator.py:339-359).
//
// The contact normal is a central difference of the SDF with d = 1e-6 (:119-134): one ulp in the local position
// moves a normal component by ~0.2 %.  Everything up to the normal therefore runs without FMA contraction and with
// correctly rounded division / square root, in the reference's operation order; the rest follows the file's flags.
#pragma once
#include "mpm_device.h"
#ifndef UD_HOST_BUILD
#include "exact_math.h"
#endif

namespace ud {

// Correctly rounded f32 sqrt / divide whatever the file's -f[no-]hip-fp32-correctly-rounded-divide-sqrt setting, as a policy of the
// SDF / collide code below:
//   CMath<false>  through f64 (53 >= 2*24 + 2 bits, so the second rounding is innocuous): right for ANY operands.  The host build (the
//                 deterministic mode's same-order CPU checker) has only this one.
//   CMath<true>   (device, round 5) exact_math.h's in-range f32 sequences -- the compiler's own IEEE expansions minus their exponent
//                 scaling: the same bits inside the stated windows (tools/check_exact_math.hip, tools/check_exact_div.hip) at about a
//                 third of the issue slots of the f64 route (v_cvt + v_rsq_f64 / v_rcp_f64 + ~8 half-rate f64 FMAs + v_cvt per root or
//                 quotient: 592 of lg_grid's 1750 static VALU instructions were f64).  It never branches: every operand goes into a
//                 RangeTrack (min of the sqrt arguments, min / max of the denominators, max and frexp-exponent range of the numerators --
//                 one or two instructions each), and collide_cell asks ONCE per cell whether any lane of the wave left a window; if so
//                 the wave runs the geometry again on the f64 route (same bits for the lanes that were in range).  So the result is the
//                 correctly rounded one for any operands, and the common case pays no per-operation branch (a guarded sqrt / divide at
//                 each of the 47 sites cost as many scalar instructions as it saved vector ones).
// Quotients over one denominator share its refined reciprocal (Den).
struct RangeTrack {
  float smin, dmin, dmax, amax;
  int emin, emax;
  __device__ __forceinline__ void init() { smin = INFINITY; dmin = INFINITY; dmax = 0.f; amax = 0.f; emin = 0; emax = 0; }
  // NaN operands slip through every test below on purpose: the in-range sequences return NaN for them, as IEEE does
  __device__ __forceinline__ bool bad() const {
    return !(smin >= 0x1p-96f) || !(dmin >= 0x1p-40f) || !(dmax <= 0x1p40f) || !(amax < 0x1p60f) || emin < -59 || emax > 60;
  }
};
template <bool FAST> struct CMath;
template <> struct CMath<false> {
  struct Den { float d; };
  static __device__ __forceinline__ float sqrt_(float x, RangeTrack&) { return (float)sqrt((double)x); }
  static __device__ __forceinline__ float sqrt_floor(float x, RangeTrack&) { return (float)sqrt((double)x); }
  static __device__ __forceinline__ Den den(float d, RangeTrack&) { return Den{d}; }
  static __device__ __forceinline__ float div(float a, const Den& q, RangeTrack&) { return (float)((double)a / (double)q.d); }
};
#ifndef UD_HOST_BUILD
template <> struct CMath<true> {
  struct Den { float d, rd; };
  static __device__ __forceinline__ float sqrt_(float x, RangeTrack& t) { t.smin = fminf(t.smin, x); return sqrt_rn_inrange(x); }
  // x = (a sum of squares) + 1e-12f: never below 1e-12 > 2^-96 (or NaN / inf, which the in-range sequence handles) -- nothing to track
  static __device__ __forceinline__ float sqrt_floor(float x, RangeTrack&) { return sqrt_rn_inrange(x); }
  static __device__ __forceinline__ Den den(float d, RangeTrack& t) {
    t.dmin = fminf(t.dmin, __builtin_fabsf(d)); t.dmax = fmaxf(t.dmax, __builtin_fabsf(d));
    return Den{d, div_prep(d)};
  }
  static __device__ __forceinline__ float div(float a, const Den& q, RangeTrack& t) {
    const int e = __builtin_amdgcn_frexp_expf(a);          // 0 for +-0 (and for inf / NaN: amax catches inf)
    t.emin = min(t.emin, e); t.emax = max(t.emax, e); t.amax = fmaxf(t.amax, __builtin_fabsf(a));
    return div_rn_prepped(a, q.d, q.rd);
  }
};
__device__ __forceinline__ float sqrt_rte(float x) { return sqrt_rn(x); }            // one-off sites (primc_finish): guarded per call
struct DivDen { float d, rd; bool ok; };
__device__ __forceinline__ DivDen div_den(float d) { return DivDen{d, div_prep(d), div_den_inrange(d)}; }
__device__ __forceinline__ float div_rte(float a, const DivDen& q) { return div_rn_shared(a, q.d, q.rd, q.ok); }
#else
__device__ __forceinline__ float sqrt_rte(float x) { return (float)sqrt((double)x); }
struct DivDen { float d; };
__device__ __forceinline__ DivDen div_den(float d) { return DivDen{d}; }
__device__ __forceinline__ float div_rte(float a, const DivDen& q) { return (float)((double)a / (double)q.d); }
#endif

// exp for the softness falloff.  The fast kernels take the platform's expf; the deterministic mode's builds (UD_MPM_EXACT on the device,
// UD_HOST_BUILD for the same-order CPU checker) need the SAME bits from hipcc and from the host compiler: range reduction and a degree-6
// polynomial in plain IEEE operations (neither build contracts), scaled by an exact ldexp.  Within 2 ulp of expf on the range that reaches
// it; results below the normal range are flushed to zero (influence 1e-38 either way).
#if defined(UD_MPM_EXACT) || defined(UD_HOST_BUILD)
__device__ __forceinline__ float ud_expf(float x) {
  if (!(x < 88.7f)) return x != x ? x : INFINITY;
  if (x < -87.f) return 0.f;
  const float kf = rintf(x * 1.44269504f);
  float r = x - kf * 0.693359375f;            // ln 2 in two pieces: the first has 9 significant bits, its product with |k| <= 128 is exact
  r = r - kf * -2.12194440e-4f;
  float p = 1.3888889e-3f;
  p = p * r + 8.3333333e-3f;
  p = p * r + 4.1666668e-2f;
  p = p * r + 1.6666667e-1f;
  p = p * r + 0.5f;
  const float y = 1.f + (r + (r * r) * p);
  return ldexpf(y, (int)kf);
}
#else
__device__ __forceinline__ float ud_expf(float x) { return expf(x); }
#endif

struct PrimC {            // the primitive as the grid op of substep f sees it (rows f and f + 1, clamped: Q5); uniform
  float p0[3], r0[4], p1[3], r1[4], iq[4], nq, size[3], soft, mu;
  int kind;               // 0 box, 1 container (the reference's process-global set_sdf, primitives.py:26-28)
};
struct PrimCGrad {        // cotangents one cell adds to the primitive's leaves
  float p0[3], r0[4], p1[3], r1[4], size[3], mu;
};
constexpr int UD_PRIMC_NGRAD = 18;

__device__ __forceinline__ void primc_finish(PrimC& pc) {   // inv_trans_batch :106-107
#pragma clang fp contract(off)
  const float c0 = pc.r0[0], c1 = -pc.r0[1], c2 = -pc.r0[2], c3 = -pc.r0[3];
  pc.nq = sqrt_rte(c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3) + 1e-12f;
  const DivDen dn = div_den(pc.nq);
  pc.iq[0] = div_rte(c0, dn); pc.iq[1] = div_rte(c1, dn);
  pc.iq[2] = div_rte(c2, dn); pc.iq[3] = div_rte(c3, dn);
}

__device__ __forceinline__ void qrot_x(const float* q, const float* v, float* o) {  // :95-102, no contraction
#pragma clang fp contract(off)
  float uv0 = q[2] * v[2] - q[3] * v[1], uv1 = q[3] * v[0] - q[1] * v[2], uv2 = q[1] * v[1] - q[2] * v[0];
  float w0 = q[2] * uv2 - q[3] * uv1, w1 = q[3] * uv0 - q[1] * uv2, w2 = q[1] * uv1 - q[2] * uv0;
  o[0] = v[0] + 2.f * (q[0] * uv0 + w0);
  o[1] = v[1] + 2.f * (q[0] * uv1 + w1);
  o[2] = v[2] + 2.f * (q[0] * uv2 + w2);
}

template <bool FAST>
__device__ __forceinline__ float box_sdf_x(const float* size, float p0, float p1, float p2, RangeTrack& t) {  // box.py:6-18
#pragma clang fp contract(off)
  const float q0 = clipf(fabsf(p0) - size[0], 0.f, INFINITY);
  const float q1 = clipf(fabsf(p1) - size[1], 0.f, INFINITY);
  const float q2 = clipf(fabsf(p2) - size[2], 0.f, INFINITY);
  const float out = CMath<FAST>::sqrt_floor(q0 * q0 + q1 * q1 + q2 * q2 + 1e-12f, t);
  float tmp = q1 > q2 ? q1 : q2;
  tmp = q0 > tmp ? q0 : tmp;
  tmp = clipf(tmp, -INFINITY, 0.f);
  return out + tmp;
}

// accumulates the cotangents of p and size for a cotangent gout of box_sdf(size, p)
__device__ __forceinline__ void box_sdf_bwd(const float* size, float p0, float p1, float p2, float gout, float* gp, float* gsize) {
  const float p[3] = {p0, p1, p2};
  float xr[3], q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { xr[a] = fabsf(p[a]) - size[a]; q[a] = clipf(xr[a], 0.f, INFINITY); }
  const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + 1e-12f);
  const bool s12 = q[1] > q[2];
  const float q12 = s12 ? q[1] : q[2];
  const bool s0 = q[0] > q12;
  const float qsel = s0 ? q[0] : q12;
  const float gt = gout * clip_grad(qsel, -INFINITY, 0.f);
  const bool sel[3] = {s0, !s0 && s12, !s0 && !s12};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float gq = gout * q[a] / len + (sel[a] ? gt : 0.f);
    const float gx = gq * clip_grad(xr[a], 0.f, INFINITY);
    const float sg = p[a] > 0.f ? 1.f : (p[a] < 0.f ? -1.f : 0.f);
    gp[a] += gx * sg;
    gsize[a] -= gx;
  }
}

// container.py:8-16 -- cut hollow sphere, size = (r, h, t)
template <bool FAST>
__device__ __forceinline__ float container_sdf_x(const float* size, float p0, float p1, float p2, RangeTrack& tr) {
#pragma clang fp contract(off)
  const float r = size[0], h = size[1], t = size[2];
  const float w = CMath<FAST>::sqrt_(r * r - h * h, tr);
  const float q0 = CMath<FAST>::sqrt_floor(p0 * p0 + p2 * p2 + 1e-12f, tr), q1 = p1;
  const bool mask = h * q0 < w * q1;
  const float d0 = q0 - w, d1 = q1 - h;
  const float val1 = CMath<FAST>::sqrt_floor(d0 * d0 + d1 * d1 + 1e-12f, tr) - t;
  const float val2 = fabsf(CMath<FAST>::sqrt_floor(q0 * q0 + q1 * q1 + 1e-12f, tr) - r) - t;
  return mask ? val1 : val2;
}

__device__ __forceinline__ void container_sdf_bwd(const float* size, float p0, float p1, float p2, float gout, float* gp, float* gsize) {
  const float r = size[0], h = size[1];
  const float w = sqrtf(r * r - h * h);
  const float q0 = sqrtf(p0 * p0 + p2 * p2 + 1e-12f), q1 = p1;
  const bool mask = h * q0 < w * q1;
  float gq0, gq1, gw = 0.f, gr = 0.f, gh = 0.f;
  if (mask) {
    const float d0 = q0 - w, d1 = q1 - h, L1 = sqrtf(d0 * d0 + d1 * d1 + 1e-12f);
    gq0 = gout * d0 / L1; gq1 = gout * d1 / L1; gw = -gq0; gh = -gq1;
  } else {
    const float L2 = sqrtf(q0 * q0 + q1 * q1 + 1e-12f), dd = L2 - r;
    const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
    gq0 = gout * sg * q0 / L2; gq1 = gout * sg * q1 / L2; gr = -gout * sg;
  }
  gr += gw * r / w; gh -= gw * h / w;
  gp[0] += gq0 * p0 / q0; gp[2] += gq0 * p2 / q0; gp[1] += gq1;
  gsize[0] += gr; gsize[1] += gh; gsize[2] -= gout;
}

// (kind as a template parameter of the collide code was tried: no faster, and `#pragma clang fp contract(off)` stopped
// being honoured inside the instantiations -- the upright-bowl test, which needs the SDF path bit-faithful, failed)
#ifndef UD_DIAG_BOX_ONLY
#define UD_DIAG_BOX_ONLY 0   // timing-only diagnostic build (never shipped): drop the container branch at every SDF call site
#endif
template <bool FAST>
__device__ __forceinline__ float prim_sdf_x(int kind, const float* size, float p0, float p1, float p2, RangeTrack& t) {
  if (UD_DIAG_BOX_ONLY) return box_sdf_x<FAST>(size, p0, p1, p2, t);
  return kind == 1 ? container_sdf_x<FAST>(size, p0, p1, p2, t) : box_sdf_x<FAST>(size, p0, p1, p2, t);
}
__device__ __forceinline__ void prim_sdf_bwd(int kind, const float* size, float p0, float p1, float p2, float gout, float* gp, float* gsize) {
  if (UD_DIAG_BOX_ONLY) { box_sdf_bwd(size, p0, p1, p2, gout, gp, gsize); return; }
  if (kind == 1) container_sdf_bwd(size, p0, p1, p2, gout, gp, gsize);
  else box_sdf_bwd(size, p0, p1, p2, gout, gp, gsize);
}

// qrot adjoint: accumulates into gq[4], gv[3]
__device__ __forceinline__ void qrot_bwd(const float* q, const float* v, const float* go, float* gq, float* gv) {
  const float uv0 = q[2] * v[2] - q[3] * v[1], uv1 = q[3] * v[0] - q[1] * v[2], uv2 = q[1] * v[1] - q[2] * v[0];
  gq[0] += 2.f * (go[0] * uv0 + go[1] * uv1 + go[2] * uv2);
  float gu0 = 2.f * q[0] * go[0], gu1 = 2.f * q[0] * go[1], gu2 = 2.f * q[0] * go[2];
  const float h0 = 2.f * go[0], h1 = 2.f * go[1], h2 = 2.f * go[2];        // cotangent of uuv
  gq[1] += uv1 * h2 - uv2 * h1; gq[2] += uv2 * h0 - uv0 * h2; gq[3] += uv0 * h1 - uv1 * h0;   // uv x g_uuv
  gu0 += h1 * q[3] - h2 * q[2]; gu1 += h2 * q[1] - h0 * q[3]; gu2 += h0 * q[2] - h1 * q[1];   // g_uuv x qv
  gq[1] += v[1] * gu2 - v[2] * gu1; gq[2] += v[2] * gu0 - v[0] * gu2; gq[3] += v[0] * gu1 - v[1] * gu0;   // v x g_uv
  gv[0] += go[0] + (gu1 * q[3] - gu2 * q[2]);
  gv[1] += go[1] + (gu2 * q[1] - gu0 * q[3]);
  gv[2] += go[2] + (gu0 * q[2] - gu1 * q[1]);
}

struct CollideRec {
  float rel[3], loc[3], e, infl, n[3], len, nl[3], D[3], cv[3], iv[3], nc, m, vt[3], vtn, arg, c, vtp[3];
  bool flag;
};

// the part of collide_batch that sees the SDF (:156-166): local position, distance -> influence, finite-difference normal, collider velocity.
// rec != nullptr (the backward, round 5): the seven SDF evaluations are not repeated -- `rec` = (e, n[3]) is what the forward's grid op
// computed for this cell and primitive and left beside the grid checkpoint (the same bits), everything else follows from it.
template <bool FAST>
__device__ __forceinline__ void collide_geom(const PrimC& pc, float dt, const float* gp, CollideRec& r, RangeTrack& t, const float* rec = nullptr) {
#pragma clang fp contract(off)
#pragma unroll
  for (int a = 0; a < 3; ++a) r.rel[a] = gp[a] - pc.p0[a];
  qrot_x(pc.iq, r.rel, r.loc);
  if (rec) {
    r.e = rec[0]; r.n[0] = rec[1]; r.n[1] = rec[2]; r.n[2] = rec[3];
  } else {
    const float dist = prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0], r.loc[1], r.loc[2], t);
    r.e = ud_expf(-dist * pc.soft);
    const float d = 1.e-6f, k = 500000.f;   // (0.5 / d)
    r.n[0] = k * (prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0] + d, r.loc[1], r.loc[2], t) - prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0] + (-d), r.loc[1], r.loc[2], t));
    r.n[1] = k * (prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0], r.loc[1] + d, r.loc[2], t) - prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0], r.loc[1] + (-d), r.loc[2], t));
    r.n[2] = k * (prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0], r.loc[1], r.loc[2] + d, t) - prim_sdf_x<FAST>(pc.kind, pc.size, r.loc[0], r.loc[1], r.loc[2] + (-d), t));
  }
  r.infl = clipf(r.e, -INFINITY, 1.f);
  r.len = CMath<FAST>::sqrt_floor(r.n[0] * r.n[0] + r.n[1] * r.n[1] + r.n[2] * r.n[2] + 1e-12f, t);
  const typename CMath<FAST>::Den dl = CMath<FAST>::den(r.len, t);
#pragma unroll
  for (int a = 0; a < 3; ++a) r.nl[a] = CMath<FAST>::div(r.n[a], dl, t);
  qrot_x(pc.r0, r.nl, r.D);
  float np_[3];
  qrot_x(pc.r1, r.loc, np_);
  const typename CMath<FAST>::Den dd = CMath<FAST>::den(dt, t);
#pragma unroll
  for (int a = 0; a < 3; ++a) r.cv[a] = CMath<FAST>::div((np_[a] + pc.p1[a]) - gp[a], dd, t);
}

// collide_batch (:154-182) for the cell at world position gp: v -> vo
__device__ __forceinline__ void collide_cell(const PrimC& pc, float dt, const float* gp, const float* v, float* vo, CollideRec& r, const float* rec = nullptr) {
  RangeTrack t;
#ifdef UD_HOST_BUILD
  collide_geom<false>(pc, dt, gp, r, t, rec);
#else
  t.init();
  collide_geom<true>(pc, dt, gp, r, t, rec);
  if (__builtin_amdgcn_ballot_w64(t.bad()) != 0) collide_geom<false>(pc, dt, gp, r, t, rec);   // wave-uniform and rare: an operand outside the in-range windows
#endif
#pragma unroll
  for (int a = 0; a < 3; ++a) r.iv[a] = v[a] - r.cv[a];
  r.nc = r.iv[0] * r.D[0] + r.iv[1] * r.D[1] + r.iv[2] * r.D[2];
  r.m = clipf(r.nc, -INFINITY, 0.f);
#pragma unroll
  for (int a = 0; a < 3; ++a) r.vt[a] = r.iv[a] - r.m * r.D[a];
  const float vt_dot = r.vt[0] * r.vt[0] + r.vt[1] * r.vt[1] + r.vt[2] * r.vt[2];
  r.vtn = sqrtf(vt_dot + 1e-12f);
  r.arg = r.vtn + r.nc * pc.mu;
  r.c = clipf(r.arg, 1e-12f, INFINITY);
  r.flag = (r.nc < 0.f) && (sqrtf(vt_dot) > 1e-12f);
  const float fl = r.flag ? 1.f : 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float vtf = r.vt[a] / r.vtn * r.c;
    r.vtp[a] = vtf * fl + r.vt[a] * (1.f - fl);
    vo[a] = r.cv[a] + r.iv[a] * (1.f - r.infl) + r.vtp[a] * r.infl;
  }
}

// adjoint: gout (cotangent of vo) -> gv (cotangent of v); pg receives this cell's share of the primitive's cotangents.
// r: the record collide_cell left for this cell
__device__ __forceinline__ void collide_cell_bwd(const PrimC& pc, float dt, const CollideRec& r, const float* gout,
                                                 float* gv, PrimCGrad& pg) {
#pragma unroll
  for (int a = 0; a < 3; ++a) { pg.p0[a] = 0.f; pg.p1[a] = 0.f; pg.size[a] = 0.f; }
#pragma unroll
  for (int a = 0; a < 4; ++a) { pg.r0[a] = 0.f; pg.r1[a] = 0.f; }
  float gcv[3], giv[3], gvtp[3], ginfl = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gcv[a] = gout[a]; giv[a] = gout[a] * (1.f - r.infl); gvtp[a] = gout[a] * r.infl;
    ginfl += gout[a] * (r.vtp[a] - r.iv[a]);
  }
  float gvt[3], gnc = 0.f;
  {
    const float fl = r.flag ? 1.f : 0.f;
    float gc = 0.f, gu_vt = 0.f, gvtn = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float gvtf = gvtp[a] * fl;
      gvt[a] = gvtp[a] * (1.f - fl);
      gc += gvtf * (r.vt[a] / r.vtn);
      const float gu = gvtf * r.c;
      gvt[a] += gu / r.vtn;
      gu_vt += gu * r.vt[a];
    }
    gvtn -= gu_vt / (r.vtn * r.vtn);
    const float garg = gc * clip_grad(r.arg, 1e-12f, INFINITY);
    gvtn += garg; gnc += garg * pc.mu; pg.mu = garg * r.nc;
#pragma unroll
    for (int a = 0; a < 3; ++a) gvt[a] += gvtn * r.vt[a] / r.vtn;
  }
  float gD[3], gm = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) { giv[a] += gvt[a]; gm -= gvt[a] * r.D[a]; gD[a] = -r.m * gvt[a]; }
  gnc += gm * clip_grad(r.nc, -INFINITY, 0.f);
#pragma unroll
  for (int a = 0; a < 3; ++a) { giv[a] += gnc * r.D[a]; gD[a] += gnc * r.iv[a]; }
#pragma unroll
  for (int a = 0; a < 3; ++a) { gv[a] = giv[a]; gcv[a] -= giv[a]; }
  float gloc[3] = {0.f, 0.f, 0.f}, gnp[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { gnp[a] = gcv[a] / dt; pg.p1[a] += gnp[a]; }
  qrot_bwd(pc.r1, r.loc, gnp, pg.r1, gloc);
  float gnl[3] = {0.f, 0.f, 0.f};
  qrot_bwd(pc.r0, r.nl, gD, pg.r0, gnl);
  const float dotn = gnl[0] * r.n[0] + gnl[1] * r.n[1] + gnl[2] * r.n[2];
  const float glen = -dotn / (r.len * r.len);
  const float d = 1.e-6f, k = 500000.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float gn = gnl[a] / r.len + glen * r.n[a] / r.len;
    const float i0 = r.loc[0] + (a == 0 ? d : 0.f), i1 = r.loc[1] + (a == 1 ? d : 0.f), i2 = r.loc[2] + (a == 2 ? d : 0.f);
    const float d0 = r.loc[0] + (a == 0 ? -d : 0.f), d1 = r.loc[1] + (a == 1 ? -d : 0.f), d2 = r.loc[2] + (a == 2 ? -d : 0.f);
    prim_sdf_bwd(pc.kind, pc.size, i0, i1, i2, k * gn, gloc, pg.size);
    prim_sdf_bwd(pc.kind, pc.size, d0, d1, d2, -(k * gn), gloc, pg.size);
  }
  const float ge = ginfl * clip_grad(r.e, -INFINITY, 1.f);
  const float gdist = -(ge * r.e) * pc.soft;
  prim_sdf_bwd(pc.kind, pc.size, r.loc[0], r.loc[1], r.loc[2], gdist, gloc, pg.size);
  float giq[4] = {0.f, 0.f, 0.f, 0.f}, grel[3] = {0.f, 0.f, 0.f};
  qrot_bwd(pc.iq, r.rel, gloc, giq, grel);
#pragma unroll
  for (int a = 0; a < 3; ++a) pg.p0[a] -= grel[a];
  const float cq[4] = {pc.r0[0], -pc.r0[1], -pc.r0[2], -pc.r0[3]};
  const float nrm = pc.nq - 1e-12f;
  const float dq = giq[0] * cq[0] + giq[1] * cq[1] + giq[2] * cq[2] + giq[3] * cq[3];
  const float gnq = -dq / (pc.nq * pc.nq);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float gcq = giq[a] / pc.nq + gnq * cq[a] / nrm;
    pg.r0[a] += (a == 0) ? gcq : -gcq;
  }
}

}  // namespace ud
