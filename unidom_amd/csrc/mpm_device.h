// Device functions shared by the MPM kernels (mpm.hip: one workgroup per env, LDS cell table;
// mpm_large.hip: many workgroups per env, dense grid in HBM).  Reference citations: see mpm.hip.
#pragma once
#include "common.h"

#ifndef UD_MPM_ABLATE
#define UD_MPM_ABLATE 0   // timing-only diagnostic builds (never shipped): bit0 no SVD, bit1 no scatter, bit2 no grid op, bit3 no g2p, bit4 no clear, bit5 no insert
#endif

#define UD_MAX_PRIM 4

namespace ud {

struct MpmConst {
  int N, Np, n_grid, res[3], steps;
  float dt, dx, inv_dx, p_mass, p_vol, stress_c, dx2, dtg[3];
  int H, logH, nthreads;
  int position_control;            // 1: position_control_batch, 0: collide_batch (soft contact)
  float prim_friction, prim_softness;   // PrimitiveState.friction / .softness (collide_batch only)
  float prim_friction_each[4], prim_softness_each[4];   // per primitive (filled from the scalars where the conf leaves them unset)
  int n_prim, sdf_kind;            // primitives per env (collide_batch: 1..UD_MAX_PRIM); 0 box SDF, 1 container SDF
  int gck;                         // many-workgroup path: grid-checkpoint records per particle and substep (0 = recompute in the backward)
  int sort;                        // many-workgroup path: re-order the particles by cell inside the handle at every step
  int det;                         // ud_mpm_conf.deterministic: the forward sums every cell in particle order (mpm_det.hip)
};

// ---- 3x3 helpers (row-major float[9]) ------------------------------------------------------------
__device__ __forceinline__ void m_mul(const float* A, const float* B, float* R) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void m_mul_bt(const float* A, const float* B, float* R) {  // A * B^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
__device__ __forceinline__ void m_mul_at(const float* A, const float* B, float* R) {  // A^T * B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// The Jacobi rotation's sqrt / reciprocal / rsqrt: the 1-ulp hardware instructions by default; UD_MPM_EXACT (mpm_det.hip, compiled with
// -ffp-contract=off and correctly rounded divide / sqrt, and the host build of the same source) takes the IEEE operations instead, so
// that the deterministic mode computes the same bits on the GPU and on the CPU.
#ifdef UD_HOST_BUILD
#define UD_WAVE_ANY(x) (x)
#else
#define UD_WAVE_ANY(x) __any(x)
#endif
#if defined(UD_MPM_EXACT) || defined(UD_HOST_BUILD)
#define UD_FSQRT(x) sqrtf(x)
#define UD_FRCP(x) (1.f / (x))
#define UD_FRSQ(x) (1.f / sqrtf(x))
#else
#define UD_FSQRT(x) __builtin_amdgcn_sqrtf(x)
#define UD_FRCP(x) __builtin_amdgcn_rcpf(x)
#define UD_FRSQ(x) __builtin_amdgcn_rsqf(x)
#endif

// One-sided Jacobi rotation of columns p, q.  With al = |a_p|^2, be = |a_q|^2, ga = a_p . a_q the textbook angle is zeta = (be - al) / (2 ga),
// t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), c = 1 / sqrt(1 + t^2) -- with the thresholds' sqrt(al be) five transcendental instructions
// (quarter rate) and a division chain per rotation.  The same t without forming zeta: t = sign(d g) |g| / (|d| + sqrt(d^2 + g^2)), d = be - al,
// g = 2 ga: ONE sqrt, ONE rcp, ONE rsq; the thresholds compare squares.  The Jacobi sweeps were ~38 % of the VALU instructions of the one-lane
// particle kernels (tools/isa_budget.py on lg_g2p_p2g<1>: 355 of 3252 static instructions, run three to four times per particle), which the
// counters show VALU-bound at ~70 % of issue (tools/pmc_budget.sh, rope at n_grid 256); round 4, as in plb_common.h for f64.
#define UD_JROT(p, q)                                                                                       \
  {                                                                                                         \
    const float al = a[p] * a[p] + a[3 + p] * a[3 + p] + a[6 + p] * a[6 + p];                                \
    const float be = a[q] * a[q] + a[3 + q] * a[3 + q] + a[6 + q] * a[6 + q];                                \
    const float ga = a[p] * a[q] + a[3 + p] * a[3 + q] + a[6 + p] * a[6 + q];                                \
    const float ab_ = al * be, gg_ = ga * ga;                                                                \
    const bool rot = !done && gg_ > 2.25e-16f * ab_;        /* |ga| > 1.5e-8 sqrt(al be) */                  \
    any_rot |= gg_ > 1e-8f * ab_;                           /* |ga| > 1e-4 sqrt(al be) */                    \
    const float g2_ = rot ? ga + ga : 1.f, d_ = be - al;                                                     \
    const float h_ = UD_FSQRT(d_ * d_ + g2_ * g2_);                                                          \
    float t = fabsf(g2_) * UD_FRCP(fabsf(d_) + h_);                                                          \
    t = ((d_ < 0.f) != (g2_ < 0.f)) ? -t : t;                                                                \
    float cs = UD_FRSQ(1.f + t * t), sn = cs * t;  /* cs^2+sn^2 = 1 to round-off whatever t is */            \
    cs = rot ? cs : 1.f; sn = rot ? sn : 0.f;                                                                \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                          \
      float ap = a[i * 3 + p], aq = a[i * 3 + q];                                                            \
      a[i * 3 + p] = cs * ap - sn * aq; a[i * 3 + q] = sn * ap + cs * aq;                                    \
      float vp = vv[i * 3 + p], vq = vv[i * 3 + q];                                                          \
      vv[i * 3 + p] = cs * vp - sn * vq; vv[i * 3 + q] = sn * vp + cs * vq;                                  \
    }                                                                                                        \
  }

#define UD_CSWAP(p, q)                                                                  \
  if (sv[p] < sv[q]) {                                                                  \
    float ts = sv[p]; sv[p] = sv[q]; sv[q] = ts;                                        \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                     \
      float t1 = a[i * 3 + p]; a[i * 3 + p] = a[i * 3 + q]; a[i * 3 + q] = t1;          \
      float t2 = vv[i * 3 + p]; vv[i * 3 + p] = vv[i * 3 + q]; vv[i * 3 + q] = t2;      \
    }                                                                                   \
  }

// One-sided Jacobi (Hestenes) SVD of a 3x3: A = U diag(S) Vh, S descending >= 0.  The reference calls LAPACK
// (third party); only U S Vh, U Vh and S -- gauge-invariant -- enter the dynamics.  The rotation angle may be
// approximate (v_rcp/v_rsq/v_sqrt, 1 ulp): each Givens pair (cs, sn) is orthonormal to round-off regardless.
#ifndef UD_SVD_SWEEPS
#define UD_SVD_SWEEPS 4
#endif
__device__ __forceinline__ void svd3(const float* A, float* U, float* S, float* Vh) {
  float a[9], vv[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = A[i];
  // Cyclic Jacobi converges quadratically: a sweep whose three normalised off-diagonal products were all below 1e-4 leaves them below
  // 1e-8 -- under the rotation threshold, so every later sweep would be the identity.  Leave after such a sweep, per WAVE on the device
  // (when none of its lanes had a larger product): a rope's F is near a rotation and its particles are done in two sweeps; the chain of
  // dependent rotations is the longest serial stretch of the pre-pass, which at one or two waves per SIMD is what a launch waits for
  // (lanes probe).
  bool done = false;
#pragma unroll 1
  for (int sweep = 0; sweep < UD_SVD_SWEEPS; ++sweep) {   // 4 sweeps reach f32 round-off for |F - I| up to O(1) (measured)
    bool any_rot = false;
    UD_JROT(0, 1)
    UD_JROT(0, 2)
    UD_JROT(1, 2)
    done = done || !any_rot;              // this matrix rotates no more, whatever its wave goes on to do: the result depends on the matrix alone
    if (!UD_WAVE_ANY(!done)) break;
  }
  float sv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) sv[j] = sqrtf(a[j] * a[j] + a[3 + j] * a[3 + j] + a[6 + j] * a[6 + j]);
  UD_CSWAP(0, 1)
  UD_CSWAP(1, 2)
  UD_CSWAP(0, 1)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S[j] = sv[j];
    float inv = sv[j] > FLT_MIN ? 1.f / sv[j] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { U[i * 3 + j] = a[i * 3 + j] * inv; Vh[j * 3 + i] = vv[i * 3 + j]; }
  }
}

__device__ __forceinline__ float safe_inv(float x) { return x / (x * x + 1e-12f); }

// svd_safe_batch.py:65-102 (real 3x3): cotangents (dU, dS, dVh) -> dA
__device__ __forceinline__ void svd3_bwd(const float* U, const float* S, const float* Vh, const float* dU,
                                         const float* dS, const float* dVh, float* dA) {
  float UtdU[9], VtdV[9];
  m_mul_at(U, dU, UtdU);     // Ut @ dU
  m_mul_bt(Vh, dVh, VtdV);   // Vt @ Hc(dVh),  Vt = Cc(Vh) = Vh
  float S2[3] = {S[0] * S[0], S[1] * S[1], S[2] * S[2]};
  float Si[3] = {safe_inv(S[0]), safe_inv(S[1]), safe_inv(S[2])};
  float JJ[9], KK[9];        // (J + J^H) * S  and  S * (K + K^H) handled below
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float Fij = (i == j) ? 0.f : safe_inv(S2[j] - S2[i]);
      JJ[i * 3 + j] = Fij * UtdU[i * 3 + j];
      KK[i * 3 + j] = Fij * VtdV[i * 3 + j];
    }
  // M = dS(diag) + (J+J^T) colscale S + rowscale S (K+K^T)   (the L - L^H term vanishes: L is a real diagonal)
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float t = (JJ[i * 3 + j] + JJ[j * 3 + i]) * S[j] + S[i] * (KK[i * 3 + j] + KK[j * 3 + i]);
      M[i * 3 + j] = t + ((i == j) ? dS[i] : 0.f);
    }
  float UM[9];
  m_mul(U, M, UM);
  m_mul(UM, Vh, dA);
  // projector terms: Pc_U_perp @ (dU * S_inv) @ Vt + (Uc * S_inv) @ dVh @ Pc_V_perp
  float Pu[9], Pv[9], T1[9], T2[9], T3[9];
  m_mul_bt(U, U, Pu);
  m_mul_at(Vh, Vh, Pv);
#pragma unroll
  for (int i = 0; i < 9; ++i) { Pu[i] = ((i % 4 == 0) ? 1.f : 0.f) - Pu[i]; Pv[i] = ((i % 4 == 0) ? 1.f : 0.f) - Pv[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { T1[i * 3 + j] = dU[i * 3 + j] * Si[j]; T2[i * 3 + j] = U[i * 3 + j] * Si[j]; }
  m_mul(Pu, T1, T3);
  m_mul(T3, Vh, T1);
#pragma unroll
  for (int i = 0; i < 9; ++i) dA[i] += T1[i];
  m_mul(T2, dVh, T3);
  m_mul(T3, Pv, T1);
#pragma unroll
  for (int i = 0; i < 9; ++i) dA[i] += T1[i];
}

// ---- primitive helpers -----------------------------------------------------------------------------
__device__ __forceinline__ void qrot(const float* q, const float* v, float* o) {  // :95-102
  float uv0 = q[2] * v[2] - q[3] * v[1], uv1 = q[3] * v[0] - q[1] * v[2], uv2 = q[1] * v[1] - q[2] * v[0];
  float w0 = q[2] * uv2 - q[3] * uv1, w1 = q[3] * uv0 - q[1] * uv2, w2 = q[1] * uv1 - q[2] * uv0;
  o[0] = v[0] + 2.f * (q[0] * uv0 + w0);
  o[1] = v[1] + 2.f * (q[0] * uv1 + w1);
  o[2] = v[2] + 2.f * (q[0] * uv2 + w2);
}

struct PrimF {          // primitive 0 at substep f (uniform)
  float pos[3], iq[4], size[3], pv[3];
  float friction;       // state.friction (ground), not the primitive's own
};

__device__ __forceinline__ float box_sdf(const float* size, const float* gp) {  // box.py:6-18
  float q0 = clipf(fabsf(gp[0]) - size[0], 0.f, INFINITY);
  float q1 = clipf(fabsf(gp[1]) - size[1], 0.f, INFINITY);
  float q2 = clipf(fabsf(gp[2]) - size[2], 0.f, INFINITY);
  float out = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + 1e-12f);
  float tmp = q1 > q2 ? q1 : q2;
  tmp = q0 > tmp ? q0 : tmp;
  tmp = clipf(tmp, -INFINITY, 0.f);
  return out + tmp;
}

struct CellRec {
  float v1[3];
  bool ctrl, fric, bnd[3];
};

template <bool REC>
__device__ __forceinline__ void grid_tail(const MpmConst& c, float friction, int ci, int cj, int ck, float* v, float* vo, CellRec* rec);

// grid op of one cell (:283-313): (m, mv) -> v.  REC: keep what the adjoint needs.
template <bool REC>
__device__ __forceinline__ void grid_op(const MpmConst& c, const PrimF& pf, int ci, int cj, int ck, float m,
                                        const float* mv, float* vo, CellRec* rec) {
  float v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) v[a] = ((m > 0.f) ? mv[a] / m : mv[a]) + c.dtg[a];
  float gp[3] = {(float)ci * c.dx, (float)cj * c.dx, (float)ck * c.dx};
  float d[3] = {gp[0] - pf.pos[0], gp[1] - pf.pos[1], gp[2] - pf.pos[2]}, loc[3];
  qrot(pf.iq, d, loc);
  const bool ctrl = box_sdf(pf.size, loc) < pf.size[0] * 1.5f;   // :232-239
#pragma unroll
  for (int a = 0; a < 3; ++a) v[a] = ctrl ? pf.pv[a] / c.dt : v[a];
  if (REC) rec->ctrl = ctrl;
  grid_tail<REC>(c, pf.friction, ci, cj, ck, v, vo, rec);
}

// ground friction (:297-307) and boundary (:310-313) of one cell: v (after the primitive op) -> vo
template <bool REC>
__device__ __forceinline__ void grid_tail(const MpmConst& c, float friction, int ci, int cj, int ck, float* v, float* vo, CellRec* rec) {
  if (REC) { rec->v1[0] = v[0]; rec->v1[1] = v[1]; rec->v1[2] = v[2]; }
  const bool fric = (cj < 3) && (v[1] <= 0.f);                    // :297-307
  {
    float g0 = (float)ci, g1 = (float)cj, g2 = (float)ck;
    float lin = v[1] + 1e-30f;
    float vit0 = v[0] - lin * 0.f - g0 * 1e-30f, vit1 = v[1] - lin * 1.f - g1 * 1e-30f, vit2 = v[2] - lin * 0.f - g2 * 1e-30f;
    float e0 = vit0 + 1e-12f, e1 = vit1 + 1e-12f, e2 = vit2 + 1e-12f;
    float lit = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
    float s = clipf(1.f + friction * lin / lit, 0.f, INFINITY);
    float f0 = s * (vit0 + g0 * 1e-30f), f2 = s * (vit2 + g2 * 1e-30f);
    v[0] = fric ? f0 : v[0];
    v[1] = fric ? 0.f : v[1];
    v[2] = fric ? f2 : v[2];
  }
  if (REC) rec->fric = fric;
  const int id[3] = {ci, cj, ck};
#pragma unroll
  for (int a = 0; a < 3; ++a) {                                    // :310-313 (Q8: n_grid, not res)
    const bool b = (id[a] < 3 && v[a] < 0.f) || (id[a] > c.n_grid - 3 && v[a] > 0.f);
    if (REC) rec->bnd[a] = b;
    vo[a] = b ? 0.f : v[a];
  }
}

// scatter index rule (Q5/Q9): negative wraps, out-of-range dropped (-1); gather rule: negative wraps, clamp
// cell key = i | j << 10 | k << 20 (res <= 1024 per axis): decoding needs no integer division
__device__ __forceinline__ int cell_scatter(const MpmConst& c, int i, int j, int k) {
  i += (i < 0) ? c.res[0] : 0; j += (j < 0) ? c.res[1] : 0; k += (k < 0) ? c.res[2] : 0;
  if (i < 0 || i >= c.res[0] || j < 0 || j >= c.res[1] || k < 0 || k >= c.res[2]) return -1;
  return i | (j << 10) | (k << 20);
}
__device__ __forceinline__ int cell_gather(const MpmConst& c, int i, int j, int k) {
  i += (i < 0) ? c.res[0] : 0; j += (j < 0) ? c.res[1] : 0; k += (k < 0) ? c.res[2] : 0;
  i = min(max(i, 0), c.res[0] - 1); j = min(max(j, 0), c.res[1] - 1); k = min(max(k, 0), c.res[2] - 1);
  return i | (j << 10) | (k << 20);
}

__device__ __forceinline__ float sel3(const float* w, int d, int i) {  // w[i*3+d] without dynamic register indexing
  return (i == 0) ? w[d] : ((i == 1) ? w[3 + d] : w[6 + d]);
}
__device__ __forceinline__ void decode_cell(const MpmConst& c, int cell, int& ci, int& cj, int& ck) {
  ci = cell & 1023; cj = (cell >> 10) & 1023; ck = (cell >> 20) & 1023;
}

// ---- particle pre-pass (:233-268) ---------------------------------------------------------------------
struct Pre {
  int base[3];
  float fx[3], w[9];           // w[k*3+d]
  float Fn[9], affine[9];
};
struct PreB {                   // extras the adjoint needs
  float U[9], Vh[9], sig_raw[3], sig[3], Jd, mu, la, A[9];
};

// svd_out / svd_in (many-workgroup path, UD_SVD_ROWS floats per particle at stride `svd_stride`: U[9], S[3], Vh[9]): the forward hands the
// factors of this substep's F to the checkpoint right after the Jacobi iteration (no register lives longer for it), the backward
// takes them from there instead of iterating again -- the same bits, the longest serial stretch of its pre-pass gone.
#define UD_SVD_ROWS 21
// LIQ (many-workgroup path): a liquid particle (material 0: mu = 0, Q10) takes no SVD at all -- the stress is la J (J - 1) I, J = the
// product of the singular values = |det Fu|, and its cotangent reaches Fu through the cofactor matrix (particle_adjoint) -- exact, and
// a liquid's F is never reset: sheared without bound, it is the matrix that needs every Jacobi sweep (39 % of pour_water's lg_p2g).
__device__ __forceinline__ float det3(const float* A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
template <bool KEEP, bool LIQ = false>
__device__ __forceinline__ void particle_pre(const MpmConst& c, const float* x, const float* Cm, const float* F,
                                             float mu_s, float la_s, int material, float hard, Pre& q, PreB* kb,
                                             float* svd_out = nullptr, const float* svd_in = nullptr, long svd_stride = 0) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    q.base[d] = (int)(x[d] * c.inv_dx - 0.5f);   // truncation (:233)
    float f = x[d] * c.inv_dx - (float)q.base[d];
    q.fx[d] = f;
    q.w[0 * 3 + d] = 0.5f * ((1.5f - f) * (1.5f - f));
    q.w[1 * 3 + d] = 0.75f - (f - 1.f) * (f - 1.f);
    q.w[2 * 3 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
  }
  float IC[9], Fu[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) IC[i] = ((i % 4 == 0) ? 1.f : 0.f) + c.dt * Cm[i];
  m_mul(IC, F, Fu);                                              // :238
  float h = clipf(hard, 0.1f, 5.f);
  float mu = mu_s * h, la = la_s * h;
  if (material == 0) { mu = 0.f; la = 1.f; }                     // Q10
  float U[9], Vh[9], sr[3], sg[3];
  if (LIQ && material == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { q.Fn[i] = Fu[i]; q.affine[i] = c.p_mass * Cm[i]; }
    const float Jd = fabsf(det3(Fu));
    const float sdiag = c.stress_c * (la * Jd * (Jd - 1.f)) / c.dx2;
    q.affine[0] += sdiag; q.affine[4] += sdiag; q.affine[8] += sdiag;
    if (KEEP) { kb->Jd = Jd; kb->mu = 0.f; kb->la = la; }     // the rest of PreB is not read for a liquid (particle_adjoint<LIQ>)
    return;
  }
  if (UD_MPM_ABLATE & 1) {
    for (int i = 0; i < 9; ++i) { U[i] = (i % 4 == 0) ? 1.f : 0.f; Vh[i] = U[i]; }
    sr[0] = Fu[0]; sr[1] = Fu[4]; sr[2] = Fu[8];
  } else if (svd_in) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { U[i] = svd_in[i * svd_stride]; Vh[i] = svd_in[(12 + i) * svd_stride]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) sr[i] = svd_in[(9 + i) * svd_stride];
  } else {
    svd3(Fu, U, sr, Vh);
    if (svd_out) {
#pragma unroll
      for (int i = 0; i < 9; ++i) { svd_out[i * svd_stride] = U[i]; svd_out[(12 + i) * svd_stride] = Vh[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) svd_out[(9 + i) * svd_stride] = sr[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sg[i] = sr[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) q.Fn[i] = Fu[i];
  if (material == 2) {                                           // :250-258
    float US[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) sg[i] = clipf(sr[i], 1.f - 2.5e-2f * 10.f, 1.f + 4.5e-3f * 100.f);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) US[i * 3 + j] = U[i * 3 + j] * sg[j];
    m_mul(US, Vh, q.Fn);
  }
  float Jd = sg[0] * sg[1] * sg[2];
  float R[9], A[9], St[9];
  m_mul(U, Vh, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i] = q.Fn[i] - R[i];
  m_mul_bt(A, q.Fn, St);
  float vol = la * Jd * (Jd - 1.f);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float s = 2.f * mu * St[i] + ((i % 4 == 0) ? vol : 0.f);
    q.affine[i] = c.stress_c * s / c.dx2 + c.p_mass * Cm[i];
  }
  if (KEEP) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { kb->U[i] = U[i]; kb->Vh[i] = Vh[i]; kb->A[i] = A[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { kb->sig_raw[i] = sr[i]; kb->sig[i] = sg[i]; }
    kb->Jd = Jd; kb->mu = mu; kb->la = la;
  }
}


__device__ __forceinline__ void grid_tail_adjoint(float friction, int ci, int cj, int ck, const CellRec& rec, float* g, float& dfric);
__device__ __forceinline__ void grid_head_adjoint(float m, const float* mvv, float* g, float& gmm);

// Adjoint of grid_op for one cell: g (cotangent of the cell's output velocity, in) -> g (cotangent of mv, out),
// gmm (cotangent of m), dfric (contribution to d state.friction), dpv (contribution to d primitive v[f], valid
// when the function returns true = the cell is position-controlled).  Reverse of :283-313.
__device__ __forceinline__ bool grid_op_adjoint(const MpmConst& c, const PrimF& pf, int ci, int cj, int ck, float m,
                                                const float* mvv, float* g, float& gmm, float& dfric, float* dpv) {
  float vo[3];
  CellRec rec;
  grid_op<true>(c, pf, ci, cj, ck, m, mvv, vo, &rec);
  grid_tail_adjoint(pf.friction, ci, cj, ck, rec, g, dfric);
  if (rec.ctrl) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { dpv[d] = g[d] / c.dt; g[d] = 0.f; }
  }
  grid_head_adjoint(m, mvv, g, gmm);
  return rec.ctrl;
}

// reverse of grid_tail: g (cotangent of vo) -> cotangent of v; dfric = contribution to d state.friction
__device__ __forceinline__ void grid_tail_adjoint(float friction, int ci, int cj, int ck, const CellRec& rec, float* g, float& dfric) {
  dfric = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) g[d] = rec.bnd[d] ? 0.f : g[d];
  if (rec.fric) {
    const float g0 = (float)ci, g1 = (float)cj, g2 = (float)ck;
    const float* vv = rec.v1;
    float lin = vv[1] + 1e-30f;
    float vit[3] = {vv[0] - g0 * 1e-30f, vv[1] - lin - g1 * 1e-30f, vv[2] - g2 * 1e-30f};
    float e[3] = {vit[0] + 1e-12f, vit[1] + 1e-12f, vit[2] + 1e-12f};
    float lit = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    float arg = 1.f + friction * lin / lit;
    float sc = clipf(arg, 0.f, INFINITY);
    float qv0 = vit[0] + g0 * 1e-30f, qv2 = vit[2] + g2 * 1e-30f;
    float gs_ = g[0] * qv0 + g[2] * qv2;
    float gvit[3] = {sc * g[0], 0.f, sc * g[2]};
    float garg = gs_ * clip_grad(arg, 0.f, INFINITY);
    dfric = garg * lin / lit;
    float glin = garg * friction / lit;
    float glit = -garg * friction * lin / (lit * lit);
#pragma unroll
    for (int d = 0; d < 3; ++d) gvit[d] += glit * e[d] / lit;
    glin -= gvit[1];
    g[0] = gvit[0]; g[1] = gvit[1] + glin; g[2] = gvit[2];
  }
}

// reverse of v = where(m > 0, mv / m, mv) + dt g (:283-285): g (cotangent of v) -> cotangent of mv, gmm of m
__device__ __forceinline__ void grid_head_adjoint(float m, const float* mvv, float* g, float& gmm) {
  if (m > 0.f) {
    gmm = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) { float vn = mvv[d] / m; gmm -= g[d] * vn / m; g[d] = g[d] / m; }
  } else if (m == 0.f) {   // Q7: the mv/m branch sees cotangent 0 and 0/0 -> NaN
    g[0] = g[1] = g[2] = NAN; gmm = NAN;
  } else {                 // m < 0 (negative quadratic weights below dx/2, Q13): pass-through branch
    gmm = 0.f;
  }
}

// Adjoint of the particle pre-pass (:233-268) given the gathered stencil cotangents:
//   gw[k*3+d] (weights), gfx (fractional position), gaff (affine), gvp (momentum v) ; gx/gv/gC/gF in: cotangents of
//   the substep outputs (gF = cotangent of F_out = Fn), out: cotangents of the substep inputs.
template <bool LIQ = false>
__device__ __forceinline__ void particle_adjoint(const MpmConst& c, const Pre& q, const PreB& kb, const float* Cm, const float* F,
                                                 int material, const float* gw, float* gfx, const float* gaff, const float* gvp,
                                                 float* gx, float* gv, float* gC, float* gF, float& gmu_p, float& gla_p) {
  // weights -> fx
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float fxd = q.fx[d];
    gfx[d] += gw[0 * 3 + d] * (-(1.5f - fxd)) + gw[1 * 3 + d] * (-2.f * (fxd - 1.f)) + gw[2 * 3 + d] * (fxd - 0.5f);
  }
  float gS[9], gCn[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) { gCn[d] = gaff[d] * c.p_mass; gS[d] = gaff[d] / c.dx2 * c.stress_c; }
  if (LIQ && material == 0) {
    // stress = la J (J - 1) I with J = |det Fu|: dJ / dFu = sign(det) cof(Fu); Fn = Fu passes the next state's cotangent through
    const float* A = q.Fn;
    const float trg = gS[0] + gS[4] + gS[8];
    const float gJ = kb.la * (2.f * kb.Jd - 1.f) * trg * ((det3(A) < 0.f) ? -1.f : 1.f);
    gmu_p = 0.f; gla_p = kb.Jd * (kb.Jd - 1.f) * trg;
    float gFu[9];
    gFu[0] = gF[0] + gJ * (A[4] * A[8] - A[5] * A[7]); gFu[1] = gF[1] - gJ * (A[3] * A[8] - A[5] * A[6]); gFu[2] = gF[2] + gJ * (A[3] * A[7] - A[4] * A[6]);
    gFu[3] = gF[3] - gJ * (A[1] * A[8] - A[2] * A[7]); gFu[4] = gF[4] + gJ * (A[0] * A[8] - A[2] * A[6]); gFu[5] = gF[5] - gJ * (A[0] * A[7] - A[1] * A[6]);
    gFu[6] = gF[6] + gJ * (A[1] * A[5] - A[2] * A[4]); gFu[7] = gF[7] - gJ * (A[0] * A[5] - A[2] * A[3]); gFu[8] = gF[8] + gJ * (A[0] * A[4] - A[1] * A[3]);
    float T1[9], IC[9];
    m_mul_bt(gFu, F, T1);
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = gCn[d] + c.dt * T1[d]; IC[d] = ((d % 4 == 0) ? 1.f : 0.f) + c.dt * Cm[d]; }
    m_mul_at(IC, gFu, gF);
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = gx[d] + gfx[d] * c.inv_dx; gv[d] = gvp[d]; }
    return;
  }
  // stress = 2 mu A Fn^T + la J (J-1) I
  float AFt[9], T1[9], gA[9], gFn[9];
  m_mul_bt(kb.A, q.Fn, AFt);
  gmu_p = 0.f;
#pragma unroll
  for (int d = 0; d < 9; ++d) gmu_p += gS[d] * 2.f * AFt[d];
  m_mul(gS, q.Fn, gA);
  m_mul_at(gS, kb.A, T1);
#pragma unroll
  for (int d = 0; d < 9; ++d) { gA[d] *= 2.f * kb.mu; gFn[d] = gF[d] + 2.f * kb.mu * T1[d] + gA[d]; }
  const float trg = gS[0] + gS[4] + gS[8];
  const float gJ = kb.la * (2.f * kb.Jd - 1.f) * trg;
  gla_p = kb.Jd * (kb.Jd - 1.f) * trg;
  // R = U Vh.  Its cotangent gR enters the SVD VJP (svd_safe_batch.py:65-102) as dU = gR Vh^T and dVh = U^T gR, for which
  // U^T dU = W and V^T dV = W^T with W = U^T gR Vh^T, and the off-diagonal of the VJP's middle factor collapses to
  // F_ij (S_j - S_i) (W_ij - W_ji), F_ij = 1 / (S_j^2 - S_i^2) (regularised: safe_inv).  Evaluated literally -- W and W^T as two
  // separate products, their antisymmetric parts times S_j and S_i added, the sum times F_ij -- the roundings of the two
  // products do not cancel and F_ij amplifies them by 1 / (S_j - S_i): percent-level errors on the particles whose singular
  // values are close (F = I + 1 % noise: about one particle in a hundred), and which of them depends on how the compiler
  // schedules the products (measured: the same source with two unrelated branches added moved the worst particle's error in gF
  // from 3e-4 to 0.18 at n_grid 256).  The rotation part is therefore taken in this closed form, with the same regularised F_ij;
  // the plastic projection's U S' Vh (material 2) gets the same treatment below; svd3_bwd, the literal VJP, is kept for reference only.
  float T2[9], MR[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) MR[d] = 0.f;
  m_mul_at(kb.U, gA, T2);     // gR = -gA (A = Fn - R): W = -(T2 Vh^T); only its antisymmetric part is needed
  // S_j^2 - S_i^2 is formed as (S_j - S_i)(S_j + S_i): the difference of two close floats is exact, the difference of their
  // rounded squares is not (at a gap of 1e-5 it is off by a percent, and F_ij with it).  Three pairs (i < j); F and S_j - S_i are odd.
  const float* S = kb.sig_raw;
  float dS[3], Fx[3];   // pairs (0,1), (0,2), (1,2)
#pragma unroll
  for (int pr = 0; pr < 3; ++pr) {
    const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
    dS[pr] = S[j] - S[i];
    Fx[pr] = safe_inv(dS[pr] * (S[j] + S[i]));
    const float wa = (T2[i * 3] * kb.Vh[j * 3] - T2[j * 3] * kb.Vh[i * 3]) + (T2[i * 3 + 1] * kb.Vh[j * 3 + 1] - T2[j * 3 + 1] * kb.Vh[i * 3 + 1]) +
                     (T2[i * 3 + 2] * kb.Vh[j * 3 + 2] - T2[j * 3 + 2] * kb.Vh[i * 3 + 2]);   // -(W_ij - W_ji)
    const float m = -(Fx[pr] * dS[pr] * wa);
    MR[i * 3 + j] = m; MR[j * 3 + i] = -m;
  }
  float gsig[3] = {gJ * kb.sig[1] * kb.sig[2], gJ * kb.sig[0] * kb.sig[2], gJ * kb.sig[0] * kb.sig[1]};
  if (material == 2) {
    // Fn = U diag(s') Vh, s' = clip(S): its cotangent gFn reaches U and Vh as dU = gFn Vh^T diag(s'), dVh = diag(s') U^T gFn, i.e.
    // U^T dU = P diag(s') and V^T dV = P^T diag(s') with P = U^T gFn Vh^T -- again one product instead of two, the VJP's
    // off-diagonal becomes F_ij [P_ij (s'_j S_j - s'_i S_i) + P_ji (S_i s'_j - S_j s'_i)], its diagonal the clipped singular
    // values' cotangent.  With c = s' - S (exactly zero where the clip is inactive) the two brackets are
    // (S_j c_j - S_i c_i) + (S_j - S_i)(S_j + S_i) and S_i c_j - S_j c_i: an unclipped pair contributes F_ij x P_ij, x = S_j^2 - S_i^2.
    float P[9];
    m_mul_at(kb.U, gFn, T1);
    m_mul_bt(T1, kb.Vh, P);
    const float cl[3] = {kb.sig[0] - S[0], kb.sig[1] - S[1], kb.sig[2] - S[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      gsig[i] += P[i * 4];
      gsig[i] *= clip_grad(kb.sig_raw[i], 1.f - 2.5e-2f * 10.f, 1.f + 4.5e-3f * 100.f);
    }
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
      const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
      const float Aij = (S[j] * cl[j] - S[i] * cl[i]) + dS[pr] * (S[j] + S[i]);
      const float Bij = S[i] * cl[j] - S[j] * cl[i];
      MR[i * 3 + j] += Fx[pr] * (P[i * 3 + j] * Aij + P[j * 3 + i] * Bij);
      MR[j * 3 + i] += Fx[pr] * (P[j * 3 + i] * Aij + P[i * 3 + j] * Bij);   // F, A, B are odd in (i, j): the signs cancel
    }
#pragma unroll
    for (int d = 0; d < 9; ++d) gFn[d] = 0.f;   // Fn = U s' Vh: gFn reaches Fu only through the SVD
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) MR[i * 4] = gsig[i];   // the VJP is U (diag(gsig) + off-diagonal) Vh: U, Vh are orthogonal, the projector terms of :95-102 vanish
  m_mul(kb.U, MR, T1);
  m_mul(T1, kb.Vh, T2);
  float* gFu = gFn;       // cotangent of Fu = (I + dt C) F
#pragma unroll
  for (int d = 0; d < 9; ++d) gFu[d] += T2[d];
  float IC[9];
  m_mul_bt(gFu, F, T1);
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[d] = gCn[d] + c.dt * T1[d]; IC[d] = ((d % 4 == 0) ? 1.f : 0.f) + c.dt * Cm[d]; }
  m_mul_at(IC, gFu, gF);
#pragma unroll
  for (int d = 0; d < 3; ++d) { gx[d] = gx[d] + gfx[d] * c.inv_dx; gv[d] = gvp[d]; }
}

}  // namespace ud
