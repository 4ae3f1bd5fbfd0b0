// ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
// Never linked into, imported by, or measured as the product path.
//
// CPU restatement (dense `res` grid, like the source) of the reference's MLS-MPM `step` and its adjoint:
//   /root/reference/DaXBench/daxbench/core/engine/mpm_simulator.py
//     substep :223-330 (p2g_micro :178-194, g2p_micro :196-221)   step :413-429   copy_frame :365-373
//     norm_grad_state / norm_grad :375-411     substep_bwd_loss (which leaves are differentiated) :339-356
//   /root/reference/DaXBench/daxbench/core/engine/svd_safe_batch.py  svd :19-51, _svd_bwd :65-102
//   /root/reference/DaXBench/daxbench/core/engine/primitives/primitives.py
//     forward_kinematics :185-194  set_action/set_velocity :212-229  position_control_batch :232-239
//     sdf_batch :112-114  inv_trans_batch :105-109  qrot_batch :95-102  qmul :73-81  w2quat :84-92  length :68-70
//   /root/reference/DaXBench/daxbench/core/engine/primitives/box.py  _sdf_batch :6-18
// Pinned by the reference's own recorded trajectory expert_demo/whip_rope/demo_0.pkl (tests/test_oracle_mpm.py);
// the adjoint is validated against torch.autograd on the line-by-line twin and f64 finite differences.
// The SVD itself is LAPACK in the reference (third-party); here a one-sided Jacobi (Hestenes) -- only the
// gauge-invariant products U S Vh, U Vh and S enter the dynamics.
//     normal_batch / _normal_batch (finite-difference normal) :117-141   collider_v_batch :144-151
//     collide_batch :154-182
//   /root/reference/DaXBench/daxbench/core/engine/primitives/container.py  _sdf_batch :8-16 (cut hollow sphere)
// Scope: one box primitive in position-control mode (whip_rope); any number of primitives with the box or the
// container SDF in soft-contact mode (collide_batch: shape_rope, pour_water).
// The soft-contact adjoint is validated against torch.autograd on the twin in f64 (tests/test_oracle_mpm.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "cloth_oracle.hpp"  // clipf, clip_grad, nan_to_num

namespace oracle {

template <class T> struct M3 { T a[3][3]; };

template <class T> inline M3<T> m3_zero() { M3<T> r; std::memset(&r, 0, sizeof(r)); return r; }
template <class T> inline M3<T> m3_eye() { M3<T> r = m3_zero<T>(); r.a[0][0] = r.a[1][1] = r.a[2][2] = 1; return r; }
template <class T> inline M3<T> mul(const M3<T>& A, const M3<T>& B) {
  M3<T> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[i][0] * B.a[0][j] + A.a[i][1] * B.a[1][j] + A.a[i][2] * B.a[2][j];
  return r;
}
template <class T> inline M3<T> tr(const M3<T>& A) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[j][i]; return r; }
template <class T> inline M3<T> add(const M3<T>& A, const M3<T>& B) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[i][j] + B.a[i][j]; return r; }
template <class T> inline M3<T> sub(const M3<T>& A, const M3<T>& B) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[i][j] - B.a[i][j]; return r; }
template <class T> inline M3<T> scale(const M3<T>& A, T s) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[i][j] * s; return r; }
template <class T> inline M3<T> load9(const T* p) { M3<T> r; for (int i = 0; i < 9; ++i) r.a[i / 3][i % 3] = p[i]; return r; }
template <class T> inline void store9(T* p, const M3<T>& A) { for (int i = 0; i < 9; ++i) p[i] = A.a[i / 3][i % 3]; }

// One-sided Jacobi SVD: A = U diag(S) Vh, S descending, S >= 0.
template <class T>
void svd3(const M3<T>& Ain, M3<T>& U, T S[3], M3<T>& Vh) {
  M3<T> A = Ain, V = m3_eye<T>();
  const T tiny = std::numeric_limits<T>::min();
  for (int sweep = 0; sweep < 12; ++sweep) {
    T off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        T al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += A.a[i][p] * A.a[i][p]; be += A.a[i][q] * A.a[i][q]; ga += A.a[i][p] * A.a[i][q]; }
        off = std::max(off, std::abs(ga) / std::sqrt(al * be + tiny));
        if (std::abs(ga) <= std::numeric_limits<T>::epsilon() * T(0.125) * std::sqrt(al * be)) continue;
        T zeta = (be - al) / (T(2) * ga);
        T t = (zeta >= 0 ? T(1) : T(-1)) / (std::abs(zeta) + std::sqrt(T(1) + zeta * zeta));
        T c = T(1) / std::sqrt(T(1) + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          T ap = A.a[i][p], aq = A.a[i][q];
          A.a[i][p] = c * ap - s * aq; A.a[i][q] = s * ap + c * aq;
          T vp = V.a[i][p], vq = V.a[i][q];
          V.a[i][p] = c * vp - s * vq; V.a[i][q] = s * vp + c * vq;
        }
      }
    if (off <= std::numeric_limits<T>::epsilon()) break;
  }
  T sv[3];
  for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(A.a[0][j] * A.a[0][j] + A.a[1][j] * A.a[1][j] + A.a[2][j] * A.a[2][j]);
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int x, int y) { return sv[x] > sv[y]; });
  for (int jj = 0; jj < 3; ++jj) {
    int j = ord[jj];
    S[jj] = sv[j];
    T inv = sv[j] > tiny ? T(1) / sv[j] : T(0);
    for (int i = 0; i < 3; ++i) { U.a[i][jj] = A.a[i][j] * inv; Vh.a[jj][i] = V.a[i][j]; }
  }
}

template <class T> inline T safe_inv(T x, T eps) { return x / (x * x + eps); }

// svd_safe_batch.py:65-102 for one real 3x3 matrix
template <class T>
M3<T> svd3_bwd(const M3<T>& U, const T S[3], const M3<T>& Vh, const M3<T>& dU, const T dS[3], const M3<T>& dVh, T eps = T(1e-12)) {
  M3<T> Ut = tr(U), Vt = Vh;                 // Cc(Vh) = Vh (real)
  M3<T> Vt_dV = mul(Vt, tr(dVh));
  T S2[3], Sinv[3];
  for (int i = 0; i < 3; ++i) { S2[i] = S[i] * S[i]; Sinv[i] = safe_inv(S[i], eps); }
  M3<T> Fm, J, K, L = m3_zero<T>();
  M3<T> UtdU = mul(Ut, dU);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Fm.a[i][j] = (i == j) ? T(0) : safe_inv(S2[j] - S2[i], eps);
      J.a[i][j] = Fm.a[i][j] * UtdU.a[i][j];
      K.a[i][j] = Fm.a[i][j] * Vt_dV.a[i][j];
      if (i == j) L.a[i][j] = Vt_dV.a[i][j];
    }
  M3<T> I = m3_eye<T>();
  M3<T> Pu = sub(I, mul(U, Ut));
  M3<T> Pv = sub(I, mul(tr(Vh), Vt));
  auto colscale = [](const M3<T>& A, const T s[3]) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[i][j] = A.a[i][j] * s[j]; return r; };
  M3<T> t1 = mul(colscale(U, dS), Vt);
  M3<T> t2 = mul(mul(U, colscale(add(J, tr(J)), S)), Vt);
  M3<T> t3 = mul(mul(colscale(U, S), add(K, tr(K))), Vt);
  M3<T> t4 = scale(mul(mul(colscale(U, Sinv), sub(L, tr(L))), Vt), T(0.5));
  M3<T> t5 = mul(mul(Pu, colscale(dU, Sinv)), Vt);
  M3<T> t6 = mul(mul(colscale(U, Sinv), dVh), Pv);
  return add(add(add(t1, t2), add(t3, t4)), add(t5, t6));
}

template <class T>
struct MpmParams {
  int N, n_grid, res[3], steps;
  T dt, dx, inv_dx, p_mass, p_vol;
  T stress_c;      // T(-dt*p_vol*4)   (:267)
  T dx2;           // T(dx**2)
  T dtg[3];        // T(dt)*T(gravity) (:285)
  int position_control;
  int n_prim = 1, sdf_kind = 0;                       // sdf_kind: 0 box (box.py), 1 container (container.py); one per process in the reference (set_sdf)
  T prim_friction = T(0.1), prim_softness = T(666);   // PrimitiveState.friction / .softness (primitives.py:31-60), same for all primitives ...
  std::vector<T> prim_friction_each, prim_softness_each;   // ... unless set per primitive (create_primitive passes them per primitive, mpm_env.py:201-217)
  std::vector<int> material;  // [N]
  std::vector<T> h;           // [N] hardness, clipped to [0.1,5] at use (:241)
};

template <class T>
struct PrimS {   // one primitive: position [steps*3], rotation [steps*4] (w,x,y,z), v, w [steps*3], size[3]
  std::vector<T> ppos, prot, pv, pw;
  T psize[3] = {0, 0, 0};
  void alloc(int steps) { ppos.assign(steps * 3, 0); prot.assign(steps * 4, 0); pv.assign(steps * 3, 0); pw.assign(steps * 3, 0); }
};

template <class T>
struct MpmState {
  std::vector<T> x, v, C, F, J;                 // [N*3],[N*3],[N*9],[N*9],[N]
  std::vector<PrimS<T>> prims;                  // PrimitiveState leaves the dynamics touch
  T friction, mu, lamda;
  void alloc(int N, int steps, int n_prim = 1) {
    x.assign(N * 3, 0); v.assign(N * 3, 0); C.assign(N * 9, 0); F.assign(N * 9, 0); J.assign(N, 0);
    prims.assign(n_prim, PrimS<T>());
    for (auto& p : prims) p.alloc(steps);
    friction = mu = lamda = 0;
  }
};

template <class T> inline int clampi(int i, int n) { return std::min(std::max(i, 0), n - 1); }

template <class T> inline void qrot(const T q[4], const T v[3], T out[3]) {  // :95-102
  T uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  T uuv[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
  for (int a = 0; a < 3; ++a) out[a] = v[a] + T(2) * (q[0] * uv[a] + uuv[a]);
}

template <class T> inline T box_sdf(const T size[3], const T gp[3]) {  // box.py:6-18
  T q[3];
  for (int a = 0; a < 3; ++a) q[a] = clipf(std::abs(gp[a]) - size[a], T(0), std::numeric_limits<T>::infinity());
  T out = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + T(1e-12));
  T tmp = q[1] > q[2] ? q[1] : q[2];
  tmp = q[0] > tmp ? q[0] : tmp;
  tmp = clipf(tmp, -std::numeric_limits<T>::infinity(), T(0));
  return out + tmp;
}

template <class T> inline T sdf_at(const T size[3], const T pos[3], const T rot[4], const T gp[3]) {  // :105-114
  T iq[4] = {rot[0], -rot[1], -rot[2], -rot[3]};
  T n = std::sqrt(iq[0] * iq[0] + iq[1] * iq[1] + iq[2] * iq[2] + iq[3] * iq[3]) + T(1e-12);
  for (int a = 0; a < 4; ++a) iq[a] = iq[a] / n;
  T d[3] = {gp[0] - pos[0], gp[1] - pos[1], gp[2] - pos[2]}, loc[3];
  qrot(iq, d, loc);
  return box_sdf(size, loc);
}

// ---- adjoint helpers -------------------------------------------------------------------------------------
// qrot adjoint: accumulates into gq[4], gv[3]
template <class T> inline void qrot_bwd(const T q[4], const T v[3], const T go[3], T gq[4], T gv[3]) {
  const T qv[3] = {q[1], q[2], q[3]};
  auto cross = [](const T a[3], const T b[3], T o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
  T uv[3];
  cross(qv, v, uv);
  gq[0] += T(2) * (go[0] * uv[0] + go[1] * uv[1] + go[2] * uv[2]);
  T guv[3] = {T(2) * q[0] * go[0], T(2) * q[0] * go[1], T(2) * q[0] * go[2]};
  const T guuv[3] = {T(2) * go[0], T(2) * go[1], T(2) * go[2]};
  T t[3];
  cross(uv, guuv, t);                       // uuv = qv x uv: g_qv += uv x g_uuv ; g_uv += g_uuv x qv
  for (int a = 0; a < 3; ++a) gq[1 + a] += t[a];
  cross(guuv, qv, t);
  for (int a = 0; a < 3; ++a) guv[a] += t[a];
  cross(v, guv, t);                         // uv = qv x v: g_qv += v x g_uv ; g_v += g_uv x qv
  for (int a = 0; a < 3; ++a) gq[1 + a] += t[a];
  cross(guv, qv, t);
  for (int a = 0; a < 3; ++a) gv[a] += go[a] + t[a];
}

// box_sdf adjoint (box.py:6-18): accumulates into gp[3], gsize[3]
template <class T> inline void box_sdf_bwd(const T size[3], const T p[3], T gout, T gp[3], T gsize[3]) {
  const T inf = std::numeric_limits<T>::infinity();
  T xr[3], q[3];
  for (int a = 0; a < 3; ++a) { xr[a] = std::abs(p[a]) - size[a]; q[a] = clipf(xr[a], T(0), inf); }
  const T len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + T(1e-12));
  const int s1 = q[1] > q[2] ? 1 : 2;
  const int sel = q[0] > q[s1] ? 0 : s1;
  const T gt = gout * clip_grad(q[sel], -inf, T(0));
  for (int a = 0; a < 3; ++a) {
    const T gq = gout * q[a] / len + (a == sel ? gt : T(0));
    const T gx = gq * clip_grad(xr[a], T(0), inf);
    const T sg = p[a] > 0 ? T(1) : (p[a] < 0 ? T(-1) : T(0));
    gp[a] += gx * sg;
    gsize[a] -= gx;
  }
}

// container.py:8-16 -- cut hollow sphere, size = (r, h, t)
template <class T> inline T container_sdf(const T size[3], const T p[3]) {
  const T r = size[0], h = size[1], t = size[2];
  const T w = std::sqrt(r * r - h * h);
  const T q0 = std::sqrt(p[0] * p[0] + p[2] * p[2] + T(1e-12)), q1 = p[1];
  const bool mask = h * q0 < w * q1;
  const T d0 = q0 - w, d1 = q1 - h;
  const T val1 = std::sqrt(d0 * d0 + d1 * d1 + T(1e-12)) - t;
  const T val2 = std::abs(std::sqrt(q0 * q0 + q1 * q1 + T(1e-12)) - r) - t;
  return mask ? val1 : val2;
}
template <class T> inline void container_sdf_bwd(const T size[3], const T p[3], T gout, T gp[3], T gsize[3]) {
  const T r = size[0], h = size[1], t = size[2];
  const T w = std::sqrt(r * r - h * h);
  const T q0 = std::sqrt(p[0] * p[0] + p[2] * p[2] + T(1e-12)), q1 = p[1];
  const bool mask = h * q0 < w * q1;
  T gq0, gq1, gw = 0, gr = 0, gh = 0;
  if (mask) {
    const T d0 = q0 - w, d1 = q1 - h, L1 = std::sqrt(d0 * d0 + d1 * d1 + T(1e-12));
    gq0 = gout * d0 / L1; gq1 = gout * d1 / L1; gw = -gq0; gh = -gq1;
  } else {
    const T L2 = std::sqrt(q0 * q0 + q1 * q1 + T(1e-12)), dd = L2 - r;
    const T sg = dd > 0 ? T(1) : (dd < 0 ? T(-1) : T(0));
    gq0 = gout * sg * q0 / L2; gq1 = gout * sg * q1 / L2; gr = -gout * sg;
  }
  gr += gw * r / w; gh -= gw * h / w;
  gp[0] += gq0 * p[0] / q0; gp[2] += gq0 * p[2] / q0; gp[1] += gq1;
  gsize[0] += gr; gsize[1] += gh; gsize[2] -= gout;
}
template <class T> inline T prim_sdf(int kind, const T size[3], const T p[3]) { return kind == 1 ? container_sdf(size, p) : box_sdf(size, p); }
template <class T> inline void prim_sdf_bwd(int kind, const T size[3], const T p[3], T gout, T gp[3], T gsize[3]) {
  if (kind == 1) container_sdf_bwd(size, p, gout, gp, gsize); else box_sdf_bwd(size, p, gout, gp, gsize);
}

// one primitive as the grid op of substep f sees it (rows f and f+1 of position / rotation, clamped: Q5)
template <class T> struct PrimCtx { T p0[3], r0[4], p1[3], r1[4], size[3], softness, friction; int kind; };
template <class T> struct PrimGrad {
  T p0[3], r0[4], p1[3], r1[4], size[3], friction;
  PrimGrad() { std::memset(this, 0, sizeof(*this)); }
};

template <class T> struct CollideRec {
  T iq[4], nq, rel[3], loc[3], e, infl, n[3], len, nl[3], D[3], cv[3], iv[3], nc, m, vt[3], vtn, arg, c, vtp[3];
  bool flag;
};

// collide_batch (:154-182) for one grid cell at world position gp: v -> v_out
template <class T>
inline void collide_cell(const PrimCtx<T>& pc, T dt, const T gp[3], const T v[3], T vout[3], CollideRec<T>& r) {
  const T inf = std::numeric_limits<T>::infinity();
  // inv_trans_batch (:105-109)
  const T cq[4] = {pc.r0[0], -pc.r0[1], -pc.r0[2], -pc.r0[3]};
  r.nq = std::sqrt(cq[0] * cq[0] + cq[1] * cq[1] + cq[2] * cq[2] + cq[3] * cq[3]) + T(1e-12);
  for (int a = 0; a < 4; ++a) r.iq[a] = cq[a] / r.nq;
  for (int a = 0; a < 3; ++a) r.rel[a] = gp[a] - pc.p0[a];
  qrot(r.iq, r.rel, r.loc);
  const T dist = prim_sdf(pc.kind, pc.size, r.loc);
  r.e = std::exp(-dist * pc.softness);
  r.infl = clipf(r.e, -inf, T(1));
  // _normal_batch (:117-134): central differences, d = 1e-6, in the primitive's frame
  const T d = T(1.e-6), k = T(0.5 / 1.e-6);
  for (int a = 0; a < 3; ++a) {
    T inc[3] = {r.loc[0], r.loc[1], r.loc[2]}, dec[3] = {r.loc[0], r.loc[1], r.loc[2]};
    inc[a] = inc[a] + d; dec[a] = dec[a] + (-d);
    r.n[a] = k * (prim_sdf(pc.kind, pc.size, inc) - prim_sdf(pc.kind, pc.size, dec));
  }
  r.len = std::sqrt(r.n[0] * r.n[0] + r.n[1] * r.n[1] + r.n[2] * r.n[2] + T(1e-12));
  for (int a = 0; a < 3; ++a) r.nl[a] = r.n[a] / r.len;
  qrot(pc.r0, r.nl, r.D);                                           // normal_batch :137-141
  // collider_v_batch (:144-151): relative_pos is the same expression as loc
  T np_[3];
  qrot(pc.r1, r.loc, np_);
  for (int a = 0; a < 3; ++a) r.cv[a] = ((np_[a] + pc.p1[a]) - gp[a]) / dt;
  for (int a = 0; a < 3; ++a) r.iv[a] = v[a] - r.cv[a];
  r.nc = r.iv[0] * r.D[0] + r.iv[1] * r.D[1] + r.iv[2] * r.D[2];
  r.m = clipf(r.nc, -inf, T(0));
  for (int a = 0; a < 3; ++a) r.vt[a] = r.iv[a] - r.m * r.D[a];
  const T vt_dot = r.vt[0] * r.vt[0] + r.vt[1] * r.vt[1] + r.vt[2] * r.vt[2];
  r.vtn = std::sqrt(vt_dot + T(1e-12));
  r.arg = r.vtn + r.nc * pc.friction;
  r.c = clipf(r.arg, T(1e-12), inf);
  r.flag = (r.nc < 0) && (std::sqrt(vt_dot) > T(1e-12));
  const T fl = r.flag ? T(1) : T(0);
  for (int a = 0; a < 3; ++a) {
    const T vtf = r.vt[a] / r.vtn * r.c;
    r.vtp[a] = vtf * fl + r.vt[a] * (T(1) - fl);
    vout[a] = r.cv[a] + r.iv[a] * (T(1) - r.infl) + r.vtp[a] * r.infl;
  }
}

// adjoint of collide_cell: gout (cotangent of v_out) -> gv (cotangent of v), accumulates the primitive's cotangents
template <class T>
inline void collide_cell_bwd(const PrimCtx<T>& pc, T dt, const T gp[3], const T v[3], const T gout[3], T gv[3], PrimGrad<T>& pg) {
  const T inf = std::numeric_limits<T>::infinity();
  CollideRec<T> r;
  T vo[3];
  collide_cell(pc, dt, gp, v, vo, r);
  T gcv[3], giv[3], gvtp[3], ginfl = 0;
  for (int a = 0; a < 3; ++a) {
    gcv[a] = gout[a]; giv[a] = gout[a] * (T(1) - r.infl); gvtp[a] = gout[a] * r.infl;
    ginfl += gout[a] * (r.vtp[a] - r.iv[a]);
  }
  T gvt[3] = {0, 0, 0}, gnc = 0;
  {
    const T fl = r.flag ? T(1) : T(0);
    T gvtf[3], u[3], gc = 0, gu_vt = 0;
    for (int a = 0; a < 3; ++a) { gvtf[a] = gvtp[a] * fl; gvt[a] = gvtp[a] * (T(1) - fl); u[a] = r.vt[a] / r.vtn; gc += gvtf[a] * u[a]; }
    T gvtn = 0;
    for (int a = 0; a < 3; ++a) { const T gu = gvtf[a] * r.c; gvt[a] += gu / r.vtn; gu_vt += gu * r.vt[a]; }
    gvtn -= gu_vt / (r.vtn * r.vtn);
    const T garg = gc * clip_grad(r.arg, T(1e-12), inf);
    gvtn += garg; gnc += garg * pc.friction; pg.friction += garg * r.nc;
    for (int a = 0; a < 3; ++a) gvt[a] += gvtn * r.vt[a] / r.vtn;
  }
  T gD[3], gm = 0;
  for (int a = 0; a < 3; ++a) { giv[a] += gvt[a]; gm -= gvt[a] * r.D[a]; gD[a] = -r.m * gvt[a]; }
  gnc += gm * clip_grad(r.nc, -inf, T(0));
  for (int a = 0; a < 3; ++a) { giv[a] += gnc * r.D[a]; gD[a] += gnc * r.iv[a]; }
  for (int a = 0; a < 3; ++a) { gv[a] = giv[a]; gcv[a] -= giv[a]; }
  // collider velocity
  T gloc[3] = {0, 0, 0}, gnp[3];
  for (int a = 0; a < 3; ++a) { gnp[a] = gcv[a] / dt; pg.p1[a] += gnp[a]; }
  qrot_bwd(pc.r1, r.loc, gnp, pg.r1, gloc);
  // normal: D = qrot(r0, nl), nl = n / len, n by central differences of the sdf
  T gnl[3] = {0, 0, 0};
  qrot_bwd(pc.r0, r.nl, gD, pg.r0, gnl);
  T dotn = gnl[0] * r.n[0] + gnl[1] * r.n[1] + gnl[2] * r.n[2];
  const T glen = -dotn / (r.len * r.len);
  const T d = T(1.e-6), k = T(0.5 / 1.e-6);
  for (int a = 0; a < 3; ++a) {
    const T gn = gnl[a] / r.len + glen * r.n[a] / r.len;
    T inc[3] = {r.loc[0], r.loc[1], r.loc[2]}, dec[3] = {r.loc[0], r.loc[1], r.loc[2]};
    inc[a] = inc[a] + d; dec[a] = dec[a] + (-d);
    prim_sdf_bwd(pc.kind, pc.size, inc, k * gn, gloc, pg.size);
    prim_sdf_bwd(pc.kind, pc.size, dec, -(k * gn), gloc, pg.size);
  }
  // influence
  const T ge = ginfl * clip_grad(r.e, -inf, T(1));
  const T gdist = -(ge * r.e) * pc.softness;
  prim_sdf_bwd(pc.kind, pc.size, r.loc, gdist, gloc, pg.size);
  // loc = qrot(iq, gp - p0), iq = conj(r0) / (|conj(r0)| + 1e-12)
  T giq[4] = {0, 0, 0, 0}, grel[3] = {0, 0, 0};
  qrot_bwd(r.iq, r.rel, gloc, giq, grel);
  for (int a = 0; a < 3; ++a) pg.p0[a] -= grel[a];
  const T cq[4] = {pc.r0[0], -pc.r0[1], -pc.r0[2], -pc.r0[3]};
  const T nrm = r.nq - T(1e-12);
  T dq = 0;
  for (int a = 0; a < 4; ++a) dq += giq[a] * cq[a];
  const T gnq = -dq / (r.nq * r.nq);
  for (int a = 0; a < 4; ++a) {
    const T gcq = giq[a] / r.nq + gnq * cq[a] / nrm;
    pg.r0[a] += (a == 0) ? gcq : -gcq;
  }
}

// forward_kinematics (:185-194) in place on (ppos, prot)
template <class T> void fk(int f, int steps, PrimS<T>& P_) {
  std::vector<T>&ppos = P_.ppos, &prot = P_.prot;
  const std::vector<T>&pv = P_.pv, &pw = P_.pw;
  const int fc = clampi<T>(f, steps);
  if (f + 1 < steps && f + 1 >= 0)
    for (int a = 0; a < 3; ++a) ppos[(f + 1) * 3 + a] = ppos[fc * 3 + a] + pv[fc * 3 + a];
  for (auto& p : ppos) p = clipf(p, T(-2), T(2));
  // rotation[f+1] = qmul(w2quat(w[f]), rotation[f])
  const T* w = &pw[fc * 3];
  T ang = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) + T(1e-12);
  T sn = std::sin(ang / 2);
  T q[4] = {std::cos(ang / 2), w[0] / ang * sn, w[1] / ang * sn, w[2] / ang * sn};
  const T* r = &prot[fc * 4];
  // terms = outer(r, q): terms[i][j] = r[i]*q[j]
  T o[4] = {r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3],
            r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2],
            r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1],
            r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0]};
  T nn = clipf(std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]), T(1e-12), std::numeric_limits<T>::infinity());
  if (f + 1 < steps && f + 1 >= 0)
    for (int a = 0; a < 4; ++a) prot[(f + 1) * 4 + a] = o[a] / nn;
}

// per-particle quantities of the p2g pre-pass
template <class T>
struct Pre {
  int base[3];
  T fx[3], w[3][3];
  M3<T> Fu, Fn, U, Vh, R, A, affine;   // Fu=(I+dtC)F ; Fn = after plastic projection ; A = Fn - R
  T sig_raw[3], sig[3], J, mu, la;
};

template <class T>
void particle_pre(const MpmParams<T>& pr, const MpmState<T>& st, int p, Pre<T>& q) {
  for (int d = 0; d < 3; ++d) {
    q.base[d] = (int)(st.x[p * 3 + d] * pr.inv_dx - T(0.5));  // :233 truncation
    q.fx[d] = st.x[p * 3 + d] * pr.inv_dx - (T)q.base[d];
    T f = q.fx[d];
    q.w[0][d] = T(0.5) * ((T(1.5) - f) * (T(1.5) - f));
    q.w[1][d] = T(0.75) - (f - T(1)) * (f - T(1));
    q.w[2][d] = T(0.5) * ((f - T(0.5)) * (f - T(0.5)));
  }
  M3<T> C = load9(&st.C[p * 9]), F = load9(&st.F[p * 9]);
  q.Fu = mul(add(m3_eye<T>(), scale(C, pr.dt)), F);  // :238
  T h = clipf(pr.h[p], T(0.1), T(5));
  q.mu = st.mu * h; q.la = st.lamda * h;
  if (pr.material[p] == 0) { q.mu = 0; q.la = 1; }   // :243-244 (Q10)
  svd3(q.Fu, q.U, q.sig_raw, q.Vh);
  for (int i = 0; i < 3; ++i) q.sig[i] = q.sig_raw[i];
  q.Fn = q.Fu;
  if (pr.material[p] == 2) {                          // :250-258
    for (int i = 0; i < 3; ++i) q.sig[i] = clipf(q.sig_raw[i], T(1 - 2.5e-2 * 10), T(1 + 4.5e-3 * 100));
    M3<T> US = q.U;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) US.a[i][j] *= q.sig[j];
    q.Fn = mul(US, q.Vh);
  }
  q.J = q.sig[0] * q.sig[1] * q.sig[2];
  q.R = mul(q.U, q.Vh);
  q.A = sub(q.Fn, q.R);
  M3<T> st_ = scale(mul(q.A, tr(q.Fn)), T(2) * q.mu);  // :265
  T vol = q.la * q.J * (q.J - T(1));
  for (int i = 0; i < 3; ++i) st_.a[i][i] += vol;        // :266
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) st_.a[i][j] = pr.stress_c * st_.a[i][j] / pr.dx2;  // :267
  q.affine = add(st_, scale(C, pr.p_mass));               // :268
}

// scatter index rule (Q5/Q9): negative wraps, out-of-range dropped -> returns -1
template <class T> inline long cell_scatter(const MpmParams<T>& pr, int i, int j, int k) {
  int id[3] = {i, j, k};
  for (int d = 0; d < 3; ++d) { if (id[d] < 0) id[d] += pr.res[d]; if (id[d] < 0 || id[d] >= pr.res[d]) return -1; }
  return ((long)id[0] * pr.res[1] + id[1]) * pr.res[2] + id[2];
}
// gather index rule: negative wraps, then clamp
template <class T> inline long cell_gather(const MpmParams<T>& pr, int i, int j, int k) {
  int id[3] = {i, j, k};
  for (int d = 0; d < 3; ++d) { if (id[d] < 0) id[d] += pr.res[d]; id[d] = std::min(std::max(id[d], 0), pr.res[d] - 1); }
  return ((long)id[0] * pr.res[1] + id[1]) * pr.res[2] + id[2];
}

template <class T>
inline PrimCtx<T> prim_ctx(const MpmParams<T>& pr, const PrimS<T>& st, int f, int ip = 0) {
  const int f0 = clampi<T>(f, pr.steps), f1 = clampi<T>(f + 1, pr.steps);
  PrimCtx<T> pc;
  for (int a = 0; a < 3; ++a) { pc.p0[a] = st.ppos[f0 * 3 + a]; pc.p1[a] = st.ppos[f1 * 3 + a]; pc.size[a] = st.psize[a]; }
  for (int a = 0; a < 4; ++a) { pc.r0[a] = st.prot[f0 * 4 + a]; pc.r1[a] = st.prot[f1 * 4 + a]; }
  pc.softness = ip < (int)pr.prim_softness_each.size() ? pr.prim_softness_each[ip] : pr.prim_softness;
  pc.friction = ip < (int)pr.prim_friction_each.size() ? pr.prim_friction_each[ip] : pr.prim_friction; pc.kind = pr.sdf_kind;
  return pc;
}

// per-cell record of the grid op (enough to run its adjoint)
template <class T>
struct CellOp {
  T v0[3];      // after normalise + gravity
  T v1[3];      // after primitive op
  T v2[3];      // after friction
  uint8_t ctrl, fric, bnd[3];
};

template <class T>
inline void grid_cell_op(const MpmParams<T>& pr, const MpmState<T>& st, int f, int ci, int cj, int ck, T m, const T mv[3],
                         T vout[3], CellOp<T>* rec) {
  T v[3];
  for (int a = 0; a < 3; ++a) v[a] = (m > 0) ? mv[a] / m : mv[a];   // :283-284
  for (int a = 0; a < 3; ++a) v[a] += pr.dtg[a];                    // :285
  if (rec) for (int a = 0; a < 3; ++a) rec->v0[a] = v[a];
  bool ctrl = false;
  if (pr.position_control) {                                        // :232-239
    const int fc = clampi<T>(f, pr.steps);
    const PrimS<T>& P0 = st.prims[0];                               // position control: one primitive (whip_rope)
    T gp[3] = {(T)ci * pr.dx, (T)cj * pr.dx, (T)ck * pr.dx};
    T dist = sdf_at(P0.psize, &P0.ppos[fc * 3], &P0.prot[fc * 4], gp);
    ctrl = dist < P0.psize[0] * T(1.5);
    if (ctrl) for (int a = 0; a < 3; ++a) v[a] = P0.pv[fc * 3 + a] / pr.dt;
  } else {                                                          // collide_batch (:154-182), primitive after primitive (:292-294)
    T gp[3] = {(T)ci * pr.dx, (T)cj * pr.dx, (T)ck * pr.dx};
    for (int i = 0; i < pr.n_prim; ++i) {
      PrimCtx<T> pc = prim_ctx(pr, st.prims[i], f, i);
      CollideRec<T> cr;
      T vo[3];
      collide_cell(pc, pr.dt, gp, v, vo, cr);
      for (int a = 0; a < 3; ++a) v[a] = vo[a];
    }
  }
  if (rec) { rec->ctrl = ctrl; for (int a = 0; a < 3; ++a) rec->v1[a] = v[a]; }
  // friction (:297-307)
  bool fric = (cj < 3) && (v[1] <= 0);
  if (fric) {
    T gi[3] = {(T)ci, (T)cj, (T)ck};
    T lin = v[1] + T(1e-30);
    T vit[3] = {v[0] - lin * T(0) - gi[0] * T(1e-30), v[1] - lin * T(1) - gi[1] * T(1e-30), v[2] - lin * T(0) - gi[2] * T(1e-30)};
    T lit = std::sqrt((vit[0] + T(1e-12)) * (vit[0] + T(1e-12)) + (vit[1] + T(1e-12)) * (vit[1] + T(1e-12)) + (vit[2] + T(1e-12)) * (vit[2] + T(1e-12)));
    T s = clipf(T(1) + st.friction * lin / lit, T(0), std::numeric_limits<T>::infinity());
    v[0] = s * (vit[0] + gi[0] * T(1e-30));
    v[2] = s * (vit[2] + gi[2] * T(1e-30));
    v[1] = 0;
  }
  if (rec) { rec->fric = fric; for (int a = 0; a < 3; ++a) rec->v2[a] = v[a]; }
  // boundary (:310-313, Q8: upper bound uses n_grid, not res)
  int id[3] = {ci, cj, ck};
  for (int a = 0; a < 3; ++a) {
    bool c = (id[a] < 3 && v[a] < 0) || (id[a] > pr.n_grid - 3 && v[a] > 0);
    if (rec) rec->bnd[a] = c;
    if (c) v[a] = 0;
  }
  for (int a = 0; a < 3; ++a) vout[a] = v[a];
}

// ---- forward substep (:223-330) ---------------------------------------------------------------------
template <class T>
void mpm_substep_fwd(const MpmParams<T>& pr, int f, const MpmState<T>& in, MpmState<T>& out, std::vector<T>& gm,
                     std::vector<T>& gv) {
  const int N = pr.N;
  const size_t G = (size_t)pr.res[0] * pr.res[1] * pr.res[2];
  gm.assign(G, 0); gv.assign(G * 3, 0);
  out = in;
  std::vector<Pre<T>> pre(N);
  for (int p = 0; p < N; ++p) {
    Pre<T>& q = pre[p];
    particle_pre(pr, in, p, q);
    store9(&out.F[p * 9], q.Fn);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {  // :178-194
      T weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      T dpos[3] = {((T)i - q.fx[0]) * pr.dx, ((T)j - q.fx[1]) * pr.dx, ((T)k - q.fx[2]) * pr.dx};
      long c = cell_scatter(pr, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      if (c < 0) continue;
      gm[c] += weight * pr.p_mass;
      for (int a = 0; a < 3; ++a) {
        T ad = q.affine.a[a][0] * dpos[0] + q.affine.a[a][1] * dpos[1] + q.affine.a[a][2] * dpos[2];
        gv[c * 3 + a] += weight * (pr.p_mass * in.v[p * 3 + a] + ad);
      }
    }
  }
  for (auto& P_ : out.prims) fk(f, pr.steps, P_);  // :277-278
  // grid op (dense, like the source)
  for (int ci = 0; ci < pr.res[0]; ++ci) for (int cj = 0; cj < pr.res[1]; ++cj) for (int ck = 0; ck < pr.res[2]; ++ck) {
    size_t c = ((size_t)ci * pr.res[1] + cj) * pr.res[2] + ck;
    T vo[3];
    grid_cell_op<T>(pr, out, f, ci, cj, ck, gm[c], &gv[c * 3], vo, nullptr);
    for (int a = 0; a < 3; ++a) gv[c * 3 + a] = vo[a];
  }
  // g2p (:196-221, :318-328)
  for (int p = 0; p < N; ++p) {
    const Pre<T>& q = pre[p];
    T nv[3] = {0, 0, 0};
    M3<T> nC = m3_zero<T>();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      T weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      T dp[3] = {(T)i - q.fx[0], (T)j - q.fx[1], (T)k - q.fx[2]};
      long c = cell_gather(pr, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      for (int a = 0; a < 3; ++a) {
        nv[a] += weight * gv[c * 3 + a];
        for (int b = 0; b < 3; ++b) nC.a[a][b] += T(4) * weight * (gv[c * 3 + a] * dp[b]) * pr.inv_dx;
      }
    }
    for (int a = 0; a < 3; ++a) { out.v[p * 3 + a] = nv[a]; out.x[p * 3 + a] = in.x[p * 3 + a] + pr.dt * nv[a]; }
    store9(&out.C[p * 9], nC);
  }
  // J (:327, Q6): C_.trace() over axes (0,1) of [N,3,3], summed -> one scalar for all particles
  T trq = 0;
  {
    T vec[3] = {0, 0, 0};
    for (int i = 0; i < std::min(3, N); ++i) for (int b = 0; b < 3; ++b) vec[b] += out.C[i * 9 + i * 3 + b];
    trq = vec[0] + vec[1] + vec[2];
  }
  for (int p = 0; p < N; ++p) out.J[p] = in.J[p] * (T(1) + pr.dt * trq);
}

// cotangents of one MPM state (same leaves the reference differentiates, :343-354; J excluded)
template <class T>
struct MpmGrad {
  std::vector<T> x, v, C, F;
  struct PrimG {
    std::vector<T> ppos, pv, prot, pw;
    T psize[3] = {0, 0, 0}, pfriction = 0;   // primitive size / friction leaves: only their norm enters (norm_grad_state)
  };
  std::vector<PrimG> prims;
  T friction = 0, mu = 0, lamda = 0;
  void alloc(int N, int steps, int n_prim = 1) {
    x.assign(N * 3, 0); v.assign(N * 3, 0); C.assign(N * 9, 0); F.assign(N * 9, 0);
    prims.assign(n_prim, PrimG());
    for (auto& p : prims) { p.ppos.assign(steps * 3, 0); p.pv.assign(steps * 3, 0); p.prot.assign(steps * 4, 0); p.pw.assign(steps * 3, 0); }
  }
};

// ---- adjoint of one substep: g (cotangent of the outputs) -> g (cotangent of the inputs), in place -----
template <class T>
void mpm_substep_bwd(const MpmParams<T>& pr, int f, const MpmState<T>& in, MpmGrad<T>& g) {
  const int N = pr.N;
  const size_t G = (size_t)pr.res[0] * pr.res[1] * pr.res[2];
  const T NaN = std::numeric_limits<T>::quiet_NaN();
  // recompute forward
  std::vector<T> gm(G, 0), gmv(G * 3, 0), gvel(G * 3, 0);
  std::vector<Pre<T>> pre(N);
  for (int p = 0; p < N; ++p) {
    Pre<T>& q = pre[p];
    particle_pre(pr, in, p, q);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      T weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      T dpos[3] = {((T)i - q.fx[0]) * pr.dx, ((T)j - q.fx[1]) * pr.dx, ((T)k - q.fx[2]) * pr.dx};
      long c = cell_scatter(pr, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      if (c < 0) continue;
      gm[c] += weight * pr.p_mass;
      for (int a = 0; a < 3; ++a) {
        T ad = q.affine.a[a][0] * dpos[0] + q.affine.a[a][1] * dpos[1] + q.affine.a[a][2] * dpos[2];
        gmv[c * 3 + a] += weight * (pr.p_mass * in.v[p * 3 + a] + ad);
      }
    }
  }
  MpmState<T> mid = in;
  for (auto& P_ : mid.prims) fk(f, pr.steps, P_);
  std::vector<CellOp<T>> rec(G);
  for (int ci = 0; ci < pr.res[0]; ++ci) for (int cj = 0; cj < pr.res[1]; ++cj) for (int ck = 0; ck < pr.res[2]; ++ck) {
    size_t c = ((size_t)ci * pr.res[1] + cj) * pr.res[2] + ck;
    grid_cell_op<T>(pr, mid, f, ci, cj, ck, gm[c], &gmv[c * 3], &gvel[c * 3], &rec[c]);
  }
  // ---- reverse ------------------------------------------------------------------------------------
  // x_out = x + dt*v_new ; J excluded
  std::vector<T> gnv(N * 3), ggv(G * 3, 0), ggm(G, 0);
  std::vector<M3<T>> gnC(N);
  for (int p = 0; p < N; ++p) {
    for (int a = 0; a < 3; ++a) gnv[p * 3 + a] = g.v[p * 3 + a] + pr.dt * g.x[p * 3 + a];
    gnC[p] = load9(&g.C[p * 9]);
  }
  // g2p adjoint: scatter onto grid velocities; weight / fx cotangents
  std::vector<T> gw(N * 9, 0), gfx(N * 3, 0);   // gw[p][k][d]
  for (int p = 0; p < N; ++p) {
    const Pre<T>& q = pre[p];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      T weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      T dp[3] = {(T)i - q.fx[0], (T)j - q.fx[1], (T)k - q.fx[2]};
      long c = cell_gather(pr, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      T gwt = 0;
      for (int a = 0; a < 3; ++a) {
        T gCd = gnC[p].a[a][0] * dp[0] + gnC[p].a[a][1] * dp[1] + gnC[p].a[a][2] * dp[2];
        ggv[c * 3 + a] += weight * gnv[p * 3 + a] + T(4) * pr.inv_dx * weight * gCd;
        gwt += gvel[c * 3 + a] * (gnv[p * 3 + a] + T(4) * pr.inv_dx * gCd);
        for (int b = 0; b < 3; ++b) gfx[p * 3 + b] -= T(4) * pr.inv_dx * weight * gnC[p].a[a][b] * gvel[c * 3 + a];
      }
      gw[p * 9 + i * 3 + 0] += gwt * q.w[j][1] * q.w[k][2];
      gw[p * 9 + j * 3 + 1] += gwt * q.w[i][0] * q.w[k][2];
      gw[p * 9 + k * 3 + 2] += gwt * q.w[i][0] * q.w[j][1];
    }
  }
  // grid-op adjoint per cell
  const int fc = clampi<T>(f, pr.steps);
  T gpv_f[3] = {0, 0, 0};
  std::vector<PrimCtx<T>> pctx;
  for (int i = 0; i < pr.n_prim; ++i) pctx.push_back(prim_ctx(pr, mid.prims[i], f, i));
  std::vector<PrimGrad<T>> pgrad(pr.n_prim);
  for (int ci = 0; ci < pr.res[0]; ++ci) for (int cj = 0; cj < pr.res[1]; ++cj) for (int ck = 0; ck < pr.res[2]; ++ck) {
    size_t c = ((size_t)ci * pr.res[1] + cj) * pr.res[2] + ck;
    const CellOp<T>& r = rec[c];
    T gvv[3] = {ggv[c * 3], ggv[c * 3 + 1], ggv[c * 3 + 2]};
    for (int a = 0; a < 3; ++a) if (r.bnd[a]) gvv[a] = 0;          // boundary
    if (r.fric) {                                                   // friction
      const T* v = r.v1;
      T gi[3] = {(T)ci, (T)cj, (T)ck};
      T lin = v[1] + T(1e-30);
      T vit[3] = {v[0] - gi[0] * T(1e-30), v[1] - lin - gi[1] * T(1e-30), v[2] - gi[2] * T(1e-30)};
      T e[3] = {vit[0] + T(1e-12), vit[1] + T(1e-12), vit[2] + T(1e-12)};
      T lit = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
      T arg = T(1) + in.friction * lin / lit;
      T s = clipf(arg, T(0), std::numeric_limits<T>::infinity());
      T qv[3] = {vit[0] + gi[0] * T(1e-30), vit[1] + gi[1] * T(1e-30), vit[2] + gi[2] * T(1e-30)};
      T gs = gvv[0] * qv[0] + gvv[2] * qv[2];
      T gvit[3] = {s * gvv[0], 0, s * gvv[2]};
      T garg = gs * clip_grad(arg, T(0), std::numeric_limits<T>::infinity());
      g.friction += garg * lin / lit;
      T glin = garg * in.friction / lit;
      T glit = -garg * in.friction * lin / (lit * lit);
      for (int a = 0; a < 3; ++a) gvit[a] += glit * e[a] / lit;
      T gin[3] = {gvit[0], gvit[1], gvit[2]};
      glin -= gvit[1];
      gin[1] += glin;
      for (int a = 0; a < 3; ++a) gvv[a] = gin[a];
    }
    if (r.ctrl) {                                                   // position control
      for (int a = 0; a < 3; ++a) { gpv_f[a] += gvv[a] / pr.dt; gvv[a] = 0; }
    }
    if (!pr.position_control) {                                     // collide_batch
      T gp[3] = {(T)ci * pr.dx, (T)cj * pr.dx, (T)ck * pr.dx}, gin[3];
      T vin[8][3];                                                  // input velocity of each primitive's collide
      for (int a = 0; a < 3; ++a) vin[0][a] = r.v0[a];
      for (int i = 0; i + 1 < pr.n_prim; ++i) { CollideRec<T> cr; collide_cell(pctx[i], pr.dt, gp, vin[i], vin[i + 1], cr); }
      for (int i = pr.n_prim - 1; i >= 0; --i) {
        collide_cell_bwd(pctx[i], pr.dt, gp, vin[i], gvv, gin, pgrad[i]);
        for (int a = 0; a < 3; ++a) gvv[a] = gin[a];
      }
    }
    // gravity: pass. normalise (:283-284, Q7)
    T m = gm[c];
    if (m > 0) {
      T gmm = 0;
      for (int a = 0; a < 3; ++a) { T vn = gmv[c * 3 + a] / m; gmm -= gvv[a] * vn / m; ggv[c * 3 + a] = gvv[a] / m; }
      ggm[c] = gmm;
    } else if (m == 0) {
      // jax: where(m>0, mv/m, mv): the mv/m branch gets cotangent 0 and 0/0 -> NaN for both mv and m (Q7)
      for (int a = 0; a < 3; ++a) ggv[c * 3 + a] = NaN;
      ggm[c] = NaN;
    } else {
      // m < 0 (negative quadratic weights when x*inv_dx < 0.5, truncating base :233): pass-through branch
      for (int a = 0; a < 3; ++a) ggv[c * 3 + a] = gvv[a];
      ggm[c] = 0;
    }
  }
  for (int ip = 0; ip < pr.n_prim; ++ip) {
  auto& G = g.prims[ip];
  const PrimS<T>& IN = in.prims[ip];
  if (!pr.position_control) {   // rows f and f+1 (clamped) of position / rotation as the grid op read them
    const int f1 = clampi<T>(f + 1, pr.steps);
    for (int a = 0; a < 3; ++a) { G.ppos[fc * 3 + a] += pgrad[ip].p0[a]; G.ppos[f1 * 3 + a] += pgrad[ip].p1[a]; G.psize[a] += pgrad[ip].size[a]; }
    for (int a = 0; a < 4; ++a) { G.prot[fc * 4 + a] += pgrad[ip].r0[a]; G.prot[f1 * 4 + a] += pgrad[ip].r1[a]; }
    G.pfriction += pgrad[ip].friction;
    // rotation' = set(rotation, f+1, qmul(w2quat(w[f]), rotation[f]))   (:190, qmul :73-81, w2quat :84-92)
    if (f + 1 < pr.steps) {
      const T inf = std::numeric_limits<T>::infinity();
      T go_[4];
      for (int a = 0; a < 4; ++a) { go_[a] = G.prot[(f + 1) * 4 + a]; G.prot[(f + 1) * 4 + a] = 0; }
      const T* w = &IN.pw[fc * 3];
      const T* rr = &IN.prot[fc * 4];
      const T s = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
      const T nrm = std::sqrt(s), ang = nrm + T(1e-12), hh = ang / 2, sn = std::sin(hh), cs = std::cos(hh);
      const T u[3] = {w[0] / ang, w[1] / ang, w[2] / ang};
      const T q[4] = {cs, u[0] * sn, u[1] * sn, u[2] * sn};
      const T o[4] = {rr[0] * q[0] - rr[1] * q[1] - rr[2] * q[2] - rr[3] * q[3],
                      rr[0] * q[1] + rr[1] * q[0] - rr[2] * q[3] + rr[3] * q[2],
                      rr[0] * q[2] + rr[1] * q[3] + rr[2] * q[0] - rr[3] * q[1],
                      rr[0] * q[3] - rr[1] * q[2] + rr[2] * q[1] + rr[3] * q[0]};
      const T oo = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
      const T nn = clipf(oo, T(1e-12), inf);
      T dot = 0;
      for (int a = 0; a < 4; ++a) dot += go_[a] * o[a];
      const T goo = -dot / (nn * nn) * clip_grad(oo, T(1e-12), inf);
      T gO[4];
      for (int a = 0; a < 4; ++a) gO[a] = go_[a] / nn + goo * o[a] / oo;
      const T gr[4] = {gO[0] * q[0] + gO[1] * q[1] + gO[2] * q[2] + gO[3] * q[3],
                       -gO[0] * q[1] + gO[1] * q[0] + gO[2] * q[3] - gO[3] * q[2],
                       -gO[0] * q[2] - gO[1] * q[3] + gO[2] * q[0] + gO[3] * q[1],
                       -gO[0] * q[3] + gO[1] * q[2] - gO[2] * q[1] + gO[3] * q[0]};
      const T gq[4] = {gO[0] * rr[0] + gO[1] * rr[1] + gO[2] * rr[2] + gO[3] * rr[3],
                       -gO[0] * rr[1] + gO[1] * rr[0] - gO[2] * rr[3] + gO[3] * rr[2],
                       -gO[0] * rr[2] + gO[1] * rr[3] + gO[2] * rr[0] - gO[3] * rr[1],
                       -gO[0] * rr[3] - gO[1] * rr[2] + gO[2] * rr[1] + gO[3] * rr[0]};
      for (int a = 0; a < 4; ++a) G.prot[fc * 4 + a] += gr[a];
      T gh = -sn * gq[0], gsn = 0, gang = 0, gw[3];
      for (int a = 0; a < 3; ++a) { gsn += gq[1 + a] * u[a]; const T gu = gq[1 + a] * sn; gw[a] = gu / ang; gang -= gu * w[a] / (ang * ang); }
      gh += cs * gsn;
      gang += gh / 2;
      // |w| = sqrt(sum w^2): at w = 0 the reference's chain rule gives 0.5/0 * 0 = NaN (laundered by nan_to_num at `step`)
      const T gs = gang * (T(0.5) / nrm);
      for (int a = 0; a < 3; ++a) G.pw[fc * 3 + a] += gw[a] + gs * (T(2) * w[a]);
    }
  }
  // primitives: FK adjoint (:185-194). position' = clip(set(position, f+1, position[f]+v[f]))
  {
    std::vector<T> gp = G.ppos;
    // clip factors evaluated on the pre-clip array
    std::vector<T> pre_clip = IN.ppos;
    if (f + 1 < pr.steps) for (int a = 0; a < 3; ++a) pre_clip[(f + 1) * 3 + a] = IN.ppos[fc * 3 + a] + IN.pv[fc * 3 + a];
    for (size_t i = 0; i < gp.size(); ++i) gp[i] *= clip_grad(pre_clip[i], T(-2), T(2));
    if (f + 1 < pr.steps) {
      for (int a = 0; a < 3; ++a) {
        T t = gp[(f + 1) * 3 + a];
        gp[(f + 1) * 3 + a] = 0;
        gp[fc * 3 + a] += t;
        G.pv[fc * 3 + a] += t;
      }
    }
    G.ppos = gp;
    if (ip == 0) for (int a = 0; a < 3; ++a) G.pv[fc * 3 + a] += gpv_f[a];
  }
  }   // primitives
  // p2g adjoint (gather) + particle pre-pass adjoint
  T gmu_tot = 0, gla_tot = 0;
  for (int p = 0; p < N; ++p) {
    const Pre<T>& q = pre[p];
    M3<T> gaff = m3_zero<T>();
    T gvp[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      long c = cell_scatter(pr, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      if (c < 0) continue;
      T weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      T dpos[3] = {((T)i - q.fx[0]) * pr.dx, ((T)j - q.fx[1]) * pr.dx, ((T)k - q.fx[2]) * pr.dx};
      T gwt = pr.p_mass * ggm[c];
      for (int a = 0; a < 3; ++a) {
        T gc = ggv[c * 3 + a];
        T ad = q.affine.a[a][0] * dpos[0] + q.affine.a[a][1] * dpos[1] + q.affine.a[a][2] * dpos[2];
        gwt += gc * (pr.p_mass * in.v[p * 3 + a] + ad);
        gvp[a] += weight * pr.p_mass * gc;
        for (int b = 0; b < 3; ++b) {
          gaff.a[a][b] += weight * gc * dpos[b];
          gfx[p * 3 + b] -= pr.dx * weight * gc * q.affine.a[a][b];
        }
      }
      gw[p * 9 + i * 3 + 0] += gwt * q.w[j][1] * q.w[k][2];
      gw[p * 9 + j * 3 + 1] += gwt * q.w[i][0] * q.w[k][2];
      gw[p * 9 + k * 3 + 2] += gwt * q.w[i][0] * q.w[j][1];
    }
    // weights -> fx -> x
    for (int d = 0; d < 3; ++d) {
      T fxd = q.fx[d];
      gfx[p * 3 + d] += gw[p * 9 + 0 * 3 + d] * (-(T(1.5) - fxd)) + gw[p * 9 + 1 * 3 + d] * (-T(2) * (fxd - T(1))) +
                        gw[p * 9 + 2 * 3 + d] * (fxd - T(0.5));
    }
    // affine = stress + p_mass*C
    M3<T> C = load9(&in.C[p * 9]), F = load9(&in.F[p * 9]);
    M3<T> gC = scale(gaff, pr.p_mass);
    M3<T> gS = gaff;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gS.a[i][j] = gS.a[i][j] / pr.dx2 * pr.stress_c;
    // stress = 2 mu A Fn^T + la J (J-1) I
    M3<T> gFn = load9(&g.F[p * 9]);    // F_out = Fn
    M3<T> AFt = mul(q.A, tr(q.Fn));
    T gmu_p = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gmu_p += gS.a[i][j] * T(2) * AFt.a[i][j];
    M3<T> gA = scale(mul(gS, q.Fn), T(2) * q.mu);
    gFn = add(gFn, scale(mul(tr(gS), q.A), T(2) * q.mu));
    T trg = gS.a[0][0] + gS.a[1][1] + gS.a[2][2];
    T gJ = q.la * (T(2) * q.J - T(1)) * trg;
    T gla_p = q.J * (q.J - T(1)) * trg;
    gFn = add(gFn, gA);
    M3<T> gR = scale(gA, T(-1));
    M3<T> gU = mul(gR, tr(q.Vh)), gVh = mul(tr(q.U), gR);
    T gsig[3] = {gJ * q.sig[1] * q.sig[2], gJ * q.sig[0] * q.sig[2], gJ * q.sig[0] * q.sig[1]};
    M3<T> gFu;
    if (pr.material[p] == 2) {
      // Fn = U diag(sig) Vh
      M3<T> US = q.U, SV = q.Vh;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { US.a[i][j] *= q.sig[j]; SV.a[i][j] *= q.sig[i]; }
      gU = add(gU, mul(gFn, tr(SV)));
      gVh = add(gVh, mul(tr(US), gFn));
      M3<T> UtG = mul(mul(tr(q.U), gFn), tr(q.Vh));
      for (int i = 0; i < 3; ++i) gsig[i] += UtG.a[i][i];
      for (int i = 0; i < 3; ++i) gsig[i] *= clip_grad(q.sig_raw[i], T(1 - 2.5e-2 * 10), T(1 + 4.5e-3 * 100));
      gFu = m3_zero<T>();
    } else {
      gFu = gFn;
    }
    gFu = add(gFu, svd3_bwd(q.U, q.sig_raw, q.Vh, gU, gsig, gVh));
    // Fu = (I + dt C) F
    gC = add(gC, scale(mul(gFu, tr(F)), pr.dt));
    M3<T> gF = mul(tr(add(m3_eye<T>(), scale(C, pr.dt))), gFu);
    T h = clipf(pr.h[p], T(0.1), T(5));
    if (pr.material[p] != 0) { gmu_tot += gmu_p * h; gla_tot += gla_p * h; }
    for (int a = 0; a < 3; ++a) {
      g.x[p * 3 + a] = g.x[p * 3 + a] + gfx[p * 3 + a] * pr.inv_dx;
      g.v[p * 3 + a] = gvp[a];
    }
    store9(&g.C[p * 9], gC);
    store9(&g.F[p * 9], gF);
  }
  g.mu += gmu_tot;
  g.lamda += gla_tot;
}

}  // namespace oracle
