// ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
// Never linked into, imported by, or measured as the product path.
//
// CPU restatement of the reference's mass-spring cloth substep and its (normalised) adjoint:
//   /root/reference/DaXBench/daxbench/core/engine/cloth_simulator.py
//     tables      :48-66      step (substep) :257-337     primitive_collision_func :198-226
//     norm_grad   :182-196    robot_step     :163-180     step_bwd_loss (what is differentiated) :235-248
// Forward arithmetic is in the reference's operation order (compile with -ffp-contract=off); the
// adjoint is a hand-derived reverse sweep of exactly that forward, validated against
// (a) torch.autograd on the line-by-line twin (oracle/twin/cloth_twin.py) and (b) f64 finite differences.
//
// PARITY STATUS: cloth x/v are "parity unpinned" by reference data (the shipped cloth demos were recorded by
// an older cloth step -- SURVEY.md F3); what the demos DO pin (primitive kinematics, state layout) is tested.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle {

template <class T>
struct ClothTables {
  int N = 0, P = 0;
  std::vector<int> nbr;  // [P*8] neighbour particle index, -1 if masked out / zero rest length (:62,:276)
  std::vector<T> L0;     // [P*8] rest length, clipped at 1e-12 (:63)
  T cell = 0;
};

template <class T>
struct ClothParams {
  T gravity_dt;   // T(gravity*dt)                      (:259)
  T gravity;      // T(gravity)                         (:278)
  T dt;           // T(dt)
  T damp;         // exp(-damping*dt) evaluated by the caller in T  (:309)
  T max_v;        // (:327)
  T small_num;    // (:281,:285)
  int substeps;   // 50 (:176)
  T n_mask;       // cloth_mask.sum() (:192)
};

template <class T>
inline T clipf(T x, T lo, T hi) { return std::min(hi, std::max(lo, x)); }

// gradient factor of jnp.clip = minimum(hi, maximum(lo, x)); ties split 0.5 (lax.max/min JVP)
template <class T>
inline T clip_grad(T x, T lo, T hi) {
  T m = std::max(lo, x);
  T f1 = (x == m) ? ((lo == m) ? T(0.5) : T(1)) : T(0);
  T a = std::min(hi, m);
  T f2 = (m == a) ? ((hi == a) ? T(0.5) : T(1)) : T(0);
  return f1 * f2;
}

template <class T>
inline T nan_to_num(T x) {
  if (std::isnan(x)) return T(0);
  if (std::isinf(x)) return x > 0 ? std::numeric_limits<T>::max() : std::numeric_limits<T>::lowest();
  return x;
}

inline void build_links(int N, const uint8_t* mask, std::vector<int>& pid, std::vector<int>& gi, std::vector<int>& gj) {
  pid.assign(N * N, -1);
  gi.clear(); gj.clear();
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (mask[i * N + j]) { pid[i * N + j] = (int)gi.size(); gi.push_back(i); gj.push_back(j); }
}

template <class T>
ClothTables<T> make_tables(int N, const uint8_t* mask) {
  static const int links[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
  ClothTables<T> t;
  t.N = N;
  std::vector<int> pid, gi, gj;
  build_links(N, mask, pid, gi, gj);
  t.P = (int)gi.size();
  t.cell = T(1.0 / N);
  t.nbr.assign(t.P * 8, -1);
  t.L0.assign(t.P * 8, T(1e-12));
  for (int p = 0; p < t.P; ++p)
    for (int l = 0; l < 8; ++l) {
      int ji = std::min(N - 1, std::max(0, gi[p] + links[l][0]));
      int jj = std::min(N - 1, std::max(0, gj[p] + links[l][1]));
      int di = ji - gi[p], dj = jj - gj[p];
      // jnp.linalg.norm on an int array -> float32; times python-float cell_size (weak type) -> float32
      float nrm = std::sqrt((float)(di * di + dj * dj));
      float ol = (float)(1.0 / N) * nrm;
      T L = (T)ol;
      if (sizeof(T) == 8) L = T(1.0 / N) * std::sqrt(T(di * di + dj * dj));
      bool nz = (ol != 0.0f);
      t.L0[p * 8 + l] = std::max(L, T(1e-12));
      t.nbr[p * 8 + l] = (nz && mask[ji * N + jj]) ? pid[ji * N + jj] : -1;
    }
  return t;
}

// Per-env forward substep. x,v: [P*3] AoS in/out. prim: [2][4], act: [2][4] (dxyz per substep, suction).
// grasp0/grasp1 (optional, [P]): masks of the gripped particles (Q3: discrete event, tested as a set).
template <class T>
void cloth_substep_fwd(const ClothTables<T>& tb, const ClothParams<T>& pr, T k, T mu, const T* x, const T* v,
                       const T* prim, const T* act, T* xo, T* vo, T* primo, uint8_t* grasp0, uint8_t* grasp1) {
  const int P = tb.P;
  const T eps = pr.small_num;
  for (int i = 0; i < P; ++i) {
    const T xi[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
    T v1[3] = {v[i * 3], v[i * 3 + 1] - pr.gravity_dt, v[i * 3 + 2]};  // :259
    T F[3] = {0, 0, 0};
    for (int l = 0; l < 8; ++l) {  // :262-277
      int j = tb.nbr[i * 8 + l];
      if (j < 0) continue;
      T r[3] = {x[j * 3] - xi[0], x[j * 3 + 1] - xi[1], x[j * 3 + 2] - xi[2]};
      T s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      T len = std::sqrt(clipf(s, T(1e-12), std::numeric_limits<T>::infinity()));
      T L0 = tb.L0[i * 8 + l];
      for (int a = 0; a < 3; ++a) F[a] += k * r[a] / len * (len - L0) / L0;  // :267-268
    }
    F[1] += -pr.gravity;  // :278
    bool fm = xi[1] <= eps;  // :281
    T cF = clipf(F[1], -std::numeric_limits<T>::infinity(), T(0));
    T muF = mu * cF * T(-1);  // :282
    T xV = v1[0], yV = v1[2];
    T sV = std::sqrt(xV * xV + yV * yV + eps);  // :285
    T dm = (fm && sV > eps) ? T(1) : T(0);      // :288
    T Ax = F[0] - dm * muF * xV / sV;           // :289
    T Az = F[2] - dm * muF * yV / sV;           // :290
    bool st = fm && (sV <= eps);                // :293
    T sF = std::sqrt(Ax * Ax + Az * Az + eps);  // :296
    T zm = (st && muF > sF) ? T(1) : T(0);      // :298
    T Bx = T(0) + (T(1) - zm) * Ax, Bz = T(0) + (T(1) - zm) * Az;  // :299-300
    T nz = (st && muF <= sF) ? T(1) : T(0);     // :302
    T R = T(1) - muF / sF;                      // :304
    T Cx = (R * Ax) * nz + Bx * (T(1) - nz);    // :305
    T Cz = (R * Az) * nz + Bz * (T(1) - nz);    // :306
    T Ff[3] = {Cx, F[1], Cz};
    T vv[3], xx[3] = {xi[0], xi[1], xi[2]};
    for (int a = 0; a < 3; ++a) vv[a] = (v1[a] + Ff[a] * pr.dt) * pr.damp;  // :308-309
    for (int g = 0; g < 2; ++g) {  // :313-314 (:198-226)
      const T* ps = prim + g * 4;
      const T* ac = act + g * 4;
      T d[3] = {xx[0] - ps[0], xx[1] - ps[1], xx[2] - ps[2]};
      T dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      bool m = dist <= ps[3];
      if (g == 0 && grasp0) grasp0[i] = m;
      if (g == 1 && grasp1) grasp1[i] = m;
      if (m) {
        T suction = ac[3];
        for (int a = 0; a < 3; ++a) { vv[a] = suction * vv[a]; xx[a] = xx[a] + ac[a] * (T(1) - suction); }
      }
    }
    for (int a = 0; a < 3; ++a) {  // :326-329
      T xc = clipf(xx[a], T(0), T(1));
      T vc = clipf(vv[a], -pr.max_v, pr.max_v);
      xo[i * 3 + a] = xc + pr.dt * vc;
      vo[i * 3 + a] = vc;
    }
  }
  for (int g = 0; g < 2; ++g) {  // :322-323
    for (int a = 0; a < 4; ++a) {
      T add = (a < 3) ? act[g * 4 + a] : T(0);
      primo[g * 4 + a] = clipf(prim[g * 4 + a] + add, T(0), T(1));
    }
  }
}

// Forward substep, operation order "v2": the same formulas re-associated into far fewer IEEE operations
//   spring      f = r * (k/L0 - k * (1/|r|))     instead of  k*r/|r|*(|r|-L0)/L0   (6 divisions + sqrt -> 1 + sqrt);
//               F = (sum over the straight links 0-3) + (sum over the diagonal links 4-7): the HIP kernel carries a
//               straight and a diagonal link in the two halves of a packed-f32 register
//   friction    t = dm*muF * (1/sV) ; A = F - t*V  instead of  F - dm*muF*V/sV   (1/sV does not wait for F)
//   the static-friction block (:293-306) is dropped: sV = sqrt(.+small_num) > small_num always (small_num < 1)
// Only +,-,*,/,sqrt, no FMA: a CPU and a GPU build of this order agree bit for bit (the default HIP forward,
// csrc/cloth.hip::substep_fwd_v2).  Against the reference order it differs by f32 round-off per substep
// (tests/test_oracle_cloth.py::test_order_v2_matches_reference_order_short_horizon).
template <class T>
void cloth_substep_fwd_v2(const ClothTables<T>& tb, const ClothParams<T>& pr, T k, T mu, const T* x, const T* v,
                          const T* prim, const T* act, T* xo, T* vo, T* primo, uint8_t* grasp0, uint8_t* grasp1) {
  const int P = tb.P;
  const T eps = pr.small_num;
  for (int i = 0; i < P; ++i) {
    const T xi[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
    T v1[3] = {v[i * 3], v[i * 3 + 1] - pr.gravity_dt, v[i * 3 + 2]};
    T Fs[3] = {0, 0, 0}, Fd[3] = {0, 0, 0};   // straight links 0-3 and diagonal links 4-7, each summed in link order
    for (int l = 0; l < 8; ++l) {
      int j = tb.nbr[i * 8 + l];
      if (j < 0) continue;
      T r[3] = {x[j * 3] - xi[0], x[j * 3 + 1] - xi[1], x[j * 3 + 2] - xi[2]};
      T s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      T len = std::sqrt(std::max(s, T(1e-12)));
      T inv = T(1) / len;
      T coef = k / tb.L0[i * 8 + l] - k * inv;
      T* Fa = (l < 4) ? Fs : Fd;
      for (int a = 0; a < 3; ++a) Fa[a] += coef * r[a];
    }
    T F[3] = {Fs[0] + Fd[0], Fs[1] + Fd[1], Fs[2] + Fd[2]};
    F[1] += -pr.gravity;
    bool fm = xi[1] <= eps;
    T cF = std::min(F[1], T(0));
    T muF = mu * cF * T(-1);
    T xV = v1[0], yV = v1[2];
    T isV = T(1) / std::sqrt(xV * xV + yV * yV + eps);
    T tf = fm ? muF * isV : T(0);
    T Ff[3] = {F[0] - tf * xV, F[1], F[2] - tf * yV};
    T vv[3], xx[3] = {xi[0], xi[1], xi[2]};
    for (int a = 0; a < 3; ++a) vv[a] = (v1[a] + Ff[a] * pr.dt) * pr.damp;
    for (int g = 0; g < 2; ++g) {
      const T* ps = prim + g * 4;
      const T* ac = act + g * 4;
      T d[3] = {xx[0] - ps[0], xx[1] - ps[1], xx[2] - ps[2]};
      T dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      bool m = dist <= ps[3];
      if (g == 0 && grasp0) grasp0[i] = m;
      if (g == 1 && grasp1) grasp1[i] = m;
      if (m) {
        T suction = ac[3];
        for (int a = 0; a < 3; ++a) { vv[a] = suction * vv[a]; xx[a] = xx[a] + ac[a] * (T(1) - suction); }
      }
    }
    for (int a = 0; a < 3; ++a) {
      T xc = clipf(xx[a], T(0), T(1));
      T vc = clipf(vv[a], -pr.max_v, pr.max_v);
      xo[i * 3 + a] = xc + pr.dt * vc;
      vo[i * 3 + a] = vc;
    }
  }
  for (int g = 0; g < 2; ++g)
    for (int a = 0; a < 4; ++a) {
      T add = (a < 3) ? act[g * 4 + a] : T(0);
      primo[g * 4 + a] = clipf(prim[g * 4 + a] + add, T(0), T(1));
    }
}

template <class T>
inline void norm_grad_bwd(T* g, int n, T n_mask) {  // :189-194
  T s = 0;
  for (int i = 0; i < n; ++i) s += g[i] * g[i];
  T nrm = std::sqrt(s);
  for (int i = 0; i < n; ++i) g[i] = nan_to_num(g[i] / nrm) / n_mask;
}

// Adjoint of one substep (what jax.grad(step_bwd_loss) returns, :235-254).
// In: state at substep input (x,v,prim,act,k,mu), cotangents of the outputs gx,gv [P*3], gprim [2][4].
// Out (overwritten): gx_in, gv_in, gprim_in; accumulated (+=): gact [2][4], gk, gmu.
template <class T>
void cloth_substep_bwd(const ClothTables<T>& tb, const ClothParams<T>& pr, bool normalize, T k, T mu, const T* x,
                       const T* v, const T* prim, const T* act, T* gx, T* gv, T* gprim, T* gact, T* gk, T* gmu,
                       std::vector<T>& scratch) {
  const int P = tb.P;
  const T eps = pr.small_num;
  const T inf = std::numeric_limits<T>::infinity();
  // ---- recompute the forward intermediates we need -------------------------------------------
  // scratch layout: v3[P*3] (after damping), x1[P*3] v4[P*3] (after gripper0), x2,v5 (after gripper1), gF[P*3]
  scratch.resize((size_t)P * 3 * 6 + P * 2);
  T* v3 = scratch.data();
  T* x1 = v3 + P * 3; T* v4 = x1 + P * 3; T* x2 = v4 + P * 3; T* v5 = x2 + P * 3; T* gF = v5 + P * 3;
  T* m0 = gF + P * 3; T* m1 = m0 + P;
  // forward pieces per particle (same arithmetic as cloth_substep_fwd)
  struct Fr { T F1, muF, cF, xV, yV, sV, dm, Ax, Az, sF, zm, nz, R; };
  std::vector<Fr> fr(P);
  for (int i = 0; i < P; ++i) {
    const T xi[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
    T v1[3] = {v[i * 3], v[i * 3 + 1] - pr.gravity_dt, v[i * 3 + 2]};
    T F[3] = {0, 0, 0};
    for (int l = 0; l < 8; ++l) {
      int j = tb.nbr[i * 8 + l];
      if (j < 0) continue;
      T r[3] = {x[j * 3] - xi[0], x[j * 3 + 1] - xi[1], x[j * 3 + 2] - xi[2]};
      T s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      T len = std::sqrt(clipf(s, T(1e-12), inf));
      T L0 = tb.L0[i * 8 + l];
      for (int a = 0; a < 3; ++a) F[a] += k * r[a] / len * (len - L0) / L0;
    }
    F[1] += -pr.gravity;
    Fr& f = fr[i];
    bool fm = xi[1] <= eps;
    f.F1 = F[1];
    f.cF = clipf(F[1], -inf, T(0));
    f.muF = mu * f.cF * T(-1);
    f.xV = v1[0]; f.yV = v1[2];
    f.sV = std::sqrt(f.xV * f.xV + f.yV * f.yV + eps);
    f.dm = (fm && f.sV > eps) ? T(1) : T(0);
    f.Ax = F[0] - f.dm * f.muF * f.xV / f.sV;
    f.Az = F[2] - f.dm * f.muF * f.yV / f.sV;
    bool st = fm && (f.sV <= eps);
    f.sF = std::sqrt(f.Ax * f.Ax + f.Az * f.Az + eps);
    f.zm = (st && f.muF > f.sF) ? T(1) : T(0);
    T Bx = T(0) + (T(1) - f.zm) * f.Ax, Bz = T(0) + (T(1) - f.zm) * f.Az;
    f.nz = (st && f.muF <= f.sF) ? T(1) : T(0);
    f.R = T(1) - f.muF / f.sF;
    T Cx = (f.R * f.Ax) * f.nz + Bx * (T(1) - f.nz);
    T Cz = (f.R * f.Az) * f.nz + Bz * (T(1) - f.nz);
    T Ff[3] = {Cx, F[1], Cz};
    T vv[3], xx[3] = {xi[0], xi[1], xi[2]};
    for (int a = 0; a < 3; ++a) { vv[a] = (v1[a] + Ff[a] * pr.dt) * pr.damp; v3[i * 3 + a] = vv[a]; }
    for (int g = 0; g < 2; ++g) {
      const T* ps = prim + g * 4;
      const T* ac = act + g * 4;
      T d[3] = {xx[0] - ps[0], xx[1] - ps[1], xx[2] - ps[2]};
      T dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      bool m = dist <= ps[3];
      (g == 0 ? m0 : m1)[i] = m ? T(1) : T(0);
      if (m) {
        T suction = ac[3];
        for (int a = 0; a < 3; ++a) { vv[a] = suction * vv[a]; xx[a] = xx[a] + ac[a] * (T(1) - suction); }
      }
      T* xs = (g == 0) ? x1 : x2;
      T* vs = (g == 0) ? v4 : v5;
      for (int a = 0; a < 3; ++a) { xs[i * 3 + a] = xx[a]; vs[i * 3 + a] = vv[a]; }
    }
  }
  // ---- reverse sweep ------------------------------------------------------------------------
  if (normalize) {  // :331-334
    norm_grad_bwd(gx, P * 3, pr.n_mask);
    norm_grad_bwd(gv, P * 3, pr.n_mask);
    norm_grad_bwd(gprim, 4, pr.n_mask);
    norm_grad_bwd(gprim + 4, 4, pr.n_mask);
  }
  // x_out = clip(x2) + dt*clip(v5)   (:326-329)
  for (int i = 0; i < P * 3; ++i) {
    T gxc = gx[i];
    T gvc = gv[i] + pr.dt * gx[i];
    gx[i] = gxc * clip_grad(x2[i], T(0), T(1));
    gv[i] = gvc * clip_grad(v5[i], -pr.max_v, pr.max_v);
  }
  // primitives (:322-323)
  for (int g = 0; g < 2; ++g)
    for (int a = 0; a < 4; ++a) {
      T add = (a < 3) ? act[g * 4 + a] : T(0);
      T t = gprim[g * 4 + a] * clip_grad(prim[g * 4 + a] + add, T(0), T(1));
      gprim[g * 4 + a] = t;
      if (a < 3) gact[g * 4 + a] += t;
    }
  // grippers in reverse order (:313-314, :198-226)
  for (int g = 1; g >= 0; --g) {
    if (normalize) { norm_grad_bwd(gx, P * 3, pr.n_mask); norm_grad_bwd(gv, P * 3, pr.n_mask); }  // :223-224
    const T* ac = act + g * 4;
    const T* msk = (g == 0) ? m0 : m1;
    const T* vin = (g == 0) ? v3 : v4;  // velocity entering this gripper
    T suction = ac[3];
    for (int i = 0; i < P; ++i) {
      if (msk[i] == T(0)) continue;
      for (int a = 0; a < 3; ++a) {
        T gvo = gv[i * 3 + a], gxo = gx[i * 3 + a];
        gact[g * 4 + 3] += vin[i * 3 + a] * gvo - gxo * ac[a];
        gact[g * 4 + a] += gxo * (T(1) - suction);
        gv[i * 3 + a] = suction * gvo;
      }
    }
  }
  // v3 = (v1 + F*dt)*damp  (:308-309) ; friction block (:281-306)
  for (int i = 0; i < P; ++i) {
    const Fr& f = fr[i];
    T gv2[3], gFf[3];
    for (int a = 0; a < 3; ++a) { gv2[a] = gv[i * 3 + a] * pr.damp; gFf[a] = gv2[a] * pr.dt; }
    T gCx = gFf[0], gCz = gFf[2];
    T gBx = gCx * (T(1) - f.nz), gBz = gCz * (T(1) - f.nz);
    T gR = gCx * f.nz * f.Ax + gCz * f.nz * f.Az;
    T gAx = gCx * f.nz * f.R + gBx * (T(1) - f.zm);
    T gAz = gCz * f.nz * f.R + gBz * (T(1) - f.zm);
    T gmuF = -gR / f.sF;
    T gsF = gR * f.muF / (f.sF * f.sF);
    gAx += gsF * f.Ax / f.sF;
    gAz += gsF * f.Az / f.sF;
    gmuF += -(gAx * f.dm * f.xV / f.sV + gAz * f.dm * f.yV / f.sV);
    T gxV = -gAx * f.dm * f.muF / f.sV, gyV = -gAz * f.dm * f.muF / f.sV;
    T gsV = (gAx * f.dm * f.muF * f.xV + gAz * f.dm * f.muF * f.yV) / (f.sV * f.sV);
    gxV += gsV * f.xV / f.sV;
    gyV += gsV * f.yV / f.sV;
    *gmu += -gmuF * f.cF;
    T gcF = -gmuF * mu;
    T gFy = gFf[1] + gcF * clip_grad(f.F1, -inf, T(0));
    gF[i * 3 + 0] = gAx; gF[i * 3 + 1] = gFy; gF[i * 3 + 2] = gAz;
    gv[i * 3 + 0] = gv2[0] + gxV;  // v1 = v - [0, g dt, 0]
    gv[i * 3 + 1] = gv2[1];
    gv[i * 3 + 2] = gv2[2] + gyV;
  }
  // spring forces (:262-277), scatter form (valid for any mask incl. lattice borders)
  for (int i = 0; i < P; ++i) {
    const T xi[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
    for (int l = 0; l < 8; ++l) {
      int j = tb.nbr[i * 8 + l];
      if (j < 0) continue;
      T r[3] = {x[j * 3] - xi[0], x[j * 3 + 1] - xi[1], x[j * 3 + 2] - xi[2]};
      T s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      T cf = clip_grad(s, T(1e-12), inf);
      T len = std::sqrt(clipf(s, T(1e-12), inf));
      T L0 = tb.L0[i * 8 + l];
      const T* g = gF + i * 3;
      T rg = r[0] * g[0] + r[1] * g[1] + r[2] * g[2];
      T c1 = (k / L0) * (T(1) - L0 / len);
      T c2 = (k / L0) * cf * L0 / (len * len * len) * rg;
      *gk += rg / len * (len - L0) / L0;
      for (int a = 0; a < 3; ++a) {
        T gr = c1 * g[a] + c2 * r[a];
        gx[j * 3 + a] += gr;
        gx[i * 3 + a] -= gr;
      }
    }
  }
}

}  // namespace oracle
