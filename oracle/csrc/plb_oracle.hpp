// ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
//
// CPU restatement (float64, dense n_grid^3 grid like the source) of the Taichi "PlasticineLab" MLS-MPM forward
// substep used by GenORM's Torus task (BASELINE config 5):
//   /root/reference/GenORM/policy/pbm/plb/engine/mpm_simulator.py
//     compute_F_tmp :91-94, svd :96-99, compute_von_mises :133-150, p2g :166-195, grid_op :200-232, g2p :234-253,
//     substep :256-268, step :438-449
//   /root/reference/GenORM/policy/pbm/plb/engine/primitive/primitives.py  Sphere :17-53 (sticky collide)
//   /root/reference/GenORM/policy/pbm/plb/engine/primitive/primive_base.py forward_kinematics :118-121
// PARITY UNPINNED: taichi is absent and the reference ships no recorded trajectory for this path; ti.svd (third
// party) is replaced by a one-sided Jacobi SVD (A = U sig V^T, sig >= 0, valid for det F > 0).  Validated against
// the literal NumPy twin (oracle/twin/plb_twin.py) and analytic known answers only.
#pragma once
#include "mpm_oracle.hpp"

namespace oracle {

struct PlbParams {
  int N, n_grid, substeps, n_prim;
  double dt, dx, inv_dx, p_mass, p_vol;
  double gravity[3], ground_friction;
  double radius[2], lo[3], hi[3];
};

inline double det3(const M3<double>& A) {
  return A.a[0][0] * (A.a[1][1] * A.a[2][2] - A.a[1][2] * A.a[2][1]) - A.a[0][1] * (A.a[1][0] * A.a[2][2] - A.a[1][2] * A.a[2][0]) +
         A.a[0][2] * (A.a[1][0] * A.a[2][1] - A.a[1][1] * A.a[2][0]);
}

// one env, one substep; pos_f / pos_f1: primitive positions at f and f+1 [n_prim][3]
inline void plb_substep(const PlbParams& pr, const double* x, const double* v, const double* C, const double* F, const double* pos_f,
                        const double* pos_f1, const double* softness, double E, double nu, double yield_stress, double* xo,
                        double* vo, double* Co, double* Fo, std::vector<double>& gm, std::vector<double>& gv) {
  const int n = pr.n_grid, N = pr.N;
  const size_t G = (size_t)n * n * n;
  gm.assign(G, 0.0); gv.assign(G * 3, 0.0);
  const double mu = E / (2 * (1 + nu)), lam = E * nu / ((1 + nu) * (1 - 2 * nu));
  struct PP { int base[3]; double fx[3], w[3][3]; };
  std::vector<PP> pp(N);
  for (int p = 0; p < N; ++p) {
    PP& q = pp[p];
    for (int d = 0; d < 3; ++d) {
      q.base[d] = (int)(x[p * 3 + d] * pr.inv_dx - 0.5);
      double f = x[p * 3 + d] * pr.inv_dx - (double)q.base[d];
      q.fx[d] = f;
      q.w[0][d] = 0.5 * (1.5 - f) * (1.5 - f); q.w[1][d] = 0.75 - (f - 1) * (f - 1); q.w[2][d] = 0.5 * (f - 0.5) * (f - 0.5);
    }
    M3<double> Cm = load9(C + p * 9), Fm = load9(F + p * 9);
    M3<double> Ft = mul(add(m3_eye<double>(), scale(Cm, pr.dt)), Fm);
    M3<double> U, Vh;
    double sig[3];
    svd3(Ft, U, sig, Vh);
    double eps[3], sum = 0;
    for (int i = 0; i < 3; ++i) { eps[i] = std::log(std::max(sig[i], 0.05)); sum += eps[i]; }
    double eh[3], nn = 0;
    for (int i = 0; i < 3; ++i) { eh[i] = eps[i] - sum / 3; nn += eh[i] * eh[i]; }
    const double ehn = std::sqrt(nn + 1e-8);
    const double dg = ehn - yield_stress / (2 * mu);
    M3<double> nF = Ft;
    if (dg > 0) {
      M3<double> US = U;
      for (int i = 0; i < 3; ++i) { const double s = std::exp(eps[i] - (dg / ehn) * eh[i]); for (int r = 0; r < 3; ++r) US.a[r][i] *= s; }
      nF = mul(US, Vh);
    }
    store9(Fo + p * 9, nF);
    const double J = det3(nF);
    M3<double> r = mul(U, Vh);
    M3<double> st = scale(mul(sub(nF, r), tr(nF)), 2 * mu);
    for (int i = 0; i < 3; ++i) st.a[i][i] += lam * J * (J - 1);
    st = scale(st, -pr.dt * pr.p_vol * 4 * pr.inv_dx * pr.inv_dx);
    M3<double> aff = add(st, scale(Cm, pr.p_mass));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      const double weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      const double dpos[3] = {(i - q.fx[0]) * pr.dx, (j - q.fx[1]) * pr.dx, (k - q.fx[2]) * pr.dx};
      const size_t c = ((size_t)(q.base[0] + i) * n + (q.base[1] + j)) * n + (q.base[2] + k);
      gm[c] += weight * pr.p_mass;
      for (int a = 0; a < 3; ++a)
        gv[c * 3 + a] += weight * (pr.p_mass * v[p * 3 + a] + aff.a[a][0] * dpos[0] + aff.a[a][1] * dpos[1] + aff.a[a][2] * dpos[2]);
    }
  }
  // grid op
  for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) for (int d_ = 0; d_ < n; ++d_) {
    const size_t c = ((size_t)a * n + b) * n + d_;
    double vv[3] = {0, 0, 0};
    if (gm[c] > 1e-12) {
      const int I[3] = {a, b, d_};
      for (int k = 0; k < 3; ++k) vv[k] = gv[c * 3 + k] / gm[c] + pr.dt * pr.gravity[k] * 30;
      const double gp[3] = {a * pr.dx, b * pr.dx, d_ * pr.dx};
      for (int pi = 0; pi < pr.n_prim; ++pi) {
        const double* P0 = pos_f + pi * 3;
        const double dist = std::sqrt((gp[0] - P0[0]) * (gp[0] - P0[0]) + (gp[1] - P0[1]) * (gp[1] - P0[1]) + (gp[2] - P0[2]) * (gp[2] - P0[2]) + 1e-14) - pr.radius[pi];
        const double soft = softness[pi];
        const double infl = std::min(std::exp(-dist * soft), 1.0);
        if (((soft > 0 && infl > 0.1) || dist <= 0.001) && soft > 0)
          for (int k = 0; k < 3; ++k) vv[k] = (pos_f1[pi * 3 + k] - P0[k]) / pr.dt;
      }
      for (int d = 0; d < 3; ++d) {
        if (I[d] < 3 && vv[d] < 0) {
          if (d != 1 || pr.ground_friction == 0) vv[d] = 0;
          else if (pr.ground_friction < 10) {
            const double lin = vv[1] + 1e-30;
            double vit[3] = {vv[0] - I[0] * 1e-30, vv[1] - lin - I[1] * 1e-30, vv[2] - I[2] * 1e-30};
            const double lit = std::sqrt(vit[0] * vit[0] + vit[1] * vit[1] + vit[2] * vit[2] + 1e-8);
            const double s = std::max(1.0 + pr.ground_friction * lin / lit, 0.0);
            for (int k = 0; k < 3; ++k) vv[k] = s * (vit[k] + I[k] * 1e-30);
            vv[1] = 0;
          } else { vv[0] = vv[1] = vv[2] = 0; }
        }
        if (I[d] > n - 3 && vv[d] > 0) vv[d] = 0;
      }
    }
    for (int k = 0; k < 3; ++k) gv[c * 3 + k] = vv[k];
  }
  // g2p
  for (int p = 0; p < N; ++p) {
    const PP& q = pp[p];
    double nv[3] = {0, 0, 0};
    M3<double> nC = m3_zero<double>();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
      const double weight = q.w[i][0] * q.w[j][1] * q.w[k][2];
      const double dp[3] = {i - q.fx[0], j - q.fx[1], k - q.fx[2]};
      const size_t c = ((size_t)(q.base[0] + i) * n + (q.base[1] + j)) * n + (q.base[2] + k);
      for (int a = 0; a < 3; ++a) {
        nv[a] += weight * gv[c * 3 + a];
        for (int b = 0; b < 3; ++b) nC.a[a][b] += 4 * pr.inv_dx * weight * gv[c * 3 + a] * dp[b];
      }
    }
    for (int a = 0; a < 3; ++a) {
      vo[p * 3 + a] = nv[a];
      xo[p * 3 + a] = std::max(std::min(x[p * 3 + a] + pr.dt * nv[a], 1.0 - 3 * pr.dx), 0.0);
    }
    store9(Co + p * 9, nC);
  }
}

}  // namespace oracle
