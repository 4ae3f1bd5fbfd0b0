// Per-particle and per-cell arithmetic of the PlasticineLab-style f64 MLS-MPM substep and of its adjoint, shared by the multi-kernel
// path (plb.hip, plb_adj.hip) and the persistent cluster kernels (plb_cluster.hip).  Reference: GenORM/policy/pbm/plb/engine/
// mpm_simulator.py compute_F_tmp :91-94, svd :96-99, compute_von_mises :133-150, p2g :166-195, grid_op :200-232, g2p :234-253,
// backward_svd :107-124 (clamp :152-161), substep_grad :271-289.
#pragma once
#include "plb_common.h"

namespace ud {

__device__ __forceinline__ void dm_mul_at(const double* A, const double* B, double* R) {   // A^T B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

__device__ __forceinline__ void plb_weights(const PlbConst& c, const double* x, int* base, double* fx, double* w, double* dw) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    base[d] = (int)(x[d] * c.inv_dx - 0.5);
    const double f = x[d] * c.inv_dx - (double)base[d];
    fx[d] = f;
    w[d] = 0.5 * (1.5 - f) * (1.5 - f); w[3 + d] = 0.75 - (f - 1) * (f - 1); w[6 + d] = 0.5 * (f - 0.5) * (f - 0.5);
    dw[d] = -(1.5 - f); dw[3 + d] = -2 * (f - 1); dw[6 + d] = f - 0.5;
  }
}
__device__ __forceinline__ void plb_weights_fwd(const PlbConst& c, const double* x, int* base, double* fx, double* w) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    base[d] = (int)(x[d] * c.inv_dx - 0.5);
    const double f = x[d] * c.inv_dx - (double)base[d];
    fx[d] = f;
    w[d] = 0.5 * (1.5 - f) * (1.5 - f); w[3 + d] = 0.75 - (f - 1) * (f - 1); w[6 + d] = 0.5 * (f - 0.5) * (f - 0.5);
  }
}
// linear index of stencil cell `cidx` (0..26) of a particle with base cell `base` (indices clamped into the grid, as the kernels always did)
__device__ __forceinline__ long plb_stencil_lin(const PlbConst& c, const int* base, int cidx) {
  const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
  const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
  return plb_lin(c, ci, cj, ck);
}

// ---- particle pre-pass: F_tmp = (I + dt C) F, its SVD, the von Mises return mapping in log-strain, the stress, the affine matrix ----------
struct PlbPre {
  double mu, lam, IC[9], Ft[9], U[9], Vh[9], sig[3], eps[3], eh[3], ehn, dg, ex[3], nF[9], J, A[9], St[9], aff[9];
  bool yields;
};
// have_svd: U, sig, Vh of q are already filled in (the adjoint reads the forward's factors from the checkpoint: same code, same inputs, same bits)
__device__ __forceinline__ void plb_prepass(const PlbConst& c, double E, double nu, double ys, const double* Cm, const double* F, PlbPre& q, bool have_svd) {
  q.mu = E / (2 * (1 + nu)); q.lam = E * nu / ((1 + nu) * (1 - 2 * nu));
#pragma unroll
  for (int i = 0; i < 9; ++i) q.IC[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c.dt * Cm[i];
  dm_mul(q.IC, F, q.Ft);
  if (!have_svd) dsvd3(q.Ft, q.U, q.sig, q.Vh);
  // Return mapping (:133-150): the deviatoric log-strain norm against yield_stress / (2 mu).  The three logarithms are needed only by a
  // particle that yields; whether it can is decided first from a bound that needs none: |log s| <= |s - 1| / min(s, 1) for s > 0, and the
  // deviator's norm is at most the norm of the strains themselves, so  sum_i (|s_i - 1| / min(s_i, 1))^2 + 1e-8 <= (ys / 2 mu)^2  means
  // "does not yield" whatever the logs are (Torus: ys / 2 mu = 0.48 -- a particle would have to be strained by a third).  Forward and
  // adjoint run this same code on the same (checkpointed) singular values: the same decision.
  const double ylim = ys / (2 * q.mu);
  double bound2 = 1e-8;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sc_ = fmax(q.sig[i], 0.05);
    const double b_ = fabs(sc_ - 1.0) * ud_rcp_nr(fmin(sc_, 1.0));
    bound2 += b_ * b_;
  }
  q.yields = false; q.ehn = 0; q.dg = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { q.eps[i] = 0; q.eh[i] = 0; }
  // ys <= 0 (an optimiser may push it there: g_ys is an exposed gradient): dg = ehn - ylim > 0 always, every particle yields -- the
  // squared comparison must not see that case
  if (!(ylim > 0 && bound2 <= ylim * ylim * (1.0 - 1e-12))) {          // (the margin covers the reciprocal's last bits)
    double sum = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { q.eps[i] = log(fmax(q.sig[i], 0.05)); sum += q.eps[i]; }
    double nn = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { q.eh[i] = q.eps[i] - sum / 3; nn += q.eh[i] * q.eh[i]; }
    q.ehn = sqrt(nn + 1e-8);
    q.dg = q.ehn - ylim;
    q.yields = q.dg > 0;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) q.nF[i] = q.Ft[i];
  q.ex[0] = 1; q.ex[1] = 1; q.ex[2] = 1;
  if (q.yields) {
    double US[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      q.ex[i] = exp(q.eps[i] - (q.dg / q.ehn) * q.eh[i]);
#pragma unroll
      for (int r = 0; r < 3; ++r) US[r * 3 + i] = q.U[r * 3 + i] * q.ex[i];
    }
    dm_mul(US, q.Vh, q.nF);
  }
  const double* nF = q.nF;
  q.J = nF[0] * (nF[4] * nF[8] - nF[5] * nF[7]) - nF[1] * (nF[3] * nF[8] - nF[5] * nF[6]) + nF[2] * (nF[3] * nF[7] - nF[4] * nF[6]);
  double R[9];
  dm_mul(q.U, q.Vh, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) q.A[i] = nF[i] - R[i];
  dm_mul_bt(q.A, nF, q.St);
  const double sc = -c.dt * c.p_vol * 4 * c.inv_dx * c.inv_dx;
#pragma unroll
  for (int i = 0; i < 9; ++i) q.aff[i] = sc * (2 * q.mu * q.St[i] + ((i % 4 == 0) ? q.lam * q.J * (q.J - 1) : 0.0)) + c.p_mass * Cm[i];
}

// ---- particle adjoint behind the p2g gather: cotangent of the affine matrix (gaff) and of F[f + 1] (g1F) -> cotangents of C and F of state f,
// and this particle's contributions to the cotangents of E, nu, yield stress.  (stress, det, U V^T, the return mapping incl. max(sig, 0.05),
// backward_svd with its clamp, F_tmp = (I + dt C) F, in reverse.)
__device__ __forceinline__ void plb_particle_adjoint(const PlbConst& c, double E, double nu, double ys, const PlbPre& q, const double* F, const double* gaff,
                                                     const double* g1F, double* gC_out, double* gF_out, double& accE, double& accNu, double& accYs) {
  const double sc = -c.dt * c.p_vol * 4 * c.inv_dx * c.inv_dx;
  const double mu = q.mu, lam = q.lam, J = q.J;
  const double* nF = q.nF;
  const double* U = q.U;
  const double* Vh = q.Vh;
  const double* sig = q.sig;
  double gC[9], Gs[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { gC[i] = c.p_mass * gaff[i]; Gs[i] = sc * gaff[i]; }
  const double trG = Gs[0] + Gs[4] + Gs[8];
  double gmu = 0, glam = J * (J - 1) * trG;
#pragma unroll
  for (int i = 0; i < 9; ++i) gmu += 2 * Gs[i] * q.St[i];
  const double gJ = lam * (2 * J - 1) * trG;
  double gM[9], gA[9], gnF[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) gM[i] = 2 * mu * Gs[i];
  dm_mul(gM, nF, gA);            // M = A nF^T: gA = gM nF
  dm_mul_at(gM, q.A, gnF);       // g(nF) = gM^T A
  const double cof[9] = {nF[4] * nF[8] - nF[5] * nF[7], nF[5] * nF[6] - nF[3] * nF[8], nF[3] * nF[7] - nF[4] * nF[6],
                         nF[2] * nF[7] - nF[1] * nF[8], nF[0] * nF[8] - nF[2] * nF[6], nF[1] * nF[6] - nF[0] * nF[7],
                         nF[1] * nF[5] - nF[2] * nF[4], nF[2] * nF[3] - nF[0] * nF[5], nF[0] * nF[4] - nF[1] * nF[3]};
#pragma unroll
  for (int i = 0; i < 9; ++i) gnF[i] += gA[i] + gJ * cof[i] + g1F[i];   // + the cotangent of F[f + 1]
  // R = U V^T (Vh = V^T):  gU = gR V = gR Vh^T,  gV = gR^T U  with gR = -gA
  double gU[9], gV[9], V[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i * 3 + j] = Vh[j * 3 + i];
  double ngA[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) ngA[i] = -gA[i];
  dm_mul(ngA, V, gU);
  dm_mul_at(ngA, U, gV);
  double gsig[3] = {0, 0, 0}, gFt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double gys = 0;
  if (q.yields) {   // nF = U diag(ex) V^T
    double T1[9], T2[9];
    dm_mul(gnF, V, T1);          // gnF V
    dm_mul_at(gnF, U, T2);       // gnF^T U
    double ge[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ge[i] = U[i] * T1[i] + U[3 + i] * T1[3 + i] + U[6 + i] * T1[6 + i];   // (U^T gnF V)_ii
#pragma unroll
      for (int r = 0; r < 3; ++r) { gU[r * 3 + i] += T1[r * 3 + i] * q.ex[i]; gV[r * 3 + i] += T2[r * 3 + i] * q.ex[i]; }
    }
    double gey[3], geps[3], geh[3];
    const double qq = q.dg / q.ehn;
    double gq = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { gey[i] = ge[i] * q.ex[i]; geps[i] = gey[i]; gq -= gey[i] * q.eh[i]; geh[i] = -qq * gey[i]; }
    // q = 1 - ys / (2 mu ehn)
    gys = -gq / (2 * mu * q.ehn);
    gmu += gq * ys / (2 * mu * mu * q.ehn);
    const double gehn = gq * ys / (2 * mu * q.ehn * q.ehn);
    double gsum = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { geh[i] += gehn * q.eh[i] / q.ehn; gsum += geh[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { geps[i] += geh[i] - gsum / 3; gsig[i] = (sig[i] > 0.05) ? geps[i] / sig[i] : 0.0; }
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) gFt[i] = gnF[i];
  }
  // ---- backward_svd (:107-124): gFt += U ((Fm * (U^T gU - gU^T U)) sig) V^T + U (sig ((Fm * (V^T gV - gV^T V)) V^T)) + U gsig V^T
  {
    double UtgU[9], VtgV[9];
    dm_mul_at(U, gU, UtgU);
    dm_mul_at(V, gV, VtgV);
    const double s2[3] = {sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2]};
    double Mm[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double val = 0.0;
        if (i != j) {
          double df = s2[j] - s2[i];
          df = (df >= 0) ? fmax(df, 1e-6) : fmin(df, -1e-6);   // clamp :152-161
          const double Fm = ud_rcp_nr(df);
          val = Fm * (UtgU[i * 3 + j] - UtgU[j * 3 + i]) * sig[j] + sig[i] * Fm * (VtgV[i * 3 + j] - VtgV[j * 3 + i]);
        } else {
          val = gsig[i];
        }
        Mm[i * 3 + j] = val;
      }
    double UM[9], add[9];
    dm_mul(U, Mm, UM);
    dm_mul(UM, Vh, add);
#pragma unroll
    for (int i = 0; i < 9; ++i) gFt[i] += add[i];
  }
  // ---- F_tmp = (I + dt C) F
  double gCF[9];
  dm_mul_bt(gFt, F, gCF);          // gFt F^T
  dm_mul_at(q.IC, gFt, gF_out);    // (I + dt C)^T gFt
#pragma unroll
  for (int i = 0; i < 9; ++i) gC_out[i] = gC[i] + c.dt * gCF[i];
  // ---- mu, lam -> E, nu
  const double a1 = 1 + nu, a2 = 1 - 2 * nu;
  accE = gmu / (2 * a1) + glam * nu / (a1 * a2);
  accNu = gmu * (-E / (2 * a1 * a1)) + glam * E * (1 + 2 * nu * nu) / (a1 * a1 * a2 * a2);
  accYs = gys;
}

// ---- grid op adjoint of one touched cell (:200-232 in reverse).  In: (m, mv) of the cell, g = cotangent of its v_out; out: ga = cotangents of
// (mv xyz, m); qs[pi] = cotangent this cell sends to the sticky sphere pi's velocity numerator (adds to gpos[f + 1][pi], subtracts from gpos[f][pi]);
// gfric = its contribution to the ground-friction cotangent.  A cell with m <= 1e-12 has no dependence on anything: ga = 0.
__device__ __forceinline__ void plb_grid_cell_adj(const PlbConst& c, long lin, double m, const double* mv, const double* gin, const double* P0, const double* soft,
                                                  double* ga, double qs[2][3], double& gfric) {
  ga[0] = 0; ga[1] = 0; ga[2] = 0; ga[3] = 0;
  if (!(m > 1e-12)) return;
  double g[3] = {gin[0], gin[1], gin[2]};
  const int n = c.n_grid;
  const int I[3] = {(int)(lin / ((long)n * n)), (int)((lin / n) % n), (int)(lin % n)};
  const double* P1 = P0 + c.np * 3;
  // forward, keeping the velocity that entered each boundary stage
  double vv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) vv[k] = (1.0 / m) * mv[k] + c.g30dt[k];
  bool stick[2] = {false, false};
  const double gp[3] = {I[0] * c.dx, I[1] * c.dx, I[2] * c.dx};
  for (int pi = 0; pi < c.np; ++pi) {
    const double d0 = gp[0] - P0[pi * 3], d1 = gp[1] - P0[pi * 3 + 1], d2 = gp[2] - P0[pi * 3 + 2];
    const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2 + 1e-14) - c.radius[pi];
    const double sf = soft[pi];
    const double infl = fmin(exp(-dist * sf), 1.0);
    if (((sf > 0 && infl > 0.1) || dist <= 0.001) && sf > 0) {
      stick[pi] = true;
#pragma unroll
      for (int k = 0; k < 3; ++k) vv[k] = (P1[pi * 3 + k] - P0[pi * 3 + k]) / c.dt;
    }
  }
  double vin[3][3];        // velocity entering stage d
  int kind[3];             // 0 nothing, 1 component zeroed, 2 friction, 3 all zeroed
  bool hiz[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int k = 0; k < 3; ++k) vin[d][k] = vv[k];
    kind[d] = 0;
    if (I[d] < 3 && vv[d] < 0) {
      if (d != 1 || c.fric == 0) { vv[d] = 0; kind[d] = 1; }
      else if (c.fric < 10) {
        const double lin_ = vv[1] + 1e-30;
        const double vit[3] = {vv[0] - I[0] * 1e-30, vv[1] - lin_ - I[1] * 1e-30, vv[2] - I[2] * 1e-30};
        const double lit = sqrt(vit[0] * vit[0] + vit[1] * vit[1] + vit[2] * vit[2] + 1e-8);
        const double s = fmax(1.0 + c.fric * lin_ / lit, 0.0);
        vv[0] = s * (vit[0] + I[0] * 1e-30); vv[2] = s * (vit[2] + I[2] * 1e-30); vv[1] = 0;
        kind[d] = 2;
      } else { vv[0] = 0; vv[1] = 0; vv[2] = 0; kind[d] = 3; }
    }
    hiz[d] = (I[d] > n - 3 && vv[d] > 0);
    if (hiz[d]) vv[d] = 0;
  }
  // reverse
#pragma unroll
  for (int d = 2; d >= 0; --d) {
    if (hiz[d]) g[d] = 0;
    if (kind[d] == 1) g[d] = 0;
    else if (kind[d] == 3) { g[0] = 0; g[1] = 0; g[2] = 0; }
    else if (kind[d] == 2) {
      const double* u = vin[d];
      const double lin_ = u[1] + 1e-30;
      const double vit[3] = {u[0] - I[0] * 1e-30, u[1] - lin_ - I[1] * 1e-30, u[2] - I[2] * 1e-30};
      const double lit = sqrt(vit[0] * vit[0] + vit[1] * vit[1] + vit[2] * vit[2] + 1e-8);
      const double arg = 1.0 + c.fric * lin_ / lit;
      const double s = fmax(arg, 0.0);
      const double gs = g[0] * (vit[0] + I[0] * 1e-30) + g[2] * (vit[2] + I[2] * 1e-30);
      double gvit[3] = {g[0] * s, 0.0, g[2] * s};
      double glin = 0, glit = 0;
      if (arg > 0) { glin = gs * c.fric / lit; glit = -gs * c.fric * lin_ / (lit * lit); gfric += gs * lin_ / lit; }
#pragma unroll
      for (int k = 0; k < 3; ++k) gvit[k] += glit * vit[k] / lit;
      glin -= gvit[1];                       // vit_y = v_y - lin - I_y 1e-30
      g[0] = gvit[0]; g[2] = gvit[2]; g[1] = gvit[1] + glin;
    }
  }
#pragma unroll
  for (int pi = 1; pi >= 0; --pi) {
    if (pi < c.np && stick[pi]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { qs[pi][k] = g[k] / c.dt; g[k] = 0; }
    }
  }
  const double im = 1.0 / m;
  ga[0] = g[0] * im; ga[1] = g[1] * im; ga[2] = g[2] * im;
  ga[3] = -(g[0] * mv[0] + g[1] * mv[1] + g[2] * mv[2]) * im * im;
}

}  // namespace ud
