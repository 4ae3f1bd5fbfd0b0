// PlasticineLab-style MLS-MPM, float64: the adjoint of one env.step, the loss kernels and the parameter gradients
// (SURVEY.md 8f rank 4).  What it replaces (reference, Taichi reverse mode):
//   /root/reference/GenORM/policy/pbm/plb/engine/mpm_simulator.py
//       substep_grad :271-289 = recompute (clear_grid, compute_F_tmp, svd, p2g, grid_op), then g2p.grad, grid_op.grad,
//       forward_kinematics.grad, p2g.grad, svd_grad :101-105 with backward_svd :107-124 (clamp :152-161), compute_F_tmp.grad
//   /root/reference/GenORM/policy/pbm/plb/engine/losses/loss.py
//       compute_loss_kernel :190-214 and its grad :216-243: density :145-148, sdf :150-153, soft / hard contact :117-140,
//       sum_up :158-162; compute_grid_m_kernel mpm_simulator.py:456-466
//   /root/reference/PlasticineLab/sim2sim/plb/engine/mpm_simulator.py:27-29,485-498  E, nu, yield_stress as differentiable
//       scalars + get_parameter_grad; optimize_ground_friction (G variant :57-58) likewise.
// Structure: the forward call keeps every substep's particle state, the primitive trajectory and the spatial order in a
// caller-owned checkpoint (ud_plb_ckpt_bytes); the backward walks the substeps in reverse, per substep
//   plb_p2g (recompute, plb.hip) -> plb_grid_keep (v_out beside (m, mv)) -> plb_g2p_adj (scatter of the v_out cotangent,
//   x cotangent through the weights) -> plb_grid_adj (cell by cell: boundary / friction / sticky sphere / normalisation)
//   -> plb_p2g_adj (gather; stress, von Mises return mapping, SVD and F update in reverse; E / nu / yield-stress sums)
//   -> plb_adj_clear,
// over the touched cells only, like the forward, and with the forward's two lane mappings (four lanes per particle while
// the launch is too small to fill the chip, one beyond).  Parity: UNPINNED -- taichi is absent and the reference
// ships no gradient of this path; the checker is torch.autograd through oracle/twin/plb_twin_torch.py (tests/test_plb.py).
#include <cstdlib>

#include "plb_device.h"

namespace ud {

// ---- grid op, kept: v_out of every touched cell into buffer 1 (buffer 0 keeps (m, mv)) ---------------------------------
__global__ void __launch_bounds__(256) plb_grid_keep(PlbArgs a) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  const int cur = a.lb, prev = cur ^ 1;
  if (t < min(a.w.count[prev * a.B + b], a.cap)) {   // the cells of substep f + 1 (the other list and buffer): its p2g adjoint has gathered them
    const long lin = a.w.list[((long)prev * a.B + b) * a.cap + t];
    double* old = plb_buf(a, prev, b) + lin * 4;
    double* g = a.w.gacc + ((long)b * a.G + lin) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { old[k] = 0.0; g[k] = 0.0; }
  }
  // this substep's cells: from the grid checkpoint when the forward kept them all (the recompute launch of plb_p2g then left at
  // once for this env), else the list plb_p2g has just rebuilt
  const int n_ck = a.w.gck_cnt ? a.w.gck_cnt[b * c.S + a.f] : 0x7fffffff;
  const bool ck = n_ck <= c.gck;
  const int n = ck ? n_ck : min(a.w.count[cur * a.B + b], a.cap);
  if (ck && t == 0) a.w.count[cur * a.B + b] = n;      // the later launches of this substep read it; nobody in this one does
  if (t >= n) return;
  long lin;
  double* cell = nullptr;
  if (ck) {
    const long r = ((long)b * c.S + a.f) * c.gck + t;
    lin = a.w.gck_lin[r];
    a.w.list[((long)cur * a.B + b) * a.cap + t] = (int)lin;
    const double* rec = a.w.gck_val + r * 4;
    cell = plb_buf(a, cur, b) + lin * 4;
    cell[0] = rec[0]; cell[1] = rec[1]; cell[2] = rec[2]; cell[3] = rec[3];
  } else {
    lin = a.w.list[((long)cur * a.B + b) * a.cap + t];
    cell = plb_buf(a, cur, b) + lin * 4;
  }
  double vv[3];
  plb_grid_cell(c, lin, cell[0], cell + 1, a.w.pos + ((long)b * (c.S + 1) + a.f) * c.np * 3, a.softness + b * c.np, vv);
  double* out = plb_vout(a, b) + lin * 4;   // never cleared: read only at cells this launch has just written
  out[0] = vv[0]; out[1] = vv[1]; out[2] = vv[2];
}

// ---- g2p adjoint (:234-253 in reverse) -------------------------------------------------------------------------------
// inputs: cotangent of state f + 1 (gstate slot `gs_in`); outputs: v_out cotangents scattered into gacc, the x cotangent
// that flows through g2p (weights, dpos, the position clamp) into gxs, and gv1 (with the advection term) kept in gstate.
// The v_out cotangents are summed per block in an LDS table first (plb_p2g's staging: open addressing over PLB_H slots, f64 LDS
// atomics), then flushed with one global atomic per touched cell and component: 81 global f64 atomics per particle were this
// kernel (30-35 us of a 105 us reverse substep on Torus, profiles/r02j_kernel_stats_torus_grad_ngrid64.csv).
template <int LANES>   // lanes per particle, as in plb_g2p: the quad splits the 27 cells 7/7/7/6 and adds its partial sums with DPP
__global__ void __launch_bounds__(256) plb_g2p_adj(PlbArgs a, int gs_in) {
  __shared__ int s_key[PlbTab<LANES>::H];
  __shared__ double s_val[PlbTab<LANES>::H * 3];   // component-major [3][PLB_H]
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const PlbConst& c = a.c;
  for (int s = threadIdx.x; s < PlbTab<LANES>::H; s += blockDim.x) { s_key[s] = -1; s_val[s] = 0; s_val[PlbTab<LANES>::H + s] = 0; s_val[2 * PlbTab<LANES>::H + s] = 0; }
  if (gid == 0) a.w.count[(a.lb ^ 1) * a.B + b] = 0;   // the other list: plb_grid_keep has just retired it, p2g of substep f - 1 refills it
  __syncthreads();
  double* gacc = a.w.gacc + (long)b * a.G * 4;
  if (p < c.N) {   // whole quads together
  const double* hi_ = plb_hist(a, b, a.hs_in);
  const double* ho = plb_hist(a, b, a.hs_out);
  const double* g1 = a.w.gstate + ((long)b * 2 + gs_in) * 24 * c.Np;
  double x[3], gx1[3], gv1[3], gC1[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) { x[d] = hi_[d * c.Np + p]; gx1[d] = g1[d * c.Np + p]; gv1[d] = g1[(3 + d) * c.Np + p]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) gC1[d] = g1[(6 + d) * c.Np + p];
  double gxp[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {   // x1 = clamp(x + dt v1, 0, 1 - 3 dx): the cotangent passes where the clamp is inactive
    const double xn = x[d] + c.dt * ho[(3 + d) * c.Np + p];
    const double pass = (xn >= 0.0 && xn <= 1.0 - 3 * c.dx) ? 1.0 : 0.0;
    gxp[d] = pass * gx1[d];
    gv1[d] += c.dt * gxp[d];
  }
  int base[3];
  double fx[3], w[9], dw[9];
  plb_weights(c, x, base, fx, w, dw);
  const double* vout = plb_vout(a, b);
  double gfx[3] = {0, 0, 0};
  const double k4 = 4 * c.inv_dx;
  const int rot = (p * LANES) % 27;   // staggered stencil walk, as in plb_p2g: neighbours never on the same table slot at once
  constexpr int TRIPS = (27 + LANES - 1) / LANES, BATCH = LANES > 1 ? TRIPS : 1;   // four / eight lanes: the lane's seven / four cells requested together
#pragma unroll 1
  for (int t0 = 0; t0 < TRIPS; t0 += BATCH) {
  double g7[BATCH][3];
  long lin7[BATCH];
#pragma unroll
  for (int t = 0; t < BATCH; ++t) {
    const int it = min(qi + LANES * (t0 + t), 26);
    const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
    lin7[t] = plb_lin(c, ci, cj, ck);
    g7[t][0] = vout[lin7[t] * 4]; g7[t][1] = vout[lin7[t] * 4 + 1]; g7[t][2] = vout[lin7[t] * 4 + 2];
  }
#pragma unroll
  for (int t = 0; t < BATCH; ++t) {
    const int it = qi + LANES * (t0 + t);
    if (it >= 27) break;
    const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const double wi = dsel3(w, 0, i), wj = dsel3(w, 1, j), wk = dsel3(w, 2, k);
    const double weight = wi * wj * wk;
    const double dp[3] = {(double)i - fx[0], (double)j - fx[1], (double)k - fx[2]};
    const long lin = lin7[t];
    const double g[3] = {g7[t][0], g7[t][1], g7[t][2]};
    unsigned s = plb_hash_t<PlbTab<LANES>::LOGH>((int)lin);
    int slot = -1;
    for (int probe = 0; probe < 64; ++probe) {
      const int cur = s_key[s];
      if (cur == (int)lin) { slot = (int)s; break; }
      if (cur == -1) {
        const int old = atomicCAS(&s_key[s], -1, (int)lin);
        if (old == -1 || old == (int)lin) { slot = (int)s; break; }
      }
      s = (s + 1) & (PlbTab<LANES>::H - 1);
    }
    double gw = 0, gdp[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double cd = gC1[r * 3] * dp[0] + gC1[r * 3 + 1] * dp[1] + gC1[r * 3 + 2] * dp[2];
      const double gcell = weight * (gv1[r] + k4 * cd);
      if (slot >= 0) __hip_atomic_fetch_add(&s_val[r * PlbTab<LANES>::H + slot], gcell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else atomicAdd(gacc + lin * 4 + r, gcell);
      gw += g[r] * (gv1[r] + k4 * cd);
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) gdp[s2] += k4 * weight * gC1[r * 3 + s2] * g[r];
    }
    gfx[0] += gw * dsel3(dw, 0, i) * wj * wk - gdp[0];
    gfx[1] += gw * wi * dsel3(dw, 1, j) * wk - gdp[1];
    gfx[2] += gw * wi * wj * dsel3(dw, 2, k) - gdp[2];
  }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) gfx[d] = plb_quad_sum<LANES>(gfx[d]);
  if (qi == 0) {
    double* gxs = a.w.gxs + (long)b * 3 * c.Np;
#pragma unroll
    for (int d = 0; d < 3; ++d) gxs[d * c.Np + p] = gxp[d] + c.inv_dx * gfx[d];
  }
  }
  __syncthreads();
  {                  // flush: four lanes per cell (a cell is 32 contiguous bytes), the fourth component is the grid-op adjoint's
    const int r = threadIdx.x & 3;
#pragma unroll 4
    for (int sl = threadIdx.x >> 2; sl < PlbTab<LANES>::H; sl += 64) {
      const int key = s_key[sl];
      if (key < 0 || r == 3) continue;
      atomicAdd(gacc + (long)key * 4 + r, s_val[r * PlbTab<LANES>::H + sl]);
    }
  }
}

// ---- grid op adjoint (:200-232 in reverse), one touched cell per lane -------------------------------------------------
__global__ void __launch_bounds__(256) plb_grid_adj(PlbArgs a) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  // per-env cotangents (sticky-sphere positions, ground friction): summed over the wave, one atomic per wave and word -- one per cell
  // put every cell near a sphere on the same few words of its env (cf. lg_grid_adj_tile, mpm_large.hip)
  double qs[2][3] = {{0, 0, 0}, {0, 0, 0}}, gfric = 0;
  const bool inlist = t < min(a.w.count[a.lb * a.B + b], a.cap);
  const long lin = inlist ? a.w.list[((long)a.lb * a.B + b) * a.cap + t] : 0;
  const double* cell = plb_buf(a, a.lb, b) + lin * 4;
  double* ga = a.w.gacc + ((long)b * a.G + lin) * 4;
  if (inlist) {
    const double g[3] = {ga[0], ga[1], ga[2]};
    double gout[4];
    plb_grid_cell_adj(c, lin, cell[0], cell + 1, g, a.w.pos + ((long)b * (c.S + 1) + a.f) * c.np * 3, a.softness + b * c.np, gout, qs, gfric);
    ga[0] = gout[0]; ga[1] = gout[1]; ga[2] = gout[2]; ga[3] = gout[3];
  }
  double* gpos = a.w.gpos + ((long)b * (c.S + 1) + a.f) * c.np * 3;
  const bool lead = (threadIdx.x & 63) == 0;
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    if (pi >= c.np) break;
    if (!__any(qs[pi][0] != 0.0 || qs[pi][1] != 0.0 || qs[pi][2] != 0.0)) continue;   // wave-uniform
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double q = plb_wave_sum(qs[pi][k]);
      if (lead && q != 0.0) { atomicAdd(gpos + c.np * 3 + pi * 3 + k, q); atomicAdd(gpos + pi * 3 + k, -q); }
    }
  }
  if (__any(gfric != 0.0)) {
    const double q = plb_wave_sum(gfric);
    if (lead && q != 0.0) atomicAdd(a.w.gpar + b * 4 + 3, q);
  }
}



// ---- p2g adjoint + particle pre-pass adjoint (:91-99, :133-195 in reverse) --------------------------------------------
template <int LANES>   // lanes per particle: every lane of the quad repeats the pre-pass (it needs the affine matrix for its cells), the
                       // gather is split 7/7/7/6 and summed with DPP, lane 0 carries on with the particle adjoint
__global__ void __launch_bounds__(128) plb_p2g_adj(PlbArgs a, int gs_in) {
  __shared__ double s_red[3][2];
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const PlbConst& c = a.c;
  double accE = 0, accNu = 0, accYs = 0;
  if (p < c.N) {
    const double* hi_ = plb_hist(a, b, a.hs_in);
    const double* g1 = a.w.gstate + ((long)b * 2 + gs_in) * 24 * c.Np;
    double* g0 = a.w.gstate + ((long)b * 2 + (gs_in ^ 1)) * 24 * c.Np;
    double x[3], v[3], Cm[9], F[9];
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = hi_[d * c.Np + p]; v[d] = hi_[(3 + d) * c.Np + p]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { Cm[d] = hi_[(6 + d) * c.Np + p]; F[d] = hi_[(15 + d) * c.Np + p]; }
    const double E = a.E[b], nu = a.nu[b], ys = a.ys[b];
    int base[3];
    double fx[3], w[9], dw[9];
    plb_weights(c, x, base, fx, w, dw);
    // ---- forward pre-pass, as plb_p2g
    PlbPre q;
    if (a.w.svd) {                  // the forward's factors of this substep's F (same code, same inputs: the same bits)
      const double* o = a.w.svd + (((long)b * c.S + a.f) * 21) * c.Np + p;
#pragma unroll
      for (int i = 0; i < 9; ++i) { q.U[i] = o[i * c.Np]; q.Vh[i] = o[(12 + i) * c.Np]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) q.sig[i] = o[(9 + i) * c.Np];
    }
    plb_prepass(c, E, nu, ys, Cm, F, q, a.w.svd != nullptr);
    const double* aff = q.aff;
    // ---- gather the cell cotangents (p2g in reverse)
    const double* gacc = a.w.gacc + (long)b * a.G * 4;
    double gv[3] = {0, 0, 0}, gaff[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gfx[3] = {0, 0, 0};
#pragma unroll 1
    for (int cidx = qi; cidx < 27; cidx += LANES) {
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const double wi = dsel3(w, 0, i), wj = dsel3(w, 1, j), wk = dsel3(w, 2, k);
      const double weight = wi * wj * wk;
      const double dp[3] = {((double)i - fx[0]) * c.dx, ((double)j - fx[1]) * c.dx, ((double)k - fx[2]) * c.dx};
      const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
      const double* gc = gacc + plb_lin(c, ci, cj, ck) * 4;
      const double gmv[3] = {gc[0], gc[1], gc[2]}, gm = gc[3];
      double gw = c.p_mass * gm, gdp[3] = {0, 0, 0};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        gw += gmv[r] * (c.p_mass * v[r] + aff[r * 3] * dp[0] + aff[r * 3 + 1] * dp[1] + aff[r * 3 + 2] * dp[2]);
        gv[r] += weight * c.p_mass * gmv[r];
#pragma unroll
        for (int s = 0; s < 3; ++s) { gaff[r * 3 + s] += weight * gmv[r] * dp[s]; gdp[s] += weight * gmv[r] * aff[r * 3 + s]; }
      }
      gfx[0] += gw * dsel3(dw, 0, i) * wj * wk - c.dx * gdp[0];
      gfx[1] += gw * wi * dsel3(dw, 1, j) * wk - c.dx * gdp[1];
      gfx[2] += gw * wi * wj * dsel3(dw, 2, k) - c.dx * gdp[2];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { gv[d] = plb_quad_sum<LANES>(gv[d]); gfx[d] = plb_quad_sum<LANES>(gfx[d]); }
#pragma unroll
    for (int d = 0; d < 9; ++d) gaff[d] = plb_quad_sum<LANES>(gaff[d]);
    if (qi == 0) {
    const double* gxs = a.w.gxs + (long)b * 3 * c.Np;
#pragma unroll
    for (int d = 0; d < 3; ++d) { g0[d * c.Np + p] = gxs[d * c.Np + p] + c.inv_dx * gfx[d]; g0[(3 + d) * c.Np + p] = gv[d]; }
    double g1F[9], gC[9], gF0[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g1F[i] = g1[(15 + i) * c.Np + p];
    plb_particle_adjoint(c, E, nu, ys, q, F, gaff, g1F, gC, gF0, accE, accNu, accYs);
#pragma unroll
    for (int i = 0; i < 9; ++i) { g0[(6 + i) * c.Np + p] = gC[i]; g0[(15 + i) * c.Np + p] = gF0[i]; }
    }
  }
  // block sums -> one atomic per block and parameter
  double vals[3] = {accE, accNu, accYs};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double v = vals[q];
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    const int lane = threadIdx.x & 63;
    double tot = 0;
    for (int r = 0; r < 4; ++r) tot += __shfl(v, r * 16);
    if (lane == 0) s_red[q][threadIdx.x >> 6] = tot;
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(a.w.gpar + b * 4 + threadIdx.x, s_red[threadIdx.x][0] + s_red[threadIdx.x][1]);
}

// zero everything the substep touched: (m, mv), v_out, cotangents; reset the list
__global__ void __launch_bounds__(256) plb_adj_clear(PlbArgs a) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(a.w.count[a.lb * a.B + b], a.cap);
  if (t < n) {
    const long lin = a.w.list[((long)a.lb * a.B + b) * a.cap + t];
    double* c0 = plb_buf(a, a.lb, b) + lin * 4;
    double* g = a.w.gacc + ((long)b * a.G + lin) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c0[k] = 0.0; g[k] = 0.0; }
  }
}
__global__ void plb_adj_reset_counts(PlbArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.Bcall) { a.w.count[b] = 0; a.w.count[a.B + b] = 0; }
}

// cotangent of the step outputs -> gstate slot (in the spatial order of the checkpoint); zero the accumulators
__global__ void __launch_bounds__(256) plb_adj_pack(PlbArgs a, int slot, const double* gx, const double* gv, const double* gC, const double* gF, const double* gpp) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  if (blockIdx.x == 0) {
    for (int e = threadIdx.x; e < (c.S + 1) * c.np * 3; e += blockDim.x)
      a.w.gpos[(long)b * (c.S + 1) * c.np * 3 + e] = (gpp && e >= c.S * c.np * 3) ? gpp[(long)b * c.np * 3 + e - c.S * c.np * 3] : 0.0;
    if (threadIdx.x < 4) a.w.gpar[b * 4 + threadIdx.x] = 0.0;
  }
  if (p >= c.N) return;
  double* g = a.w.gstate + ((long)b * 2 + slot) * 24 * c.Np;
  const int up = a.w.perm[(long)b * c.Np + p];
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) { g[d * c.Np + p] = gx ? gx[o3 + d] : 0.0; g[(3 + d) * c.Np + p] = gv ? gv[o3 + d] : 0.0; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { g[(6 + d) * c.Np + p] = gC ? gC[o9 + d] : 0.0; g[(15 + d) * c.Np + p] = gF ? gF[o9 + d] : 0.0; }
}

__global__ void __launch_bounds__(256) plb_adj_unpack(PlbArgs a, int slot, double* gx, double* gv, double* gC, double* gF) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  if (p >= c.N) return;
  const double* g = a.w.gstate + ((long)b * 2 + slot) * 24 * c.Np;
  const int up = a.w.perm[(long)b * c.Np + p];
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) { gx[o3 + d] = g[d * c.Np + p]; gv[o3 + d] = g[(3 + d) * c.Np + p]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[o9 + d] = g[(6 + d) * c.Np + p]; gF[o9 + d] = g[(15 + d) * c.Np + p]; }
}

// forward_kinematics.grad + set_action in reverse: pos[s+1] = clamp(pos[s] + pv), pv = clip(action, -1, 1) / S for primitive 0
__global__ void plb_adj_epilogue(PlbArgs a, const double* action, double* g_prim_pos0, double* g_action, double* g_E, double* g_nu, double* g_ys, double* g_fric) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.Bcall) return;   // g_action, g_E, ... are the caller's [Bcall] arrays
  const PlbConst& c = a.c;
  const double* P = a.w.pos + (long)b * (c.S + 1) * c.np * 3;
  double* G = a.w.gpos + (long)b * (c.S + 1) * c.np * 3;
  for (int pi = 0; pi < c.np; ++pi)
    for (int d = 0; d < 3; ++d) {
      const double raw = (pi == 0) ? action[b * 3 + d] : 0.0;
      const double pv = (pi == 0) ? fmin(fmax(raw, -1.0), 1.0) / (double)c.S : 0.0;
      double gpv = 0.0;
      for (int s = c.S - 1; s >= 0; --s) {
        const double un = P[(s * c.np + pi) * 3 + d] + pv;
        const double pass = (un >= c.lo[d] && un <= c.hi[d]) ? 1.0 : 0.0;
        const double g = pass * G[((s + 1) * c.np + pi) * 3 + d];
        G[(s * c.np + pi) * 3 + d] += g;
        gpv += g;
      }
      if (g_prim_pos0) g_prim_pos0[((long)b * c.np + pi) * 3 + d] = G[pi * 3 + d];
      if (pi == 0 && g_action) g_action[b * 3 + d] = (raw >= -1.0 && raw <= 1.0) ? gpv / (double)c.S : 0.0;
    }
  if (g_E) g_E[b] = a.w.gpar[b * 4];
  if (g_nu) g_nu[b] = a.w.gpar[b * 4 + 1];
  if (g_ys) g_ys[b] = a.w.gpar[b * 4 + 2];
  if (g_fric) g_fric[b] = a.w.gpar[b * 4 + 3];
}

// ---- losses (engine/losses/loss.py) -----------------------------------------------------------------------------------
// grid mass of the particles (compute_grid_m_kernel): dense [B][G], zeroed by the caller
__global__ void __launch_bounds__(256) plb_loss_mass(PlbConst c, long G, const double* x, double* gm) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.N) return;
  const double xp[3] = {x[((long)b * c.N + p) * 3], x[((long)b * c.N + p) * 3 + 1], x[((long)b * c.N + p) * 3 + 2]};
  int base[3];
  double fx[3], w[9], dw[9];
  plb_weights(c, xp, base, fx, w, dw);
#pragma unroll 1
  for (int cidx = 0; cidx < 27; ++cidx) {
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
    atomicAdd(gm + (long)b * G + plb_lin(c, ci, cj, ck), dsel3(w, 0, i) * dsel3(w, 1, j) * dsel3(w, 2, k) * c.p_mass);
  }
}

__device__ __forceinline__ double plb_block_sum(double v, double* sh) {   // 256 threads; result in thread 0
  v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
  double tot = 0;
  for (int r = 0; r < 4; ++r) tot += __shfl(v, r * 16);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = tot;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// density :145-148 and sdf :150-153 terms: lred[b][0] += sum |gm - td|, lred[b][1] += sum tsdf * gm
__global__ void __launch_bounds__(256) plb_loss_grid(long G, const double* gm, const double* td, const double* tsdf, double* lred) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  double d = 0, s = 0;
  for (long I = (long)blockIdx.x * blockDim.x + threadIdx.x; I < G; I += (long)gridDim.x * blockDim.x) {
    const double m = gm[(long)b * G + I];
    d += fabs(m - td[I]);
    s += tsdf[I] * m;
  }
  d = plb_block_sum(d, sh);
  s = plb_block_sum(s, sh);
  if (threadIdx.x == 0) { atomicAdd(lred + b * 16, d); atomicAdd(lred + b * 16 + 1, s); }
}

// contact sums per primitive pi: soft :117-135 lred[b][4 + 2 pi] += sum w(d), lred[b][5 + 2 pi] += sum d w(d);
// hard :120-124: a minimum, taken through the ordered-integer view of the non-negative double (lred[b][8 + pi], preset to +inf)
__global__ void __launch_bounds__(256) plb_loss_contact(PlbConst c, const double* x, const double* prim_pos, int soft, double* lred) {
  __shared__ double sh[4];
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  for (int pi = 0; pi < c.np; ++pi) {
    double dij = 0, sw = 0;
    if (p < c.N) {
      const double* xp = x + ((long)b * c.N + p) * 3;
      const double* pp = prim_pos + ((long)b * c.np + pi) * 3;
      const double d0 = xp[0] - pp[0], d1 = xp[1] - pp[1], d2 = xp[2] - pp[2];
      dij = fmax(sqrt(d0 * d0 + d1 * d1 + d2 * d2 + 1e-14) - c.radius[pi], 0.0);
      sw = 1.0 / (1.0 + dij * dij * 10000.0);
    }
    if (soft) {
      const double s0 = plb_block_sum(p < c.N ? sw : 0.0, sh), s1 = plb_block_sum(p < c.N ? dij * sw : 0.0, sh);
      if (threadIdx.x == 0) { atomicAdd(lred + b * 16 + 4 + 2 * pi, s0); atomicAdd(lred + b * 16 + 5 + 2 * pi, s1); }
    } else if (p < c.N) {
      atomicMin((unsigned long long*)(lred + b * 16 + 8 + pi), (unsigned long long)__double_as_longlong(dij));
    }
  }
}

__global__ void plb_loss_init(int B, double* lred) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B * 16) lred[t] = ((t & 15) == 8 || (t & 15) == 9) ? INFINITY : 0.0;
}

__global__ void plb_loss_finish(PlbConst c, int B, int soft, const double* wts, const double* lred, double* loss, double* parts) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* r = lred + b * 16;
  double contact = 0;
  for (int pi = 0; pi < c.np; ++pi) {
    const double md = soft ? r[5 + 2 * pi] / r[4 + 2 * pi] : r[8 + pi];
    contact += md * md;
  }
  loss[b] = contact * wts[0] + r[0] * wts[1] + r[1] * wts[2];
  if (parts) { parts[b * 3] = contact; parts[b * 3 + 1] = r[0]; parts[b * 3 + 2] = r[1]; }
}

// loss adjoint: g_x and g_prim_pos.  Needs gm (grid mass) and lred (the forward's sums) of the same inputs.
__global__ void __launch_bounds__(256) plb_loss_bwd_kernel(PlbConst c, long G, int soft, const double* wts, const double* x, const double* prim_pos,
                                                          const double* td, const double* tsdf, const double* gm, const double* lred,
                                                          const double* g_loss, double* g_x, double* g_pp) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.N) return;
  const double gl = g_loss[b];
  const double* xp = x + ((long)b * c.N + p) * 3;
  int base[3];
  double fx[3], w[9], dw[9];
  plb_weights(c, xp, base, fx, w, dw);
  double gfx[3] = {0, 0, 0};
#pragma unroll 1
  for (int cidx = 0; cidx < 27; ++cidx) {
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
    const long I = plb_lin(c, ci, cj, ck);
    const double m = gm[(long)b * G + I], df = m - td[I];
    const double gmI = gl * (wts[1] * ((df > 0) ? 1.0 : ((df < 0) ? -1.0 : 0.0)) + wts[2] * tsdf[I]) * c.p_mass;
    const double wi = dsel3(w, 0, i), wj = dsel3(w, 1, j), wk = dsel3(w, 2, k);
    gfx[0] += gmI * dsel3(dw, 0, i) * wj * wk;
    gfx[1] += gmI * wi * dsel3(dw, 1, j) * wk;
    gfx[2] += gmI * wi * wj * dsel3(dw, 2, k);
  }
  double gx[3] = {c.inv_dx * gfx[0], c.inv_dx * gfx[1], c.inv_dx * gfx[2]};
  const double* r = lred + b * 16;
  for (int pi = 0; pi < c.np; ++pi) {
    const double* pp = prim_pos + ((long)b * c.np + pi) * 3;
    const double d0 = xp[0] - pp[0], d1 = xp[1] - pp[1], d2 = xp[2] - pp[2];
    const double len = sqrt(d0 * d0 + d1 * d1 + d2 * d2 + 1e-14);
    const double raw = len - c.radius[pi];
    const double dij = fmax(raw, 0.0);
    double gd = 0;   // d loss / d dij for this particle
    if (soft) {
      const double nrm = r[4 + 2 * pi], md = r[5 + 2 * pi] / nrm;
      const double sw = 1.0 / (1.0 + dij * dij * 10000.0);
      const double dsw = -20000.0 * dij * sw * sw;
      // md = sum(d w) / sum(w):  d md / d d_i = (w + d w') / nrm - md w' / nrm
      gd = gl * wts[0] * 2 * md * ((sw + dij * dsw) / nrm - md * dsw / nrm);
    } else {
      const double md = r[8 + pi];
      gd = (dij == md) ? gl * wts[0] * 2 * md : 0.0;   // the minimum passes its cotangent to the argmin (ties: every one of them)
    }
    if (raw > 0.0 && gd != 0.0) {
      const double q[3] = {gd * d0 / len, gd * d1 / len, gd * d2 / len};
#pragma unroll
      for (int k = 0; k < 3; ++k) { gx[k] += q[k]; if (g_pp) atomicAdd(g_pp + ((long)b * c.np + pi) * 3 + k, -q[k]); }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) g_x[((long)b * c.N + p) * 3 + k] = gx[k];
}

}  // namespace ud

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" {

int ud_plb_step_bwd(ud_plb* h, int B, const void* ckpt, const double* softness, const double* action, const double* E, const double* nu,
                    const double* yield_stress, const double* g_x, const double* g_v, const double* g_C, const double* g_F,
                    const double* g_prim_pos, double* g_x0, double* g_v0, double* g_C0, double* g_F0, double* g_prim_pos0,
                    double* g_action, double* g_E, double* g_nu, double* g_yield_stress, double* g_ground_friction, void* stream) {
  if (!h || !ckpt || !softness || !action || !E || !nu || !yield_stress || !g_x0 || !g_v0 || !g_C0 || !g_F0) {
    ud::set_error("ud_plb_step_bwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1) { ud::set_error("ud_plb_step_bwd: B=%d", B); return UD_ERR_INVALID; }
  if (B > h->B) { ud::set_error("ud_plb_step_bwd: B=%d exceeds the handle's max_envs=%d", B, h->B); return UD_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (h->cl.per > 0)
    return plb_cluster_step_bwd(h, B, ckpt, softness, action, E, nu, yield_stress, g_x, g_v, g_C, g_F, g_prim_pos, g_x0, g_v0, g_C0, g_F0,
                                g_prim_pos0, g_action, g_E, g_nu, g_yield_stress, g_ground_friction, st);
  ud::PlbArgs a{};
  a.c = h->c; a.w = h->w; a.B = h->B; a.Bcall = B; a.f = 0; a.epoch = 0; a.cap = h->cap; a.G = h->G;
  a.softness = softness; a.E = E; a.nu = nu; a.ys = yield_stress;
  plb_bind_ckpt(a, h->c, B, const_cast<void*>(ckpt));
  a.slots = h->c.S + 1; a.lb = 0; a.ls = 0; a.lprev = 1; a.lnext = 1; a.hs_out2 = 0; a.epoch2 = 0;
  a.ck_skip = h->c.gck > 0 ? 1 : 0;
  const bool never_recompute = h->c.gck >= h->cap;          // every substep of every env is in the grid checkpoint: no recompute launch at all
  const int S = h->c.S;
  const dim3 blk(256), gp((h->c.N + 255) / 256, B), gc((h->cap + 255) / 256, B), gpa((h->c.N + 127) / 128, B);
  const int lanes = h->lanes ? h->lanes : (((long)B * h->c.N <= 16000) ? 8 : (((long)B * h->c.N < 100000) ? 4 : 1));   // as the forward (plb.hip)
  const dim3 gq((lanes * h->c.N + 255) / 256, B), gqa((lanes * h->c.N + 127) / 128, B);
  hipLaunchKernelGGL(ud::plb_adj_reset_counts, dim3((B + 63) / 64), dim3(64), 0, st, a);
  hipLaunchKernelGGL(ud::plb_adj_pack, gp, blk, 0, st, a, S & 1, g_x, g_v, g_C, g_F, g_prim_pos);
  // Five launches per reverse substep.  List and (m, mv) buffer alternate with the substep like the forward's: plb_grid_keep retires
  // the cells of substep f + 1 (their buffer and cotangent cells back to zero) beside its own work, plb_g2p_adj resets that list's
  // count -- the separate clear and count-reset launches of every substep are gone; one clear after the loop for substep 0.
  for (int f = S - 1; f >= 0; --f) {
    a.f = f; a.epoch = h->epoch++; a.hs_in = f; a.hs_out = f + 1; a.lb = f & 1; a.ls = a.lb; a.lprev = a.lb ^ 1; a.lnext = a.lprev;
    if (!never_recompute) ud::plb_launch_p2g(a, lanes, lanes > 1 ? gq : gp, st);   // recompute (m, mv) (rewrites F[f + 1] with the same values); envs with a checkpointed substep leave at once
    hipLaunchKernelGGL(ud::plb_grid_keep, gc, blk, 0, st, a);
    if (lanes == 8) hipLaunchKernelGGL(ud::plb_g2p_adj<8>, gq, blk, 0, st, a, (f + 1) & 1);
    else if (lanes == 4) hipLaunchKernelGGL(ud::plb_g2p_adj<4>, gq, blk, 0, st, a, (f + 1) & 1);
    else hipLaunchKernelGGL(ud::plb_g2p_adj<1>, gp, blk, 0, st, a, (f + 1) & 1);
    hipLaunchKernelGGL(ud::plb_grid_adj, gc, blk, 0, st, a);
    if (lanes == 8) hipLaunchKernelGGL(ud::plb_p2g_adj<8>, gqa, dim3(128), 0, st, a, (f + 1) & 1);
    else if (lanes == 4) hipLaunchKernelGGL(ud::plb_p2g_adj<4>, gqa, dim3(128), 0, st, a, (f + 1) & 1);
    else hipLaunchKernelGGL(ud::plb_p2g_adj<1>, gpa, dim3(128), 0, st, a, (f + 1) & 1);
  }
  a.lb = 0; a.ls = 0; a.lprev = 1; a.lnext = 1;
  hipLaunchKernelGGL(ud::plb_adj_clear, gc, blk, 0, st, a);
  hipLaunchKernelGGL(ud::plb_adj_reset_counts, dim3((B + 63) / 64), dim3(64), 0, st, a);
  hipLaunchKernelGGL(ud::plb_adj_unpack, gp, blk, 0, st, a, 0, g_x0, g_v0, g_C0, g_F0);
  hipLaunchKernelGGL(ud::plb_adj_epilogue, dim3((B + 63) / 64), dim3(64), 0, st, a, action, g_prim_pos0, g_action, g_E, g_nu, g_yield_stress, g_ground_friction);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_step_bwd: %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

static int plb_loss_common(ud_plb* h, int B, const double* x, const double* prim_pos, const double* target_density, const double* target_sdf,
                           const double* weights, int soft_contact, hipStream_t st) {
  if (B > h->B) { ud::set_error("ud_plb_loss: B=%d exceeds the handle's max_envs=%d", B, h->B); return UD_ERR_INVALID; }
  const ud::PlbConst& c = h->c;
  hipError_t e = hipMemsetAsync(h->gm, 0, (size_t)B * h->G * 8, st);
  if (e != hipSuccess) { ud::set_error("ud_plb_loss: memset failed"); return UD_ERR_HIP; }
  const dim3 blk(256), gp((c.N + 255) / 256, B);
  hipLaunchKernelGGL(ud::plb_loss_init, dim3((B * 16 + 255) / 256), blk, 0, st, B, h->lred);
  hipLaunchKernelGGL(ud::plb_loss_mass, gp, blk, 0, st, c, h->G, x, h->gm);
  const int gb = (int)std::min<long>((h->G + 255) / 256, 1024);
  hipLaunchKernelGGL(ud::plb_loss_grid, dim3(gb, B), blk, 0, st, h->G, (const double*)h->gm, target_density, target_sdf, h->lred);
  if (c.np > 0) hipLaunchKernelGGL(ud::plb_loss_contact, gp, blk, 0, st, c, x, prim_pos, soft_contact, h->lred);
  return UD_OK;
}

int ud_plb_loss_fwd(ud_plb* h, int B, const double* x, const double* prim_pos, const double* target_density, const double* target_sdf,
                    const double* weights, int soft_contact, double* loss, double* parts, void* stream) {
  if (!h || !x || !prim_pos || !target_density || !target_sdf || !weights || !loss) { ud::set_error("ud_plb_loss_fwd: null argument"); return UD_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int rc = plb_loss_common(h, B, x, prim_pos, target_density, target_sdf, weights, soft_contact, st);
  if (rc) return rc;
  hipLaunchKernelGGL(ud::plb_loss_finish, dim3((B + 63) / 64), dim3(64), 0, st, h->c, B, soft_contact, weights, (const double*)h->lred, loss, parts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_loss_fwd: %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

int ud_plb_loss_bwd(ud_plb* h, int B, const double* x, const double* prim_pos, const double* target_density, const double* target_sdf,
                    const double* weights, int soft_contact, const double* g_loss, double* g_x, double* g_prim_pos, void* stream) {
  if (!h || !x || !prim_pos || !target_density || !target_sdf || !weights || !g_loss || !g_x) { ud::set_error("ud_plb_loss_bwd: null argument"); return UD_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int rc = plb_loss_common(h, B, x, prim_pos, target_density, target_sdf, weights, soft_contact, st);   // recompute the grid mass and the sums
  if (rc) return rc;
  if (g_prim_pos) { if (hipMemsetAsync(g_prim_pos, 0, (size_t)B * h->c.np * 3 * 8, st) != hipSuccess) { ud::set_error("ud_plb_loss_bwd: memset failed"); return UD_ERR_HIP; } }
  const dim3 blk(256), gp((h->c.N + 255) / 256, B);
  hipLaunchKernelGGL(ud::plb_loss_bwd_kernel, gp, blk, 0, st, h->c, h->G, soft_contact, weights, x, prim_pos, target_density, target_sdf,
                     (const double*)h->gm, (const double*)h->lred, g_loss, g_x, g_prim_pos);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_loss_bwd: %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

}  // extern "C"
