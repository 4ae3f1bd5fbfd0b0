// PlasticineLab-style MLS-MPM forward step for gfx950, float64 (GenORM Torus task, BASELINE config 5).
//
// What it replaces (reference, /root/reference/GenORM/policy/pbm/plb/engine/):
//   mpm_simulator.py  step :438-449, substep :256-268 = clear_grid :69-79, compute_F_tmp :91-94, svd :96-99,
//                     p2g :166-195 (compute_von_mises :133-150), grid_op :200-232, g2p :234-253
//   primitive/primitives.py Sphere.sdf/collider_v/collide :17-53 ; primitive/primive_base.py forward_kinematics :118-121,
//                     set_velocity :185-192
// The reference drives 7 Taichi kernels per substep from Python over a dense n_grid^3 grid, one env per process.
// Here B envs run batched: per substep  clear (touched cells only) -> p2g -> grid op (touched cells only) -> g2p,
// the p2g scatter is summed per workgroup in an LDS cell table (ds_add_f64) and flushed with one
// global_atomic_add_f64 per distinct cell and component; the grid lives dense in HBM (n_grid^3 x 4 doubles per env)
// but only cells on the per-env active list are ever read, written or cleared.  No MFMA (scatter / stencil work).
// ti.svd (third party) is replaced by a one-sided Jacobi SVD in registers.  Parity: UNPINNED (see the header).
#include <cstdlib>
#include "plb_device.h"

#include <vector>

namespace ud {

// ---- kernels -------------------------------------------------------------------------------------------
// end of a step only: zero the cells of the last substep (list `lprev`, the other buffer), back to the all-zero grid invariant
__device__ __forceinline__ void plb_clear_body(const PlbArgs& a, int b, int t) {
  const int prev = a.lb ^ 1;
  if (t < min(a.w.count[a.lprev * a.B + b], a.cap)) {
    double* cell = plb_buf(a, prev, b) + (long)a.w.list[((long)a.lprev * a.B + b) * a.cap + t] * 4;
    cell[0] = 0.0; cell[1] = 0.0; cell[2] = 0.0; cell[3] = 0.0;
  }
}

// What a p2g pass targets: the (m, mv) buffer and list slot it fills, the substep it belongs to (SVD rows), the history slot that receives
// F of the NEXT state, the stamp epoch of its first-seen test.  plb_p2g fills it from the launch arguments; the fused forward kernel
// (plb_g2p_p2g) runs the pass of substep f + 1 behind the g2p of substep f.
struct P2gTarget { int buf, ls, f, hs_out, epoch; };

// compute_F_tmp + svd + von Mises + p2g (:91-99, :133-195) of one particle's lane, then the block's flush.  Every thread of the block
// calls it (the flush has barriers); `livep`: this lane has a particle, whose state is in x, v, Cm, F.  s_key / s_val: cleared, barrier passed.
template <int LANES>
__device__ __forceinline__ void plb_p2g_body(const PlbArgs& a, const P2gTarget& tg, int b, int p, int qi, bool livep, const double* x, const double* v,
                                             const double* Cm, const double* F, int* s_key, double* s_val) {
  const PlbConst& c = a.c;
  double* val = plb_buf(a, tg.buf, b);
  if (livep) {
    double* ho = plb_hist(a, b, tg.hs_out);
    int base[3];
    double fx[3], w[9];
    plb_weights_fwd(c, x, base, fx, w);
    PlbPre q;
    plb_prepass(c, a.E[b], a.nu[b], a.ys[b], Cm, F, q, false);
    const double* aff = q.aff;
    if (a.w.svd && qi == 0) {      // checkpointing forward: the factors of this substep's F, for the adjoint's pre-pass
      double* o = a.w.svd + (((long)b * c.S + tg.f) * 21) * c.Np + p;
#pragma unroll
      for (int i = 0; i < 9; ++i) { o[i * c.Np] = q.U[i]; o[(12 + i) * c.Np] = q.Vh[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) o[(9 + i) * c.Np] = q.sig[i];
    }
    if (qi == 0) {
#pragma unroll
      for (int d = 0; d < 9; ++d) ho[(15 + d) * c.Np + p] = q.nF[d];
    }
    const int rot = (p * LANES) % 27;   // staggered stencil walk (see lg_p2g, mpm_large.hip): neighbours never on the same slot at once
#pragma unroll 1
    for (int it = qi; it < 27; it += LANES) {
      const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const double weight = dsel3(w, 0, i) * dsel3(w, 1, j) * dsel3(w, 2, k);
      const double dp0 = ((double)i - fx[0]) * c.dx, dp1 = ((double)j - fx[1]) * c.dx, dp2 = ((double)k - fx[2]) * c.dx;
      const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
      const long lin = plb_lin(c, ci, cj, ck);
      double contrib[4];
      contrib[0] = weight * c.p_mass;
#pragma unroll
      for (int r = 0; r < 3; ++r) contrib[1 + r] = weight * (c.p_mass * v[r] + aff[r * 3] * dp0 + aff[r * 3 + 1] * dp1 + aff[r * 3 + 2] * dp2);
      // block-level staging
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
      if (slot >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) __hip_atomic_fetch_add(&s_val[r * PlbTab<LANES>::H + slot], contrib[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(val + lin * 4 + r, contrib[r]);
        plb_touch(a, b, lin, tg.ls, tg.epoch);
      }
    }
  }
  // flush; first-seen cells are appended to the env's active list with ONE counter atomic per block (same-address
  // global atomics on the counter, one per cell, dominated the substep in the f32 large path: mpm_large.hip)
  __shared__ int s_new, s_base;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  constexpr int PER = PlbTab<LANES>::H / 256;
  // value atomics: four lanes per cell, one per component -- a cell is 32 contiguous bytes (as in mpm_large.hip: one lane per cell
  // and component put the 64 lanes of an atomic on 64 different lines)
  {
    const int r = threadIdx.x & 3;
#pragma unroll 4
    for (int sl = threadIdx.x >> 2; sl < PlbTab<LANES>::H; sl += 64) {
      const int key = s_key[sl];
      if (key < 0) continue;
      atomicAdd(val + (long)key * 4 + r, s_val[r * PlbTab<LANES>::H + sl]);
    }
  }
  unsigned newmask = 0;
  int nnew = 0;
  {
    int old[PER], key[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {   // the returning exchanges of a lane go out together
      key[u] = s_key[threadIdx.x + u * 256];
      old[u] = tg.epoch;
      if (key[u] >= 0) old[u] = atomicExch(&a.w.stamp[(long)b * a.G + key[u]], tg.epoch);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (old[u] != tg.epoch) { newmask |= 1u << u; ++nnew; }
  }
  const int mine = nnew ? atomicAdd(&s_new, nnew) : 0;
  __syncthreads();
  const int cur = tg.ls;
  if (threadIdx.x == 0) s_base = s_new ? atomicAdd(&a.w.count[cur * a.B + b], s_new) : 0;
  __syncthreads();
  int e = s_base + mine;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (!(newmask & (1u << u))) continue;
    if (e < a.cap) a.w.list[((long)cur * a.B + b) * a.cap + e] = s_key[threadIdx.x + u * 256];
    ++e;
  }
}

template <int LANES>
__global__ void __launch_bounds__(256) plb_p2g(PlbArgs a) {
  __shared__ int s_key[PlbTab<LANES>::H];
  __shared__ double s_val[PlbTab<LANES>::H * 4];   // component-major [4][PLB_H]: slot-major rows of 32 B leave the lanes of a ds_add_f64 on 8 banks
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const PlbConst& c = a.c;
  if (a.ck_skip && a.w.gck_cnt[b * c.S + a.f] <= c.gck) return;   // adjoint: this env's substep is in the grid checkpoint (block-uniform)
  for (int s = threadIdx.x; s < PlbTab<LANES>::H; s += blockDim.x) { s_key[s] = -1; s_val[s] = 0; s_val[PlbTab<LANES>::H + s] = 0; s_val[2 * PlbTab<LANES>::H + s] = 0; s_val[3 * PlbTab<LANES>::H + s] = 0; }
  __syncthreads();
  double x[3] = {0, 0, 0}, v[3] = {0, 0, 0}, Cm[9], F[9];
  if (p < c.N) {
    const double* hi_ = plb_hist(a, b, a.hs_in);
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = hi_[d * c.Np + p]; v[d] = hi_[(3 + d) * c.Np + p]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { Cm[d] = hi_[(6 + d) * c.Np + p]; F[d] = hi_[(15 + d) * c.Np + p]; }
  }
  const P2gTarget tg{a.lb, a.ls, a.f, a.hs_out, a.epoch};
  plb_p2g_body<LANES>(a, tg, b, p, qi, p < c.N, x, v, Cm, F, s_key, s_val);
}

// grid_op (:200-232) over the touched cells
__global__ void __launch_bounds__(256) plb_grid(PlbArgs a) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  const int cur = a.lb, prev = cur ^ 1;
  if (t < min(a.w.count[a.lprev * a.B + b], a.cap)) {          // the previous substep's cells, in the other buffer: done with
    double* old = plb_buf(a, prev, b) + (long)a.w.list[((long)a.lprev * a.B + b) * a.cap + t] * 4;
    old[0] = 0.0; old[1] = 0.0; old[2] = 0.0; old[3] = 0.0;
  }
  if (t == 0 && a.lnext != a.lprev) a.w.count[a.lnext * a.B + b] = 0;   // fused forward: the third list, which the next launch's p2g pass fills
  const int n = min(a.w.count[a.ls * a.B + b], a.cap);
  if (a.w.gck_cnt && t == 0) a.w.gck_cnt[b * c.S + a.f] = n;
  if (t >= n) return;
  const long lin = a.w.list[((long)a.ls * a.B + b) * a.cap + t];
  double* cell = plb_buf(a, cur, b) + lin * 4;
  if (a.w.gck_cnt && t < c.gck) {      // grid checkpoint: (index, m, mv) before the grid op overwrites mv
    const long r = ((long)b * c.S + a.f) * c.gck + t;
    a.w.gck_lin[r] = (int)lin;
    double* o = a.w.gck_val + r * 4;
    o[0] = cell[0]; o[1] = cell[1]; o[2] = cell[2]; o[3] = cell[3];
  }
  double vv[3];
  plb_grid_cell(c, lin, cell[0], cell + 1, a.w.pos + ((long)b * (c.S + 1) + a.f) * c.np * 3, a.softness + b * c.np, vv);
  cell[1] = vv[0]; cell[2] = vv[1]; cell[3] = vv[2];
}

// g2p (:234-253)
template <int LANES>
__global__ void __launch_bounds__(256) plb_g2p(PlbArgs a) {
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const PlbConst& c = a.c;
  if (gid == 0) a.w.count[a.lprev * a.B + b] = 0;   // the other list: plb_grid has just retired it, p2g of the next substep refills it
  if (p >= c.N) return;   // whole quads leave together
  const double* hi_ = plb_hist(a, b, a.hs_in);
  double* ho = plb_hist(a, b, a.hs_out);
  double x[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) x[d] = hi_[d * c.Np + p];
  int base[3];
  double fx[3], w[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    base[d] = (int)(x[d] * c.inv_dx - 0.5);
    const double f = x[d] * c.inv_dx - (double)base[d];
    fx[d] = f;
    w[d] = 0.5 * (1.5 - f) * (1.5 - f); w[3 + d] = 0.75 - (f - 1) * (f - 1); w[6 + d] = 0.5 * (f - 0.5) * (f - 0.5);
  }
  const double* val = plb_buf(a, a.lb, b);
  double nv[3] = {0, 0, 0}, nC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int TRIPS = (27 + LANES - 1) / LANES, BATCH = LANES > 1 ? TRIPS : 1;   // four / eight lanes: the lane's seven / four cells requested together
#pragma unroll 1
  for (int t0 = 0; t0 < TRIPS; t0 += BATCH) {
    double g7[BATCH][3];
#pragma unroll
    for (int t = 0; t < BATCH; ++t) {
      const int cidx = min(qi + LANES * (t0 + t), 26);
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
      const double* cell = val + plb_lin(c, ci, cj, ck) * 4;
      g7[t][0] = cell[1]; g7[t][1] = cell[2]; g7[t][2] = cell[3];
    }
#pragma unroll
    for (int t = 0; t < BATCH; ++t) {
      const int cidx = qi + LANES * (t0 + t);
      if (cidx >= 27) break;
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const double weight = dsel3(w, 0, i) * dsel3(w, 1, j) * dsel3(w, 2, k);
      const double dp[3] = {(double)i - fx[0], (double)j - fx[1], (double)k - fx[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        nv[r] += weight * g7[t][r];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4 * c.inv_dx * weight * g7[t][r] * dp[s2];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) nv[d] = plb_quad_sum<LANES>(nv[d]);
#pragma unroll
  for (int d = 0; d < 9; ++d) nC[d] = plb_quad_sum<LANES>(nC[d]);
  if (qi != 0) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    ho[(3 + d) * c.Np + p] = nv[d];
    ho[d * c.Np + p] = fmax(fmin(x[d] + c.dt * nv[d], 1.0 - 3 * c.dx), 0.0);
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) ho[(6 + d) * c.Np + p] = nC[d];
}

// g2p of substep f, then -- the particle's new state in registers -- the p2g pass of substep f + 1 (forward only).  One launch instead
// of two per substep boundary and no state round trip.  What makes it legal: the (m, mv) buffers alternate already (the pass of f + 1
// fills the buffer plb_grid(f) has just retired), and the active lists take a third slot -- while this launch fills the list of f + 1,
// the list of f is still wanted (plb_grid(f + 1) retires it) and the one of f - 1 has just been retired: three in flight; plb_grid(f)
// resets the count of the slot this launch fills.
template <int LANES>
__global__ void __launch_bounds__(256) plb_g2p_p2g(PlbArgs a) {
  __shared__ int s_key[PlbTab<LANES>::H];
  __shared__ double s_val[PlbTab<LANES>::H * 4];
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const PlbConst& c = a.c;
  for (int s = threadIdx.x; s < PlbTab<LANES>::H; s += blockDim.x) { s_key[s] = -1; s_val[s] = 0; s_val[PlbTab<LANES>::H + s] = 0; s_val[2 * PlbTab<LANES>::H + s] = 0; s_val[3 * PlbTab<LANES>::H + s] = 0; }
  const bool livep = p < c.N;
  double x[3] = {0, 0, 0}, nv[3] = {0, 0, 0}, nC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[9];
  if (livep) {   // whole quads together
    const double* hi_ = plb_hist(a, b, a.hs_in);
    double* ho = plb_hist(a, b, a.hs_out);
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = hi_[d * c.Np + p];
#pragma unroll
    for (int d = 0; d < 9; ++d) F[d] = ho[(15 + d) * c.Np + p];   // F of state f + 1: the p2g pass of substep f wrote it
    int base[3];
    double fx[3], w[9];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      base[d] = (int)(x[d] * c.inv_dx - 0.5);
      const double f = x[d] * c.inv_dx - (double)base[d];
      fx[d] = f;
      w[d] = 0.5 * (1.5 - f) * (1.5 - f); w[3 + d] = 0.75 - (f - 1) * (f - 1); w[6 + d] = 0.5 * (f - 0.5) * (f - 0.5);
    }
    const double* val = plb_buf(a, a.lb, b);
    constexpr int TRIPS = (27 + LANES - 1) / LANES, BATCH = LANES > 1 ? TRIPS : 1;
#pragma unroll 1
    for (int t0 = 0; t0 < TRIPS; t0 += BATCH) {
      double g7[BATCH][3];
#pragma unroll
      for (int t = 0; t < BATCH; ++t) {
        const int cidx = min(qi + LANES * (t0 + t), 26);
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const int ci = min(max(base[0] + i, 0), c.n_grid - 1), cj = min(max(base[1] + j, 0), c.n_grid - 1), ck = min(max(base[2] + k, 0), c.n_grid - 1);
        const double* cell = val + plb_lin(c, ci, cj, ck) * 4;
        g7[t][0] = cell[1]; g7[t][1] = cell[2]; g7[t][2] = cell[3];
      }
#pragma unroll
      for (int t = 0; t < BATCH; ++t) {
        const int cidx = qi + LANES * (t0 + t);
        if (cidx >= 27) break;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const double weight = dsel3(w, 0, i) * dsel3(w, 1, j) * dsel3(w, 2, k);
        const double dp[3] = {(double)i - fx[0], (double)j - fx[1], (double)k - fx[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g7[t][r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4 * c.inv_dx * weight * g7[t][r] * dp[s2];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) nv[d] = plb_quad_sum<LANES>(nv[d]);     // every lane of the quad gets the sums: they all run the pre-pass below
#pragma unroll
    for (int d = 0; d < 9; ++d) nC[d] = plb_quad_sum<LANES>(nC[d]);
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = fmax(fmin(x[d] + c.dt * nv[d], 1.0 - 3 * c.dx), 0.0);
    if (qi == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { ho[(3 + d) * c.Np + p] = nv[d]; ho[d * c.Np + p] = x[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) ho[(6 + d) * c.Np + p] = nC[d];
    }
  }
  __syncthreads();   // the table clear
  const P2gTarget tg{a.lb ^ 1, a.lnext, a.f + 1, a.hs_out2, a.epoch2};
  plb_p2g_body<LANES>(a, tg, b, p, qi, livep, x, nv, nC, F, s_key, s_val);
}

// Spatial order (as lg_sort in mpm_large.hip): the Torus body is sampled with np.random (shape_maker.py:21,57), consecutive
// particles share no cells and the block-level staging of p2g does not aggregate.  One workgroup per env sorts (Morton key of
// the base cell, index) in LDS; pack / unpack go through the permutation, the caller keeps its own order.
__device__ __forceinline__ unsigned plb_morton10(unsigned v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
constexpr int PLB_SORT_MAX = 8192;
__global__ void __launch_bounds__(1024) plb_sort(PlbArgs a, const double* x, int npow2, int* perm_out) {
  extern __shared__ unsigned long long plb_sk[];
  const PlbConst& c = a.c;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < npow2; i += blockDim.x) {
    unsigned long long e = ~0ull;
    if (i < c.N) {
      unsigned key = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int cell = min(max((int)(x[((long)b * c.N + i) * 3 + d] * c.inv_dx - 0.5), 0), 1023);
        key |= plb_morton10((unsigned)cell) << d;
      }
      e = ((unsigned long long)key << 32) | (unsigned)i;
    }
    plb_sk[i] = e;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long ei = plb_sk[i], el = plb_sk[l];
          const bool up = (i & k) == 0;
          if ((ei > el) == up) { plb_sk[i] = el; plb_sk[l] = ei; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < c.N; i += blockDim.x) perm_out[(long)b * c.Np + i] = (int)(plb_sk[i] & 0xffffffffu);
}

// state into the history's slot 0 in the call's spatial order; block 0 of each env also runs the prologue (primitive positions of the whole
// step, both list counts).  order: the handle's current spatial order (arena), copied to this call's perm (the checkpoint's, for the adjoint).
__global__ void __launch_bounds__(256) plb_pack(PlbArgs a, const double* x, const double* v, const double* Cm, const double* F, const int* order,
                                                const double* prim_pos, const double* action) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const PlbConst& c = a.c;
  if (blockIdx.x == 0) {
    if (threadIdx.x < c.np * 3) {   // pos[s+1] = clamp(pos[s] + v), v = clip(action) * scale / substeps for primitive 0; one thread per coordinate
      const int pi = threadIdx.x / 3, d = threadIdx.x % 3;
      double* P = a.w.pos + (long)b * (c.S + 1) * c.np * 3;
      const double pv = (pi == 0) ? fmin(fmax(action[b * 3 + d], -1.0), 1.0) * 1.0 / (double)c.S : 0.0;
      double cur = prim_pos[(long)b * c.np * 3 + pi * 3 + d];
      P[pi * 3 + d] = cur;
      for (int s = 0; s < c.S; ++s) {
        cur = fmax(fmin(cur + pv, c.hi[d]), c.lo[d]);
        P[((s + 1) * c.np + pi) * 3 + d] = cur;
      }
    }
    if (threadIdx.x >= 64 && threadIdx.x < 67) a.w.count[(threadIdx.x - 64) * a.B + b] = 0;
  }
  if (p >= c.N) return;
  double* h = plb_hist(a, b, 0);
  const int up = order ? order[(long)b * c.Np + p] : p;
  a.w.perm[(long)b * c.Np + p] = up;
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) { h[d * c.Np + p] = x[o3 + d]; h[(3 + d) * c.Np + p] = v[o3 + d]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { h[(6 + d) * c.Np + p] = Cm[o9 + d]; h[(15 + d) * c.Np + p] = F[o9 + d]; }
}
// end of a forward call in one launch: blocks [0, nb_unpack) write the last state back in the caller's order (+ the primitive positions),
// the blocks behind them clear the last substep's cells
__global__ void __launch_bounds__(256) plb_unpack_clear(PlbArgs a, int slot, double* x, double* v, double* Cm, double* F, double* prim_o, int nb_unpack) {
  const int b = blockIdx.y;
  const PlbConst& c = a.c;
  if ((int)blockIdx.x >= nb_unpack) { plb_clear_body(a, b, ((int)blockIdx.x - nb_unpack) * blockDim.x + threadIdx.x); return; }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < c.np * 3)
    prim_o[(long)b * c.np * 3 + threadIdx.x] = a.w.pos[((long)b * (c.S + 1) + c.S) * c.np * 3 + threadIdx.x];   // copyframe(cur, 0)
  if (p >= c.N) return;
  const double* h = plb_hist(a, b, slot);
  const int up = a.w.perm[(long)b * c.Np + p];
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) { x[o3 + d] = h[d * c.Np + p]; v[o3 + d] = h[(3 + d) * c.Np + p]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[o9 + d] = h[(6 + d) * c.Np + p]; F[o9 + d] = h[(15 + d) * c.Np + p]; }
}

void plb_launch_p2g(const PlbArgs& a, int lanes, dim3 grid, hipStream_t st) {
  if (lanes == 8) hipLaunchKernelGGL(plb_p2g<8>, grid, dim3(256), 0, st, a);
  else if (lanes == 4) hipLaunchKernelGGL(plb_p2g<4>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(plb_p2g<1>, grid, dim3(256), 0, st, a);
}

}  // namespace ud

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int plb_reserve(ud_plb* h, int B, bool multi_kernel) {
  const ud::PlbConst& c = h->c;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const bool mk = multi_kernel;
  // multi-kernel path: dense double-buffered (m, mv) grid, stamps, active lists, ping-pong history, the adjoint's grids and cotangent state
  const size_t o_val = mk ? take((size_t)2 * B * h->G * 32) : 0, o_stamp = mk ? take((size_t)B * h->G * 4) : 0, o_list = mk ? take((size_t)3 * B * h->cap * 4) : 0;
  const size_t o_count = mk ? take((size_t)3 * B * 4) : 0, o_pos = mk ? take((size_t)B * (c.S + 1) * c.np * 3 * 8) : 0, o_hist = mk ? take((size_t)B * 2 * 24 * c.Np * 8) : 0;
  const size_t o_perm = mk ? take((size_t)B * c.Np * 4) : 0;
  const size_t o_gacc = mk ? take((size_t)B * h->G * 32) : 0, o_vout = mk ? take((size_t)B * h->G * 32) : 0, o_gstate = mk ? take((size_t)B * 2 * 24 * c.Np * 8) : 0;
  const size_t o_gxs = mk ? take((size_t)B * 3 * c.Np * 8) : 0, o_gpos = mk ? take((size_t)B * (c.S + 1) * c.np * 3 * 8 + 64) : 0, o_gpar = mk ? take((size_t)B * 4 * 8) : 0;
  // both paths: the spatial order, the loss kernels' grid mass and partial sums
  const size_t o_order = take((size_t)B * c.Np * 4), o_gm = take((size_t)B * h->G * 8), o_lred = take((size_t)B * 16 * 8);
  hipError_t e = hipMalloc(&h->arena, off);
  if (e != hipSuccess) { ud::set_error("ud_plb_create: hipMalloc(%zu MB) failed: %s", off >> 20, hipGetErrorString(e)); return UD_ERR_HIP; }
  e = hipMemset(h->arena, 0, off);
  if (e != hipSuccess) { ud::set_error("ud_plb_create: memset failed"); return UD_ERR_HIP; }
  char* base = (char*)h->arena;
  if (mk) {
    h->w.val = (double*)(base + o_val); h->w.stamp = (int*)(base + o_stamp); h->w.list = (int*)(base + o_list);
    h->w.count = (int*)(base + o_count); h->w.pos = (double*)(base + o_pos); h->w.hist = (double*)(base + o_hist);
    h->w.perm = (int*)(base + o_perm);
    h->w.gacc = (double*)(base + o_gacc); h->w.vout = (double*)(base + o_vout); h->w.gstate = (double*)(base + o_gstate);
    h->w.gxs = (double*)(base + o_gxs); h->w.gpos = (double*)(base + o_gpos); h->w.gpar = (double*)(base + o_gpar);
  }
  h->order = (int*)(base + o_order); h->gm = (double*)(base + o_gm); h->lred = (double*)(base + o_lred);
  h->B = B; h->epoch = 1; h->sort_B = 0;
  return UD_OK;
}

// caller-owned checkpoint of one step call: hist[B][S+1][24][Np] | pos[B][S+1][np][3] | perm[B][Np] (int) | grid checkpoint
PlbCkOff plb_ckpt_layout(const ud::PlbConst& c, int B) {
  PlbCkOff k{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  k.hist = take((size_t)B * (c.S + 1) * 24 * c.Np * 8);
  k.pos = take((size_t)B * (c.S + 1) * c.np * 3 * 8);
  k.perm = take((size_t)B * c.Np * 4);
  k.gck_cnt = take(c.gck ? (size_t)B * c.S * 4 : 0);
  k.gck_lin = take((size_t)B * c.S * c.gck * 4);
  k.gck_val = take((size_t)B * c.S * c.gck * 32);
  k.svd = take((size_t)B * c.S * 21 * c.Np * 8);
  k.total = off;
  return k;
}
void plb_bind_ckpt(ud::PlbArgs& a, const ud::PlbConst& c, int B, void* ckpt) {
  const PlbCkOff k = plb_ckpt_layout(c, B);
  char* base = (char*)ckpt;
  a.w.hist = (double*)(base + k.hist); a.w.pos = (double*)(base + k.pos); a.w.perm = (int*)(base + k.perm);
  a.w.gck_cnt = c.gck ? (int*)(base + k.gck_cnt) : nullptr;
  a.w.gck_lin = c.gck ? (int*)(base + k.gck_lin) : nullptr;
  a.w.gck_val = c.gck ? (double*)(base + k.gck_val) : nullptr;
  a.w.svd = (double*)(base + k.svd);
}

extern "C" {

int ud_plb_create(const ud_plb_conf* conf, ud_plb** out) {
  if (!conf || !out) { ud::set_error("ud_plb_create: null argument"); return UD_ERR_INVALID; }
  if (conf->n_particles < 1 || conf->n_grid < 8 || conf->n_grid > 512 || conf->substeps < 1 || conf->n_primitives < 0 || conf->n_primitives > 2) {
    ud::set_error("ud_plb_create: bad sizes"); return UD_ERR_INVALID;
  }
  if (conf->max_envs < 1) { ud::set_error("ud_plb_create: max_envs = %d (every arena is sized at create: give the largest B any call will pass)", conf->max_envs); return UD_ERR_INVALID; }
  if (conf->path < 0 || conf->path > 2 || !(conf->lanes == 0 || conf->lanes == 1 || conf->lanes == 4 || conf->lanes == 8)) {
    ud::set_error("ud_plb_create: path = %d (0 auto, 1 multi-kernel, 2 persistent), lanes = %d (0 auto, 1, 4, 8)", conf->path, conf->lanes); return UD_ERR_INVALID;
  }
  auto* h = new ud_plb();
  ud::PlbConst& c = h->c;
  c.N = conf->n_particles; c.Np = (c.N + 15) / 16 * 16; c.n_grid = conf->n_grid; c.S = conf->substeps; c.np = conf->n_primitives;
  c.dt = conf->dt; c.dx = 1.0 / conf->n_grid; c.inv_dx = (double)conf->n_grid;
  c.p_vol = (c.dx * 0.5) * (c.dx * 0.5); c.p_mass = c.p_vol * 1;
  for (int d = 0; d < 3; ++d) { c.g30dt[d] = conf->dt * conf->gravity[d] * 30; c.lo[d] = conf->lower_bound[d]; c.hi[d] = conf->upper_bound[d]; }
  c.fric = conf->ground_friction;
  c.radius[0] = conf->radius[0]; c.radius[1] = conf->radius[1];
  h->G = (long)c.n_grid * c.n_grid * c.n_grid;
  h->cap = (int)std::min<long>(h->G, (long)27 * c.N);
  c.gck = conf->grid_ckpt_cells > 0 ? (int)std::min<long>(h->cap, (long)conf->grid_ckpt_cells * c.N) : 0;
  h->lanes = conf->lanes;
  h->sort_every = conf->sort_every == 0 ? 8 : conf->sort_every;
  (void)hipFuncSetAttribute((const void*)ud::plb_sort, hipFuncAttributeMaxDynamicSharedMemorySize, ud::PLB_SORT_MAX * 8);
  // Which kernels this handle runs is decided here, once: the persistent launch per step call (plb_cluster.hip) where all parts of a
  // launch can be resident and the exchange grids fit, else the multi-kernel path.  path = 1 / 2 force one (2: an error if it cannot run).
  int per = conf->path == 1 ? 0 : plb_cluster_plan(h, conf->max_envs);
  if (conf->path == 2 && per < 1) {
    ud::set_error("ud_plb_create: path = 2 (persistent) does not fit this configuration (parts per env %d, substeps %d)", h->cl.W, c.S);
    delete h; return UD_ERR_UNSUPPORTED;
  }
  int rc = plb_reserve(h, conf->max_envs, per < 1);
  if (rc == UD_OK && per >= 1) rc = plb_cluster_reserve(h, per);
  if (rc != UD_OK) { ud_plb_destroy(h); return rc; }
  if (hipDeviceSynchronize() != hipSuccess) { ud::set_error("ud_plb_create: device synchronisation failed"); ud_plb_destroy(h); return UD_ERR_HIP; }
  *out = h;
  return UD_OK;
}

void ud_plb_destroy(ud_plb* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  if (h->cl.arena) (void)hipFree(h->cl.arena);
  delete h;
}

size_t ud_plb_ckpt_bytes(const ud_plb* h, int B) {
  if (!h || B < 1) return 0;
  return h->cl.per > 0 ? plb_cluster_ckpt_bytes(h, B) : plb_ckpt_layout(h->c, B).total;
}

int ud_plb_launch_plan(const ud_plb* h, int B) {
  if (!h || B < 1 || B > h->B) return -1;
  return h->cl.per > 0 ? 2 : 1;
}

int ud_plb_poll_timeouts(ud_plb* h, void* stream) {
  if (!h) { ud::set_error("ud_plb_poll_timeouts: null handle"); return UD_ERR_INVALID; }
  if (h->cl.per < 1) return 0;
  hipStream_t st = (hipStream_t)stream;
  int n = 0;
  if (hipMemcpyAsync(&n, h->cl.timeouts, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    ud::set_error("ud_plb_poll_timeouts: copy failed"); return UD_ERR_HIP;
  }
  if (n > 0) {
    // a part that gave up leaves its env's exchange grids, barrier words and accumulators dirty: back to the rest state before anything else runs
    (void)hipMemsetAsync(h->cl.arena, 0, h->cl.bytes, st);
    (void)hipStreamSynchronize(st);
    ud::set_error("ud_plb: %d workgroup(s) of the persistent kernels gave up waiting for a sibling (outputs of those envs are NaN); the handle's "
                  "exchange arena has been reset.  The persistent path needs every workgroup of a launch resident at once: other kernels "
                  "occupying the device for seconds can cause this", n);
  }
  return n;
}

int ud_plb_step_fwd(ud_plb* h, int B, const double* x, const double* v, const double* C, const double* F,
                    const double* prim_pos, const double* softness, const double* action, const double* E,
                    const double* nu, const double* yield_stress, double* x_out, double* v_out, double* C_out,
                    double* F_out, double* prim_pos_out, void* ckpt, void* stream) {
  if (!h || !x || !v || !C || !F || !prim_pos || !softness || !action || !E || !nu || !yield_stress || !x_out || !v_out || !C_out ||
      !F_out || !prim_pos_out) {
    ud::set_error("ud_plb_step_fwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1) { ud::set_error("ud_plb_step_fwd: B=%d", B); return UD_ERR_INVALID; }
  if (B > h->B) { ud::set_error("ud_plb_step_fwd: B=%d exceeds the handle's max_envs=%d (arenas are sized at create)", B, h->B); return UD_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  // Spatial order: any permutation is valid, only its locality ages (a particle moves a fraction of a cell per step) -- computed on the
  // first call, when more envs arrive than it covers, and every sort_every-th call (conf; default 8) after that; kept in the handle and
  // copied into every call's own perm (the checkpoint's).
  const bool sorted = h->sort_every > 0 && h->c.N <= ud::PLB_SORT_MAX;
  if (sorted && (B > h->sort_B || ++h->sort_age >= h->sort_every)) {
    int npow2 = 64;
    while (npow2 < h->c.N) npow2 <<= 1;
    ud::PlbArgs sa{};
    sa.c = h->c;
    hipLaunchKernelGGL(ud::plb_sort, dim3(B), dim3(1024), (size_t)npow2 * 8, st, sa, x, npow2, h->order);
    h->sort_B = std::max(h->sort_B, B); h->sort_age = 0;
  }
  if (h->cl.per > 0)
    return plb_cluster_step_fwd(h, B, x, v, C, F, prim_pos, softness, action, E, nu, yield_stress, x_out, v_out, C_out, F_out, prim_pos_out,
                                sorted ? (const int*)h->order : (const int*)nullptr, ckpt, st);
  ud::PlbArgs a{};
  a.c = h->c; a.w = h->w; a.B = h->B; a.Bcall = B; a.f = 0; a.epoch = 0; a.cap = h->cap; a.G = h->G;
  a.slots = 2; a.hs_in = 0; a.hs_out = 1; a.lb = 0; a.ls = 0; a.lprev = 1; a.lnext = 1; a.hs_out2 = 0; a.epoch2 = 0;
  a.ck_skip = 0; a.w.gck_cnt = nullptr; a.w.gck_lin = nullptr; a.w.gck_val = nullptr; a.w.svd = nullptr;
  if (ckpt) {   // keep every substep's particle state, the primitive trajectory, the spatial order (and the touched grid cells) for ud_plb_step_bwd
    plb_bind_ckpt(a, h->c, B, ckpt);
    a.slots = h->c.S + 1;
  }
  a.softness = softness; a.E = E; a.nu = nu; a.ys = yield_stress;
  const dim3 blk(256), gp((h->c.N + 255) / 256, B), gc((h->cap + 255) / 256, B);
  // lanes per particle in p2g / g2p: 8 while even four leave half of the SIMDs without a wave (B N <= 16 000), 4 while the launch does
  // not fill the chip, 1 beyond; ud_plb_conf.lanes forces one mapping (how the tests reach all three at their sizes)
  const int lanes = h->lanes ? h->lanes : (((long)B * h->c.N <= 16000) ? 8 : (((long)B * h->c.N < 100000) ? 4 : 1));
  const dim3 gq((lanes * h->c.N + 255) / 256, B);
  hipLaunchKernelGGL(ud::plb_pack, gp, blk, 0, st, a, x, v, C, F, sorted ? (const int*)h->order : (const int*)nullptr, prim_pos, action);
  // Per substep: plb_grid(f), then ONE particle launch: g2p(f) -> p2g(f + 1) (plb_g2p_p2g); p2g(0) opens the step, g2p(S - 1) closes it.
  const bool fused = true;
  const int S = h->c.S;
  const int ep0 = h->epoch; h->epoch += S + 1;
  auto set = [&](int f) {
    a.f = f; a.epoch = ep0 + f; a.epoch2 = ep0 + f + 1;
    a.hs_in = f % a.slots; a.hs_out = (f + 1) % a.slots; a.hs_out2 = (f + 2) % a.slots; a.lb = f & 1;
    if (fused) { a.ls = f % 3; a.lprev = (f + 2) % 3; a.lnext = (f + 1) % 3; }
    else { a.ls = a.lb; a.lprev = a.lb ^ 1; a.lnext = a.lprev; }
  };
  set(0);
#define UD_PLB_LAUNCH(K) do { if (lanes == 8) hipLaunchKernelGGL(ud::K<8>, gq, blk, 0, st, a); else if (lanes == 4) hipLaunchKernelGGL(ud::K<4>, gq, blk, 0, st, a); \
                               else hipLaunchKernelGGL(ud::K<1>, gp, blk, 0, st, a); } while (0)
  UD_PLB_LAUNCH(plb_p2g);
  for (int f = 0; f < S; ++f) {
    set(f);
    hipLaunchKernelGGL(ud::plb_grid, gc, blk, 0, st, a);
    if (fused && f + 1 < S) {
      UD_PLB_LAUNCH(plb_g2p_p2g);
    } else {
      UD_PLB_LAUNCH(plb_g2p);
      if (f + 1 < S) {
        set(f + 1);
        UD_PLB_LAUNCH(plb_p2g);
      }
    }
  }
  // back to the all-zero grid invariant: the cells of the last substep (its list is `lprev` of a substep S that never runs)
  set(S);
  hipLaunchKernelGGL(ud::plb_unpack_clear, dim3(gp.x + gc.x, B), blk, 0, st, a, h->c.S % a.slots, x_out, v_out, C_out, F_out, prim_pos_out, (int)gp.x);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_step_fwd: %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

}  // extern "C"
