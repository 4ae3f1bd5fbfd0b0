// Shared pieces of the PlasticineLab-style f64 MLS-MPM kernels: forward (plb.hip), adjoint and losses (plb_adj.hip).
#pragma once
#include "common.h"

namespace ud {


struct PlbConst {
  int N, Np, n_grid, S, np;
  int gck;        // grid checkpoint: cells kept per env and substep (0 = off; == cap: a substep can never overflow it)
  double dt, dx, inv_dx, p_mass, p_vol, g30dt[3], fric, radius[2], lo[3], hi[3];
};

struct PlbBuf {
  double* val;    // [2][B][G][4] (m, mv) -> after the grid op (m, v)
  int* stamp;     // [B][G]
  int* list;      // [3][B][cap]
  int* count;     // [3][B]
  double* pos;    // [B][S+1][np][3] primitive positions of this step (handle arena, or the caller's checkpoint)
  double* hist;   // [B][slots][24][Np] particle state per substep: slots = 2 (ping-pong) or S + 1 (checkpoint: all of them)
  int* perm;      // [B][Np] spatial order of this call: slot p of hist holds the caller's particle perm[p]
  // grid checkpoint (caller's checkpoint; ud_plb_conf.grid_ckpt_cells): the touched cells of every substep before the grid op
  int* gck_cnt;   // [B][S]        cells substep f touched (may exceed c.gck: then nothing usable was kept for it)
  int* gck_lin;   // [B][S][gck]   their linear indices, in the order of the forward's active list
  double* gck_val;// [B][S][gck][4] (m, mv)
  double* svd;    // [B][S][21][Np] U, S, Vh of every substep's F (caller's checkpoint): the adjoint's pre-pass reads them instead of iterating again
  // adjoint only (plb_adj.hip)
  double* gacc;   // [B][G][4] cotangent of the cell's v_out (xyz), then of (mv xyz, m)
  double* vout;   // [B][G][4] the cell's v_out of the substep being reversed (xyz; never cleared: read only where just written)
  double* gstate; // [B][2][24][Np] cotangent of the particle state, ping-pong over substeps
  double* gxs;    // [B][3][Np] the part of x's cotangent that g2p's adjoint produces, handed to p2g's adjoint
  double* gpos;   // [B][S+1][np][3] cotangent of the primitive positions
  double* gpar;   // [B][4] cotangents of E, nu, yield_stress, ground friction
};

struct PlbArgs {
  PlbConst c;
  PlbBuf w;
  int B, f, epoch, cap;     // B = envs the handle's arena was sized for (the arena stride), NOT the envs of this call
  int Bcall;                  // envs of this call (<= B): the bound of every per-env guard that touches caller-owned arrays
  int slots, hs_in, hs_out;   // hist slots per env; slot of this substep's input state / output state
  int lb;                     // active list and grid buffer of this substep (forward: f & 1, the other one is being retired)
  int ls, lprev;              // active-list slots: this substep's cells, the list being retired.  Two slots alternating with lb everywhere
                              // except the fused forward (plb_g2p_p2g), which fills substep f + 1's list (lnext) while f's is still read: three slots
  int lnext, hs_out2, epoch2; // fused forward only: list slot / history slot for F / stamp epoch of the p2g pass of substep f + 1
  int ck_skip;                // adjoint's recompute launch of plb_p2g: envs whose substep f is in the grid checkpoint leave at once
  long G;
  const double *softness, *E, *nu, *ys;
};

__device__ __forceinline__ double* plb_hist(const PlbArgs& a, int b, int slot) { return a.w.hist + ((long)b * a.slots + slot) * 24 * a.c.Np; }

// ---- double 3x3 helpers -----------------------------------------------------------------------------
__device__ __forceinline__ void dm_mul(const double* A, const double* B, double* R) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void dm_mul_bt(const double* A, const double* B, double* R) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}

// f64 reciprocal / square root / reciprocal square root from the hardware seeds (v_rcp_f64 / v_rsq_f64) and Newton / Goldschmidt steps on FMAs,
// for operands in the normal range (no scaling, no special-case fix-up: the compiler's IEEE sequences carry both, v_div_scale / v_div_fmas /
// v_div_fixup and two v_ldexp, and are 11-14 dependent instructions each).  Two quadratic steps from the seed (>= 14 good bits) reach the last
// one or two bits; the Jacobi rotations below need no more -- their angle only has to shrink the off-diagonal product, the factors are re-
// normalised at the end.
__device__ __forceinline__ double ud_rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double ud_sqrt_nr(double x) {      // x > 0
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}
__device__ __forceinline__ double ud_rsqrt_nr(double x) {     // x > 0
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);                                           // 0.5 / sqrt(x)
  return h + h;
}

// One-sided Jacobi rotation of columns p, q.  With al = |a_p|^2, be = |a_q|^2, ga = a_p . a_q the textbook angle is zeta = (be - al) / (2 ga),
// t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), c = 1 / sqrt(1 + t^2): three square roots and three divisions in one dependent chain (with the
// threshold's sqrt(al be)).  The same t without forming zeta: t = sign(d g) |g| / (|d| + sqrt(d^2 + g^2)) with d = be - al, g = 2 ga -- ONE square
// root, ONE reciprocal, ONE reciprocal square root; the thresholds compare squares.  The rotation was 60-70 % of the particle pre-pass, which was
// 5.7 of the persistent forward's 19 us per substep (tools/pcl_stamps.py).
#define UD_DJROT(p, q)                                                                       \
  {                                                                                          \
    const double al = a[p] * a[p] + a[3 + p] * a[3 + p] + a[6 + p] * a[6 + p];                \
    const double be = a[q] * a[q] + a[3 + q] * a[3 + q] + a[6 + q] * a[6 + q];                \
    const double ga = a[p] * a[q] + a[3 + p] * a[3 + q] + a[6 + p] * a[6 + q];                \
    const double ab_ = al * be, gg_ = ga * ga;                                                \
    const bool rot = !done && gg_ > 1e-34 * ab_;            /* |ga| > 1e-17 sqrt(al be) */    \
    big_rot |= gg_ > 9e-18 * ab_;                           /* |ga| > 3e-9 sqrt(al be) */     \
    const double g2_ = rot ? ga + ga : 1.0, d_ = be - al;                                     \
    const double h_ = ud_sqrt_nr(fma(d_, d_, g2_ * g2_));                                     \
    double t = fabs(g2_) * ud_rcp_nr(fabs(d_) + h_);                                          \
    t = ((d_ < 0.0) != (g2_ < 0.0)) ? -t : t;                                                 \
    double cs = ud_rsqrt_nr(fma(t, t, 1.0)), sn = cs * t;                                     \
    cs = rot ? cs : 1.0; sn = rot ? sn : 0.0;                                                 \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                           \
      double ap = a[i * 3 + p], aq = a[i * 3 + q];                                            \
      a[i * 3 + p] = cs * ap - sn * aq; a[i * 3 + q] = sn * ap + cs * aq;                     \
      double vp = vv[i * 3 + p], vq = vv[i * 3 + q];                                          \
      vv[i * 3 + p] = cs * vp - sn * vq; vv[i * 3 + q] = sn * vp + cs * vq;                     \
    }                                                                                         \
  }
#define UD_DCSWAP(p, q)                                                              \
  if (sv[p] < sv[q]) {                                                               \
    double ts = sv[p]; sv[p] = sv[q]; sv[q] = ts;                                    \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                  \
      double t1 = a[i * 3 + p]; a[i * 3 + p] = a[i * 3 + q]; a[i * 3 + q] = t1;      \
      double t2 = vv[i * 3 + p]; vv[i * 3 + p] = vv[i * 3 + q]; vv[i * 3 + q] = t2;  \
    }                                                                                \
  }

// A = U diag(S) Vh, S descending >= 0 (one-sided Jacobi, 6 sweeps reach f64 round-off for |F - I| = O(1))
__device__ __forceinline__ void dsvd3(const double* A, double* U, double* S, double* Vh) {
  double a[9], vv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = A[i];
  bool done = false;
#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
    // Cyclic Jacobi converges quadratically: a sweep whose three normalised off-diagonal products were all below 3e-9 leaves them
    // below 1e-17 -- under the rotation threshold, so every later sweep would be the identity.  Leave after such a sweep (per wave):
    // four sweeps instead of six on a typical Torus state, and the Jacobi iteration is most of plb_p2g's serial chain.
    bool big_rot = false;
    UD_DJROT(0, 1)
    UD_DJROT(0, 2)
    UD_DJROT(1, 2)
    done = done || !big_rot;      // per matrix, so that the result does not depend on what else is in the wave
    if (!__any(!done)) break;
  }
  double sv[3], isv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double n2 = a[j] * a[j] + a[3 + j] * a[3 + j] + a[6 + j] * a[6 + j];
    const bool pos = n2 > 1e-280;
    isv[j] = pos ? ud_rsqrt_nr(pos ? n2 : 1.0) : 0.0;
    sv[j] = pos ? n2 * isv[j] : 0.0;
  }
#define UD_DISWAP(p, q) { const bool sw_ = sv[p] < sv[q]; const double ti_ = isv[p]; isv[p] = sw_ ? isv[q] : isv[p]; isv[q] = sw_ ? ti_ : isv[q]; }
  UD_DISWAP(0, 1) UD_DCSWAP(0, 1)
  UD_DISWAP(1, 2) UD_DCSWAP(1, 2)
  UD_DISWAP(0, 1) UD_DCSWAP(0, 1)
#undef UD_DISWAP
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S[j] = sv[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) { U[i * 3 + j] = a[i * 3 + j] * isv[j]; Vh[j * 3 + i] = vv[i * 3 + j]; }
  }
}

__device__ __forceinline__ double dsel3(const double* w, int d, int i) { return (i == 0) ? w[d] : ((i == 1) ? w[3 + d] : w[6 + d]); }

__device__ __forceinline__ long plb_lin(const PlbConst& c, int i, int j, int k) { return ((long)i * c.n_grid + j) * c.n_grid + k; }

__device__ __forceinline__ void plb_touch(const PlbArgs& a, int b, long lin, int cur, int epoch) {   // cur: list slot, epoch: stamp of the pass
  const int old = atomicExch(&a.w.stamp[(long)b * a.G + lin], epoch);
  if (old != epoch) {
    const int e = atomicAdd(&a.w.count[cur * a.B + b], 1);
    if (e < a.cap) a.w.list[((long)cur * a.B + b) * a.cap + e] = (int)lin;
  }
}

// The grid values are double-buffered, buffer k belongs to active list k (substep f uses k = f & 1): the cells of substep f - 1
// can then be zeroed while substep f runs -- plb_grid(f) does it next to its own work -- instead of in a launch of their own
// between g2p(f - 1) and p2g(f).  One launch less per substep (4 -> 3) on a path whose kernels sit near the launch floor.
__device__ __forceinline__ double* plb_buf(const PlbArgs& a, int k, int b) { return a.w.val + (((long)k * a.B + b) * a.G) * 4; }
__device__ __forceinline__ double* plb_vout(const PlbArgs& a, int b) { return a.w.vout + ((long)b * a.G) * 4; }
__device__ __forceinline__ double plb_wave_sum(double v) {   // all 64 lanes get the sum
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define PLB_H 1024
#define PLB_LOGH 10
// block table size by lanes per particle: a block of 256 lanes holds 256 / LANES particles; at eight lanes (32 particles, 864 cells at the very
// most, ~150 when sorted) 512 slots halve the clear and the three flush sweeps; a full table falls back to global atomics as ever
template <int LANES> struct PlbTab { static constexpr int H = LANES == 8 ? 512 : PLB_H, LOGH = LANES == 8 ? 9 : PLB_LOGH; };
template <int LOGH>
__device__ __forceinline__ unsigned plb_hash_t(int cell) {
  unsigned h = (unsigned)cell;
  h ^= h >> 9; h *= 2654435761u; h ^= h >> 15;
  return h >> (32 - LOGH);
}
__device__ __forceinline__ unsigned plb_hash(int cell) {
  unsigned h = (unsigned)cell;
  h ^= h >> 9; h *= 2654435761u; h ^= h >> 15;
  return h >> (32 - PLB_LOGH);
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int LANES>
__device__ __forceinline__ double plb_quad_sum(double v) {
  if (LANES >= 4) {
    v += dpp_d<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);  // quad_perm [2,3,0,1]
  }
  if (LANES >= 8) v += dpp_d<0x141>(v);  // row_half_mirror: lane i <-> 7 - i of each group of eight = the other quad
  return v;
}

// grid_op for one touched cell (:200-232): (m, mv) -> v_out.  P0 / P0 + np*3: primitive positions at substeps f and f + 1.
__device__ __forceinline__ void plb_grid_cell(const PlbConst& c, long lin, double m, const double* mv, const double* P0, const double* soft, double* vv) {
  vv[0] = 0.0; vv[1] = 0.0; vv[2] = 0.0;
  if (!(m > 1e-12)) return;
  const int n = c.n_grid;
  const int I[3] = {(int)(lin / ((long)n * n)), (int)((lin / n) % n), (int)(lin % n)};
#pragma unroll
  for (int k = 0; k < 3; ++k) vv[k] = (1.0 / m) * mv[k] + c.g30dt[k];
  const double gp[3] = {I[0] * c.dx, I[1] * c.dx, I[2] * c.dx};
  const double* P1 = P0 + c.np * 3;
  for (int pi = 0; pi < c.np; ++pi) {                                    // Sphere.collide (sticky), primitives.py:46-53
    const double d0 = gp[0] - P0[pi * 3], d1 = gp[1] - P0[pi * 3 + 1], d2 = gp[2] - P0[pi * 3 + 2];
    const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2 + 1e-14) - c.radius[pi];
    const double sf = soft[pi];
    const double infl = fmin(exp(-dist * sf), 1.0);
    if (((sf > 0 && infl > 0.1) || dist <= 0.001) && sf > 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) vv[k] = (P1[pi * 3 + k] - P0[pi * 3 + k]) / c.dt;   // collider_v, identity rotations
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (I[d] < 3 && vv[d] < 0) {
      if (d != 1 || c.fric == 0) vv[d] = 0;
      else if (c.fric < 10) {
        const double lin_ = vv[1] + 1e-30;
        const double vit[3] = {vv[0] - I[0] * 1e-30, vv[1] - lin_ - I[1] * 1e-30, vv[2] - I[2] * 1e-30};
        const double lit = sqrt(vit[0] * vit[0] + vit[1] * vit[1] + vit[2] * vit[2] + 1e-8);
        const double s = fmax(1.0 + c.fric * lin_ / lit, 0.0);
        vv[0] = s * (vit[0] + I[0] * 1e-30); vv[2] = s * (vit[2] + I[2] * 1e-30); vv[1] = 0;
      } else { vv[0] = 0; vv[1] = 0; vv[2] = 0; }
    }
    if (I[d] > n - 3 && vv[d] > 0) vv[d] = 0;
  }
}

void plb_launch_p2g(const PlbArgs& a, int lanes, dim3 grid, hipStream_t st);   // plb_p2g<lanes> (plb.hip), for the adjoint's recompute

}  // namespace ud

// ---- host side, shared by plb.hip, plb_adj.hip and plb_cluster.hip ------------------------------------------
struct PlbCluster {     // persistent path (plb_cluster.hip): exchange grids for `per` envs per launch, allocated at create
  void* arena = nullptr;
  size_t bytes = 0;
  double* cg[3] = {nullptr, nullptr, nullptr};
  unsigned* bar = nullptr;
  double* gposacc = nullptr;
  double* gpar = nullptr;
  int* timeouts = nullptr;
  int W = 0, per = 0;   // parts per env; envs per launch (0 = this handle runs the multi-kernel path)
};
struct ud_plb {
  ud::PlbConst c;
  int B = 0, cap = 0, epoch = 1;   // B = envs every arena is sized for (ud_plb_conf.max_envs): fixed at create
  long G = 0;
  ud::PlbBuf w{};
  double* gm = nullptr;     // [B][G] grid mass of the loss kernels
  double* lred = nullptr;   // [B][16] loss partial sums
  void* arena = nullptr;
  int* order = nullptr;     // [B][Np] the handle's current spatial order (both paths)
  int sort_B = 0, sort_age = 0;   // envs the spatial order covers, forward calls since it was computed
  int lanes = 0;            // multi-kernel path: lanes per particle forced by the conf (0 = by launch size)
  int sort_every = 8;       // forward calls between two sorts (<= 0: never sort)
  PlbCluster cl;
};
// every arena of the handle, once, at create (no allocation and no host synchronisation in any step call)
int plb_reserve(ud_plb* h, int B, bool multi_kernel);
struct PlbCkOff { size_t hist, pos, perm, gck_cnt, gck_lin, gck_val, svd, total; };
PlbCkOff plb_ckpt_layout(const ud::PlbConst& c, int B);
void plb_bind_ckpt(ud::PlbArgs& a, const ud::PlbConst& c, int B, void* ckpt);
// persistent path (plb_cluster.hip)
int plb_cluster_plan(ud_plb* h, int max_envs);
int plb_cluster_reserve(ud_plb* h, int per);
size_t plb_cluster_ckpt_bytes(const ud_plb* h, int B);
int plb_cluster_step_fwd(ud_plb* h, int B, const double* x, const double* v, const double* C, const double* F, const double* prim_pos,
                         const double* softness, const double* action, const double* E, const double* nu, const double* ys, double* xo,
                         double* vo, double* Co, double* Fo, double* prim_o, const int* order, void* ckpt, hipStream_t st);
int plb_cluster_step_bwd(ud_plb* h, int B, const void* ckpt, const double* softness, const double* action, const double* E, const double* nu,
                         const double* ys, const double* g_x, const double* g_v, const double* g_C, const double* g_F, const double* g_prim_pos,
                         double* g_x0, double* g_v0, double* g_C0, double* g_F0, double* g_prim_pos0, double* g_action, double* g_E, double* g_nu,
                         double* g_ys, double* g_fric, hipStream_t st);
