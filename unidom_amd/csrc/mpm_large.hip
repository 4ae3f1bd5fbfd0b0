// MLS-MPM `step` for environments too large for one workgroup (N > 128 particles: the scaled whip_rope configs
// n_grid 128 / 256 -> N = 798 / 6675): many workgroups per env, the `res` grid dense in HBM, a handful of small
// kernels per substep.  Same arithmetic (shared device functions, mpm_device.h) and the same C ABI as mpm.hip.
//
// Per substep the forward runs  grid op (+ retiring the previous substep's cells) -> [g2p -> p2g of the next substep] (two launches:
// lg_grid, lg_g2p_p2g; round 3 -- rounds 1-2: clear+FK -> p2g -> grid op -> g2p, four launches), or, for
// solids with one primitive below the chip-filling size, ONE persistent launch per step call (mpm_cluster.h); the backward either
// clear -> p2g (recompute) -> grid op (recompute) -> g2p-adjoint -> grid-op adjoint -> p2g-adjoint (+ FK adjoint in extra blocks)
// or, with the grid checkpoint,  restore -> g2p-adjoint -> grid-op adjoint -> p2g-adjoint (+ FK adjoint), which for four lanes per
// particle and one primitive is two launches (lg_gadj_restore, lg_padj_gadj); ud_mpm_conf.deterministic: the forward is mpm_det.hip's, the
// backward this file's kernels with every arrival-ordered sum replaced by an ordered one (the `c.det` branch of mpm_large_step_bwd):
//   * p2g scatters with global_atomic_add_f32 into one float4 (m, mv) per cell and marks cells in a bitmap
//     (one bit per cell); the first toucher appends the cell to the env's ACTIVE LIST, so the grid op and the clear of
//     the next substep visit only touched cells (never the 32^3..128^3 dense volume the reference sweeps ~10x);
//   * particle state history (24 floats/particle/substep, SoA) doubles as the backward's checkpoint;
//   * no host synchronisation: all launches go to the caller's stream in order.
// Scatters are staged per workgroup in an LDS cell table; particles can be kept in Morton order inside the handle (lg_sort);
// small launches split the envs over two streams.
#include <cstdlib>

#include "mpm_device.h"
#include "mpm_large.h"
#include "mpm_det_host.h"
#include "mpm_collide.h"

namespace ud {

struct LargeBuf {
  float4* val;      // [B][G]  (m, mvx, mvy, mvz); after the grid op (fwd): (m, vx, vy, vz)
  float4* val2;     // [B][G]  the forward's second (m, mv) grid: odd substeps (ls3 mode: see LargeArgs)
  float4* vel;      // [B][G]  bwd: grid velocity after the grid op
  float4* gacc;     // [B][G]  bwd: cotangent of grid velocity -> (g_mv xyz, g_m)
  unsigned* bits;   // [B][W32]  one bit per cell: in the active list of the substep in flight (cleared with the list)
  int* list;        // [3][B][cap]
  int* count;       // [3][B]
  // primitive arrays carry a primitive axis: [B][P][...], P = c.n_prim (1 in position-control mode)
  float* ppos;      // [B][P][S*3]  primitive position (evolving)
  float* prot;      // [B][P][S*4]
  float* ppin;      // [B][P][S*3]  input position array (bwd clip factors)
  float* trq;       // [B][S]    Q6 scalar per substep
  float* gppos;     // [B][P][S*3]  bwd
  float* gpv;       // [B][P][S*3]  bwd
  float* acc;       // [B][4]    bwd: friction, mu, lamda accumulators
  float* pscr;      // [B][3][Np] bwd: the g2p adjoint's cotangent of fx, per particle, for lg_p2g_adj
  float* hist;      // [B][2][24][Np] ping-pong state when the caller passes no checkpoint
  float* gstate;    // [B][24][Np] bwd: cotangent state (gx,gv,gC,gF) SoA
  float* grot;      // [B][P][S*4]  bwd, soft contact: cotangent of the rotation array
  float* gpw;       // [B][P][S*3]  bwd, soft contact: cotangent of the angular velocity rows
  float* gpsz;      // [B][P][4]    bwd, soft contact: cotangents of primitive size[3] and friction (enter the norm only)
  int* perm;        // [B][Np]      spatial order of a forward without checkpoint (sort_particles)
};

struct LargeArgs {
  MpmConst c;
  LargeBuf w;
  const int* material;
  const float* hard;
  int B, f, cap;
  // Forward without a clear launch (ls3 != 0): the (m, mv) grid alternates between val and val2 with the substep (vb), lg_grid(f) retires
  // the cells of substep f - 1 in the OTHER buffer and clears its own list's bits beside its work, and the active lists take three
  // slots (ls: this substep's; lprev: the one being retired; lnext: the one the next substep's p2g fills, its count reset by lg_grid).
  // ls3 == 0 (every backward launch): one grid, two lists alternating with f, lg_clear_fk in front of every substep.
  int ls3, vb, ls, lprev, lnext;
  int gpar;               // backward with the grid checkpoint, fused kernels: the cotangent grid of substep f is w.gacc for even f and
                          // w.val (untouched by that backward otherwise) for odd f, so that substep f - 1's scatter can run beside f's gather
  long W32;               // bitmap words per env
  int b0;                 // first env of this launch (env groups on separate streams)
  long G;
  // grid checkpoint (ud_mpm_conf.grid_ckpt_cells > 0): the forward appends one record {key, m, mv[3], v[3]} per active
  // cell and substep to a per-env pool inside the caller's checkpoint, the backward restores from it instead of running
  // p2g + grid op again.  gck_idx[f] = first record of substep f ([S+1] ints per env), pool holds gck_budget records.
  float* gck_base;        // = checkpoint base (env stride hist_stride_b); nullptr = off
  long gck_off_idx, gck_off_pool;
  long gck_off_crec;      // soft contact, multi-kernel forward (round 5): beside every grid-checkpoint record, one float4 (e, n[3]) per primitive =
                          // exp(-dist * softness) and the finite-difference normal of collide_batch for that cell; 0 = none.  The grid-op adjoint reads
                          // them instead of evaluating seven SDFs per cell and primitive three times over (collide_geom).
  int gck_budget;
  int* status;
  // spatial order (ud_mpm_conf.sort_particles): slot p of the SoA history holds the caller's particle perm[p]; nullptr = as given
  const int* perm;        // [B][perm_stride]
  long perm_stride;
  int svd_rows;           // the history records carry the SVD factors of each substep's F (rows 24 .. 44; ck_layout): the backward reads them
  const float* hist_in;   // state at substep f      [B][*][24][Np] with stride
  float* hist_out;        // state at substep f + 1
  long hist_stride_b;     // floats between envs
  const float *psize, *friction, *mu, *lamda, *action;
  // logical launch shape of the per-substep kernels: (nbx blocks per env) x (Bg envs), issued as a ONE-dimensional grid (lg_bid)
  int nbx, Bg, xcd;
  // deterministic backward (ud_mpm_conf.deterministic, position control; null otherwise): every sum that an atomic would order by arrival goes
  // to an array instead and is added up in a fixed order by a kernel of its own (mpm_det.hip)
  float* det_cellred;     // [B][det_capc][det_K] per listed cell (the list is sorted): what the grid-op adjoint adds to the env's cotangents --
                          // ground friction, then the controlled velocity xyz (position control) or 18 values per primitive (soft contact)
  int det_capc, det_K;
  float* det_pacc;        // [B][2][Np]  per particle: mu / lamda cotangents, accumulated over the substeps by the particle's own thread
  float* det_normpart;    // [B][LG_NORM_PARTS] per block of lg_bwd_norm: its share of the squared norm
};
constexpr int LG_NORM_PARTS = 64;

// Block -> (block within the env, env) of the per-substep kernels.  Workgroups are dealt round-robin over the 8 XCDs in linear order, and
// each XCD has its own L2: with the natural (x, env) order the 13-27 blocks of one env land on all eight, and every L2 fetches the env's
// grid lines (vel, cotangent grid) for itself -- measured on the rope at n_grid 128: ~215 of the 420 KB the backward's particle kernel
// fetches per env and substep were those refetches (tools/pmc_large.sh; 128-B lines of eight z-neighbours x 2 grids x 8 XCDs).  With
// xcd != 0 (Bg >= 8) ids congruent mod 8 belong to one env -- the mapping of the cluster kernels (clm_decode) -- so an env's grid lives
// in ONE L2.  Speed and traffic only, never correctness.  Fewer than 8 envs keep the natural order (an env per XCD would idle the rest).
struct LgB { int x, y; bool ok; };
__device__ __forceinline__ LgB lg_bid(const LargeArgs& a) {
  const int id = blockIdx.x;
  LgB o;
  if (a.xcd) {
    const int j = id >> 3;
    o.y = (j / a.nbx) * 8 + (id & 7);
    o.x = j % a.nbx;
  } else {
    o.y = id / a.nbx;
    o.x = id % a.nbx;
  }
  o.ok = o.y < a.Bg;
  return o;
}

// ---- agent-scope accesses (persistent cluster kernels, mpm_cluster.h): sc1 loads bypass the CU's L1, sc1 stores write through
__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int COH> __device__ __forceinline__ float ld_f(const float* p) { return COH ? ldc(p) : *p; }
template <int COH> __device__ __forceinline__ void st_f(float* p, float v) { if (COH) stc(p, v); else *p = v; }
// p += v: a read-modify-write, or (COH == 2: other workgroups add to the same word at the same time) an atomic
template <int COH> __device__ __forceinline__ void add_f(float* p, float v) { if (COH == 2) { if (v != 0.f) atomicAdd(p, v); } else st_f<COH>(p, ld_f<COH>(p) + v); }

__device__ __forceinline__ long cell_lin(const MpmConst& c, int key) {
  int ci, cj, ck;
  decode_cell(c, key, ci, cj, ck);
  return ((long)ci * c.res[1] + cj) * c.res[2] + ck;
}

__device__ __forceinline__ int lg_ls(const LargeArgs& a) { return a.ls3 ? a.ls : (a.f & 1); }
__device__ __forceinline__ float4* lg_val(const LargeArgs& a) { return (a.ls3 && a.vb) ? a.w.val2 : a.w.val; }
__device__ __forceinline__ float4* lg_gacc(const LargeArgs& a, int f) { return (a.gpar && (f & 1)) ? a.w.val : a.w.gacc; }
// active-list slot of substep f in the backward: two slots alternating with f; three in the fused (two-launch) backward, so that the keys of
// substep f + 2 are still there when the restore of f zeroes that substep's cotangent cells (4 B per cell instead of the 32-B record lines)
__device__ __forceinline__ int lg_bslot(const LargeArgs& a, int f) { return a.gpar ? ((f % 3) + 3) % 3 : (f & 1); }

// caller's index of the particle in slot p
__device__ __forceinline__ int user_index(const LargeArgs& a, int b, int p) { return a.perm ? a.perm[(long)b * a.perm_stride + p] : p; }

__device__ __forceinline__ int* gck_idx(const LargeArgs& a, int b) { return (int*)(a.gck_base + (long)b * a.hist_stride_b + a.gck_off_idx); }
__device__ __forceinline__ float4* gck_pool(const LargeArgs& a, int b) { return (float4*)(a.gck_base + (long)b * a.hist_stride_b + a.gck_off_pool); }
__device__ __forceinline__ float4* gck_crec(const LargeArgs& a, int b) { return (float4*)(a.gck_base + (long)b * a.hist_stride_b + a.gck_off_crec); }

__device__ __forceinline__ void touch(const LargeArgs& a, int b, int key, long lin) {
  const unsigned bit = 1u << (lin & 31);
  const unsigned old = atomicOr(&a.w.bits[(long)b * a.W32 + (lin >> 5)], bit);
  if (!(old & bit)) {
    const int cur = lg_ls(a);
    const int e = atomicAdd(&a.w.count[cur * a.B + b], 1);
    if (e < a.cap) a.w.list[((long)cur * a.B + b) * a.cap + e] = key;
  }
}

__device__ __forceinline__ void load_prim_f(const LargeArgs& a, int b, int f, PrimF& pf, float* pv) {
  const int S = a.c.steps, fc = min(max(f, 0), S - 1);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    pv[d] = clipf(a.action[b * 6 + d], -1.f, 1.f) * 1.f / (float)S;
    pf.pos[d] = a.w.ppos[(long)b * S * 3 + fc * 3 + d]; pf.size[d] = a.psize[b * 3 + d]; pf.pv[d] = pv[d];
  }
  const float* r = a.w.prot + (long)b * S * 4 + fc * 4;
  float r0 = r[0], r1 = -r[1], r2 = -r[2], r3 = -r[3];
  float n = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + 1e-12f;
  pf.iq[0] = r0 / n; pf.iq[1] = r1 / n; pf.iq[2] = r2 / n; pf.iq[3] = r3 / n;
  pf.friction = a.friction[b];
}

__device__ __forceinline__ void load_prim(const LargeArgs& a, int b, PrimF& pf, float* pv) { load_prim_f(a, b, a.f, pf, pv); }

// soft contact: rows f and f + 1 (clamped, Q5) of the primitive arrays
__device__ __forceinline__ void load_primc_f(const LargeArgs& a, int b, int ip, int f, PrimC& pc) {
  const int S = a.c.steps, f0 = min(max(f, 0), S - 1), f1 = min(max(f + 1, 0), S - 1);
  const long bp = (long)b * a.c.n_prim + ip;
  const float* pp = a.w.ppos + bp * S * 3;
  const float* pr = a.w.prot + bp * S * 4;
#pragma unroll
  for (int d = 0; d < 3; ++d) { pc.p0[d] = pp[f0 * 3 + d]; pc.p1[d] = pp[f1 * 3 + d]; pc.size[d] = a.psize[bp * 3 + d]; }
#pragma unroll
  for (int d = 0; d < 4; ++d) { pc.r0[d] = pr[f0 * 4 + d]; pc.r1[d] = pr[f1 * 4 + d]; }
  pc.soft = a.c.prim_softness_each[ip]; pc.mu = a.c.prim_friction_each[ip]; pc.kind = a.c.sdf_kind;
  primc_finish(pc);
}
__device__ __forceinline__ void load_primc(const LargeArgs& a, int b, int ip, PrimC& pc) { load_primc_f(a, b, ip, a.f, pc); }

__device__ __forceinline__ void load_state(const float* h, int Np, int p, float* x, float* v, float* Cm, float* F) {
#pragma unroll
  for (int d = 0; d < 3; ++d) { x[d] = h[d * Np + p]; v[d] = h[(3 + d) * Np + p]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[d] = h[(6 + d) * Np + p]; F[d] = h[(15 + d) * Np + p]; }
}


// ---- block-level LDS staging of scatter-adds (north_star: "p2g staged through LDS with per-cell atomics localised
// by ... particle blocks"): consecutive particles are spatial neighbours (lattice seeding order), so the 256
// particles of a block touch a few hundred distinct cells; they are summed in an LDS cell table first and each
// distinct cell is then flushed with ONE global atomic per component.  That removes the ~50-way same-address
// contention plain global atomics suffer on a compact body (measured: 17 us -> see DESIGN.md per env-substep).
#define LG_SCATTER_T 128   // threads per workgroup in the two scatter kernels

// Diagnostic build only (-DUD_LG_STAMPS, tools/lg_stamps.sh): s_memtime at the phase boundaries of the three particle kernels,
// summed over wave 0 of every block into ud_lg_stamps[kernel][phase] ([..][7] = number of blocks); vector-memory waits are
// forced at the stamps (s_waitcnt vmcnt(0)) so that a load's latency is billed to the phase that issued it.
#ifdef UD_LG_STAMPS
__device__ unsigned long long ud_lg_stamps[4][8];
#define LG_STAMP_BEGIN unsigned long long lg_t0_ = __builtin_amdgcn_s_memtime();
#define LG_STAMP(K, PH) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&ud_lg_stamps[K][PH], t_ - lg_t0_); lg_t0_ = t_; \
      if ((PH) == 0) atomicAdd(&ud_lg_stamps[K][7], 1ull); } } while (0)
#else
#define LG_STAMP_BEGIN
#define LG_STAMP(K, PH) do {} while (0)
#endif
#ifndef LG_LOGH1
#define LG_LOGH1 9
#endif
// staging-table slots per block.  32 particles (4 lanes each) touch ~150-250 distinct cells; 128 particles (1 lane each)
// touch 500-900, but a table that holds them all costs more to clear and flush than the spill to global atomics
// it avoids (measured, n_grid 256: 2048 slots 47.8 k substeps/s, 512 slots 69 k).
template <int LANES> struct LgTable { static constexpr int LOGH = LANES == 4 ? 9 : LG_LOGH1, H = 1 << LOGH; };
// staged values are float64: ds_add_f32 retires ~20x slower than ds_add_f64 on gfx950 (tools/ubench_lds_atomic.hip)
// val is component-major ([4][H]): slot-major rows of 4 doubles (32 B) put the 64 lanes of one ds_add_f64 on 8 of the 64 banks
struct BlockTable { int* key; double* val; };   // key[H], val[4][H]

// The particle kernels use the small path's lane mapping: lane = 4*particle + q, the quad splits the 27 stencil cells
// 7/7/7/6 and reduces with DPP.  One lane per particle left the chip mostly idle (N = 798, 32 envs: 400 waves on
// 1024 SIMDs, each walking 27 cells serially).
// LANES = 4 when the launch is small enough to need the parallelism (B*N below ~100 k particles: 1.2x at n_grid 128),
// LANES = 1 (one lane per particle, serial 27-cell walk, no redundant pre-pass) once the chip is full (n_grid 256).
template <int LANES>
__device__ __forceinline__ float lg_quad_sum(float v) {
  if (LANES == 4) {
    v += dpp_f<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);  // quad_perm [2,3,0,1]
  }
  return v;
}
template <int LANES> constexpr size_t lg_table_bytes() { return (size_t)LgTable<LANES>::H * (sizeof(int) + 4 * sizeof(double)); }   // dynamic LDS
template <int H>
__device__ __forceinline__ BlockTable bt_make() {
  extern __shared__ double lg_smem[];
  return BlockTable{(int*)(lg_smem + H * 4), lg_smem};
}

template <int LOGH>
__device__ __forceinline__ unsigned lg_hash(int cell) {
  unsigned h = (unsigned)cell;
  h ^= h >> 9; h *= 2654435761u; h ^= h >> 15;
  return h >> (32 - LOGH);
}

template <int H>
__device__ __forceinline__ void bt_clear(const BlockTable& t) {
  for (int s = threadIdx.x; s < H; s += blockDim.x) {
    t.key[s] = -1;
    t.val[s] = 0.0; t.val[H + s] = 0.0; t.val[2 * H + s] = 0.0; t.val[3 * H + s] = 0.0;
  }
}

// returns the slot of `cell`, or -1 when the table is full (the caller then falls back to a global atomic)
template <int H, int LOGH>
__device__ __forceinline__ int bt_slot(const BlockTable& t, int cell) {
  unsigned s = lg_hash<LOGH>(cell);
  for (int probe = 0; probe < 64; ++probe) {
    const int cur = t.key[s];
    if (cur == cell) return (int)s;
    if (cur == -1) {
      const int old = atomicCAS(&t.key[s], -1, cell);
      if (old == -1 || old == cell) return (int)s;
    }
    s = (s + 1) & (H - 1);
  }
  return -1;
}

// ---- block window: the table addressed directly --------------------------------------------------------
// The particles of a block are spatial neighbours (lattice order, or sort_particles), so their stencils often fit a small box:
// when the block's base cells span at most 6 per axis the table IS the 8x8x8 cells starting at the smallest base cell,
// slot = offset inside it (z fastest, like the grid in memory: a row of 8 slots is 128 contiguous bytes of cells) -- no key
// compare, no CAS, nothing returned from LDS inside the walk.  A block whose particles are
// spread wider keeps the open-addressing table above.  Measured (tools/abl_p2g.sh, window forced on / off): n_grid-256 rope
// lg_p2g 83.8 us with the hash only, 76.9 us adaptive; pour_soup 107 either way (few of its blocks qualify), and 161 us with the
// window forced on (cells outside it go to HBM atomics) -- hence the per-block choice.
struct BlockWin { int on, ox, oy, oz; };
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_min_i(int v) {   // DPP butterflies inside the rows of 16, the four row minima through readlane (cf. wave_sum)
  v = min(v, dpp_i<0xB1>(v));    // quad_perm [1,0,3,2]
  v = min(v, dpp_i<0x4E>(v));    // quad_perm [2,3,0,1]
  v = min(v, dpp_i<0x141>(v));   // row_half_mirror
  v = min(v, dpp_i<0x140>(v));   // row_mirror
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// every thread of the block calls this (it holds the barrier that also publishes bt_clear)
__device__ __forceinline__ BlockWin bt_window(const MpmConst& c, bool live, const int* base) {
  __shared__ int s_lo[3], s_hi[3];
  if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = 0x7fffffff; }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int bw = base[d] + ((base[d] < 0) ? c.res[d] : 0);     // negatives wrap (Q9): far from the rest, such a block keeps the hash
    const int lo = wave_min_i(live ? bw : 0x7fffffff), nhi = wave_min_i(live ? -bw : 0x7fffffff);
    if ((threadIdx.x & 63) == 0) { atomicMin(&s_lo[d], lo); atomicMin(&s_hi[d], nhi); }
  }
  __syncthreads();
  BlockWin w;
  w.ox = s_lo[0]; w.oy = s_lo[1]; w.oz = s_lo[2];
  w.on = (s_lo[0] != 0x7fffffff) && (-s_hi[0] - s_lo[0] <= 5) && (-s_hi[1] - s_lo[1] <= 5) && (-s_hi[2] - s_lo[2] <= 5);
  if (UD_MPM_ABLATE & 1024) w.on = 1;   // timing only: window forced on (outside cells go to HBM atomics) / off
  if (UD_MPM_ABLATE & 2048) w.on = 0;
  return w;
}
// slot of a packed cell key inside the window (LgTable<>::H == 512), or -1 outside (clamped / wrapped stencil cells)
__device__ __forceinline__ int bt_win_slot(const BlockWin& w, int key) {
  const unsigned rx = (unsigned)((key & 1023) - w.ox), ry = (unsigned)(((key >> 10) & 1023) - w.oy), rz = (unsigned)(((key >> 20) & 1023) - w.oz);
  return ((rx | ry | rz) < 8u) ? (int)(rz | (ry << 3) | (rx << 6)) : -1;
}
template <int H, int LOGH>
__device__ __forceinline__ int bt_find(const BlockTable& t, const BlockWin& w, int cell) {
  static_assert(H == 512, "the block window is 8x8x8 slots");
  if (w.on) {
    const int s = bt_win_slot(w, cell);
    if (s >= 0) t.key[s] = cell;           // every writer stores the same value; the flush reads it after the barrier
    return s;
  }
  return bt_slot<H, LOGH>(t, cell);
}

// ---- forward kernels ---------------------------------------------------------------------------------
// The kernels over an env's active-cell list do not know its length on the host (`cap` = 54 N cells at most; 2-5 k of pour_soup's
// 412 k are ever active).  One 256-cell tile per block for `cap` cells meant ~51 k blocks per launch there, an eighth of that with
// eight tiles per block still 6 432, of which ~280 found work: lg_grid_adj took 43 us for 14 us of work inside its blocks
// (tools/lg_stamps.sh) -- the rest is the dispatch of blocks that read the count and leave.  Now: at most LG_CELL_BLOCKS blocks
// per env, block x walks tiles x, x + gridDim.x, ... until one lies past the list -- the first 24 tiles (6 144 cells) still go to
// different blocks (giving a block consecutive tiles put them all on one or two blocks per env and made lg_grid_adj 4x slower);
// longer lists take a second trip.
constexpr int LG_CELL_BLOCKS = 24;
__host__ __device__ constexpr int lg_cell_blocks(int cap) { return (cap + 255) / 256 < LG_CELL_BLOCKS ? (cap + 255) / 256 : LG_CELL_BLOCKS; }

// clear the cells the previous substep touched; block 0 of each env also runs forward_kinematics (:185-194)
__global__ void __launch_bounds__(256) lg_clear_fk(LargeArgs a, int do_fk, int clear_bwd) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0;
  const int prev = (a.f + 1) & 1, cur = a.f & 1;   // works for f ascending (forward) and descending (backward)
  const int n = min(a.w.count[prev * a.B + b], a.cap);
  for (int u = 0;; ++u) {
    const int t = (u * a.nbx + lgb_.x) * 256 + threadIdx.x;
    if (t - (int)threadIdx.x >= n) break;
    if (t < n) {
      const long lin = cell_lin(a.c, a.w.list[((long)prev * a.B + b) * a.cap + t]);
      a.w.val[(long)b * a.G + lin] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.w.bits[(long)b * a.W32 + (lin >> 5)] = 0u;   // every bit set in this word belongs to a cell of this list
      if (clear_bwd) a.w.gacc[(long)b * a.G + lin] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (lgb_.x == 0) {
    const int S = a.c.steps, f = a.f, tid = threadIdx.x;
    if (tid == 0) a.w.count[cur * a.B + b] = 0;
    if (tid == 0 && a.gck_base && !clear_bwd) {   // forward: records of substep f start where those of f - 1 end
      int* idx = gck_idx(a, b);
      idx[f] = (f == 0) ? 0 : idx[f - 1] + n;
    }
    if (do_fk)
    for (int ip = 0; ip < a.c.n_prim; ++ip) {
      const long bp = (long)b * a.c.n_prim + ip;
      float* pp = a.w.ppos + bp * S * 3;
      float* pr = a.w.prot + bp * S * 4;
      for (int e0 = 0; e0 < S * 3; e0 += blockDim.x) {   // read all, then write all (one block per env)
        const int e = e0 + tid;
        float pending = 0.f;
        if (e < S * 3) {
          const int row = e / 3, d = e - row * 3;
          const float pva = clipf(a.action[bp * 6 + d], -1.f, 1.f) * 1.f / (float)S;
          pending = (row == f + 1) ? (pp[min(f, S - 1) * 3 + d] + pva) : pp[e];
        }
        __syncthreads();
        if (e < S * 3) pp[e] = clipf(pending, -2.f, 2.f);
        __syncthreads();
      }
      if (tid == 0 && f + 1 < S) {
        float pw[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) pw[d] = clipf(a.action[bp * 6 + 3 + d], -1.f, 1.f) * 1.f / (float)S;
        float ang = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) + 1e-12f;
        float sn = sinf(ang / 2.f);
        float q[4] = {cosf(ang / 2.f), pw[0] / ang * sn, pw[1] / ang * sn, pw[2] / ang * sn};
        const float* r = pr + f * 4;
        float o0 = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
        float o1 = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
        float o2 = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
        float o3 = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
        float nn = clipf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3), 1e-12f, INFINITY);
        float* w = pr + (f + 1) * 4;
        w[0] = o0 / nn; w[1] = o1 / nn; w[2] = o2 / nn; w[3] = o3 / nn;
      }
    }
  }
}

// particle pre-pass + scatter (:233-274).  store_F: write F_out into the next history record (forward only)
// `reg`: nullptr = the particle's state comes from the history record hist_in; else x[3], v[3], C[9] in registers (the fused forward kernel:
// g2p of the previous substep has just produced them) and only F is read from hist_in.  Every thread of the block calls it (barriers inside).
template <int LANES>
__device__ __forceinline__ void lg_p2g_body(const LargeArgs& a, int store_F, const float* reg) {
  const LgB lgb_ = lg_bid(a);

  constexpr int TH = LgTable<LANES>::H, TLOG = LgTable<LANES>::LOGH;
  const BlockTable bt = bt_make<TH>();
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const MpmConst& c = a.c;
  LG_STAMP_BEGIN
  bt_clear<TH>(bt);
  float4* val = lg_val(a) + (long)b * a.G;
  const bool live = p < c.N;
  Pre q;
  float v[3] = {0.f, 0.f, 0.f};
  q.base[0] = q.base[1] = q.base[2] = 0;
  if (live) {
    float x[3], Cm[9], F[9];
    if (reg) {
      const float* hi = a.hist_in + (long)b * a.hist_stride_b;
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = reg[d]; v[d] = reg[3 + d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { Cm[d] = reg[6 + d]; F[d] = hi[(15 + d) * c.Np + p]; }
    } else {
      load_state(a.hist_in + (long)b * a.hist_stride_b, c.Np, p, x, v, Cm, F);
    }
    const int up = user_index(a, b, p);
    LG_STAMP(0, 0);   // table clear + state loads
    // (store_F = the caller's checkpoint is being written: the SVD factors go into this substep's record, one lane of the quad)
    float* svd_o = (store_F && a.svd_rows && qi == 0) ? const_cast<float*>(a.hist_in) + (long)b * a.hist_stride_b + (long)24 * c.Np + p : nullptr;
    particle_pre<false, true>(c, x, Cm, F, a.mu[b], a.lamda[b], a.material[up], a.hard[up], q, nullptr, svd_o, nullptr, c.Np);
    if (store_F && qi == 0) {
      float* ho = a.hist_out + (long)b * a.hist_stride_b;
#pragma unroll
      for (int d = 0; d < 9; ++d) ho[(15 + d) * c.Np + p] = q.Fn[d];
    }
  }
  LG_STAMP(0, 1);     // pre-pass (F update, SVD, stress)
  const BlockWin win = bt_window(c, live, q.base);
  LG_STAMP(0, 2);     // window reduction + its two barriers (arrival skew of the block's waves)
  if (live) {
    // The walk is staggered -- each particle starts at a different cell / column -- so that the lanes of a run of particles
    // sharing a base cell (sorted or lattice-seeded neighbours; pour_soup's vegetable cloud has ~26 per cell) never add to the
    // same table slot at once.
    // Measured on pour_soup, 32 envs x 7631 particles (tools/abl_p2g.sh, tools/pmc_large.sh, profiles/r01i_pmc_large_pour_soup.csv):
    // with one stencil cell instead of 27 this kernel takes 30 us, with 27 it took 158: staggering the walk and the
    // component-major table brought 107, the block window nothing here (the liquid's blocks span more than 6 cells and keep the
    // hash) but 8 % on the lattice-seeded ropes, the fast walk below 97 (rope at n_grid 256: step forward 9.2 -> 8.5 ms).
    // Of the 97 us the table atomics are ~4 and the flush ~19; the counters say 3.1 k VALU + 1.4 k SALU + 244 LDS instructions per
    // wave -- VALU issue is about a quarter of the duration, LDS busy + bank conflicts about a fifth.  What bounds the rest
    // (3.75 waves per SIMD in a single round: latency of the state loads, the SVD chain and the flush's HBM atomics) is not
    // separated yet.
    // Fast walk: when the block window is on and the particle's stencil is interior (no wrap, drop or clamp on any axis), the
    // 27 cells are slot0 + 64 i + 8 j + k / key0 + i + (j << 10) + (k << 20): nine (j, k) columns walked dynamically (staggered:
    // their slots fall on nine different bank pairs), the three i cells of a column unrolled with their weights and affine
    // terms hoisted.
    const bool interior = win.on && !(UD_MPM_ABLATE & 64) && q.base[0] >= 0 && q.base[1] >= 0 && q.base[2] >= 0 &&
                          q.base[0] + 2 < c.res[0] && q.base[1] + 2 < c.res[1] && q.base[2] + 2 < c.res[2];
    if (interior) {
      const int key0 = q.base[0] | (q.base[1] << 10) | (q.base[2] << 20);
      const int slot0 = (q.base[2] - win.oz) | ((q.base[1] - win.oy) << 3) | ((q.base[0] - win.ox) << 6);
      float wx[3], ax[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        wx[i] = q.w[i * 3];
        const float dp0 = ((float)i - q.fx[0]) * c.dx;
#pragma unroll
        for (int r = 0; r < 3; ++r) ax[r * 3 + i] = q.affine[r * 3] * dp0;
      }
      const int rot9 = (p * LANES) % 9;
#pragma unroll 1
      for (int it = qi; it < 9; it += LANES) {
        const int col = it + rot9 >= 9 ? it + rot9 - 9 : it + rot9;
        const int j = col / 3, k = col - 3 * j;
        const float wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
        const float dp1 = ((float)j - q.fx[1]) * c.dx, dp2 = ((float)k - q.fx[2]) * c.dx;
        float br[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) br[r] = c.p_mass * v[r] + q.affine[r * 3 + 1] * dp1 + q.affine[r * 3 + 2] * dp2;
        const int sl = slot0 + 8 * j + k, key = key0 + (j << 10) + (k << 20);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float wgt = wx[i] * wj * wk;
          bt.key[sl + 64 * i] = key + i;
          __hip_atomic_fetch_add(&bt.val[sl + 64 * i], (double)(wgt * c.p_mass), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
          for (int r = 0; r < 3; ++r)
            __hip_atomic_fetch_add(&bt.val[(1 + r) * TH + sl + 64 * i], (double)(wgt * (br[r] + ax[r * 3 + i])), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    } else {
    const int rot = (p * LANES) % 27;
#pragma unroll 1
    for (int it = qi; it < ((UD_MPM_ABLATE & 64) ? 1 : 27); it += LANES) {
      const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const float weight = sel3(q.w, 0, i) * sel3(q.w, 1, j) * sel3(q.w, 2, k);
      const int sc = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      const int gc = cell_gather(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      if (sc >= 0) {
        const float dp0 = ((float)i - q.fx[0]) * c.dx, dp1 = ((float)j - q.fx[1]) * c.dx, dp2 = ((float)k - q.fx[2]) * c.dx;
        const int sl = (UD_MPM_ABLATE & 512) ? (int)lg_hash<TLOG>(sc) : bt_find<TH, TLOG>(bt, win, sc);   // 512: timing only, no lookup
        float contrib[4];
        contrib[0] = weight * c.p_mass;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float ad = q.affine[r * 3] * dp0 + q.affine[r * 3 + 1] * dp1 + q.affine[r * 3 + 2] * dp2;
          contrib[1 + r] = weight * (c.p_mass * v[r] + ad);
        }
        if (UD_MPM_ABLATE & 256) {          // timing only: the arithmetic without the LDS atomics
          if (contrib[0] + contrib[1] + contrib[2] + contrib[3] == 1.2345e30f) bt.val[sl] = 1.0;
        } else if (sl >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) __hip_atomic_fetch_add(&bt.val[r * TH + sl], (double)contrib[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {   // block table full: straight to HBM
          float* cell = (float*)(val + cell_lin(c, sc));
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(cell + r, contrib[r]);
          touch(a, b, sc, cell_lin(c, sc));
        }
      }
      if (gc != sc) touch(a, b, gc, cell_lin(c, gc));   // Q5: a clamped gather cell takes part with m = 0
    }
    }
  }
  // flush.  Value atomics: four lanes per cell, one per component -- a cell is 16 contiguous bytes and, in a block window, eight
  // consecutive slots are 128 contiguous bytes, the shape global float atomics want (one lane per cell and component meant 64
  // lanes on 64 different 64-byte lines: each atomic its own memory-side request, and 4 x the bytes in WRITE_SIZE).
  // Cells this substep sees for the first time join the env's active list: first toucher = whoever sets the cell's bit in the
  // env's bitmap -- one returning atomic OR per row of eight slots in a block window (per cell otherwise) instead of an
  // exchange per cell on a 4-byte stamp -- and the appends of a block are aggregated: one atomicAdd on the env's counter per
  // block (~2 k same-address atomics per env-substep at n_grid 256, one per active cell, were 80 % of the whole step).
  LG_STAMP(0, 3);     // the 27-cell walk
  __shared__ int s_new, s_base;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  LG_STAMP(0, 4);     // barrier before the flush
  if (UD_MPM_ABLATE & 128) return;
  {
    const int r = threadIdx.x & 3;
#pragma unroll 4
    for (int sl = threadIdx.x >> 2; sl < TH; sl += LG_SCATTER_T / 4) {
      const int key = bt.key[sl];
      if (key < 0) continue;
      atomicAdd((float*)(val + cell_lin(c, key)) + r, (float)bt.val[r * TH + sl]);
    }
  }
  unsigned* bits = a.w.bits + (long)b * a.W32;
  const int cur = lg_ls(a);
  int* list = a.w.list + ((long)cur * a.B + b) * a.cap;
  if (win.on) {             // block-uniform.  64 rows of eight slots: wave 0 alone, no barrier
    if (threadIdx.x >= 64) return;
    const int row = threadIdx.x;
    unsigned m8 = 0;
#pragma unroll
    for (int z = 0; z < 8; ++z) m8 |= (bt.key[row * 8 + z] >= 0 ? 1u : 0u) << z;
    unsigned fresh = 0;
    if (m8) {
      const long lin0 = ((long)(win.ox + (row >> 3)) * c.res[1] + (win.oy + (row & 7))) * c.res[2] + win.oz;
      const int sh = (int)(lin0 & 31);
      unsigned* wd = bits + (lin0 >> 5);
      const unsigned lo = atomicOr(wd, m8 << sh);
      const unsigned hi = (sh > 24) ? atomicOr(wd + 1, m8 >> (32 - sh)) : 0u;
      const unsigned old8 = ((lo >> sh) | ((sh > 24) ? hi << (32 - sh) : 0u)) & 0xffu;
      fresh = m8 & ~old8;
    }
    LG_STAMP(0, 5);     // flush: value atomics + bitmap
    // offsets inside the block's share of the list: an inclusive wave scan of the per-row counts (no LDS counter: a relaxed
    // load of one "after every lane's add" has no ordering against the adds of the lanes in the other arm of `nnew ?`)
    const int nnew = __popc(fresh);
    int incl = nnew;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (threadIdx.x >= d) incl += up; }
    const int mine = incl - nnew, total = __shfl(incl, 63);
    int base = 0;
    if (threadIdx.x == 0 && total) base = atomicAdd(&a.w.count[cur * a.B + b], total);
    base = __shfl(base, 0);
    int e = base + mine;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
      if (!(fresh & (1u << z))) continue;
      if (e < a.cap) list[e] = bt.key[row * 8 + z];
      ++e;
    }
    LG_STAMP(0, 6);     // list append
    return;
  }
  constexpr int PER = TH / LG_SCATTER_T;
  unsigned fresh = 0;
  int nnew = 0;
  {
    unsigned old[PER], bit[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {   // the returning ORs of a lane go out together
      const int key = bt.key[threadIdx.x + u * LG_SCATTER_T];
      bit[u] = 0u; old[u] = 0u;
      if (key >= 0) {
        const long lin = cell_lin(c, key);
        bit[u] = 1u << (lin & 31);
        old[u] = atomicOr(bits + (lin >> 5), bit[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (bit[u] & ~old[u]) { fresh |= 1u << u; ++nnew; }
  }
  LG_STAMP(0, 5);     // flush: value atomics + bitmap
  const int mine = nnew ? atomicAdd(&s_new, nnew) : 0;
  __syncthreads();
  if (threadIdx.x == 0) s_base = s_new ? atomicAdd(&a.w.count[cur * a.B + b], s_new) : 0;
  __syncthreads();
  int e = s_base + mine;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (!(fresh & (1u << u))) continue;
    if (e < a.cap) list[e] = bt.key[threadIdx.x + u * LG_SCATTER_T];
    ++e;
  }
  LG_STAMP(0, 6);     // list append
}

template <int LANES>
__global__ void __launch_bounds__(LG_SCATTER_T) lg_p2g(LargeArgs a, int store_F) { if (!lg_bid(a).ok) return; lg_p2g_body<LANES>(a, store_F, nullptr); }

// grid op over the active cells (:283-313).  to_vel: write the velocity to w.vel (backward) instead of in place
__device__ __forceinline__ void lg_grid_cell(const LargeArgs& a, int b, int t, int to_vel) {
  const int cur = lg_ls(a);
  if (a.ls3 && t < min(a.w.count[a.lprev * a.B + b], a.cap)) {      // the previous substep's cells, in the other grid: g2p has read them
    const long lp = cell_lin(a.c, a.w.list[((long)a.lprev * a.B + b) * a.cap + t]);
    (a.vb ? a.w.val : a.w.val2)[(long)b * a.G + lp] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (t >= min(a.w.count[cur * a.B + b], a.cap)) return;
  const int key = a.w.list[((long)cur * a.B + b) * a.cap + t];
  int ci, cj, ck;
  decode_cell(a.c, key, ci, cj, ck);
  const long lin = ((long)ci * a.c.res[1] + cj) * a.c.res[2] + ck;
  if (a.ls3) a.w.bits[(long)b * a.W32 + (lin >> 5)] = 0u;            // this substep's p2g is done with the bitmap (every bit of the word is this list's)
  const float4 mv = lg_val(a)[(long)b * a.G + lin];
  const float mvv[3] = {mv.y, mv.z, mv.w};
  float vo[3];
  if (a.c.position_control) {
    PrimF pf;
    float pv[3];
    load_prim(a, b, pf, pv);
    grid_op<false>(a.c, pf, ci, cj, ck, mv.x, mvv, vo, nullptr);
  } else {                                                          // collide_batch (primitives.py:154-182)
    float v0[3], v1[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) v0[d] = ((mv.x > 0.f) ? mvv[d] / mv.x : mvv[d]) + a.c.dtg[d];
    const float gp[3] = {(float)ci * a.c.dx, (float)cj * a.c.dx, (float)ck * a.c.dx};
    // collide records for the backward (gck_off_crec): this cell's row sits at the position of its grid-checkpoint record
    float4* crec = nullptr;
    if (!to_vel && a.gck_base && a.gck_off_crec) {
      const int pos = gck_idx(a, b)[a.f] + t;
      if (pos < a.gck_budget) crec = gck_crec(a, b) + (long)pos * a.c.n_prim;
    }
#pragma unroll 1
    for (int ip = 0; ip < a.c.n_prim; ++ip) {                       // primitive after primitive (mpm_simulator.py:292-294)
      PrimC pc;
      load_primc(a, b, ip, pc);
      CollideRec cr;
      collide_cell(pc, a.c.dt, gp, v0, v1, cr);
      if (crec) crec[ip] = make_float4(cr.e, cr.n[0], cr.n[1], cr.n[2]);
#pragma unroll
      for (int d = 0; d < 3; ++d) v0[d] = v1[d];
    }
    grid_tail<false>(a.c, a.friction[b], ci, cj, ck, v1, vo, nullptr);
  }
  if (to_vel) a.w.vel[(long)b * a.G + lin] = make_float4(vo[0], vo[1], vo[2], 0.f);
  else lg_val(a)[(long)b * a.G + lin] = make_float4(mv.x, vo[0], vo[1], vo[2]);
  if (!to_vel && a.gck_base) {
    const int pos = gck_idx(a, b)[a.f] + t;
    if (pos < a.gck_budget) {
      float4* r = gck_pool(a, b) + (long)pos * 2;
      r[0] = make_float4(__builtin_bit_cast(float, key), mv.x, mv.y, mv.z);
      r[1] = make_float4(mv.w, vo[0], vo[1], vo[2]);
    } else if (a.status) {
      atomicOr(&a.status[b], 1);   // (OR: other workgroups flag the same word) pool exhausted: the backward of this env is invalid (UD_ERR_OVERFLOW, reported like the LDS table's)
    }
  }
}
__global__ void __launch_bounds__(256) lg_grid(LargeArgs a, int to_vel) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0;
  int n = min(a.w.count[lg_ls(a) * a.B + b], a.cap);
  if (a.ls3) {
    if (lgb_.x == 0 && threadIdx.x == 0) {
      a.w.count[a.lnext * a.B + b] = 0;                              // the list the next substep's p2g fills (nobody reads it in this launch)
      if (a.gck_base && a.f < a.c.steps) gck_idx(a, b)[a.f + 1] = gck_idx(a, b)[a.f] + n;   // records of substep f + 1 start where these end
    }
    n = max(n, min(a.w.count[a.lprev * a.B + b], a.cap));
    if (a.f >= a.c.steps) {                                           // the launch after the last substep: only the retiring
      for (int u = 0;; ++u) {
        const int t = (u * a.nbx + lgb_.x) * 256 + threadIdx.x;
        if (t - (int)threadIdx.x >= n) break;
        if (t < min(a.w.count[a.lprev * a.B + b], a.cap))
          (a.vb ? a.w.val : a.w.val2)[(long)b * a.G + cell_lin(a.c, a.w.list[((long)a.lprev * a.B + b) * a.cap + t])] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
  }
  for (int u = 0;; ++u) {
    const int base = (u * a.nbx + lgb_.x) * 256;
    if (base >= n) break;
    lg_grid_cell(a, b, base + threadIdx.x, to_vel);
  }
}

// g2p + advect (:196-221, :318-328)
template <int LANES>
__global__ void __launch_bounds__(256) lg_g2p(LargeArgs a) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const MpmConst& c = a.c;
  if (p >= c.N) return;   // whole quads leave together
  const float* hi = a.hist_in + (long)b * a.hist_stride_b;
  float* ho = a.hist_out + (long)b * a.hist_stride_b;
  float x[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) x[d] = hi[d * c.Np + p];
  int base[3];
  float fx[3], w[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    base[d] = (int)(x[d] * c.inv_dx - 0.5f);
    const float f = x[d] * c.inv_dx - (float)base[d];
    fx[d] = f;
    w[d] = 0.5f * ((1.5f - f) * (1.5f - f)); w[3 + d] = 0.75f - (f - 1.f) * (f - 1.f); w[6 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
  }
  const float4* val = lg_val(a) + (long)b * a.G;
  float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (LANES == 1) {   // nine (i, j) columns, the three k cells of a column (neighbours in memory) in flight together -- see lg_g2p_adj
#pragma unroll 1
    for (int col = 0; col < 9; ++col) {
      const int i = col / 3, j = col - 3 * i;
      const float wij = sel3(w, 0, i) * sel3(w, 1, j);
      float4 g4[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) g4[k] = val[cell_lin(c, cell_gather(c, base[0] + i, base[1] + j, base[2] + k))];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float weight = wij * w[k * 3 + 2];
        const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
        const float g[3] = {g4[k].y, g4[k].z, g4[k].w};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
        }
      }
    }
  } else {   // four lanes per particle: the lane's seven cells (qi, qi + 4, ...) requested together, then used
    float4 g7[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int cidx = min(qi + 4 * t, 26);
      g7[t] = val[cell_lin(c, cell_gather(c, base[0] + cidx / 9, base[1] + (cidx / 3) % 3, base[2] + cidx % 3))];
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int cidx = qi + 4 * t;
      if (cidx >= 27) break;
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const float weight = sel3(w, 0, i) * sel3(w, 1, j) * sel3(w, 2, k);
      const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
      const float g[3] = {g7[t].y, g7[t].z, g7[t].w};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        nv[r] += weight * g[r];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) nv[d] = lg_quad_sum<LANES>(nv[d]);
#pragma unroll
  for (int d = 0; d < 9; ++d) nC[d] = lg_quad_sum<LANES>(nC[d]);
  if (qi != 0) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = x[d] + c.dt * nv[d]; ho[(3 + d) * c.Np + p] = nv[d]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) ho[(6 + d) * c.Np + p] = nC[d];
  const int up = user_index(a, b, p);
  if (up < 3) {   // Q6: row p of (the caller's) particle p; three adds per substep on a zero-initialised word
    const float r0 = nC[0] + nC[1] + nC[2], r1 = nC[3] + nC[4] + nC[5], r2 = nC[6] + nC[7] + nC[8];
    atomicAdd(&a.w.trq[(long)b * c.steps + a.f], (up == 0) ? r0 : ((up == 1) ? r1 : r2));
  }
}

// Forward, ls3 mode: g2p of substep f and -- the particle's new x, v, C in registers -- the p2g pass of substep f + 1 in ONE launch
// (hist_out2: where that pass stores F of state f + 2).  The pass fills the other (m, mv) grid and the list slot lnext, both made
// ready by lg_grid(f); its bitmap is clean because lg_grid(f) cleared the bits of substep f's list.
template <int LANES>
__global__ void __launch_bounds__(LG_SCATTER_T) lg_g2p_p2g(LargeArgs a, float* hist_out2) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const MpmConst& c = a.c;
  float reg[15];
#pragma unroll
  for (int d = 0; d < 15; ++d) reg[d] = 0.f;
  if (p < c.N) {   // whole quads together
    const float* hi = a.hist_in + (long)b * a.hist_stride_b;
    float* ho = a.hist_out + (long)b * a.hist_stride_b;
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = hi[d * c.Np + p];
    int base[3];
    float fx[3], w[9];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      base[d] = (int)(x[d] * c.inv_dx - 0.5f);
      const float f = x[d] * c.inv_dx - (float)base[d];
      fx[d] = f;
      w[d] = 0.5f * ((1.5f - f) * (1.5f - f)); w[3 + d] = 0.75f - (f - 1.f) * (f - 1.f); w[6 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
    }
    const float4* val = lg_val(a) + (long)b * a.G;
    float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (LANES == 1) {
#pragma unroll 1
      for (int col = 0; col < 9; ++col) {
        const int i = col / 3, j = col - 3 * i;
        const float wij = sel3(w, 0, i) * sel3(w, 1, j);
        float4 g4[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) g4[k] = val[cell_lin(c, cell_gather(c, base[0] + i, base[1] + j, base[2] + k))];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float weight = wij * w[k * 3 + 2];
          const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
          const float g[3] = {g4[k].y, g4[k].z, g4[k].w};
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            nv[r] += weight * g[r];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
          }
        }
      }
    } else {
      float4 g7[7];
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int cidx = min(qi + 4 * t, 26);
        g7[t] = val[cell_lin(c, cell_gather(c, base[0] + cidx / 9, base[1] + (cidx / 3) % 3, base[2] + cidx % 3))];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int cidx = qi + 4 * t;
        if (cidx >= 27) break;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const float weight = sel3(w, 0, i) * sel3(w, 1, j) * sel3(w, 2, k);
        const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
        const float g[3] = {g7[t].y, g7[t].z, g7[t].w};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) nv[d] = lg_quad_sum<LANES>(nv[d]);     // every lane of the quad holds the sums: they all run the pre-pass
#pragma unroll
    for (int d = 0; d < 9; ++d) nC[d] = lg_quad_sum<LANES>(nC[d]);
#pragma unroll
    for (int d = 0; d < 3; ++d) { reg[d] = x[d] + c.dt * nv[d]; reg[3 + d] = nv[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) reg[6 + d] = nC[d];
    if (qi == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = reg[d]; ho[(3 + d) * c.Np + p] = nv[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) ho[(6 + d) * c.Np + p] = nC[d];
      const int up = user_index(a, b, p);
      if (up < 3) {   // Q6, as lg_g2p
        const float r0 = nC[0] + nC[1] + nC[2], r1 = nC[3] + nC[4] + nC[5], r2 = nC[6] + nC[7] + nC[8];
        atomicAdd(&a.w.trq[(long)b * c.steps + a.f], (up == 0) ? r0 : ((up == 1) ? r1 : r2));
      }
    }
  }
  LargeArgs n = a;                    // the p2g pass of substep f + 1
  n.f = a.f + 1; n.vb = a.vb ^ 1; n.ls = a.lnext;
  n.hist_in = a.hist_out; n.hist_out = hist_out2;
  lg_p2g_body<LANES>(n, 1, reg);
}

// Spatial order for bodies whose particles arrive in no particular order (the uniformly sampled liquid of pour_water: the
// block-level staging of p2g only pays when consecutive particles share cells -- lg_p2g 77 us there against 23 us for a
// lattice-seeded rope).  One workgroup per env: Morton key of each particle's base cell, bitonic sort of (key, index) in LDS,
// perm[slot] = caller's index.  Purely internal: pack / unpack / material lookups / the Q6 trace go through perm, the
// caller sees its own order; only the summation order of the scatters changes.
__device__ __forceinline__ unsigned morton10(unsigned v) {   // spread the low 10 bits: b9..b0 -> bits 27,24,...,0
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
constexpr int LG_SORT_MAX = 8192;   // particles per env the LDS sort holds (64 KB of 8-byte entries)
__global__ void __launch_bounds__(1024) lg_sort(MpmConst c, int b0, const float* x, int* perm, long perm_stride, int npow2) {
  extern __shared__ unsigned long long lg_sk[];
  const int b = blockIdx.x + b0, tid = threadIdx.x;
  for (int i = tid; i < npow2; i += blockDim.x) {
    unsigned long long e = ~0ull;
    if (i < c.N) {
      unsigned key = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float xv = nan_to_num(x[((long)b * c.N + i) * 3 + d]);
        const int cell = min(max((int)(xv * c.inv_dx - 0.5f), 0), 1023);
        key |= morton10((unsigned)cell) << d;
      }
      e = ((unsigned long long)key << 32) | (unsigned)i;
    }
    lg_sk[i] = e;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long ei = lg_sk[i], el = lg_sk[l];
          const bool up = (i & k) == 0;
          if ((ei > el) == up) { lg_sk[i] = el; lg_sk[l] = ei; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < c.N; i += blockDim.x) perm[(long)b * perm_stride + i] = (int)(lg_sk[i] & 0xffffffffu);
}

// AoS boundary <-> SoA history; nan_to_num on the way in (norm_grad_state fwd, :377-381)
__global__ void __launch_bounds__(256) lg_pack(MpmConst c, int b0, const float* x, const float* v, const float* Cm, const float* F,
                                               float* hist, long stride_b, int sanitize, const int* perm, long perm_stride) {
  const int b = blockIdx.y + b0, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.N) return;
  float* h = hist + (long)b * stride_b;
  const int up = perm ? perm[(long)b * perm_stride + p] : p;
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    h[d * c.Np + p] = sanitize ? nan_to_num(x[o3 + d]) : x[o3 + d];
    h[(3 + d) * c.Np + p] = sanitize ? nan_to_num(v[o3 + d]) : v[o3 + d];
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    h[(6 + d) * c.Np + p] = sanitize ? nan_to_num(Cm[o9 + d]) : Cm[o9 + d];
    h[(15 + d) * c.Np + p] = sanitize ? nan_to_num(F[o9 + d]) : F[o9 + d];
  }
}

__global__ void __launch_bounds__(256) lg_unpack(MpmConst c, int b0, const float* hist, long stride_b, float* x, float* v, float* Cm,
                                                 float* F, const int* perm, long perm_stride) {
  const int b = blockIdx.y + b0, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.N) return;
  const float* h = hist + (long)b * stride_b;
  const int up = perm ? perm[(long)b * perm_stride + p] : p;
  const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
  for (int d = 0; d < 3; ++d) { x[o3 + d] = h[d * c.Np + p]; v[o3 + d] = h[(3 + d) * c.Np + p]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[o9 + d] = h[(6 + d) * c.Np + p]; F[o9 + d] = h[(15 + d) * c.Np + p]; }
}

// forward prologue / epilogue for the primitive arrays, J and the outputs set_action writes
__global__ void __launch_bounds__(256) lg_prim_in(LargeArgs a, const float* ppos, const float* prot) {
  const int b = blockIdx.x + a.b0, ip = blockIdx.y, S = a.c.steps;   // grid (B, n_prim)
  const long bp = (long)b * a.c.n_prim + ip;
  for (int e = threadIdx.x; e < S * 3; e += blockDim.x) { a.w.ppos[bp * S * 3 + e] = ppos[bp * S * 3 + e]; a.w.ppin[bp * S * 3 + e] = ppos[bp * S * 3 + e]; }
  for (int e = threadIdx.x; e < S * 4; e += blockDim.x) a.w.prot[bp * S * 4 + e] = prot[bp * S * 4 + e];
  if (ip != 0) return;
  for (int e = threadIdx.x; e < S; e += blockDim.x) a.w.trq[(long)b * S + e] = 0.f;
  if (threadIdx.x < 2) a.w.count[threadIdx.x * a.B + b] = 0;
}

__global__ void __launch_bounds__(256) lg_fwd_out(LargeArgs a, const float* J, float* Jo, float* ppos_o, float* prot_o, float* pv_o,
                                                  float* pw_o, float* ck_tail, long ck_stride_b) {
  const int b = blockIdx.x + a.b0, ip = blockIdx.y, S = a.c.steps, N = a.c.N;   // grid (B, n_prim + blocks of 256 particles)
  if (ip >= a.c.n_prim) {       // J of 256 particles (one block per env walked 7631 of them in 88 us)
    const int p = (ip - a.c.n_prim) * 256 + threadIdx.x;
    if (p < N) {
      float Jp = nan_to_num(J[(long)b * N + p]);
      for (int f = 0; f < S; ++f) Jp = Jp * (1.f + a.c.dt * a.w.trq[(long)b * S + f]);   // :327
      Jo[(long)b * N + p] = Jp;
    }
    return;
  }
  const long bp = (long)b * a.c.n_prim + ip;
  float* tail = ck_tail ? ck_tail + (long)b * ck_stride_b + (long)ip * S * 10 : nullptr;   // per primitive: position, rotation, input position
  for (int e = threadIdx.x; e < S * 3; e += blockDim.x) {
    const int row = e / 3, d = e - row * 3;
    const float* pp = a.w.ppos + bp * S * 3;
    ppos_o[bp * S * 3 + e] = (row == 0) ? pp[(S - 1) * 3 + d] : pp[e];       // copy_frame(steps, 0), Q5
    pv_o[bp * S * 3 + e] = clipf(a.action[bp * 6 + d], -1.f, 1.f) * 1.f / (float)S;
    pw_o[bp * S * 3 + e] = clipf(a.action[bp * 6 + 3 + d], -1.f, 1.f) * 1.f / (float)S;
    if (tail) { tail[e] = pp[e]; tail[S * 7 + e] = a.w.ppin[bp * S * 3 + e]; }
  }
  for (int e = threadIdx.x; e < S * 4; e += blockDim.x) {
    const int row = e / 4, d = e - row * 4;
    const float* pr = a.w.prot + bp * S * 4;
    prot_o[bp * S * 4 + e] = (row == 0) ? pr[(S - 1) * 4 + d] : pr[e];
    if (tail) tail[S * 3 + e] = pr[e];
  }
}

// ---- backward kernels ---------------------------------------------------------------------------------
__device__ __forceinline__ float ppos_preclip_g(const float* pp, const float* pin, int f, int j, int a, float pva) {
  if (j == f + 1) return pp[f * 3 + a] + pva;
  if (j <= f) return (f == 0) ? pin[j * 3 + a] : pp[j * 3 + a];
  return (f == 0) ? pin[j * 3 + a] : clipf(pin[j * 3 + a], -2.f, 2.f);
}

// FK adjoint of substep f (one block per env)
// Runs in the extra blocks of lg_p2g_adj (one 256-thread block per env and primitive): both only need the grid-op adjoint
// of this substep, neither needs the other, and a launch of its own cost 5 us per reverse substep for microseconds of work.
// COH = 1: the cotangent arrays are read and written with agent-scope accesses (the persistent cluster kernel with two barriers per
// substep: other workgroups of the same launch add to them with atomics BETWEEN two calls).  COH = 2 (one barrier per substep: the
// other workgroups' grid-op adjoint of substep f - 1 adds to rows f - 1 and f WHILE this runs): every update of a row those atomics
// may touch is itself an atomic add -- row f receives t, row f + 1 (complete, nobody adds to it any more) is consumed -- and the
// whole-array clip factor, 1 for every coordinate inside (-2, 2), is applied only where it is not 1: exact unless a primitive
// coordinate sits at or beyond +-2 (outside every env's domain), where that multiplication may lose a concurrent add.
template <int COH>
__device__ __forceinline__ void fk_adj_block_f(const LargeArgs& a, long b /* (env, primitive) row of the primitive arrays */, int f) {
  const int S = a.c.steps;
  const float* pp = a.w.ppos + b * S * 3;
  const float* pin = a.w.ppin + b * S * 3;
  float* gp = a.w.gppos + b * S * 3;
  float* gpv = a.w.gpv + b * S * 3;
  for (int e0 = 0; e0 < S * 3; e0 += blockDim.x) {
    const int e = e0 + threadIdx.x;
    float val = 0.f, t = 0.f;
    if (e < S * 3) {
      const int row = e / 3, d = e - row * 3;
      const float pva = clipf(a.action[b * 6 + d], -1.f, 1.f) * 1.f / (float)S;
      val = ld_f<COH>(gp + e) * clip_grad(ppos_preclip_g(pp, pin, f, row, d, pva), -2.f, 2.f);
      if (f + 1 < S) {
        if (row == f + 1) val = 0.f;
        if (row == f) { t = ld_f<COH>(gp + e + 3) * clip_grad(ppos_preclip_g(pp, pin, f, f + 1, d, pva), -2.f, 2.f); val += t; }
      }
    }
    __syncthreads();
    if (COH == 2) {
      if (e < S * 3) {
        const int row = e / 3, d = e - row * 3;
        const float pva = clipf(a.action[b * 6 + d], -1.f, 1.f) * 1.f / (float)S;
        const float cgr = clip_grad(ppos_preclip_g(pp, pin, f, row, d, pva), -2.f, 2.f);
        if (f + 1 < S && row == f + 1) stc(gp + e, 0.f);                                   // consumed (its t went to row f below)
        else if (cgr != 1.f) stc(gp + e, ldc(gp + e) * cgr + t);                           // degenerate: see above
        else if (t != 0.f) atomicAdd(gp + e, t);
        if (t != 0.f) atomicAdd(gpv + e, t);
      }
    } else if (e < S * 3) { st_f<COH>(gp + e, val); if (!COH || t != 0.f) st_f<COH>(gpv + e, ld_f<COH>(gpv + e) + t); }
    __syncthreads();
  }
  // soft contact: rotation' = set(rotation, f+1, qmul(w2quat(w[f]), rotation[f]))  (primitives.py:190, :73-92)
  if (!a.c.position_control && threadIdx.x == 0 && f + 1 < S) {
    float* gr_ = a.w.grot + b * S * 4;
    const float* rr = a.w.prot + b * S * 4 + f * 4;
    float go[4], w[3];
#pragma unroll
    for (int d = 0; d < 4; ++d) { go[d] = ld_f<COH>(gr_ + (f + 1) * 4 + d); st_f<COH>(gr_ + (f + 1) * 4 + d, 0.f); }
#pragma unroll
    for (int d = 0; d < 3; ++d) w[d] = clipf(a.action[b * 6 + 3 + d], -1.f, 1.f) * 1.f / (float)S;
    const float s2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const float nrm = sqrtf(s2), ang = nrm + 1e-12f, hh = ang / 2.f, sn = sinf(hh), cs = cosf(hh);
    const float u[3] = {w[0] / ang, w[1] / ang, w[2] / ang};
    const float q[4] = {cs, u[0] * sn, u[1] * sn, u[2] * sn};
    const float o[4] = {rr[0] * q[0] - rr[1] * q[1] - rr[2] * q[2] - rr[3] * q[3],
                        rr[0] * q[1] + rr[1] * q[0] - rr[2] * q[3] + rr[3] * q[2],
                        rr[0] * q[2] + rr[1] * q[3] + rr[2] * q[0] - rr[3] * q[1],
                        rr[0] * q[3] - rr[1] * q[2] + rr[2] * q[1] + rr[3] * q[0]};
    const float oo = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    const float nn = clipf(oo, 1e-12f, INFINITY);
    const float dot = go[0] * o[0] + go[1] * o[1] + go[2] * o[2] + go[3] * o[3];
    const float goo = -dot / (nn * nn) * clip_grad(oo, 1e-12f, INFINITY);
    float gO[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) gO[d] = go[d] / nn + goo * o[d] / oo;
    add_f<COH>(gr_ + f * 4 + 0, gO[0] * q[0] + gO[1] * q[1] + gO[2] * q[2] + gO[3] * q[3]);
    add_f<COH>(gr_ + f * 4 + 1, -gO[0] * q[1] + gO[1] * q[0] + gO[2] * q[3] - gO[3] * q[2]);
    add_f<COH>(gr_ + f * 4 + 2, -gO[0] * q[2] - gO[1] * q[3] + gO[2] * q[0] + gO[3] * q[1]);
    add_f<COH>(gr_ + f * 4 + 3, -gO[0] * q[3] + gO[1] * q[2] - gO[2] * q[1] + gO[3] * q[0]);
    const float gq[4] = {gO[0] * rr[0] + gO[1] * rr[1] + gO[2] * rr[2] + gO[3] * rr[3],
                         -gO[0] * rr[1] + gO[1] * rr[0] - gO[2] * rr[3] + gO[3] * rr[2],
                         -gO[0] * rr[2] + gO[1] * rr[3] + gO[2] * rr[0] - gO[3] * rr[1],
                         -gO[0] * rr[3] - gO[1] * rr[2] + gO[2] * rr[1] + gO[3] * rr[0]};
    float gh = -sn * gq[0], gsn = 0.f, gang = 0.f, gw[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { gsn += gq[1 + d] * u[d]; const float gu = gq[1 + d] * sn; gw[d] = gu / ang; gang -= gu * w[d] / (ang * ang); }
    gh += cs * gsn;
    gang += gh / 2.f;
    // |w| = sqrt(sum w^2): at w = 0 the reference's chain rule is 0.5/0 * 0 = NaN, laundered by nan_to_num at `step`
    const float gs = gang * (0.5f / nrm);
#pragma unroll
    for (int d = 0; d < 3; ++d) { float* pw_ = a.w.gpw + b * S * 3 + f * 3 + d; st_f<COH>(pw_, ld_f<COH>(pw_) + (gw[d] + gs * (2.f * w[d]))); }   // only this block writes gpw
  }
}
__device__ __forceinline__ void fk_adj_block(const LargeArgs& a, long b) { fk_adj_block_f<0>(a, b, a.f); }

// g2p adjoint: scatter cotangents onto the grid velocity, keep the weight / fx partials per particle
template <int LANES>
__global__ void __launch_bounds__(LG_SCATTER_T) lg_g2p_adj(LargeArgs a) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  constexpr int TH = LgTable<LANES>::H, TLOG = LgTable<LANES>::LOGH;
  const BlockTable bt = bt_make<TH>();
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const MpmConst& c = a.c;
  LG_STAMP_BEGIN
  bt_clear<TH>(bt);
  float4* gacc = lg_gacc(a, a.f) + (long)b * a.G;
  const bool live = p < c.N;
  const float* hi = a.hist_in + (long)b * a.hist_stride_b;
  int base[3] = {0, 0, 0};
  float fx[3], w[9];
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float xd = hi[d * c.Np + p];
      base[d] = (int)(xd * c.inv_dx - 0.5f);
      const float f = xd * c.inv_dx - (float)base[d];
      fx[d] = f;
      w[d] = 0.5f * ((1.5f - f) * (1.5f - f)); w[3 + d] = 0.75f - (f - 1.f) * (f - 1.f); w[6 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
    }
  }
  LG_STAMP(1, 0);     // table clear + position loads
  // the cotangents are loaded ahead of the window reduction: their latency runs under its two barriers
  const float* gs = a.w.gstate + (long)b * 24 * c.Np;
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f}, gC[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) gC[d] = 0.f;
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = gs[d * c.Np + p]; gv[d] = gs[(3 + d) * c.Np + p]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) gC[d] = gs[(6 + d) * c.Np + p];
  }
  const BlockWin win = bt_window(c, live, base);
  LG_STAMP(1, 1);     // window reduction + barriers
  if (live) {
  float gnv[3], gw[9], gfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 3; ++d) gnv[d] = gv[d] + c.dt * gx[d];
#pragma unroll
  for (int d = 0; d < 9; ++d) gw[d] = 0.f;
  const float4* vel = a.w.vel + (long)b * a.G;
  // (Round 2 tried copying the window's 8 x 8 x 8 velocities into LDS once per block and reading the 27 cells from there: slower
  // everywhere -- rope at n_grid 256 backward 11.1 -> 12.5 ms, pour_soup 5.7 -> 6.1 -- the extra barrier and the 512 staged
  // loads cost more than the L2 gathers they replace.)
  if (LANES == 1) {
    // Two walks over the nine (i, j) columns, three k cells -- neighbours in memory, the grid is z-fastest -- per trip.
    // The scatter into the block table is staggered per particle (no two lanes of a run of particles sharing a base cell on the
    // same table slot at once) and needs no grid velocity.  The weight / fx partials need the velocity and want the opposite:
    // every lane on the same column, so that the particles of a cell ask for the same 48 bytes (one cell per trip, staggered,
    // kept a single 16-B gather in flight per wave, 64 different lines per instruction: the walk was 45 % of the kernel,
    // profiles/r02e_lg_stamps_pour_soup.txt).  Measured on pour_soup: 88.7 us as it was, 72.3 with three loads per trip
    // (staggered), 78.2 with the next column prefetched, 87.3 with a whole i plane (nine loads, 190 registers) in flight.
    const int rot9 = p % 9;
    const bool interior = win.on && base[0] >= 0 && base[1] >= 0 && base[2] >= 0 &&
                          base[0] + 2 < c.res[0] && base[1] + 2 < c.res[1] && base[2] + 2 < c.res[2];
    if (interior) {   // as lg_p2g's fast walk: slots by arithmetic, nine (j, k) columns on nine bank pairs, the three i cells unrolled
      const int key0 = base[0] | (base[1] << 10) | (base[2] << 20);
      const int slot0 = (base[2] - win.oz) | ((base[1] - win.oy) << 3) | ((base[0] - win.ox) << 6);
#pragma unroll 1
      for (int it = 0; it < 9; ++it) {
        const int col = it + rot9 >= 9 ? it + rot9 - 9 : it + rot9;
        const int j = col / 3, k = col - 3 * j;
        const float wj = sel3(w, 1, j), wk = sel3(w, 2, k);
        const float dp1 = (float)j - fx[1], dp2 = (float)k - fx[2];
        const int sl = slot0 + 8 * j + k, key = key0 + (j << 10) + (k << 20);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float weight = w[i * 3] * wj * wk;
          const float dp0 = (float)i - fx[0];
          bt.key[sl + 64 * i] = key + i;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float gCd = gC[r * 3] * dp0 + gC[r * 3 + 1] * dp1 + gC[r * 3 + 2] * dp2;
            const float gcell = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
            __hip_atomic_fetch_add(&bt.val[r * TH + sl + 64 * i], (double)gcell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    } else {
#pragma unroll 1
    for (int it = 0; it < 9; ++it) {
      const int col = it + rot9 >= 9 ? it + rot9 - 9 : it + rot9;
      const int i = col / 3, j = col - 3 * i;
      const float wij = sel3(w, 0, i) * sel3(w, 1, j);
      const float dp0 = (float)i - fx[0], dp1 = (float)j - fx[1];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float weight = wij * w[k * 3 + 2];
        const float dp2 = (float)k - fx[2];
        const int gkey = cell_gather(c, base[0] + i, base[1] + j, base[2] + k);
        const int sl = bt_find<TH, TLOG>(bt, win, gkey);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gCd = gC[r * 3] * dp0 + gC[r * 3 + 1] * dp1 + gC[r * 3 + 2] * dp2;
          const float gcell = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
          if (sl >= 0) __hip_atomic_fetch_add(&bt.val[r * TH + sl], (double)gcell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else atomicAdd((float*)(gacc + cell_lin(c, gkey)) + r, gcell);
        }
      }
    }
    }
    LG_STAMP(1, 2);     // scatter walk (one-lane kernel)
#pragma unroll 1
    for (int col = 0; col < 9; ++col) {
      const int i = col / 3, j = col - 3 * i;
      const float wi = sel3(w, 0, i), wj = sel3(w, 1, j);
      const float dp0 = (float)i - fx[0], dp1 = (float)j - fx[1];
      float4 v4[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) v4[k] = vel[cell_lin(c, cell_gather(c, base[0] + i, base[1] + j, base[2] + k))];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float wk = w[k * 3 + 2];
        const float weight = wi * wj * wk;
        const float dp[3] = {dp0, dp1, (float)k - fx[2]};
        const float vv[3] = {v4[k].x, v4[k].y, v4[k].z};
        float gwt = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gCd = gC[r * 3] * dp[0] + gC[r * 3 + 1] * dp[1] + gC[r * 3 + 2] * dp[2];
          gwt += vv[r] * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gC[r * 3 + s2] * vv[r];
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
          gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
        }
        gw[k * 3 + 2] += gwt * wi * wj;
      }
    }
  } else {
  const int rot = (p * LANES) % 27;   // staggered stencil walk, as in lg_p2g: no two lanes of a run on the same table slot
  // four lanes per particle: the lane's seven cells requested together, then used
  int key7[7];
  float4 v7[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int it = min(qi + 4 * t, 26);
    const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
    key7[t] = cell_gather(c, base[0] + cidx / 9, base[1] + (cidx / 3) % 3, base[2] + cidx % 3);
    v7[t] = vel[cell_lin(c, key7[t])];
  }
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int it = qi + 4 * t;
    if (it >= 27) break;
    const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const float wi = sel3(w, 0, i), wj = sel3(w, 1, j), wk = sel3(w, 2, k);
    const float weight = wi * wj * wk;
    const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
    const int gkey = key7[t];
    const long lin = cell_lin(c, gkey);
    const float vv[3] = {v7[t].x, v7[t].y, v7[t].z};
    float gwt = 0.f;
    const int sl = bt_find<TH, TLOG>(bt, win, gkey);      // one lookup per cell (bt_add per component repeated it three times)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gCd = gC[r * 3] * dp[0] + gC[r * 3 + 1] * dp[1] + gC[r * 3 + 2] * dp[2];
      const float gcell = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
      if (sl >= 0) __hip_atomic_fetch_add(&bt.val[r * TH + sl], (double)gcell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else atomicAdd((float*)(gacc + lin) + r, gcell);
      gwt += vv[r] * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gC[r * 3 + s2] * vv[r];
    }
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
      gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
      gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
    }
  }
  }
  LG_STAMP(1, 3);     // the 27-cell walk (velocity gathers from HBM / L2 + table adds)
#pragma unroll
  for (int d = 0; d < 9; ++d) gw[d] = lg_quad_sum<LANES>(gw[d]);
#pragma unroll
  for (int d = 0; d < 3; ++d) gfx[d] = lg_quad_sum<LANES>(gfx[d]);
  if (qi == 0) {
    // What lg_p2g_adj needs of these partials is one number per axis: the weights enter the particle adjoint only through
    // d w / d fx (mpm_device.h, "weights -> fx"), a linear map this kernel can apply itself -- 3 floats per particle to the scratch
    // instead of 12 (and SoA: coalesced), 17.6 MB less per pour_soup substep both ways.
    float* ps = a.w.pscr + (long)b * 3 * c.Np + p;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      ps[d * c.Np] = gfx[d] + gw[0 * 3 + d] * (-(1.5f - fx[d])) + gw[1 * 3 + d] * (-2.f * (fx[d] - 1.f)) + gw[2 * 3 + d] * (fx[d] - 0.5f);
  }
  }
  LG_STAMP(1, 4);     // partials to the particle scratch
  __syncthreads();
  LG_STAMP(1, 5);     // barrier before the flush
  {                   // four lanes per cell as in lg_p2g (the fourth component, g_m, is the grid-op adjoint's)
    const int r = threadIdx.x & 3;
#pragma unroll 4
    for (int sl = threadIdx.x >> 2; sl < TH; sl += LG_SCATTER_T / 4) {
      const int key = bt.key[sl];
      if (key < 0 || r == 3) continue;
      atomicAdd((float*)(gacc + cell_lin(c, key)) + r, (float)bt.val[r * TH + sl]);
    }
  }
  LG_STAMP(1, 6);     // flush
}

// Deterministic backward: the g2p adjoint of one particle per lane with NO accumulation across particles -- what the particle adds to the cell
// its offset (i, j, k) gathers from goes to contrib[27][Np] (x, y, z; w = 0), and mpm_det.hip sums every cell over (offset, particle) in that
// order (det_cells_kernel<1>); the weight / fx partials are the particle's own (27 cells in (i, j, k) order) and go to the scratch as in lg_g2p_adj.
__global__ void __launch_bounds__(256) lg_g2p_adj_det(LargeArgs a, float4* contrib) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0, p = lgb_.x * blockDim.x + threadIdx.x;
  const MpmConst& c = a.c;
  if (p >= c.N) return;
  const float* hi = a.hist_in + (long)b * a.hist_stride_b;
  int base[3];
  float fx[3], w[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float xd = hi[d * c.Np + p];
    base[d] = (int)(xd * c.inv_dx - 0.5f);
    const float f = xd * c.inv_dx - (float)base[d];
    fx[d] = f;
    w[d] = 0.5f * ((1.5f - f) * (1.5f - f)); w[3 + d] = 0.75f - (f - 1.f) * (f - 1.f); w[6 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
  }
  const float* gs = a.w.gstate + (long)b * 24 * c.Np;
  float gnv[3], gC[9], gw[9], gfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 3; ++d) gnv[d] = gs[(3 + d) * c.Np + p] + c.dt * gs[d * c.Np + p];
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[d] = gs[(6 + d) * c.Np + p]; gw[d] = 0.f; }
  const float4* vel = a.w.vel + (long)b * a.G;
  float4* out = contrib + (long)b * 27 * c.Np + p;
#pragma unroll 1
  for (int cidx = 0; cidx < 27; ++cidx) {
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const float wi = sel3(w, 0, i), wj = sel3(w, 1, j), wk = sel3(w, 2, k);
    const float weight = wi * wj * wk;
    const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
    const float4 v4 = vel[cell_lin(c, cell_gather(c, base[0] + i, base[1] + j, base[2] + k))];
    const float vv[3] = {v4.x, v4.y, v4.z};
    float gcell[3], gwt = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gCd = gC[r * 3] * dp[0] + gC[r * 3 + 1] * dp[1] + gC[r * 3 + 2] * dp[2];
      gcell[r] = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
      gwt += vv[r] * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gC[r * 3 + s2] * vv[r];
    }
    out[(long)cidx * c.Np] = make_float4(gcell[0], gcell[1], gcell[2], 0.f);
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
      gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
      gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
    }
  }
  float* ps = a.w.pscr + (long)b * 3 * c.Np + p;
#pragma unroll
  for (int d = 0; d < 3; ++d)
    ps[d * c.Np] = gfx[d] + gw[0 * 3 + d] * (-(1.5f - fx[d])) + gw[1 * 3 + d] * (-2.f * (fx[d] - 1.f)) + gw[2 * 3 + d] * (fx[d] - 0.5f);
}

// grid checkpoint -> dense arrays of the backward for substep f: velocity after the grid op, zeroed cotangent, the cell list.
// It also zeroes the cotangent cells of substep f + 1 (just consumed by its p2g adjoint), and a last call with f = -1 does
// only that for substep 0: the handle's gacc grid is all-zero again afterwards, which the recomputing backward relies on
// (a handle may serve both modes: a step whose pool overflowed falls back to recomputing).
__device__ __forceinline__ void lg_restore_tile(const LargeArgs& a, int b, int t) {
  const int S = a.c.steps;
  const int* idx = gck_idx(a, b);
  if (a.f + 1 < S) {                                  // cells of substep f + 1: done with
    const int first = idx[a.f + 1], n = min(min(idx[a.f + 2], a.gck_budget) - first, a.cap);
    if (t < n) {
      const int key = __builtin_bit_cast(int, gck_pool(a, b)[(long)(first + t) * 2].x);
      a.w.gacc[(long)b * a.G + cell_lin(a.c, key)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (a.f < 0) return;
  const int cur = a.f & 1;
  const int first = idx[a.f], n = min(min(idx[a.f + 1], a.gck_budget) - first, a.cap);
  if (t == 0) a.w.count[cur * a.B + b] = max(n, 0);
  if (t >= n) return;
  const float4* r = gck_pool(a, b) + (long)(first + t) * 2;
  const float4 r0 = r[0], r1 = r[1];
  const int key = __builtin_bit_cast(int, r0.x);
  const long lin = cell_lin(a.c, key);
  a.w.list[((long)cur * a.B + b) * a.cap + t] = key;
  a.w.vel[(long)b * a.G + lin] = make_float4(r1.y, r1.z, r1.w, 0.f);
  a.w.gacc[(long)b * a.G + lin] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void __launch_bounds__(256) lg_restore(LargeArgs a) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  const int b = lgb_.y + a.b0, S = a.c.steps;
  const int* idx = gck_idx(a, b);                   // the longer of the two lists this launch walks (records of f + 1 and of f)
  int n = 1;                                        // tile 0 always runs: it publishes the count
  if (a.f + 1 < S) n = max(n, min(idx[a.f + 2], a.gck_budget) - idx[a.f + 1]);
  if (a.f >= 0) n = max(n, min(idx[a.f + 1], a.gck_budget) - idx[a.f]);
  n = min(n, a.cap);
  for (int u = 0;; ++u) {
    const int base = (u * a.nbx + lgb_.x) * 256;
    if (base >= n) break;
    lg_restore_tile(a, b, base + threadIdx.x);
  }
}

// (m, mv) of active cell t of substep f: from the grid checkpoint when there is one, else from the recomputed dense grid
__device__ __forceinline__ float4 cell_mass_momentum(const LargeArgs& a, int b, int t, long lin) {
  if (a.gck_base) {
    const float4* r = gck_pool(a, b) + (long)(gck_idx(a, b)[a.f] + t) * 2;
    const float4 r0 = r[0], r1 = r[1];
    return make_float4(r0.y, r0.z, r0.w, r1.x);
  }
  return a.w.val[(long)b * a.G + lin];
}

// grid-op adjoint over the active cells
// DET (a template parameter, so that the default kernels' code and registers stay what they were: 163 VGPRs = three waves per SIMD; with the
// branch at run time the kernel took 181 and pour_water's grid-op adjoint 29 us instead of 20): the deterministic backward's variant.
// REC (a template parameter for the same reason): the forward left (e, n) of every cell and primitive beside the grid checkpoint
// (LargeArgs::gck_off_crec) -- the collide chain is re-run from those instead of from the SDFs.
template <bool DET, bool REC = false>
__device__ __forceinline__ void lg_grid_adj_tile(const LargeArgs& a, int b, int tile_base, float (*red)[UD_PRIMC_NGRAD]) {
  const int t = tile_base + threadIdx.x;
  const int cur = lg_bslot(a, a.f);
  const bool live = t < min(a.w.count[cur * a.B + b], a.cap);
  if (a.c.position_control) {
    // The friction and controlled-velocity cotangents are per ENV: one atomic per cell put every ground-layer cell of an env on
    // one word (the envs' words share four cache lines) and the memory side serialises them -- 11 us for a launch with 3 us of
    // work on the rope at n_grid 128.  Summed over the wave first: one atomic per wave and word.
    if (tile_base >= min(a.w.count[cur * a.B + b], a.cap)) return;   // block-uniform
    float dfric = 0.f, dpv[3] = {0.f, 0.f, 0.f};
    if (live) {
      const int key = a.w.list[((long)cur * a.B + b) * a.cap + t];
      int ci, cj, ck;
      decode_cell(a.c, key, ci, cj, ck);
      const long lin = ((long)ci * a.c.res[1] + cj) * a.c.res[2] + ck;
      const float4 mv = cell_mass_momentum(a, b, t, lin);
      const float mvv[3] = {mv.y, mv.z, mv.w};
      const float4 g4 = lg_gacc(a, a.f)[(long)b * a.G + lin];
      float g[3] = {g4.x, g4.y, g4.z}, gmm, pv[3], dp[3];
      PrimF pf;
      load_prim(a, b, pf, pv);
      if (grid_op_adjoint(a.c, pf, ci, cj, ck, mv.x, mvv, g, gmm, dfric, dp)) { dpv[0] = dp[0]; dpv[1] = dp[1]; dpv[2] = dp[2]; }
      lg_gacc(a, a.f)[(long)b * a.G + lin] = make_float4(g[0], g[1], g[2], gmm);
    }
    if (DET) {                    // deterministic backward: per cell, summed in a fixed order by det_reduce_cells_kernel
      if (live && t < a.det_capc) {
        float* cr = a.det_cellred + ((long)b * a.det_capc + t) * 4;
        cr[0] = dfric; cr[1] = dpv[0]; cr[2] = dpv[1]; cr[3] = dpv[2];
      }
      return;
    }
    const float sf = wave_sum(dfric), s0 = wave_sum(dpv[0]), s1 = wave_sum(dpv[1]), s2 = wave_sum(dpv[2]);
    if ((threadIdx.x & 63) == 0) {
      if (sf != 0.f) atomicAdd(&a.w.acc[b * 4 + 0], sf);
      float* gp = a.w.gpv + (long)b * a.c.steps * 3 + a.f * 3;
      if (s0 != 0.f) atomicAdd(gp + 0, s0);
      if (s1 != 0.f) atomicAdd(gp + 1, s1);
      if (s2 != 0.f) atomicAdd(gp + 2, s2);
    }
    return;
  }
  // ---- soft contact: collide_batch of each primitive in turn (forward), reversed here --------------------------------
  // The grid covers `cap` cells per env, most blocks hold no active cell: leave block-uniformly (safe for the barriers
  // below) before any of the collide arithmetic.
  if (tile_base >= min(a.w.count[cur * a.B + b], a.cap)) return;
  LG_STAMP_BEGIN
  const int P = a.c.n_prim, S = a.c.steps, f0 = min(max(a.f, 0), S - 1), f1 = min(max(a.f + 1, 0), S - 1);
  int ci = 0, cj = 0, ck = 0;
  long lin = 0;
  float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
  float g[3] = {0.f, 0.f, 0.f}, gp[3] = {0.f, 0.f, 0.f}, dfric_cell = 0.f;   // dfric_cell: this cell's ground-friction cotangent
  if (live) {
    const int key = a.w.list[((long)cur * a.B + b) * a.cap + t];
    decode_cell(a.c, key, ci, cj, ck);
    lin = ((long)ci * a.c.res[1] + cj) * a.c.res[2] + ck;
    mv = cell_mass_momentum(a, b, t, lin);
    const float4 g4 = lg_gacc(a, a.f)[(long)b * a.G + lin];
    g[0] = g4.x; g[1] = g4.y; g[2] = g4.z;
    gp[0] = (float)ci * a.c.dx; gp[1] = (float)cj * a.c.dx; gp[2] = (float)ck * a.c.dx;
  }
  const float mvv[3] = {mv.y, mv.z, mv.w};
  float v0[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) v0[d] = ((mv.x > 0.f) ? mvv[d] / mv.x : mvv[d]) + a.c.dtg[d];
  LG_STAMP(3, 0);     // list / checkpoint / cotangent loads
  // Reverse walk over the primitives.  Primitive ip's input velocity is recomputed by running the chain 0..ip again
  // (one extra collide for two primitives) -- loops are kept rolled: one copy of the collide code in the kernel.
#pragma unroll 1
  for (int ip = P - 1; ip >= 0; --ip) {
    float pgv[UD_PRIMC_NGRAD];
#pragma unroll
    for (int d = 0; d < UD_PRIMC_NGRAD; ++d) pgv[d] = 0.f;
    if (live) {                  // lanes past the env's active list only take part in the reductions below
      PrimC pc;
      CollideRec cr;
      float vi[3] = {v0[0], v0[1], v0[2]}, v1[3];
#pragma unroll 1
      for (int j = 0; j <= ip; ++j) {
        load_primc(a, b, j, pc);
        if (REC) {
          const float4 r4 = gck_crec(a, b)[(long)(gck_idx(a, b)[a.f] + t) * P + j];
          const float rec[4] = {r4.x, r4.y, r4.z, r4.w};
          collide_cell(pc, a.c.dt, gp, vi, v1, cr, rec);
        } else {
          collide_cell(pc, a.c.dt, gp, vi, v1, cr);
        }
        if (j < ip) { vi[0] = v1[0]; vi[1] = v1[1]; vi[2] = v1[2]; }
      }
      if (ip == P - 1) {         // v1 = velocity after the last primitive: ground friction + boundary, reversed first
        CellRec rec;
        float vo[3], dfric;
        grid_tail<true>(a.c, a.friction[b], ci, cj, ck, v1, vo, &rec);
        grid_tail_adjoint(a.friction[b], ci, cj, ck, rec, g, dfric);
        dfric_cell = dfric;
      }
      PrimCGrad pg;
      float gin[3];
      collide_cell_bwd(pc, a.c.dt, cr, g, gin, pg);
#pragma unroll
      for (int d = 0; d < 3; ++d) { g[d] = gin[d]; pgv[d] = pg.p0[d]; pgv[7 + d] = pg.p1[d]; pgv[14 + d] = pg.size[d]; }
#pragma unroll
      for (int d = 0; d < 4; ++d) { pgv[3 + d] = pg.r0[d]; pgv[10 + d] = pg.r1[d]; }
      pgv[17] = pg.mu;
    }
    LG_STAMP(3, 1);   // collide chain forward + this primitive's adjoint
    if (DET) {                    // deterministic backward: the cell's 18 values to its row, summed in a fixed order by det_reduce_cells_kernel
      if (live && t < a.det_capc) {
        float* cr = a.det_cellred + ((long)b * a.det_capc + t) * a.det_K + 1 + ip * UD_PRIMC_NGRAD;
#pragma unroll
        for (int d = 0; d < UD_PRIMC_NGRAD; ++d) cr[d] = pgv[d];
      }
      continue;                   // (block-uniform)
    }
    // this primitive's cotangents: wave sums, then one set of atomics per block onto rows f and f + 1 (clamped)
    {   // 16 of the 18 values in two transposed butterflies (eight wave totals per pass: common.h), the last two one by one
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      const float va[8] = {pgv[0], pgv[1], pgv[2], pgv[3], pgv[4], pgv[5], pgv[6], pgv[7]};
      const float vb[8] = {pgv[8], pgv[9], pgv[10], pgv[11], pgv[12], pgv[13], pgv[14], pgv[15]};
      const float wa = wave_sum8_t(va, lane), wb = wave_sum8_t(vb, lane);
      const float s16 = wave_sum(pgv[16]), s17 = wave_sum(pgv[17]);
      if ((lane & 0x2C) == 0) {                       // lanes 0-3 and 16-19 hold the totals of values 0-3 and 4-7 of a pass
        const int j = ((lane >> 2) & 4) | (lane & 3);
        red[wv][j] = wa; red[wv][8 + j] = wb;
      }
      if (lane == 0) { red[wv][16] = s16; red[wv][17] = s17; }
    }
    __syncthreads();
    if (threadIdx.x < UD_PRIMC_NGRAD) {
      const int d = threadIdx.x;
      const long bp = (long)b * P + ip;
      const float tot = red[0][d] + red[1][d] + red[2][d] + red[3][d];
      float* dst = (d < 3)    ? a.w.gppos + bp * S * 3 + f0 * 3 + d
                   : (d < 7)  ? a.w.grot + bp * S * 4 + f0 * 4 + (d - 3)
                   : (d < 10) ? a.w.gppos + bp * S * 3 + f1 * 3 + (d - 7)
                   : (d < 14) ? a.w.grot + bp * S * 4 + f1 * 4 + (d - 10)
                              : a.w.gpsz + bp * 4 + (d - 14);
      if (tot != 0.f) atomicAdd(dst, tot);
    }
    __syncthreads();
    LG_STAMP(3, 2);   // wave sums, barriers, the block's atomics
  }
  if (DET) {
    if (live && t < a.det_capc) a.det_cellred[((long)b * a.det_capc + t) * a.det_K] = dfric_cell;
  } else {           // one atomic per wave onto the env's friction word (see the position-control branch)
    const float sf = wave_sum(dfric_cell);
    if ((threadIdx.x & 63) == 0 && sf != 0.f) atomicAdd(&a.w.acc[b * 4 + 0], sf);
  }
  if (live) {
    float gmm;
    grid_head_adjoint(mv.x, mvv, g, gmm);
    lg_gacc(a, a.f)[(long)b * a.G + lin] = make_float4(g[0], g[1], g[2], gmm);
  }
  LG_STAMP(3, 3);     // head adjoint + store
}
template <bool DET, bool REC = false>
__device__ __forceinline__ void lg_grid_adj_body(const LargeArgs& a) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  __shared__ float red[4][UD_PRIMC_NGRAD];
  const int b = lgb_.y + a.b0, n = min(a.w.count[lg_bslot(a, a.f) * a.B + b], a.cap);
  for (int u = 0;; ++u) {          // block-uniform trip count: the tiles' reductions hold barriers
    const int base = (u * a.nbx + lgb_.x) * 256;
    if (base >= n) break;
    lg_grid_adj_tile<DET, REC>(a, b, base, red);
  }
}
// amdgpu_waves_per_eu(3): at most 168 VGPRs.  These kernels run a few thousand dependent instructions per cell on launches of ~2 waves per
// SIMD; at 169+ registers a SIMD holds two waves instead of three and pour_water's grid-op adjoint takes 29 us instead of 20 (round 4's "cliff")
#define LG_W3 __attribute__((amdgpu_waves_per_eu(3)))
__global__ void __launch_bounds__(256) LG_W3 lg_grid_adj(LargeArgs a) { lg_grid_adj_body<false>(a); }
__global__ void __launch_bounds__(256) LG_W3 lg_grid_adj_rec(LargeArgs a) { lg_grid_adj_body<false, true>(a); }
__global__ void __launch_bounds__(256) lg_grid_adj_det(LargeArgs a) { lg_grid_adj_body<true>(a); }

// p2g adjoint (gather) + particle pre-pass adjoint: cotangent state at substep f+1 -> at substep f (in place)
template <int LANES>
__global__ void __launch_bounds__(256, 2) lg_p2g_adj(LargeArgs a, int particle_blocks) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  if (lgb_.x >= particle_blocks) {   // the last n_prim blocks of each env: FK adjoint of this substep
    fk_adj_block(a, (long)(lgb_.y + a.b0) * a.c.n_prim + (lgb_.x - particle_blocks));
    return;
  }
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid / LANES, qi = gid % LANES;
  const MpmConst& c = a.c;
  // mu / lamda cotangents: summed per block in LDS, ONE global atomic per block and parameter.  (One per particle -- 7631
  // same-address atomics per env and substep on pour_soup -- serialises at the memory side: round 2 measurement below.)
  __shared__ float s_par[2];
  if (threadIdx.x < 2) s_par[threadIdx.x] = 0.f;
  __syncthreads();
  LG_STAMP_BEGIN
  if (p < c.N) {   // whole quads leave together
  float x[3], v[3], Cm[9], F[9];
  load_state(a.hist_in + (long)b * a.hist_stride_b, c.Np, p, x, v, Cm, F);
  Pre q;
  PreB kb;
  const int up = user_index(a, b, p);
  const int material = a.material[up];
  LG_STAMP(2, 0);     // state loads
  particle_pre<true, true>(c, x, Cm, F, a.mu[b], a.lamda[b], material, a.hard[up], q, &kb, nullptr,
                     a.svd_rows ? a.hist_in + (long)b * a.hist_stride_b + (long)24 * c.Np + p : nullptr, c.Np);
  LG_STAMP(2, 1);     // pre-pass with the adjoint's extras
  float* gs = a.w.gstate + (long)b * 24 * c.Np;
  float gx[3], gv[3], gC[9], gF[9];
  // one lane per particle: loaded after the gather instead -- 24 registers the walk's loads in flight can use (243 VGPRs, two
  // waves per SIMD; ahead of it the kernel needs 278 and drops to one)
  // Of the cotangent of state f + 1 only x's and F's rows are read: v_{f+1} and C_{f+1} are outputs of g2p, their cotangents were consumed
  // by the g2p adjoint of this substep (the previous launch); the particle adjoint overwrites gv, gC (mpm_device.h::particle_adjoint).
  // 12 of the 24 rows: a quarter of this kernel's bytes where the chip is full (rope at n_grid 256).
  if (LANES != 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = gs[d * c.Np + p]; gv[d] = 0.f; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = 0.f; gF[d] = gs[(15 + d) * c.Np + p]; }
  }
  const float* ps = a.w.pscr + (long)b * 3 * c.Np + p;
  float gw[9], gfx[3], gaff[9], gvp[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 9; ++d) { gw[d] = 0.f; gaff[d] = 0.f; }
#pragma unroll
  for (int d = 0; d < 3; ++d) gfx[d] = (qi == 0) ? ps[d * c.Np] : 0.f;   // the g2p adjoint's share (already through d w / d fx) enters the quad sum once
  const float4* gacc = lg_gacc(a, a.f) + (long)b * a.G;
  LG_STAMP(2, 2);     // cotangent + scratch loads
  // One lane per particle: the 27 cells as nine (i, j) columns, the three k cells of a column -- neighbours in memory, the grid is
  // z-fastest -- loaded together before any of them is used (one cell per trip left a single 16-B gather in flight per wave, at
  // two waves per SIMD: the walk was 63 % of this kernel, profiles/r02e_lg_stamps_pour_soup.txt).  (A whole i plane, nine loads,
  // in flight needs 67 more registers than two waves per SIMD have: spilled it gained 4 us of 52 on pour_soup; parking the SVD
  // factors in LDS across the gather instead (48 us) moved gF by 14 % on a test: in hindsight the schedule dependence of the SVD
  // cotangent as it was then evaluated (particle_adjoint, fixed since), not the park; not retried.)
  if (LANES == 1) {
#pragma unroll 1
    for (int col = 0; col < 9; ++col) {
      const int i = col / 3, j = col - 3 * i;
      const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j);
      const float dp0 = ((float)i - q.fx[0]) * c.dx, dp1 = ((float)j - q.fx[1]) * c.dx;
      int sc[3];
      float4 g4[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) sc[k] = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) g4[k] = sc[k] >= 0 ? gacc[cell_lin(c, sc[k])] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (sc[k] < 0) continue;
        const float wk = q.w[k * 3 + 2];
        const float weight = wi * wj * wk;
        const float dpos[3] = {dp0, dp1, ((float)k - q.fx[2]) * c.dx};
        const float gcv[3] = {g4[k].x, g4[k].y, g4[k].z};
        float gwt = c.p_mass * g4[k].w;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float ad = q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2];
          gwt += gcv[r] * (c.p_mass * v[r] + ad);
          gvp[r] += weight * c.p_mass * gcv[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) {
            gaff[r * 3 + s2] += weight * gcv[r] * dpos[s2];
            gfx[s2] -= c.dx * weight * gcv[r] * q.affine[r * 3 + s2];
          }
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
          gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
        }
        gw[k * 3 + 2] += gwt * wi * wj;
      }
    }
  } else {
#pragma unroll 1
  for (int cidx = qi; cidx < 27; cidx += LANES) {
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const int sc = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
    if (sc < 0) continue;
    const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
    const float weight = wi * wj * wk;
    const float dpos[3] = {((float)i - q.fx[0]) * c.dx, ((float)j - q.fx[1]) * c.dx, ((float)k - q.fx[2]) * c.dx};
    const float4 g4 = gacc[cell_lin(c, sc)];
    const float gcv[3] = {g4.x, g4.y, g4.z};
    float gwt = c.p_mass * g4.w;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float ad = q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2];
      gwt += gcv[r] * (c.p_mass * v[r] + ad);
      gvp[r] += weight * c.p_mass * gcv[r];
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        gaff[r * 3 + s2] += weight * gcv[r] * dpos[s2];
        gfx[s2] -= c.dx * weight * gcv[r] * q.affine[r * 3 + s2];
      }
    }
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
      gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
      gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
    }
  }
  }
#pragma unroll
  for (int d = 0; d < 9; ++d) { gw[d] = lg_quad_sum<LANES>(gw[d]); gaff[d] = lg_quad_sum<LANES>(gaff[d]); }
#pragma unroll
  for (int d = 0; d < 3; ++d) { gfx[d] = lg_quad_sum<LANES>(gfx[d]); gvp[d] = lg_quad_sum<LANES>(gvp[d]); }
  LG_STAMP(2, 3);     // the 27-cell gather
  if (qi == 0) {
  if (LANES == 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = gs[d * c.Np + p]; gv[d] = 0.f; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = 0.f; gF[d] = gs[(15 + d) * c.Np + p]; }
    // Fn is formed again here (as particle_pre forms it) instead of being held across the gather: nine registers fewer there
    if (material == 2) {
      float US[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) US[i * 3 + j] = kb.U[i * 3 + j] * kb.sig[j];
      m_mul(US, kb.Vh, q.Fn);
    } else {
      float IC[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) IC[i] = ((i % 4 == 0) ? 1.f : 0.f) + c.dt * Cm[i];
      m_mul(IC, F, q.Fn);
    }
  }
  float gmu_p, gla_p;
  particle_adjoint<true>(c, q, kb, Cm, F, material, gw, gfx, gaff, gvp, gx, gv, gC, gF, gmu_p, gla_p);
  if (material != 0) {
    const float h = clipf(a.hard[up], 0.1f, 5.f);
    if (a.det_pacc) {            // deterministic backward: the particle's own running sums, reduced in a fixed order at the end of the step
      float* pa = a.det_pacc + (long)b * 2 * c.Np;
      pa[p] += gmu_p * h; pa[c.Np + p] += gla_p * h;
    } else {
      atomicAdd(&s_par[0], gmu_p * h);
      atomicAdd(&s_par[1], gla_p * h);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { gs[d * c.Np + p] = gx[d]; gs[(3 + d) * c.Np + p] = gv[d]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { gs[(6 + d) * c.Np + p] = gC[d]; gs[(15 + d) * c.Np + p] = gF[d]; }
  }
  LG_STAMP(2, 4);     // particle adjoint (stress, SVD VJP) + stores
  }
  __syncthreads();
  if (threadIdx.x < 2 && s_par[threadIdx.x] != 0.f) atomicAdd(&a.w.acc[b * 4 + 1 + threadIdx.x], s_par[threadIdx.x]);
}

// ---- backward with the grid checkpoint, four lanes per particle: TWO launches per reverse substep instead of four ------------
// The four kernels (restore, g2p adjoint, grid-op adjoint, p2g adjoint) sit at 4-14 us each on 37 k lanes: launch ramps and
// chains of dependent round trips, not work (profiles/r02j_kernel_stats_shape_rope.csv).  Two of the hand-overs are per PARTICLE
// (the p2g adjoint of substep f produces the cotangent state that the g2p adjoint of substep f - 1 scatters) and two per CELL
// (the grid-op adjoint of f and the restore of f - 1 touch different lists), so:
//   K1(f) = lg_gadj_restore : grid-op adjoint of substep f  ||  restore of substep f - 1 (other blocks of the same launch)
//   K2(f) = lg_padj_gadj    : p2g adjoint + particle adjoint of f, then -- cotangent state in registers -- g2p adjoint of f - 1
// The cotangent grids of consecutive substeps must then be different arrays (K2 gathers f's while it scatters f - 1's):
// a.gpar, even substeps in w.gacc, odd ones in w.val (which this backward does not touch otherwise).  Buffer f & 1 was last
// used by substep f + 2: the restore of f also zeroes those cells (consumed by K2(f + 2)), and two trailing calls (f = -1, -2)
// leave both grids all-zero, the invariant every other path of the handle relies on.
__device__ __forceinline__ void lg_restore_par_tile(const LargeArgs& a, int b, int f, int t) {
  const int S = a.c.steps;
  const int* idx = gck_idx(a, b);
  if (f + 2 < S) {                                  // cells of substep f + 2, same buffer: consumed by its p2g adjoint
    const int first = idx[f + 2], n = min(min(idx[f + 3], a.gck_budget) - first, a.cap);
    if (t < n) {
      const int key = a.w.list[((long)lg_bslot(a, f + 2) * a.B + b) * a.cap + t];      // written by the restore of f + 2 (three launches ago), slot not reused since
      lg_gacc(a, f)[(long)b * a.G + cell_lin(a.c, key)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (f < 0) return;
  const int cur = lg_bslot(a, f);
  const int first = idx[f], n = min(min(idx[f + 1], a.gck_budget) - first, a.cap);
  if (t == 0) a.w.count[cur * a.B + b] = max(n, 0);
  if (t >= n) return;
  const float4* r = gck_pool(a, b) + (long)(first + t) * 2;
  const float4 r0 = r[0], r1 = r[1];
  const int key = __builtin_bit_cast(int, r0.x);
  const long lin = cell_lin(a.c, key);
  a.w.list[((long)cur * a.B + b) * a.cap + t] = key;
  a.w.vel[(long)b * a.G + lin] = make_float4(r1.y, r1.z, r1.w, 0.f);   // (its cotangent cells are zero already: the buffer's last user was f + 2)
}
__device__ __forceinline__ void lg_restore_par(const LargeArgs& a, int b, int f, int nblocks, int block) {
  const int S = a.c.steps;
  const int* idx = gck_idx(a, b);
  int n = 1;                                        // tile 0 always runs: it publishes the count
  if (f + 2 < S) n = max(n, min(idx[f + 3], a.gck_budget) - idx[f + 2]);
  if (f >= 0) n = max(n, min(idx[f + 1], a.gck_budget) - idx[f]);
  n = min(n, a.cap);
  for (int u = 0;; ++u) {
    const int base = (u * nblocks + block) * 256;
    if (base >= n) break;
    lg_restore_par_tile(a, b, f, base + threadIdx.x);
  }
}
// blocks [0, nb): grid-op adjoint of substep a.f (none when a.f is not a substep: the first and the last launch of a step);
// blocks [nb, 2 nb): restore of substep a.f - 1
__global__ void __launch_bounds__(256) LG_W3 lg_gadj_restore(LargeArgs a, int nb) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  __shared__ float red[4][UD_PRIMC_NGRAD];
  const int b = lgb_.y + a.b0;
  if (lgb_.x < nb) {
    if (a.f < 0 || a.f >= a.c.steps) return;
    const int n = min(a.w.count[lg_bslot(a, a.f) * a.B + b], a.cap);
    for (int u = 0;; ++u) {          // block-uniform trip count: the tiles' reductions hold barriers
      const int base = (u * nb + lgb_.x) * 256;
      if (base >= n) break;
      lg_grid_adj_tile<false>(a, b, base, red);
    }
    return;
  }
  lg_restore_par(a, b, a.f - 1, nb, lgb_.x - nb);
}

// K2: p2g adjoint + particle adjoint of substep a.f, then the g2p adjoint of substep a.f - 1 (hist_prev = its input state; nullptr:
// none, the last launch).  do_a == 0 (the first launch): only the g2p adjoint, of substep a.f - 1 = S - 1, cotangents from w.gstate.
__global__ void __launch_bounds__(LG_SCATTER_T) __attribute__((amdgpu_waves_per_eu(2))) lg_padj_gadj(LargeArgs a, int particle_blocks, int do_a, const float* hist_prev) {
  const LgB lgb_ = lg_bid(a);
  if (!lgb_.ok) return;
  if (lgb_.x >= particle_blocks) {   // the last n_prim blocks of each env: FK adjoint of this substep
    if (do_a) fk_adj_block(a, (long)(lgb_.y + a.b0) * a.c.n_prim + (lgb_.x - particle_blocks));
    return;
  }
  constexpr int TH = LgTable<4>::H, TLOG = LgTable<4>::LOGH;
  const BlockTable bt = bt_make<TH>();
  const int b = lgb_.y + a.b0, gid = lgb_.x * blockDim.x + threadIdx.x, p = gid >> 2, qi = gid & 3;
  const MpmConst& c = a.c;
  __shared__ float s_par[2];
  if (threadIdx.x < 2) s_par[threadIdx.x] = 0.f;
  if (hist_prev) bt_clear<TH>(bt);
  __syncthreads();
  const bool live = p < c.N;
  // What crosses a launch boundary of the cotangent state: x's and F's rows only.  v_{f+1} and C_{f+1} are outputs of g2p: their cotangents
  // are consumed by part B (the g2p adjoint) of the launch that produced them, from registers; the particle adjoint overwrites them.  So a
  // launch that runs part A reads 12 of the 24 rows and, while more launches follow, writes those 12; the first launch (B only) reads x, v,
  // C (15 rows), the last one (no B) writes all 24 for lg_bwd_norm / lg_bwd_out.
  float* gs = a.w.gstate + (long)b * 24 * c.Np;
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f}, gC[9], gF[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[d] = 0.f; gF[d] = 0.f; }
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) gx[d] = gs[d * c.Np + p];
    if (do_a) {
#pragma unroll
      for (int d = 0; d < 9; ++d) gF[d] = gs[(15 + d) * c.Np + p];
    } else {
#pragma unroll
      for (int d = 0; d < 3; ++d) gv[d] = gs[(3 + d) * c.Np + p];
#pragma unroll
      for (int d = 0; d < 9; ++d) gC[d] = gs[(6 + d) * c.Np + p];
    }
  }
  // ---- A: p2g adjoint (gather) + particle adjoint of substep f: cotangent state at f + 1 -> at f ----
  if (do_a && live) {
    float x[3], v[3], Cm[9], F[9];
    load_state(a.hist_in + (long)b * a.hist_stride_b, c.Np, p, x, v, Cm, F);
    Pre q;
    PreB kb;
    const int up = user_index(a, b, p);
    const int material = a.material[up];
    particle_pre<true, true>(c, x, Cm, F, a.mu[b], a.lamda[b], material, a.hard[up], q, &kb, nullptr,
                       a.svd_rows ? a.hist_in + (long)b * a.hist_stride_b + (long)24 * c.Np + p : nullptr, c.Np);
    const float* ps = a.w.pscr + (long)b * 3 * c.Np + p;
    float gw[9], gfx[3], gaff[9], gvp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 9; ++d) { gw[d] = 0.f; gaff[d] = 0.f; }
#pragma unroll
    for (int d = 0; d < 3; ++d) gfx[d] = (qi == 0) ? ps[d * c.Np] : 0.f;   // the g2p adjoint's share (already through d w / d fx) enters the quad sum once
    const float4* gacc = lg_gacc(a, a.f) + (long)b * a.G;
#pragma unroll 1
    for (int cidx = qi; cidx < 27; cidx += 4) {
      const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
      const int sc = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
      if (sc < 0) continue;
      const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
      const float weight = wi * wj * wk;
      const float dpos[3] = {((float)i - q.fx[0]) * c.dx, ((float)j - q.fx[1]) * c.dx, ((float)k - q.fx[2]) * c.dx};
      const float4 g4 = gacc[cell_lin(c, sc)];
      const float gcv[3] = {g4.x, g4.y, g4.z};
      float gwt = c.p_mass * g4.w;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float ad = q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2];
        gwt += gcv[r] * (c.p_mass * v[r] + ad);
        gvp[r] += weight * c.p_mass * gcv[r];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          gaff[r * 3 + s2] += weight * gcv[r] * dpos[s2];
          gfx[s2] -= c.dx * weight * gcv[r] * q.affine[r * 3 + s2];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
        gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
        gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
      }
    }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gw[d] = lg_quad_sum<4>(gw[d]); gaff[d] = lg_quad_sum<4>(gaff[d]); }
#pragma unroll
    for (int d = 0; d < 3; ++d) { gfx[d] = lg_quad_sum<4>(gfx[d]); gvp[d] = lg_quad_sum<4>(gvp[d]); }
    float gmu_p, gla_p;     // every lane of the quad: the g2p adjoint below wants the result in all four
    particle_adjoint<true>(c, q, kb, Cm, F, material, gw, gfx, gaff, gvp, gx, gv, gC, gF, gmu_p, gla_p);
    if (qi == 0) {
      if (material != 0) {
        const float h = clipf(a.hard[up], 0.1f, 5.f);
        atomicAdd(&s_par[0], gmu_p * h);
        atomicAdd(&s_par[1], gla_p * h);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) gs[d * c.Np + p] = gx[d];
#pragma unroll
      for (int d = 0; d < 9; ++d) gs[(15 + d) * c.Np + p] = gF[d];
      if (!hist_prev) {               // the last launch of the step: the whole cotangent state
#pragma unroll
        for (int d = 0; d < 3; ++d) gs[(3 + d) * c.Np + p] = gv[d];
#pragma unroll
        for (int d = 0; d < 9; ++d) gs[(6 + d) * c.Np + p] = gC[d];
      }
    }
  }
  // ---- B: g2p adjoint of substep f - 1 (lg_g2p_adj<4>), its input cotangents in registers ----
  if (hist_prev) {                    // kernel-uniform
    const int fb = a.f - 1;
    const float* hp = hist_prev + (long)b * a.hist_stride_b;
    int base[3] = {0, 0, 0};
    float fx[3] = {0.f, 0.f, 0.f}, w[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) w[d] = 0.f;
    if (live) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float xd = hp[d * c.Np + p];
        base[d] = (int)(xd * c.inv_dx - 0.5f);
        const float f = xd * c.inv_dx - (float)base[d];
        fx[d] = f;
        w[d] = 0.5f * ((1.5f - f) * (1.5f - f)); w[3 + d] = 0.75f - (f - 1.f) * (f - 1.f); w[6 + d] = 0.5f * ((f - 0.5f) * (f - 0.5f));
      }
    }
    const BlockWin win = bt_window(c, live, base);
    float4* gaccb = lg_gacc(a, fb) + (long)b * a.G;
    if (live) {
      float gnv[3], gw[9], gfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 3; ++d) gnv[d] = gv[d] + c.dt * gx[d];
#pragma unroll
      for (int d = 0; d < 9; ++d) gw[d] = 0.f;
      const float4* vel = a.w.vel + (long)b * a.G;
      const int rot = (p * 4) % 27;
      int key7[7];
      float4 v7[7];
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int it = min(qi + 4 * t, 26);
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        key7[t] = cell_gather(c, base[0] + cidx / 9, base[1] + (cidx / 3) % 3, base[2] + cidx % 3);
        v7[t] = vel[cell_lin(c, key7[t])];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int it = qi + 4 * t;
        if (it >= 27) break;
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const float wi = sel3(w, 0, i), wj = sel3(w, 1, j), wk = sel3(w, 2, k);
        const float weight = wi * wj * wk;
        const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
        const int gkey = key7[t];
        const float vv[3] = {v7[t].x, v7[t].y, v7[t].z};
        float gwt = 0.f;
        const int sl = bt_find<TH, TLOG>(bt, win, gkey);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gCd = gC[r * 3] * dp[0] + gC[r * 3 + 1] * dp[1] + gC[r * 3 + 2] * dp[2];
          const float gcell = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
          if (sl >= 0) __hip_atomic_fetch_add(&bt.val[r * TH + sl], (double)gcell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else atomicAdd((float*)(gaccb + cell_lin(c, gkey)) + r, gcell);
          gwt += vv[r] * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gC[r * 3 + s2] * vv[r];
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
          gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
          gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
        }
      }
#pragma unroll
      for (int d = 0; d < 9; ++d) gw[d] = lg_quad_sum<4>(gw[d]);
#pragma unroll
      for (int d = 0; d < 3; ++d) gfx[d] = lg_quad_sum<4>(gfx[d]);
      if (qi == 0) {
        float* ps = a.w.pscr + (long)b * 3 * c.Np + p;
#pragma unroll
        for (int d = 0; d < 3; ++d)
          ps[d * c.Np] = gfx[d] + gw[0 * 3 + d] * (-(1.5f - fx[d])) + gw[1 * 3 + d] * (-2.f * (fx[d] - 1.f)) + gw[2 * 3 + d] * (fx[d] - 0.5f);
      }
    }
    __syncthreads();
    {
      const int r = threadIdx.x & 3;
#pragma unroll 4
      for (int sl = threadIdx.x >> 2; sl < TH; sl += LG_SCATTER_T / 4) {
        const int key = bt.key[sl];
        if (key < 0 || r == 3) continue;
        atomicAdd((float*)(gaccb + cell_lin(c, key)) + r, (float)bt.val[r * TH + sl]);
      }
    }
  }
  __syncthreads();
  if (do_a && threadIdx.x < 2 && s_par[threadIdx.x] != 0.f) atomicAdd(&a.w.acc[b * 4 + 1 + threadIdx.x], s_par[threadIdx.x]);
}

// backward prologue: cotangent state, primitive arrays from the checkpoint tail, copy_frame adjoint
__global__ void __launch_bounds__(256) lg_bwd_in(LargeArgs a, const float* ck_tail, long ck_stride_b, const float* gppos,
                                                 const float* gprot) {
  const int b = blockIdx.x + a.b0, ip = blockIdx.y, S = a.c.steps;   // grid (B, n_prim)
  const long bp = (long)b * a.c.n_prim + ip;
  const float* tail = ck_tail + (long)b * ck_stride_b + (long)ip * S * 10;
  for (int e = threadIdx.x; e < S * 3; e += blockDim.x) {
    a.w.ppos[bp * S * 3 + e] = tail[e];
    a.w.ppin[bp * S * 3 + e] = tail[S * 7 + e];
    a.w.gpv[bp * S * 3 + e] = 0.f;
    a.w.gpw[bp * S * 3 + e] = 0.f;
    const int row = e / 3, d = e - row * 3;
    float g = gppos[bp * S * 3 + e];
    if (S > 1) {
      if (row == 0) g = 0.f;
      if (row == S - 1) g += gppos[bp * S * 3 + d];
    }
    a.w.gppos[bp * S * 3 + e] = g;
  }
  for (int e = threadIdx.x; e < S * 4; e += blockDim.x) {
    a.w.prot[bp * S * 4 + e] = tail[S * 3 + e];
    const int row = e / 4, d = e - row * 4;
    float g = gprot ? gprot[bp * S * 4 + e] : 0.f;          // copy_frame adjoint: rotation[0] <- rotation[steps - 1]
    if (S > 1 && gprot) {
      if (row == 0) g = 0.f;
      if (row == S - 1) g += gprot[bp * S * 4 + d];
    }
    a.w.grot[bp * S * 4 + e] = g;
  }
  if (threadIdx.x < 4) a.w.gpsz[bp * 4 + threadIdx.x] = 0.f;
  if (ip != 0) return;
  if (threadIdx.x < 4) a.w.acc[b * 4 + threadIdx.x] = 0.f;
  if (threadIdx.x < 3) a.w.count[threadIdx.x * a.B + b] = 0;      // (three list slots in the fused backward: lg_bslot)
}

// backward epilogue: set_action adjoint, norm_grad_state clip (the whole cotangent state of an env scaled to norm 1 when above),
// un-sort + SoA -> the caller's arrays.  One workgroup per env took 0.47 ms at 7631 particles (5 % of a pour_soup update): the
// particle part now runs 256 particles per block -- lg_bwd_norm sums the squares into acc[b][3] (zeroed by lg_bwd_in), lg_bwd_out
// scales and writes -- block 0 of each env takes the primitive arrays and the scalars along.
// set_action adjoint (primitives.py:212-229): thread (ip, d) sums its component over the substeps -> sga (action), sgs (its norm share)
__device__ __forceinline__ void lg_action_adjoint(const LargeArgs& a, int b, int clip, float* sga, float* sgs) {
  const int S = a.c.steps, P = a.c.n_prim, tid = threadIdx.x;
  if (tid < 6 * P) {
    const int ip = tid / 6, d = tid - ip * 6;
    const long bp = (long)b * P + ip;
    const float ac = clipf(a.action[bp * 6 + d], -1.f, 1.f);
    float g1 = 0.f, g2 = 0.f;
    // position control: nothing reaches the rotation chain and action[3:6] is reported as 0 (see unidom_hip.h)
    const float* src = (d < 3) ? a.w.gpv + bp * S * 3 + d : (a.c.position_control ? nullptr : a.w.gpw + bp * S * 3 + (d - 3));
    if (src)
      for (int j = 0; j < S; ++j) { const float t = src[j * 3]; g1 += t * 1.f / (float)S; g2 += t * ac / (float)S; }
    g1 *= clip_grad(a.action[bp * 6 + d], -1.f, 1.f);
    if (clip) { g1 = nan_to_num(g1 + 0.f); g2 = nan_to_num(g2); }
    sga[tid] = g1; sgs[tid] = g2;
  }
  __syncthreads();
}

// clip only: nan_to_num the cotangent state in place (norm_grad_state) and add its squares to acc[b][3]
__global__ void __launch_bounds__(256) lg_bwd_norm(LargeArgs a) {
  __shared__ float red[4], sga[6 * UD_MAX_PRIM], sgs[6 * UD_MAX_PRIM];
  const int b = blockIdx.y + a.b0, S = a.c.steps, N = a.c.N, Np = a.c.Np, P = a.c.n_prim, tid = threadIdx.x;
  const int p = blockIdx.x * 256 + tid;
  float* gs = a.w.gstate + (long)b * 24 * Np;
  float s2 = 0.f;
  if (p < N) {
#pragma unroll 8
    for (int r = 0; r < 24; ++r) { const float t = nan_to_num(gs[r * Np + p] + 0.f); gs[r * Np + p] = t; s2 += t * t; }
  }
  if (blockIdx.x == 0) {
    lg_action_adjoint(a, b, 1, sga, sgs);
    float* gp = a.w.gppos + (long)b * P * S * 3;     // all primitives of this env, contiguous
    float* gr = a.w.grot + (long)b * P * S * 4;
    for (int e = tid; e < P * S * 3; e += blockDim.x) { const float t = nan_to_num(gp[e] + 0.f); gp[e] = t; s2 += t * t; }
    if (!a.c.position_control)
      for (int e = tid; e < P * S * 4; e += blockDim.x) { const float t = nan_to_num(gr[e] + 0.f); gr[e] = t; s2 += t * t; }
    if (tid == 0) {
      const float tf = nan_to_num(a.w.acc[b * 4 + 0]), tm = nan_to_num(a.w.acc[b * 4 + 1]), tl = nan_to_num(a.w.acc[b * 4 + 2]);
      s2 += tf * tf + tm * tm + tl * tl;
      for (int d = 0; d < 6 * P; ++d) s2 += sgs[d] * sgs[d];
      if (!a.c.position_control)
        for (int d = 0; d < 4 * P; ++d) { const float t = nan_to_num(a.w.gpsz[(long)b * P * 4 + d]); s2 += t * t; }
    }
  }
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  if (tid == 0) {
    const float part = (red[0] + red[1]) + (red[2] + red[3]);
    if (a.det_normpart) a.det_normpart[b * LG_NORM_PARTS + blockIdx.x] = part;     // added up in block order by lg_bwd_out
    else atomicAdd(&a.w.acc[b * 4 + 3], part);
  }
}

__global__ void __launch_bounds__(256) lg_bwd_out(LargeArgs a, int clip, float* gx0, float* gv0, float* gC0, float* gF0, float* gppos0,
                                                  float* gfric, float* gmu, float* glam, float* gaction, float* grot0) {
  __shared__ float sga[6 * UD_MAX_PRIM], sgs[6 * UD_MAX_PRIM];
  const int b = blockIdx.y + a.b0, S = a.c.steps, N = a.c.N, Np = a.c.Np, P = a.c.n_prim, tid = threadIdx.x;
  const int p = blockIdx.x * 256 + tid;
  const float* gs = a.w.gstate + (long)b * 24 * Np;
  float n2 = a.w.acc[b * 4 + 3];
  if (a.det_normpart && clip) {
    n2 = 0.f;
    for (int k = 0; k < (N + 255) / 256; ++k) n2 += a.det_normpart[b * LG_NORM_PARTS + k];
  }
  const float sn = clip ? sqrtf(n2) : 0.f;
  const bool sc = clip && !(sn < 1.f);
  if (p < N) {
    const int up = user_index(a, b, p);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      gx0[((long)b * N + up) * 3 + d] = sc ? gs[d * Np + p] / sn : gs[d * Np + p];
      gv0[((long)b * N + up) * 3 + d] = sc ? gs[(3 + d) * Np + p] / sn : gs[(3 + d) * Np + p];
    }
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      gC0[((long)b * N + up) * 9 + d] = sc ? gs[(6 + d) * Np + p] / sn : gs[(6 + d) * Np + p];
      gF0[((long)b * N + up) * 9 + d] = sc ? gs[(15 + d) * Np + p] / sn : gs[(15 + d) * Np + p];
    }
  }
  if (blockIdx.x != 0) return;
  lg_action_adjoint(a, b, clip, sga, sgs);
  const float* gp = a.w.gppos + (long)b * P * S * 3;
  const float* gr = a.w.grot + (long)b * P * S * 4;
  for (int e = tid; e < P * S * 3; e += blockDim.x) gppos0[(long)b * P * S * 3 + e] = sc ? gp[e] / sn : gp[e];
  if (grot0)
    for (int e = tid; e < P * S * 4; e += blockDim.x) grot0[(long)b * P * S * 4 + e] = a.c.position_control ? 0.f : (sc ? gr[e] / sn : gr[e]);
  if (tid == 0) {
    float tf = a.w.acc[b * 4 + 0], tm = a.w.acc[b * 4 + 1], tl = a.w.acc[b * 4 + 2];
    if (clip) { tf = nan_to_num(tf); tm = nan_to_num(tm); tl = nan_to_num(tl); }
    gfric[b] = sc ? tf / sn : tf; gmu[b] = sc ? tm / sn : tm; glam[b] = sc ? tl / sn : tl;
  }
  if (tid < 6 * P) {
    float anrm = 0.f;
    if (clip) {
      float n2 = 0.f;
      for (int d = 0; d < 6 * P; ++d) n2 += sga[d] * sga[d];
      anrm = sqrtf(n2);
    }
    gaction[(long)b * P * 6 + tid] = (clip && !(anrm < 1.f)) ? sga[tid] / anrm : sga[tid];
  }
}

}  // namespace ud
#include "mpm_cluster.h"
namespace ud {

struct MpmLarge {
  MpmConst c;
  LgTune t{};
  const int* d_material;
  const float* d_hard;
  int B = 0, cap = 0;  // B = envs the arenas hold (LgTune::max_envs): fixed at create
  long W32 = 0;        // bitmap words per env
  long G = 0;
  LargeBuf w{};
  void* arena = nullptr;
  size_t arena_bytes = 0;
  // Small launches (a few hundred workgroups of latency-bound work) leave most of the chip idle and every kernel waits
  // for the previous one: the envs are split into groups that run the same kernel sequence on separate streams, forked
  // from and joined back into the caller's stream with events (no host synchronisation).
  static constexpr int MAX_GROUPS = 4;   // (6 and 8 groups measured: shape_rope backward 5.0 -> 9.4 ms, profiles/r03c_fused_bwd_groups.txt)
  hipStream_t side[MAX_GROUPS - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS - 1] = {};
  // persistent cluster forward (mpm_cluster.h): rotating grids for `cl.Bl` envs per launch, allocated at create where the handle can take it
  ClusterGrid cl{};
  void* cl_arena = nullptr;
  size_t cl_zero_bytes = 0, cl_own_off = 0, cl_own_words = 0, cl_bytes = 0;
  void* det_arena = nullptr; // deterministic forward (mpm_det.hip): flag [B][G] int, pre [B][27][Np], trq3 [B][S][3]
  size_t det_bytes = 0, det_off[14] = {};
  int det_epoch = 0;
  bool has_liquid = false;   // some particle has material 0
  int n_cu = 0, occ_fwd[2] = {0, 0};   // CUs; resident parts per CU of the persistent forward (occupancy query), [0] 64-lane, [1] 128-lane parts
};

#ifndef LG_GROUPS
#define LG_GROUPS 2
#endif
// number of env groups for a launch of B envs.  Measured (1x MI355X, substeps/s forward+backward, groups 1 / 2 / 4): rope at n_grid 256
// (32 envs x 6675, position control, one lane per particle) 123 k / 144 k / 151 k -- its grid kernels are short launches of a few hundred
// blocks and leave the chip to the other groups' particle kernels; pour_soup (32 x 7631, soft contact) 76.0 k / 77.5 k / 74.5 k; the
// four-lane launches are best at 2 (rope at n_grid 128 +6 %, shape_rope +3 %) and lose 5-20 % at 4 -- except pour_water (two container
// primitives: the grid kernels, not the particle kernels, carry its substep): 231 k in one group, 199-227 k from run to run in two.
static int lg_groups(const MpmLarge* L, int B) {
  const int forced = L->t.env_groups;                             // ud_mpm_conf.tune_env_groups (diagnostics, counter passes)
  if (!L->ev_fork || L->c.det) return 1;                          // (the deterministic mode runs on the caller's stream alone)
  const long particles = (long)B * L->c.N;
  int want = 1;
  if (particles <= 50000) want = L->c.n_prim >= 2 ? 1 : LG_GROUPS;     // four-lane kernels, up to 200 k lanes (two collide passes per cell: below)
  if (particles >= 100000) want = L->c.position_control ? 4 : LG_GROUPS;   // one-lane kernels (measurements above)
  if (forced > 0) want = forced;
  if (want <= 1 || B < 2 * want) return 1;
  return want < MpmLarge::MAX_GROUPS ? want : MpmLarge::MAX_GROUPS;
}

static int reserve(MpmLarge* L, int B);
static int clm_lanes(const MpmLarge* L);
static int clm_envs_per_launch(const MpmLarge* L, int B, int T);
static int clm_reserve(MpmLarge* L, int Bl);

MpmLarge* mpm_large_create(const MpmConst& c, const int* d_material, const float* d_hard, bool has_liquid, const LgTune& tune) {
  auto* L = new MpmLarge();
  L->c = c; L->t = tune; L->d_material = d_material; L->d_hard = d_hard; L->has_liquid = has_liquid;
  L->G = (long)c.res[0] * c.res[1] * c.res[2];
  L->cap = (int)std::min<long>(L->G, (long)54 * c.N);
  L->W32 = (L->G + 31) / 32 + 1;   // + 1: a row of eight cells may straddle into the word behind the last
  // dynamic LDS of the staging kernels (set explicitly so that a table above the 64 KB default keeps working)
  (void)hipFuncSetAttribute((const void*)lg_p2g<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<1>());
  (void)hipFuncSetAttribute((const void*)lg_p2g<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<4>());
  (void)hipFuncSetAttribute((const void*)lg_g2p_p2g<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<1>());
  (void)hipFuncSetAttribute((const void*)lg_g2p_p2g<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<4>());
  (void)hipFuncSetAttribute((const void*)lg_g2p_adj<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<1>());
  (void)hipFuncSetAttribute((const void*)lg_g2p_adj<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<4>());
  (void)hipFuncSetAttribute((const void*)lg_padj_gadj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_table_bytes<4>());
  (void)hipFuncSetAttribute((const void*)lg_sort, hipFuncAttributeMaxDynamicSharedMemorySize, LG_SORT_MAX * 8);
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&L->n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    for (int i = 0; i < 2; ++i) {   // i = 0: 64-lane parts, 1: 128-lane parts
      const void* kf = i ? (const void*)clm_fwd_kernel<128> : (const void*)clm_fwd_kernel<64>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&L->occ_fwd[i], kf, i ? 128 : 64, 0) != hipSuccess) L->occ_fwd[i] = 0;
    }
    (void)hipGetLastError();
  }
  bool ok = hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming) == hipSuccess;
  for (int g = 0; g < MpmLarge::MAX_GROUPS - 1; ++g) {
    ok = ok && hipStreamCreateWithFlags(&L->side[g], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L->ev_join[g], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { if (L->ev_fork) (void)hipEventDestroy(L->ev_fork); L->ev_fork = nullptr; (void)hipGetLastError(); }   // single-stream fallback
  // every arena of the handle, once, for LgTune::max_envs envs: no step call allocates or synchronises the host
  int rc = reserve(L, tune.max_envs);
  if (rc == UD_OK) {
    const int per = clm_envs_per_launch(L, tune.max_envs, clm_lanes(L));
    if (per > 0) rc = clm_reserve(L, per);
  }
  if (rc == UD_OK && c.det) {
    if (c.N > LG_SORT_MAX) { set_error("ud_mpm_create: deterministic mode sorts an env's particles in one workgroup's LDS: n_particles <= %d", LG_SORT_MAX); rc = UD_ERR_UNSUPPORTED; }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t Bm = (size_t)tune.max_envs;
    L->det_off[0] = take(Bm * L->G * 4);                 // flag
    L->det_off[1] = take(Bm * 27 * c.Np * 4);            // pre
    L->det_off[2] = take(Bm * c.steps * 3 * 4);          // trq3
    L->det_off[3] = take(Bm * c.Np * 4);                 // bkey
    L->det_off[4] = take(Bm * c.Np * 4);                 // order
    L->det_off[5] = take(Bm * L->G * 8);                 // brange
    L->det_off[6] = take(Bm * L->G * 4);                 // bflag
    L->det_off[7] = take(Bm * 4);                        // nirr
    L->det_off[8] = take(Bm * L->cap * 4);               // list
    L->det_off[9] = take(Bm * 4);                        // count
    L->det_off[10] = take(Bm * 27 * c.Np * 16);          // contrib
    L->det_off[11] = take(Bm * (size_t)std::min(L->cap, 32768) * (c.position_control ? 4 : 1 + 18 * c.n_prim) * 4);   // cellred (backward)
    L->det_off[12] = take(Bm * 2 * c.Np * 4);            // pacc
    L->det_off[13] = take(Bm * LG_NORM_PARTS * 4);       // normpart
    L->det_bytes = off;
    if (rc == UD_OK && (hipMalloc(&L->det_arena, L->det_bytes) != hipSuccess || hipMemset(L->det_arena, 0, L->det_bytes) != hipSuccess)) { set_error("ud_mpm_create (deterministic): hipMalloc failed"); rc = UD_ERR_HIP; }
  }
  if (rc == UD_OK && hipDeviceSynchronize() != hipSuccess) { set_error("ud_mpm_create: device synchronisation failed"); rc = UD_ERR_HIP; }
  if (rc != UD_OK) { mpm_large_destroy(L); return nullptr; }
  return L;
}

void mpm_large_destroy(MpmLarge* L) {
  if (!L) return;
  if (L->det_arena) (void)hipFree(L->det_arena);
  if (L->arena) (void)hipFree(L->arena);
  if (L->cl_arena) (void)hipFree(L->cl_arena);
  for (int g = 0; g < MpmLarge::MAX_GROUPS - 1; ++g) {
    if (L->side[g]) (void)hipStreamDestroy(L->side[g]);
    if (L->ev_join[g]) (void)hipEventDestroy(L->ev_join[g]);
  }
  if (L->ev_fork) (void)hipEventDestroy(L->ev_fork);
  delete L;
}

// checkpoint layout per env (floats): particle history [(S+1)][24][Np] | primitive tail [P][S*10] | grid checkpoint:
// record index [S+1] (ints, padded to 4) and the record pool [budget][8] | spatial order [Np] (ints)
static int lg_lanes(const MpmLarge* L, int B);
// The SVD factors ride in the history records where a launch waits for one wave's serial chain (the four-lane regime); where the chip
// is full (one lane per particle: B N >= 100 000) they cost 84 B of traffic per particle-substep each way and save nothing
// (rope at n_grid 256: 250 k -> 225-246 k substeps/s with them) -- so the record layout depends on the envs of the call, which
// ud_mpm_ckpt_bytes, the forward and the backward all know.
static bool lg_svd_rows(const MpmLarge* L, int B) { return !L->c.det && lg_lanes(L, B) == 4; }
struct CkLayout { long rec, off_tail, off_idx, off_pool, off_crec, off_perm, stride; int budget; };
static int clm_envs_per_launch(const MpmLarge* L, int B, int T);
static int clm_lanes(const MpmLarge* L);
// collide records beside the grid checkpoint: soft contact, grid checkpoint on, and the multi-kernel forward (the one whose grid op is
// lg_grid) serves this call shape -- forward, backward and ud_mpm_ckpt_bytes decide it from the handle and B alike
static bool lg_crec(const MpmLarge* L, int B) {
  const MpmConst& c = L->c;
  if (c.position_control || c.det || c.gck <= 0 || L->t.collide_records < 0) return false;
  return !(L->cl_arena && clm_envs_per_launch(L, B, clm_lanes(L)) > 0);
}
static CkLayout ck_layout(const MpmLarge* L, int B) {
  const MpmConst& c = L->c;
  CkLayout k{};
  const long S = c.steps;
  k.rec = (long)(24 + (lg_svd_rows(L, B) ? UD_SVD_ROWS : 0)) * c.Np;      // state rows + (four-lane regime) the SVD factors of the substep's F
  k.off_tail = (S + 1) * k.rec;
  k.off_idx = k.off_tail + (long)c.n_prim * S * 10;
  k.off_idx = (k.off_idx + 3) / 4 * 4;                         // float4 alignment of the pool behind it
  const long nidx = c.gck > 0 ? (S + 1 + 3) / 4 * 4 : 0;
  k.off_pool = k.off_idx + nidx;
  const long budget = c.gck > 0 ? S * (long)c.gck * c.N : 0;   // records per env and launch: gck cells per particle and substep on average
  k.budget = (int)std::min<long>(budget, 0x7fffffff / 2);
  k.off_crec = k.off_pool + (long)k.budget * 8;                // [budget][n_prim] float4 (lg_crec), or empty
  k.off_perm = k.off_crec + (lg_crec(L, B) ? (long)k.budget * 4 * c.n_prim : 0);   // [Np] ints: the spatial order of this launch (sort_particles)
  k.stride = k.off_perm + (c.sort ? (long)c.Np : 0);
  return k;
}

size_t mpm_large_ckpt_bytes(const MpmLarge* L, int B) {
  return (size_t)B * (size_t)ck_layout(L, B).stride * sizeof(float);
}

static int reserve(MpmLarge* L, int B) {
  const MpmConst& c = L->c;
  const long G = L->G, S = c.steps;
  const size_t BP = (size_t)B * c.n_prim;   // rows of the primitive arrays
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_val = take((size_t)B * G * 16), o_val2 = take((size_t)B * G * 16), o_vel = take((size_t)B * G * 16), o_gacc = take((size_t)B * G * 16);
  const size_t o_bits = take((size_t)B * L->W32 * 4), o_list = take((size_t)3 * B * L->cap * 4), o_count = take((size_t)3 * B * 4);
  const size_t o_ppos = take(BP * S * 3 * 4), o_prot = take(BP * S * 4 * 4), o_ppin = take(BP * S * 3 * 4);
  const size_t o_trq = take((size_t)B * S * 4), o_gppos = take(BP * S * 3 * 4), o_gpv = take(BP * S * 3 * 4);
  const size_t o_acc = take((size_t)B * 4 * 4), o_pscr = take((size_t)B * c.Np * 3 * 4);
  const size_t o_hist = take((size_t)B * 2 * (24 + UD_SVD_ROWS) * c.Np * 4), o_gstate = take((size_t)B * 24 * c.Np * 4);
  const size_t o_grot = take(BP * S * 4 * 4), o_gpw = take(BP * S * 3 * 4), o_gpsz = take(BP * 4 * 4);
  const size_t o_perm = take((size_t)B * c.Np * 4);
  hipError_t e = hipMalloc(&L->arena, off);
  if (e != hipSuccess) { set_error("ud_mpm_create (many-workgroup path): hipMalloc(%zu MB) failed: %s", off >> 20, hipGetErrorString(e)); L->B = 0; return UD_ERR_HIP; }
  e = hipMemset(L->arena, 0, off);   // grid cells, bitmap and counters start at zero
  if (e != hipSuccess) { set_error("ud_mpm (large path): memset failed"); return UD_ERR_HIP; }
  char* base = (char*)L->arena;
  L->w.val = (float4*)(base + o_val); L->w.val2 = (float4*)(base + o_val2); L->w.vel = (float4*)(base + o_vel); L->w.gacc = (float4*)(base + o_gacc);
  L->w.bits = (unsigned*)(base + o_bits); L->w.list = (int*)(base + o_list); L->w.count = (int*)(base + o_count);
  L->w.ppos = (float*)(base + o_ppos); L->w.prot = (float*)(base + o_prot); L->w.ppin = (float*)(base + o_ppin);
  L->w.trq = (float*)(base + o_trq); L->w.gppos = (float*)(base + o_gppos); L->w.gpv = (float*)(base + o_gpv);
  L->w.acc = (float*)(base + o_acc); L->w.pscr = (float*)(base + o_pscr); L->w.hist = (float*)(base + o_hist);
  L->w.gstate = (float*)(base + o_gstate);
  L->w.grot = (float*)(base + o_grot); L->w.gpw = (float*)(base + o_gpw); L->w.gpsz = (float*)(base + o_gpsz);
  L->w.perm = (int*)(base + o_perm);
  L->arena_bytes = off; L->B = B;
  return UD_OK;
}

static LargeArgs base_args(MpmLarge* L, int B, const float* psize, const float* friction, const float* mu, const float* lamda,
                           const float* action) {
  LargeArgs a{};
  a.c = L->c; a.w = L->w; a.material = L->d_material; a.hard = L->d_hard; a.B = L->B; a.f = 0; a.cap = L->cap; a.G = L->G; a.W32 = L->W32;
  a.hist_in = nullptr; a.hist_out = nullptr; a.hist_stride_b = 0; a.b0 = 0;
  a.gck_base = nullptr; a.gck_off_idx = 0; a.gck_off_pool = 0; a.gck_budget = 0; a.status = nullptr; a.gpar = 0;
  a.svd_rows = 0; a.ls3 = 0; a.vb = 0; a.ls = 0; a.lprev = 0; a.lnext = 0; a.nbx = 1; a.Bg = 0; a.xcd = 0;
  a.perm = nullptr; a.perm_stride = 0;
  a.det_cellred = nullptr; a.det_capc = 0; a.det_K = 0; a.det_pacc = nullptr; a.det_normpart = nullptr;
  a.psize = psize; a.friction = friction; a.mu = mu; a.lamda = lamda; a.action = action;
  (void)B;
  return a;
}

// env groups: group g covers envs [b0, b0 + Bg) on stream s (group 0 = the caller's stream)
struct LgGroup { int b0, Bg; hipStream_t s; };
static int lg_fork(MpmLarge* L, int B, hipStream_t st, LgGroup* grp, int want = 0) {   // want > 0: that many groups where lg_groups allows any
  int G = lg_groups(L, B);
  if (want > 0 && L->ev_fork && L->t.env_groups <= 0) G = (B >= 2 * want) ? std::min(want, (int)MpmLarge::MAX_GROUPS) : 1;
  for (int g = 0; g < G; ++g) {
    const int b0 = (int)((long)B * g / G), b1 = (int)((long)B * (g + 1) / G);
    grp[g] = LgGroup{b0, b1 - b0, g == 0 ? st : L->side[g - 1]};
  }
  if (G > 1) {
    (void)hipEventRecord(L->ev_fork, st);
    for (int g = 1; g < G; ++g) (void)hipStreamWaitEvent(grp[g].s, L->ev_fork, 0);
  }
  return G;
}
static void lg_join(MpmLarge* L, int G, hipStream_t st, const LgGroup* grp) {
  for (int g = 1; g < G; ++g) {
    (void)hipEventRecord(L->ev_join[g - 1], grp[g].s);
    (void)hipStreamWaitEvent(st, L->ev_join[g - 1], 0);
  }
}

// lanes per particle: 4 while the launch is too small to fill the chip, 1 beyond (see LgTable).  ud_mpm_conf.tune_lanes = 1 | 4 fixes it at
// create -- a diagnostic, and how the tests reach the one-lane kernels at sizes their CPU oracle can follow.
static int lg_lanes(const MpmLarge* L, int B) {
  if (L->t.lanes == 1 || L->t.lanes == 4) return L->t.lanes;
  return ((long)B * L->c.N < 100000) ? 4 : 1;
}

// backward with the grid checkpoint: two launches per reverse substep (lg_gadj_restore, lg_padj_gadj) or four -- measurements at the call
static bool lg_two_launch_bwd(const MpmLarge* L, int lanes) { return lanes == 4 && L->c.n_prim == 1 && L->t.bwd_two_launch >= 0; }

// ---- persistent cluster forward (mpm_cluster.h) -------------------------------------------------------------------
// Taken when the launch does not fill the chip (the four-lane regime) and the body's parts fit: envs per launch =
// 8 * floor(resident parts per XCD / parts per env), the parts of an env sharing an XCD under round-robin placement.
// lanes per part: 128 (default: 32 particles per part -- half the parts, less duplicated grid work; what a part's 512-slot table
// cannot hold goes to the env's HBM grid directly, clm_scatter) or 64 (tune_cluster_part_lanes = 64: 16 particles, never spills)
static int clm_lanes(const MpmLarge* L) { return L->t.cluster_part_lanes == 64 ? 64 : 128; }
static int clm_parts(const MpmConst& c, int T) { return (c.N + T / 4 - 1) / (T / 4); }
// Which bodies: by default solids with one primitive -- lattice-seeded ropes, whose 32 consecutive particles stay within a few cells
// of each other.  Liquids (material 0: sampled uniformly, they mix) spread a part over more cells than its table holds, and
// with several primitives every part repeats the collide chains of the cells it shares (measured, pour_water: the multi-kernel
// path is faster): both keep the multi-kernel path.  tune_cluster = 1 takes the cluster forward wherever it fits, -1 never.
static int clm_envs_per_launch(const MpmLarge* L, int B, int T) {
  if (L->t.cluster < 0 || L->c.det) return 0;
  if (L->t.cluster == 0 && (L->has_liquid || L->c.n_prim > 1)) return 0;
  const int i = T == 128 ? 1 : 0;
  const int occ = L->occ_fwd[i];
  if (occ <= 0 || L->n_cu < 8 || lg_lanes(L, B) != 4) return 0;
  const int W = clm_parts(L->c, T);
  const int per_xcd = ((L->n_cu / 8) * occ) / W;
  if (per_xcd < 1) return 0;
  int per = std::min(B, 8 * per_xcd);
  if (L->t.cluster_envs > 0) per = std::min(per, L->t.cluster_envs);   // tests: several launches per call
  return per;
}

static int clm_reserve(MpmLarge* L, int Bl) {
  const size_t cells = (size_t)Bl * L->G;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  size_t o_cg[3], o_own[3];
  for (int i = 0; i < 3; ++i) o_cg[i] = take(cells * 16);
  const size_t zero_bytes = off;                         // grids rest at zero, the owner stamps at INT_MAX
  for (int i = 0; i < 3; ++i) o_own[i] = take(cells * 4);
  const size_t o_bar = take((size_t)Bl * CLM_BAR_STRIDE * 4);
  hipError_t e = hipMalloc(&L->cl_arena, off);
  if (e != hipSuccess) { set_error("ud_mpm_create (cluster path): hipMalloc(%zu MB) failed: %s", off >> 20, hipGetErrorString(e)); return UD_ERR_HIP; }
  char* base = (char*)L->cl_arena;
  L->cl_zero_bytes = zero_bytes; L->cl_own_off = o_own[0]; L->cl_own_words = (o_bar - o_own[0]) / 4; L->cl_bytes = off;
  e = hipMemset(base, 0, off);
  if (e == hipSuccess) e = hipMemsetD32((hipDeviceptr_t)(base + L->cl_own_off), 0x7fffffff, L->cl_own_words);
  if (e != hipSuccess) { set_error("ud_mpm_create (cluster path): memset failed"); return UD_ERR_HIP; }
  for (int i = 0; i < 3; ++i) { L->cl.cg[i] = (float4*)(base + o_cg[i]); L->cl.own[i] = (int*)(base + o_own[i]); }
  L->cl.bar = (unsigned*)(base + o_bar);
  L->cl.Bl = Bl;
  return UD_OK;
}

// every arena back to its rest state, asynchronously on `st`: after a device-side time-out (status bit 4) a part may have left rotating
// grids, owner stamps, active lists or cotangent grids dirty
int mpm_large_reset(MpmLarge* L, hipStream_t st) {
  hipError_t e = hipSuccess;
  if (L->arena) e = hipMemsetAsync(L->arena, 0, L->arena_bytes, st);
  if (e == hipSuccess && L->cl_arena) {
    e = hipMemsetAsync(L->cl_arena, 0, L->cl_bytes, st);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)((char*)L->cl_arena + L->cl_own_off), 0x7fffffff, L->cl_own_words, st);
  }
  if (e == hipSuccess && L->det_arena) { e = hipMemsetAsync(L->det_arena, 0, L->det_bytes, st); L->det_epoch = 0; }
  if (e != hipSuccess) { set_error("ud_mpm_reset: %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

__global__ void lg_ckpt_cells(const float* ckpt, long stride_b, long off_idx, int S, int B, int* cells) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) cells[b] = ((const int*)(ckpt + (long)b * stride_b + off_idx))[S];     // gck_idx(b)[S]: one past the last record of the step
}
int mpm_large_ckpt_cells(const MpmLarge* L, int B, const float* ckpt, int* cells, hipStream_t st) {
  if (B > L->B) { set_error("ud_mpm_ckpt_cells: B=%d exceeds the handle's max_envs=%d", B, L->B); return UD_ERR_INVALID; }
  const CkLayout ck = ck_layout(L, B);
  if (ck.budget <= 0) return hipMemsetAsync(cells, 0, (size_t)B * sizeof(int), st) == hipSuccess ? UD_OK : UD_ERR_HIP;
  hipLaunchKernelGGL(lg_ckpt_cells, dim3((B + 63) / 64), dim3(64), 0, st, ckpt, ck.stride, ck.off_idx, L->c.steps, B, cells);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}

int mpm_large_plan(MpmLarge* L, int B) {
  if (B > L->B) return -1;
  int plan = 1;
  if (L->cl_arena && clm_envs_per_launch(L, B, clm_lanes(L))) plan |= 2;
  if (ck_layout(L, B).budget > 0 && lg_two_launch_bwd(L, lg_lanes(L, B))) plan |= 4;
  return plan;
}

// launch of a per-substep kernel: logical shape (NBX blocks per env) x (BG envs) as a one-dimensional grid, XCD-aware where there are
// at least eight envs (lg_bid); `a` (the LargeArgs of the calling function) carries the shape
#define LG_LAUNCH(K, NBX, BG, BLK, SH, ST, ...) do { a.nbx = (int)(NBX); a.Bg = (int)(BG); a.xcd = a.Bg >= 8 ? 1 : 0;                      \
    hipLaunchKernelGGL(K, dim3(a.xcd ? 8u * (unsigned)a.nbx * (unsigned)((a.Bg + 7) / 8) : (unsigned)a.nbx * (unsigned)a.Bg), BLK, SH, ST, a, ##__VA_ARGS__); } while (0)

// the deterministic mode's arrays (mpm_det.hip): scratch in the handle's det arena, grids shared with the default kernels
static DetArgs lg_det_args(MpmLarge* L, int B, const float* psize, const float* friction, const float* mu, const float* lamda, const float* action) {
  DetArgs d{};
  d.c = L->c; d.B = B; d.G = L->G; d.material = L->d_material; d.hard = L->d_hard;
  d.ppos = L->w.ppos; d.prot = L->w.prot; d.psize = psize; d.friction = friction; d.mu = mu; d.lamda = lamda; d.action = action;
  d.vel = (float*)L->w.vel;
  char* db = (char*)L->det_arena;
  d.flag = (int*)(db + L->det_off[0]); d.pre = (float*)(db + L->det_off[1]); d.trq3 = (float*)(db + L->det_off[2]);
  d.bkey = (int*)(db + L->det_off[3]); d.order = (int*)(db + L->det_off[4]); d.brange = (void*)(db + L->det_off[5]); d.bflag = (int*)(db + L->det_off[6]);
  d.nirr = (int*)(db + L->det_off[7]); d.list = (int*)(db + L->det_off[8]); d.count = (int*)(db + L->det_off[9]); d.cap = L->cap; d.contrib = (float*)(db + L->det_off[10]);
  d.cellred = (float*)(db + L->det_off[11]); d.pacc = (float*)(db + L->det_off[12]);
  d.capc = std::min(L->cap, 32768); d.K = L->c.position_control ? 4 : 1 + 18 * L->c.n_prim;
  d.gppos = L->w.gppos; d.grot = L->w.grot; d.gpsz = L->w.gpsz;
  d.trq = L->w.trq;
  return d;
}

int mpm_large_step_fwd(MpmLarge* L, int B, const float* x, const float* v, const float* C, const float* F, const float* J,
                       const float* ppos, const float* prot, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, float* xo, float* vo, float* Co, float* Fo, float* Jo, float* ppos_o,
                       float* prot_o, float* pv_o, float* pw_o, float* ckpt, int* status, hipStream_t st) {
  if (B > L->B) { set_error("ud_mpm_step_fwd: B=%d exceeds the handle's max_envs=%d (arenas are sized at create)", B, L->B); return UD_ERR_INVALID; }
  int rc = UD_OK;
  const MpmConst& c = L->c;
  const int S = c.steps, N = c.N;
  const dim3 blk(256), blks(LG_SCATTER_T);
  const int lanes = lg_lanes(L, B);                      // lanes per particle in the four particle kernels
  LargeArgs a = base_args(L, B, psize, friction, mu, lamda, action);
  a.B = L->B;
  // history: in the caller's checkpoint when there is one, otherwise the handle's ping-pong pair
  float* hist = ckpt ? ckpt : L->w.hist;
  const CkLayout ck = ck_layout(L, B);
  const long rec = ck.rec;
  const long stride_b = ckpt ? ck.stride : 2 * rec;
  a.hist_stride_b = stride_b;
  a.svd_rows = (ckpt && lg_svd_rows(L, B)) ? 1 : 0;       // the SVD factors ride in the checkpoint's records (ck_layout), for the backward
  if (status) (void)hipMemsetAsync(status, 0, (size_t)B * sizeof(int), st);   // before the launches: lg_grid may flag an env
  if (ckpt && ck.budget > 0) { a.gck_base = ckpt; a.gck_off_idx = ck.off_idx; a.gck_off_pool = ck.off_pool; a.gck_budget = ck.budget; a.status = status; }
  if (a.gck_base && lg_crec(L, B)) a.gck_off_crec = ck.off_crec;
  // spatial order of this launch: into the checkpoint (the backward needs the same one) or the handle's arena
  const bool sort = c.sort && N <= LG_SORT_MAX;
  int* perm = nullptr;
  long perm_stride = 0;
  if (sort) {
    perm = ckpt ? (int*)(ckpt + ck.off_perm) : L->w.perm;
    perm_stride = ckpt ? stride_b : c.Np;
    a.perm = perm; a.perm_stride = perm_stride;
  }
  int npow2 = 64;
  while (npow2 < N) npow2 <<= 1;
  if (c.det) {
    // deterministic forward (mpm_det.hip): the data movement around it is this file's, the arithmetic is compiled there
    a.b0 = 0; a.f = 0;
    hipLaunchKernelGGL(lg_prim_in, dim3(B, c.n_prim), blk, 0, st, a, ppos, prot);
    hipLaunchKernelGGL(lg_pack, dim3((N + 255) / 256, B), blk, 0, st, c, 0, x, v, C, F, hist, stride_b, 1, (const int*)nullptr, 0L);
    DetArgs d = lg_det_args(L, B, psize, friction, mu, lamda, action);
    d.hist = hist; d.rec = rec; d.stride_b = stride_b; d.pingpong = ckpt ? 0 : 1;
    rc = mpm_det_forward(d, &L->det_epoch, st);
    if (rc) { set_error("ud_mpm_step_fwd (deterministic): launch failed"); return rc; }
    a.f = S;
    const float* last = hist + (ckpt ? (long)S * rec : (long)(S & 1) * rec);
    float* tail = ckpt ? ckpt + ck.off_tail : nullptr;
    hipLaunchKernelGGL(lg_unpack, dim3((N + 255) / 256, B), blk, 0, st, c, 0, last, stride_b, xo, vo, Co, Fo, (const int*)nullptr, 0L);
    hipLaunchKernelGGL(lg_fwd_out, dim3(B, c.n_prim + (N + 255) / 256), blk, 0, st, a, J, Jo, ppos_o, prot_o, pv_o, pw_o, tail, stride_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("ud_mpm_step_fwd (deterministic): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
    return UD_OK;
  }
  const int clT = clm_lanes(L);
  if (const int per = L->cl_arena ? std::min(clm_envs_per_launch(L, B, clT), L->cl.Bl) : 0) {
    // persistent cluster kernel: one launch runs all S substeps of `per` envs (mpm_cluster.h)
    a.status = status;                                    // gck_base stays: the cluster forward writes the grid checkpoint too
    const float* last = hist + (ckpt ? (long)S * rec : (long)(S & 1) * rec);
    float* tail = ckpt ? ckpt + ck.off_tail : nullptr;
    for (int b0 = 0; b0 < B; b0 += per) {
      const int Bl = std::min(per, B - b0);
      a.b0 = b0; a.f = 0;
      hipLaunchKernelGGL(lg_prim_in, dim3(Bl, c.n_prim), blk, 0, st, a, ppos, prot);
      if (sort) hipLaunchKernelGGL(lg_sort, dim3(Bl), dim3(1024), (size_t)npow2 * 8, st, c, b0, x, perm, perm_stride, npow2);
      hipLaunchKernelGGL(lg_pack, dim3((N + 255) / 256, Bl), blk, 0, st, c, b0, x, v, C, F, hist, stride_b, 1, (const int*)perm, perm_stride);
      hipLaunchKernelGGL(lg_fk_all, dim3(Bl, c.n_prim), dim3(64), 0, st, a);
      (void)hipMemsetAsync(L->cl.bar, 0, (size_t)Bl * CLM_BAR_STRIDE * sizeof(unsigned), st);
      ClusterGrid g = L->cl;
      g.Bl = Bl; g.W = clm_parts(c, clT);
      const long last_off = ckpt ? (long)S * rec : (long)(S & 1) * rec;
      if (clT == 128) hipLaunchKernelGGL(clm_fwd_kernel<128>, dim3(clm_grid(Bl, g.W)), dim3(128), 0, st, a, g, hist, rec, ckpt ? 1 : 0, last_off);
      else hipLaunchKernelGGL(clm_fwd_kernel<64>, dim3(clm_grid(Bl, g.W)), dim3(64), 0, st, a, g, hist, rec, ckpt ? 1 : 0, last_off);
      a.f = S;
      hipLaunchKernelGGL(lg_unpack, dim3((N + 255) / 256, Bl), blk, 0, st, c, b0, last, stride_b, xo, vo, Co, Fo, (const int*)perm, perm_stride);
      hipLaunchKernelGGL(lg_fwd_out, dim3(Bl, c.n_prim + (N + 255) / 256), blk, 0, st, a, J, Jo, ppos_o, prot_o, pv_o, pw_o, tail, stride_b);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("ud_mpm_step_fwd (cluster path): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
    return UD_OK;
  }
  LgGroup grp[MpmLarge::MAX_GROUPS];
  const int G = lg_fork(L, B, st, grp);
  for (int g = 0; g < G; ++g) {
    a.b0 = grp[g].b0;
    hipLaunchKernelGGL(lg_prim_in, dim3(grp[g].Bg, c.n_prim), blk, 0, grp[g].s, a, ppos, prot);
    if (sort) hipLaunchKernelGGL(lg_sort, dim3(grp[g].Bg), dim3(1024), (size_t)npow2 * 8, grp[g].s, c, a.b0, x, perm, perm_stride, npow2);
    hipLaunchKernelGGL(lg_pack, dim3((N + 255) / 256, grp[g].Bg), blk, 0, grp[g].s, c, a.b0, x, v, C, F, hist, stride_b, 1, (const int*)perm, perm_stride);
  }
  // Two launches per substep: the grid op (which also retires the previous substep's cells and its own list's bits: no clear launch)
  // and ONE particle launch, g2p(f) -> p2g(f + 1) (lg_g2p_p2g); p2g(0) opens the step, g2p(S - 1) closes it -- the (m, mv) grid
  // alternates between two arrays, the active lists between three (LargeArgs, ls3); forward kinematics of the whole step once (lg_fk_all).
  for (int g = 0; g < G; ++g) {
    a.b0 = grp[g].b0; a.f = 0; a.ls3 = 1;
    hipLaunchKernelGGL(lg_fk_all, dim3(grp[g].Bg, c.n_prim), dim3(64), 0, grp[g].s, a);
  }
  for (int f = 0; f <= S; ++f) {
    a.f = f;
    a.hist_in = hist + (ckpt ? (long)f * rec : (long)(f & 1) * rec);
    a.hist_out = hist + (ckpt ? (long)(f + 1) * rec : (long)((f + 1) & 1) * rec);
    a.ls3 = 1; a.vb = f & 1; a.ls = f % 3; a.lprev = (f + 2) % 3; a.lnext = (f + 1) % 3;
    for (int g = 0; g < G; ++g) {
      const int Bg = grp[g].Bg;
      hipStream_t s = grp[g].s;
      a.b0 = grp[g].b0;
      const dim3 gc(lg_cell_blocks(L->cap), Bg), gs((lanes * N + LG_SCATTER_T - 1) / LG_SCATTER_T, Bg), gq((lanes * N + 255) / 256, Bg);
      if (f == S) { LG_LAUNCH(lg_grid, gc.x, (int)gc.y, blk, 0, s, 0); continue; }   // retires the last substep's cells: both grids all-zero again
      if (f == 0) {                                          // the p2g pass of substep f >= 1 rides behind g2p(f - 1) in lg_g2p_p2g
        if (lanes == 4) LG_LAUNCH(lg_p2g<4>, gs.x, (int)gs.y, blks, lg_table_bytes<4>(), s, 1); else LG_LAUNCH(lg_p2g<1>, gs.x, (int)gs.y, blks, lg_table_bytes<1>(), s, 1);
      }
      LG_LAUNCH(lg_grid, gc.x, (int)gc.y, blk, 0, s, 0);
      if (f + 1 < S) {
        float* ho2 = hist + (ckpt ? (long)(f + 2) * rec : (long)(f & 1) * rec);
        if (lanes == 4) LG_LAUNCH(lg_g2p_p2g<4>, gs.x, (int)gs.y, blks, lg_table_bytes<4>(), s, ho2); else LG_LAUNCH(lg_g2p_p2g<1>, gs.x, (int)gs.y, blks, lg_table_bytes<1>(), s, ho2);
      } else {
        if (lanes == 4) LG_LAUNCH(lg_g2p<4>, gq.x, (int)gq.y, blk, 0, s); else LG_LAUNCH(lg_g2p<1>, gq.x, (int)gq.y, blk, 0, s);
      }
    }
  }
  a.ls3 = 0;
  a.f = S;
  const float* last = hist + (ckpt ? (long)S * rec : (long)(S & 1) * rec);
  float* tail = ckpt ? ckpt + ck.off_tail : nullptr;
  for (int g = 0; g < G; ++g) {
    a.b0 = grp[g].b0;
    hipLaunchKernelGGL(lg_unpack, dim3((N + 255) / 256, grp[g].Bg), blk, 0, grp[g].s, c, a.b0, last, stride_b, xo, vo, Co, Fo, (const int*)perm, perm_stride);
    hipLaunchKernelGGL(lg_fwd_out, dim3(grp[g].Bg, c.n_prim + (N + 255) / 256), blk, 0, grp[g].s, a, J, Jo, ppos_o, prot_o, pv_o, pw_o, tail, stride_b);
  }
  lg_join(L, G, st, grp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("ud_mpm_step_fwd (large path): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

int mpm_large_step_bwd(MpmLarge* L, int B, const float* ckpt, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, const float* gx, const float* gv, const float* gC, const float* gF,
                       const float* gppos, const float* gprot, int clip, float* gx0, float* gv0, float* gC0, float* gF0, float* gppos0,
                       float* grot0, float* gfric, float* gmu, float* glam, float* gaction, int* status, hipStream_t st) {
  if (B > L->B) { set_error("ud_mpm_step_bwd: B=%d exceeds the handle's max_envs=%d", B, L->B); return UD_ERR_INVALID; }
  const MpmConst& c = L->c;
  const int S = c.steps, N = c.N, Np = c.Np;
  const dim3 blk(256), blks(LG_SCATTER_T);
  const int lanes = lg_lanes(L, B);                      // lanes per particle in the four particle kernels
  LargeArgs a = base_args(L, B, psize, friction, mu, lamda, action);
  const CkLayout ck = ck_layout(L, B);
  const long rec = ck.rec;
  const long stride_b = ck.stride;
  a.hist_stride_b = stride_b;
  a.svd_rows = lg_svd_rows(L, B) ? 1 : 0;                 // the forward's SVD factors ride in the records: read, not iterated again
  // restore the grid from the checkpoint instead of recomputing p2g + grid op -- unless the caller saw the forward flag a
  // pool overflow and asks for the recomputing backward (clip bit 1)
  const bool gck = ck.budget > 0 && !(clip & 2);
  clip &= 1;
  if (gck) { a.gck_base = const_cast<float*>(ckpt); a.gck_off_idx = ck.off_idx; a.gck_off_pool = ck.off_pool; a.gck_budget = ck.budget; }
  if (gck && lg_crec(L, B)) a.gck_off_crec = ck.off_crec;
  if (c.sort && N <= LG_SORT_MAX) { a.perm = (const int*)(ckpt + ck.off_perm); a.perm_stride = stride_b; }   // the forward's order
  if (status) (void)hipMemsetAsync(status, 0, (size_t)B * sizeof(int), st);
  // The backward is the multi-kernel path, restoring the grid from the checkpoint that either forward wrote.  (A persistent cluster
  // backward -- recomputing the grid, or restoring it from per-part records -- was built in round 3 and measured no faster: 256 VGPRs +
  // scratch at two waves per SIMD, and its long chains are single-wave latency either way; a three-launch form where the two-launch one
  // does not apply was measured slower.  Both are gone: git history)
  // Two launches per reverse substep (lg_gadj_restore, lg_padj_gadj) where the backward restores the grid, four lanes work on a
  // particle and one primitive touches the grid.  Measured on 1x MI355X, 32 envs, backward ms per step, four-kernel / two-launch
  // (profiles/r03c_fused_bwd_groups.txt): rope at n_grid 128 (position control) 2.88 / 2.45 in one env group, 2.76 / 2.92 in two;
  // shape_rope (soft contact: the grid-op adjoint is the long launch and overlaps the other groups' particle launches)
  // 6.08 / 6.00 in one, 5.98 / 5.65 in two, 7.24 / 5.25 in four; pour_water (two container primitives) 1.35 / 1.61: kept on four.
  const bool fused = gck && lg_two_launch_bwd(L, lanes);
  LgGroup grp[MpmLarge::MAX_GROUPS];
  const int G = lg_fork(L, B, st, grp, fused ? (c.position_control ? 1 : 4) : 0);
  for (int g = 0; g < G; ++g) {
    a.b0 = grp[g].b0;
    hipLaunchKernelGGL(lg_bwd_in, dim3(grp[g].Bg, c.n_prim), blk, 0, grp[g].s, a, ckpt + ck.off_tail, stride_b, gppos, gprot);
    hipLaunchKernelGGL(lg_pack, dim3((N + 255) / 256, grp[g].Bg), blk, 0, grp[g].s, c, a.b0, gx, gv, gC, gF, L->w.gstate, (long)24 * Np, 0, a.perm, a.perm_stride);
  }
  if (c.det) {
    // Deterministic backward: the recomputing backward with every arrival-ordered sum replaced -- the grid of substep f comes
    // from the deterministic forward's own kernels (ordered (m, mv) sums), the g2p adjoint's scatter is an ordered sum per cell over (offset,
    // particle) (lg_g2p_adj_det -> det_cells_kernel<1>), the per-env cotangents of the grid-op adjoint, the mu / lamda cotangents and the clip's
    // norm are added up in a fixed order.  One lane per particle, one stream.  Two calls on the same inputs return the same bits.
    DetArgs d = lg_det_args(L, B, psize, friction, mu, lamda, action);
    d.hist = const_cast<float*>(ckpt); d.rec = rec; d.stride_b = stride_b; d.pingpong = 0; d.bwd = 1;
    d.val_out = (float*)L->w.val; d.gacc = (float*)L->w.gacc; d.acc = L->w.acc; d.gpv = L->w.gpv; d.status = status;
    char* db = (char*)L->det_arena;
    a.det_cellred = d.cellred; a.det_capc = d.capc; a.det_K = d.K; a.det_pacc = (float*)(db + L->det_off[12]); a.det_normpart = (float*)(db + L->det_off[13]);
    a.b0 = 0; a.gck_base = nullptr; a.svd_rows = 0; a.perm = nullptr;
    const dim3 gc(lg_cell_blocks(L->cap), B), gq((N + 255) / 256, B), gqf(gq.x + c.n_prim, B);
    int rc = UD_OK;
    for (int f = S - 1; f >= 0 && rc == UD_OK; --f) {
      a.f = f; a.hist_in = ckpt + (long)f * rec;
      d.keylist = L->w.list + (long)(f & 1) * L->B * L->cap; d.keycount = L->w.count + (long)(f & 1) * L->B;
      rc = mpm_det_bwd_recompute(d, f, &L->det_epoch, st);
      {
        hipStream_t s = st;
        LG_LAUNCH(lg_g2p_adj_det, gq.x, (int)gq.y, blk, 0, s, (float4*)d.contrib);
        if (rc == UD_OK) rc = mpm_det_bwd_gcells(d, f, L->det_epoch, st);
        LG_LAUNCH(lg_grid_adj_det, gc.x, (int)gc.y, blk, 0, s);
        if (rc == UD_OK) rc = mpm_det_bwd_reduce_cells(d, f, st);
        LG_LAUNCH(lg_p2g_adj<1>, gqf.x, (int)gqf.y, blk, 0, s, (int)gq.x);
        if (rc == UD_OK) rc = mpm_det_bwd_clear(d, st);
      }
    }
    if (rc == UD_OK) rc = mpm_det_bwd_reduce_particles(d, st);
    a.f = -1;
    const dim3 gp((N + 255) / 256, B);
    if (clip) hipLaunchKernelGGL(lg_bwd_norm, gp, blk, 0, st, a);
    hipLaunchKernelGGL(lg_bwd_out, gp, blk, 0, st, a, clip, gx0, gv0, gC0, gF0, gppos0, gfric, gmu, glam, gaction, grot0);
    const hipError_t e = hipGetLastError();
    if (rc != UD_OK || e != hipSuccess) { set_error("ud_mpm_step_bwd (deterministic): %s", rc != UD_OK ? "launch failed" : hipGetErrorString(e)); return rc != UD_OK ? rc : UD_ERR_HIP; }
    return UD_OK;
  }
  if (fused) {
    a.gpar = 1;
    for (int g = 0; g < G; ++g) {
      const int Bg = grp[g].Bg;
      hipStream_t s = grp[g].s;
      a.b0 = grp[g].b0;
      const int nb = lg_cell_blocks(L->cap);
      const dim3 gc2(2 * nb, Bg);
      const int pb = (4 * N + LG_SCATTER_T - 1) / LG_SCATTER_T;
      const dim3 gk2(pb + c.n_prim, Bg);
      // prologue: restore of substep S - 1, then its g2p adjoint alone (cotangents from w.gstate)
      a.f = S; a.hist_in = ckpt;
      LG_LAUNCH(lg_gadj_restore, gc2.x, (int)gc2.y, blk, 0, s, nb);
      LG_LAUNCH(lg_padj_gadj, gk2.x, (int)gk2.y, blks, lg_table_bytes<4>(), s, pb, 0, ckpt + (long)(S - 1) * rec);
      for (int f = S - 1; f >= 0; --f) {
        a.f = f; a.hist_in = ckpt + (long)f * rec;
        LG_LAUNCH(lg_gadj_restore, gc2.x, (int)gc2.y, blk, 0, s, nb);
        LG_LAUNCH(lg_padj_gadj, gk2.x, (int)gk2.y, blks, lg_table_bytes<4>(), s, pb, 1, f > 0 ? ckpt + (long)(f - 1) * rec : nullptr);
      }
      // lg_padj_gadj(0) consumed the cotangent cells of substep 0: "restore" of substep -2 zeroes them (those of substep 1 went beside
      // the grid-op adjoint of substep 0) -- both grids all-zero again
      a.f = -1;
      LG_LAUNCH(lg_gadj_restore, gc2.x, (int)gc2.y, blk, 0, s, nb);
    }
  } else
  for (int f = S - 1; f >= -1; --f) {
    // list parity: cur = f & 1, "previous" = (f + 1) & 1 = the substep processed just before (f + 1) -- same rule as forward
    a.f = f;
    a.hist_in = ckpt + (long)max(f, 0) * rec;
    for (int g = 0; g < G; ++g) {
      const int Bg = grp[g].Bg;
      hipStream_t s = grp[g].s;
      a.b0 = grp[g].b0;
      const dim3 gc(lg_cell_blocks(L->cap), Bg), gs((lanes * N + LG_SCATTER_T - 1) / LG_SCATTER_T, Bg), gq((lanes * N + 255) / 256, Bg);
      if (gck) {          // the dense val grid is never touched: nothing to recompute; gacc is handed back all-zero
        LG_LAUNCH(lg_restore, gc.x, (int)gc.y, blk, 0, s);
        if (f < 0) continue;
      } else {
        LG_LAUNCH(lg_clear_fk, gc.x, (int)gc.y, blk, 0, s, 0, 1);
        if (f < 0) continue;
        if (lanes == 4) LG_LAUNCH(lg_p2g<4>, gs.x, (int)gs.y, blks, lg_table_bytes<4>(), s, 0); else LG_LAUNCH(lg_p2g<1>, gs.x, (int)gs.y, blks, lg_table_bytes<1>(), s, 0);
        LG_LAUNCH(lg_grid, gc.x, (int)gc.y, blk, 0, s, 1);
      }
      if (lanes == 4) LG_LAUNCH(lg_g2p_adj<4>, gs.x, (int)gs.y, blks, lg_table_bytes<4>(), s); else LG_LAUNCH(lg_g2p_adj<1>, gs.x, (int)gs.y, blks, lg_table_bytes<1>(), s);
      if (a.gck_off_crec) LG_LAUNCH(lg_grid_adj_rec, gc.x, (int)gc.y, blk, 0, s); else LG_LAUNCH(lg_grid_adj, gc.x, (int)gc.y, blk, 0, s);
      const dim3 gqf(gq.x + c.n_prim, Bg);   // + one block per primitive: the FK adjoint
      if (lanes == 4) LG_LAUNCH(lg_p2g_adj<4>, gqf.x, (int)gqf.y, blk, 0, s, (int)gq.x); else LG_LAUNCH(lg_p2g_adj<1>, gqf.x, (int)gqf.y, blk, 0, s, (int)gq.x);
    }
  }
  a.f = -1;
  for (int g = 0; g < G; ++g) {
    a.b0 = grp[g].b0;
    const dim3 gp((N + 255) / 256, grp[g].Bg);
    if (clip) hipLaunchKernelGGL(lg_bwd_norm, gp, blk, 0, grp[g].s, a);
    hipLaunchKernelGGL(lg_bwd_out, gp, blk, 0, grp[g].s, a, clip, gx0, gv0, gC0, gF0, gppos0, gfric, gmu, glam, gaction, grot0);
  }
  lg_join(L, G, st, grp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("ud_mpm_step_bwd (large path): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

}  // namespace ud

#ifdef UD_LG_STAMPS
extern "C" int ud_debug_lg_stamps(unsigned long long* out32, int reset) {   // diagnostic builds only: [4][8] counters
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(ud::ud_lg_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ud::ud_lg_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
