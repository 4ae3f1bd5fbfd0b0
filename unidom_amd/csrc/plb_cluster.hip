// PlasticineLab-style f64 MLS-MPM (GenORM Torus, BASELINE config 5): ONE persistent launch per step call and direction.
//
// What it replaces: the 2 launches per forward substep and 5 per reverse substep of plb.hip / plb_adj.hip for launches that do not fill
// the chip (8 envs x 1000 particles: rocprofv3 showed 23 + 6 us per forward substep and 48 us per reverse one, four of the seven kernels at
// the launch floor, 2.4-2.5x the algorithmic HBM bytes: profiles/r03g_kernel_stats_torus_grad_ngrid{64,128}.csv).  Reference:
// GenORM/policy/pbm/plb/engine/mpm_simulator.py step :438-449 (substeps x substep :256-268), substep_grad :271-289.
//
// Mapping (the design of mpm_cluster.h, in f64): an env is cut into W = ceil(N / 32) PARTS of 32 consecutive particles of the call's
// spatial order (Morton key of the base cell, plb_sort); one 256-lane workgroup per part, EIGHT lanes per particle (the octet splits the
// 27 stencil cells 4/4/4/3...), the particle state -- and in the backward its cotangent -- in REGISTERS for the whole launch; the part's
// cells in an LDS hash table of 1024 slots (>= 32 x 27: it cannot overflow).  What crosses a part boundary is the grid: per substep
// every part adds its table to a dense HBM grid with f64 atomics (memory-side on gfx950), the parts of the env meet at ONE barrier, and
// every part reads the summed cells of ITS OWN table back and runs the grid op on them (redundantly where parts share cells: same inputs,
// same result).  No active list, no stamps, no counters.
//   forward    pre-pass (SVD, return mapping) -> p2g into LDS -> flush -> BARRIER -> read back + grid op -> g2p from LDS        1 barrier / substep
//   backward   restore the part's cells from ITS records in the checkpoint -> grid op -> g2p adjoint into LDS -> flush of the v_out
//              cotangents -> BARRIER -> read back + grid-op adjoint -> p2g adjoint (gather from LDS) + particle adjoint            1 barrier / substep
// The forward keeps, per part and substep, the summed (m, mv) of the cells in its table (40 B each: key, slot, m, mv and the part's share
// of the cell's mass) next to the particle records and the SVD factors; the backward never scatters the forward grid again.  Per-env
// cotangents of the grid op (sticky-sphere positions, ground friction) are booked by every part that holds the cell, weighted with its
// share of the cell's mass (the shares of a cell sum to one) -- no owner election.  Grid buffers rotate by three so that nobody adds into a
// buffer another part may still be zeroing (flush f, read f after barrier f, zero f after barrier f +- 1, next flush three substeps on).
//
// Hand-off rules (MI355X_MICROARCH.md, inter-workgroup visibility) as in mpm_cluster.h: every access to data another part may have written
// inside this launch is an agent-scope operation (memory-side atomics, L1-bypassing loads, write-through zero stores); each wave drains its
// traffic (s_waitcnt vmcnt(0)) before the workgroup barrier in front of the arrival; one lane per part arrives with a returning atomic add,
// the last arriver publishes the phase in a generation word that the others poll (bounded: a part that gives up flags the handle's
// time-out counter, ud_plb_poll_timeouts; every wave still reaches the end of the kernel).  Progress needs every part of a launch resident
// at once: the host cuts a call into launches that fit (occupancy query x CUs at create), back to back on the caller's stream.
#include "plb_device.h"

namespace ud {

constexpr int PCL_PP = 32, PCL_LANES = 8, PCL_T = 256;
constexpr int PCL_H = 1024, PCL_LOGH = 10;      // >= PCL_PP * 27 = 864 cells a part can touch at the very most
constexpr int PCL_REC = PCL_PP * 27;            // cell records per part and substep (checkpoint)
constexpr int PCL_SMAX = 128;                   // substeps per step call the primitive trajectory in LDS is sized for
constexpr int PCL_BAR_STRIDE = 64, PCL_BAR_GEN = 32, PCL_BAR_EXIT = 48;
constexpr unsigned PCL_SPIN = 1u << 22;         // polls (~1 us each) before a part gives up

// Diagnostic build only (-DUD_PCL_STAMPS, tools/pcl_stamps.sh): s_memtime at the phase boundaries of the two kernels, summed over thread 0
// of every part into ud_pcl_stamps[kernel][phase] ([..][15] = parts); memory waits are forced at the stamps so that a load's latency is
// billed to the phase that issued it.
#ifdef UD_PCL_STAMPS
__device__ unsigned long long ud_pcl_stamps[2][16];
// the sums stay in thread 0's registers until the end of the kernel (an atomic per stamp would put its own round trip into the next phase)
#define PCL_STAMP_BEGIN unsigned long long pcl_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pcl_t0_ = __builtin_amdgcn_s_memtime();
#define PCL_STAMP(K, PH) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pcl_acc_[PH] += t_ - pcl_t0_; pcl_t0_ = t_; } } while (0)
#define PCL_STAMP_END(K) do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&ud_pcl_stamps[K][i_], pcl_acc_[i_]); atomicAdd(&ud_pcl_stamps[K][15], 1ull); } } while (0)
#else
#define PCL_STAMP_BEGIN
#define PCL_STAMP(K, PH) do {} while (0)
#define PCL_STAMP_END(K) do {} while (0)
#endif

struct PclCk {          // the caller's checkpoint of one step call, bound to pointers (plb_cluster_ckpt_layout)
  double* hist;         // [B][S+1][24][Np]  particle state at the start of every substep (+ the final one), in the call's spatial order
  double* pos;          // [B][S+1][np][3]   primitive trajectory
  int* perm;            // [B][Np]           slot p of hist holds the caller's particle perm[p]
  double* svd;          // [B][S][21][Np]    U, sig, Vh of every substep's F_tmp
  int* rec_cnt;         // [B][S][W]         cells in the part's table
  int2* rec_meta;       // [B][S][W][REC]    (linear cell index, table slot)
  double* rec_val;      // [B][S][W][REC][5] (m, mv xyz summed over the env; this part's share of m)
};

struct PclArgs {
  PlbConst c;
  long G;
  int W, Bl, b0;                       // parts per env; envs of this launch; first env of the launch inside the call
  const double *x, *v, *C, *F, *prim_pos, *softness, *action, *E, *nu, *ys;
  const int* order;                    // [B][Np] the handle's current spatial order (caller's particle index per slot), or null = identity
  double *xo, *vo, *Co, *Fo, *prim_o;  // forward outputs / backward: cotangents of the inputs (g_x0, g_v0, g_C0, g_F0, g_prim_pos0)
  const double *gx, *gv, *gC, *gF, *gpp;   // backward: cotangents of the step outputs (any may be null)
  double *g_action, *g_E, *g_nu, *g_ys, *g_fric;
  PclCk ck;
  int keep;                            // forward: a checkpoint is written
  double* cg[3];                       // [Bl][G][4] rotating exchange grids: (m, mv) forward, cotangent of v_out backward (rest state: zero)
  unsigned* bar;                       // [Bl][PCL_BAR_STRIDE]: arrival counter, generation word, exit counter (rest state: zero)
  double* gposacc;                     // [Bl][S+1][np][3] backward: cotangent of the primitive trajectory from the grid op (rest state: zero)
  double* gpar;                        // [Bl][4] backward: E, nu, yield stress, ground friction (rest state: zero)
  int* timeouts;                       // handle-wide count of parts that gave up waiting
};

// ---- agent-scope accesses ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pcl_ld(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void pcl_st(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pcl_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void pcl_decode(int W, int& bl, int& w) {   // ids congruent mod 8 share an XCD's L2 under round-robin placement (speed only)
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  bl = (j / W) * 8 + xcd;
  w = j % W;
}
__host__ inline int pcl_grid(int Bl, int W) { return 8 * W * ((Bl + 7) / 8); }

// The parts of an env meet (mpm_cluster.h::clm_barrier), in two halves so that work which needs nothing from the siblings can run between
// them: pcl_arrive -- every wave has drained its global traffic, one lane adds to the env's counter, the part whose add completes the count
// publishes the phase; pcl_wait -- that lane polls the generation word, the others join it at a workgroup barrier.  pcl_wait returns false
// once the env is dead (a part gave up).
__device__ __forceinline__ void pcl_arrive(unsigned* bar, unsigned phase, unsigned W) {
  pcl_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == phase * W) __hip_atomic_store(bar + PCL_BAR_GEN, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ bool pcl_wait(unsigned* bar, unsigned phase, int* s_dead) {
  if (threadIdx.x == 0) {
    for (unsigned spins = 0; __hip_atomic_load(bar + PCL_BAR_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase; ++spins) {
      if (spins > PCL_SPIN) { *s_dead = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  return *s_dead == 0;
}
__device__ __forceinline__ bool pcl_barrier(unsigned* bar, unsigned phase, unsigned W, int* s_dead) {
  pcl_arrive(bar, phase, W);
  return pcl_wait(bar, phase, s_dead);
}
// End of a launch: the last part of the env to leave puts the barrier words back to their rest state (no memset between launches).
__device__ __forceinline__ void pcl_exit(unsigned* bar, unsigned W) {
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(bar + PCL_BAR_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == W) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(bar + PCL_BAR_GEN, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(bar + PCL_BAR_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- the part's LDS cell table ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pcl_find(int* key, int cell) {      // insert-or-find; the table cannot fill up (PCL_H >= PCL_REC)
  unsigned s = plb_hash_t<PCL_LOGH>(cell);
  for (int probe = 0; probe < PCL_H; ++probe) {
    const int cur = key[s];
    if (cur == cell) return (int)s;
    if (cur == -1) {
      const int old = atomicCAS(&key[s], -1, cell);
      if (old == -1 || old == cell) return (int)s;
    }
    s = (s + 1) & (PCL_H - 1);
  }
  return 0;
}
__device__ __forceinline__ int pcl_lookup(const int* key, int cell) {   // a cell the walk of this substep (or the restored records) put there
  unsigned s = plb_hash_t<PCL_LOGH>(cell);
  for (int probe = 0; probe < PCL_H; ++probe) {
    if (key[s] == cell) return (int)s;
    s = (s + 1) & (PCL_H - 1);
  }
  return 0;
}
// occupied slots as a dense (key, slot) list; every thread calls it, it ends with a workgroup barrier (mpm_cluster.h::clm_compact)
__device__ __forceinline__ int pcl_compact(const int* key, int* klist, unsigned short* slist, int* s_n) {
  for (int s0 = 0; s0 < PCL_H; s0 += PCL_T) {
    const int sl = s0 + (int)threadIdx.x, k = key[sl];
    const bool occ = k >= 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
    const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(s_n, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (occ) { klist[base + pre] = k; slist[base + pre] = (unsigned short)sl; }
  }
  __syncthreads();
  return *s_n;
}

// primitive trajectory of the whole step into LDS (forward_kinematics :118-121 + set_velocity :185-192 as plb_pack writes it):
// P[s + 1] = clamp(P[s] + v), v = clip(action) / S for primitive 0
__device__ __forceinline__ void pcl_trajectory(const PclArgs& a, int b, double* s_pos, double* out /* checkpoint rows of this env, or null */) {
  const PlbConst& c = a.c;
  if ((int)threadIdx.x < c.np * 3) {
    const int pi = threadIdx.x / 3, d = threadIdx.x % 3;
    const double pv = (pi == 0) ? fmin(fmax(a.action[b * 3 + d], -1.0), 1.0) * 1.0 / (double)c.S : 0.0;
    double cur = a.prim_pos[(long)b * c.np * 3 + pi * 3 + d];
    s_pos[pi * 3 + d] = cur;
    if (out) out[pi * 3 + d] = cur;
    for (int s = 0; s < c.S; ++s) {
      cur = fmax(fmin(cur + pv, c.hi[d]), c.lo[d]);
      s_pos[((s + 1) * c.np + pi) * 3 + d] = cur;
      if (out) out[((s + 1) * c.np + pi) * 3 + d] = cur;
    }
  }
}

__device__ __forceinline__ double* pcl_hist(const PclArgs& a, int b, int slot) { return a.ck.hist + ((long)b * (a.c.S + 1) + slot) * 24 * a.c.Np; }
__device__ __forceinline__ long pcl_rec(const PclArgs& a, int b, int f, int w) { return (((long)b * a.c.S + f) * a.W + w) * PCL_REC; }

// ---- forward ----------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PCL_T) pcl_fwd_kernel(const PclArgs a) {
  __shared__ int s_key[PCL_H], s_klist[2][PCL_REC];
  __shared__ unsigned short s_slist[PCL_REC];
  __shared__ double s_val[4 * PCL_H];              // component-major (m, mv) sums of the part; after the read-back: m | v_out xyz of the env
  __shared__ double s_pos[(PCL_SMAX + 1) * 6];
  __shared__ int s_dead, s_n, s_poison;
  int bl, w;
  pcl_decode(a.W, bl, w);
  if (bl >= a.Bl) return;
  const PlbConst& c = a.c;
  const int b = a.b0 + bl, tid = threadIdx.x, p = w * PCL_PP + (tid >> 3), qi = tid & 7;
  const bool live = p < c.N;
  const int S = c.S;
  // a time-out nobody has polled yet (ud_plb_poll_timeouts) leaves barrier words and exchange grids of its env dirty: until the poll has
  // reset them, every launch on the handle declines to run -- its parts touch no barrier, no grid, and every output is NaN
  if (tid == 0) { s_dead = 0; s_poison = __hip_atomic_load(a.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
  pcl_trajectory(a, b, s_pos, (a.keep && w == 0) ? a.ck.pos + (long)b * (S + 1) * c.np * 3 : nullptr);
  const int up = live ? (a.order ? a.order[(long)b * c.Np + p] : p) : 0;
  double x[3] = {0, 0, 0}, v[3] = {0, 0, 0}, Cm[9], F[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[d] = 0; F[d] = (d % 4 == 0) ? 1.0 : 0.0; }
  if (live) {
    const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
    for (int d = 0; d < 3; ++d) { x[d] = a.x[o3 + d]; v[d] = a.v[o3 + d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { Cm[d] = a.C[o9 + d]; F[d] = a.F[o9 + d]; }
    if (a.keep && qi == 0) a.ck.perm[(long)b * c.Np + p] = up;
  }
  const double E = a.E[b], nu = a.nu[b], ys = a.ys[b];
  const double* soft = a.softness + b * c.np;
  unsigned* bar = a.bar + (long)bl * PCL_BAR_STRIDE;
  int nprev = 0;
  __syncthreads();
  bool alive = s_poison == 0;
  PCL_STAMP_BEGIN
  PCL_STAMP(0, 0);                                              // prologue: trajectory, state loads
  for (int f = 0; f < S && alive; ++f) {
    int* klist = s_klist[f & 1];
    const int* kprev = s_klist[(f + 1) & 1];
    double* gcur = a.cg[f % 3] + (long)bl * a.G * 4;
    double* gold = a.cg[(f + 2) % 3] + (long)bl * a.G * 4;    // substep f - 1's buffer
    // ---- table clear (after the first substep: only the slots the previous substep used -- s_slist still lists them), pre-pass ----
    if (f == 0) {
      for (int s = tid; s < PCL_H; s += PCL_T) { s_key[s] = -1; s_val[s] = 0.0; s_val[PCL_H + s] = 0.0; s_val[2 * PCL_H + s] = 0.0; s_val[3 * PCL_H + s] = 0.0; }
    } else {
      for (int e = tid; e < nprev; e += PCL_T) {
        const int s = s_slist[e];
        s_key[s] = -1; s_val[s] = 0.0; s_val[PCL_H + s] = 0.0; s_val[2 * PCL_H + s] = 0.0; s_val[3 * PCL_H + s] = 0.0;
      }
    }
    if (tid == 0) s_n = 0;
    PCL_STAMP(0, 9);                                            // (diagnostic split of phase 1) table clear
    int base[3] = {0, 0, 0};
    double fx[3], wgt[9];
    PlbPre q;
    if (live) {
      plb_weights_fwd(c, x, base, fx, wgt);
      plb_prepass(c, E, nu, ys, Cm, F, q, false);
      PCL_STAMP(0, 10);                                         // (diagnostic split) weights + pre-pass
      if (a.keep && qi == 0) {      // record f: the state this substep starts from, and the factors of its F_tmp
        double* ho = pcl_hist(a, b, f);
#pragma unroll
        for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = x[d]; ho[(3 + d) * c.Np + p] = v[d]; }
#pragma unroll
        for (int d = 0; d < 9; ++d) { ho[(6 + d) * c.Np + p] = Cm[d]; ho[(15 + d) * c.Np + p] = F[d]; }
        double* o = a.ck.svd + (((long)b * S + f) * 21) * c.Np + p;
#pragma unroll
        for (int i = 0; i < 9; ++i) { o[i * c.Np] = q.U[i]; o[(12 + i) * c.Np] = q.Vh[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) o[(9 + i) * c.Np] = q.sig[i];
      }
    }
    __syncthreads();
    PCL_STAMP(0, 1);                                            // table clear, pre-pass, record stores
    // ---- p2g into the table (the lane's four cells keep their slots for the gather below) ----
    const int rot = (p * PCL_LANES) % 27;     // staggered stencil walk: neighbouring particles never on the same slot at once
    int slot4[4] = {0, 0, 0, 0};
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int it = qi + PCL_LANES * t;
        if (it >= 27) break;
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const double weight = dsel3(wgt, 0, i) * dsel3(wgt, 1, j) * dsel3(wgt, 2, k);
        const double dp0 = ((double)i - fx[0]) * c.dx, dp1 = ((double)j - fx[1]) * c.dx, dp2 = ((double)k - fx[2]) * c.dx;
        const int sl = pcl_find(s_key, (int)plb_stencil_lin(c, base, cidx));
        slot4[t] = sl;
        __hip_atomic_fetch_add(&s_val[sl], weight * c.p_mass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 3; ++r)
          __hip_atomic_fetch_add(&s_val[(1 + r) * PCL_H + sl], weight * (c.p_mass * v[r] + q.aff[r * 3] * dp0 + q.aff[r * 3 + 1] * dp1 + q.aff[r * 3 + 2] * dp2),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    PCL_STAMP(0, 2);                                            // p2g walk
    const int n = pcl_compact(s_key, klist, s_slist, &s_n);
    PCL_STAMP(0, 3);                                            // compaction
    // ---- flush: four lanes per cell, one per component (a cell is 32 contiguous bytes) ----
    {
      const int r = tid & 3;
      for (int e = tid >> 2; e < n; e += PCL_T / 4)
        __hip_atomic_fetch_add(gcur + (long)klist[e] * 4 + r, s_val[r * PCL_H + s_slist[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PCL_STAMP(0, 4);                                            // flush atomics (drained)
    alive = pcl_barrier(bar, (unsigned)(f + 1), (unsigned)a.W, &s_dead);
    if (!alive) break;
    PCL_STAMP(0, 5);                                            // barrier
    // ---- read the summed cells back, (record,) grid op; zero the cells of substep f - 1 ----
    for (int e0 = 0; e0 < n; e0 += PCL_T) {
      const int e = e0 + tid;
      if (e < n) {
        const int k = klist[e], sl = s_slist[e];
        const double* cell = gcur + (long)k * 4;
        const double m = pcl_ld(cell), mv[3] = {pcl_ld(cell + 1), pcl_ld(cell + 2), pcl_ld(cell + 3)};
        if (a.keep) {
          const long r = pcl_rec(a, b, f, w) + e;
          a.ck.rec_meta[r] = make_int2(k, sl);
          double* o = a.ck.rec_val + r * 5;
          o[0] = m; o[1] = mv[0]; o[2] = mv[1]; o[3] = mv[2];
          o[4] = (m > 1e-12) ? s_val[sl] / m : 0.0;      // this part's share of the cell's mass (the shares of a cell sum to one)
        }
        double vv[3];
        plb_grid_cell(c, k, m, mv, s_pos + f * c.np * 3, soft, vv);
        s_val[sl] = m; s_val[PCL_H + sl] = vv[0]; s_val[2 * PCL_H + sl] = vv[1]; s_val[3 * PCL_H + sl] = vv[2];
      }
    }
    if (a.keep && tid == 0) a.ck.rec_cnt[((long)b * S + f) * a.W + w] = n;
    for (int e = tid; e < nprev; e += PCL_T) {
      double* cell = gold + (long)kprev[e] * 4;
      pcl_st(cell, 0.0); pcl_st(cell + 1, 0.0); pcl_st(cell + 2, 0.0); pcl_st(cell + 3, 0.0);
    }
    nprev = n;
    __syncthreads();
    PCL_STAMP(0, 6);                                            // read-back, records, grid op, zeroing
    // ---- g2p + advect ----
    if (live) {
      double nv[3] = {0, 0, 0}, nC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int it = qi + PCL_LANES * t;
        if (it >= 27) break;
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const int sl = slot4[t];
        const double g3[3] = {s_val[PCL_H + sl], s_val[2 * PCL_H + sl], s_val[3 * PCL_H + sl]};
        const double weight = dsel3(wgt, 0, i) * dsel3(wgt, 1, j) * dsel3(wgt, 2, k);
        const double dp[3] = {(double)i - fx[0], (double)j - fx[1], (double)k - fx[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g3[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4 * c.inv_dx * weight * g3[r] * dp[s2];
        }
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) nv[d] = plb_quad_sum<PCL_LANES>(nv[d]);
#pragma unroll
      for (int d = 0; d < 9; ++d) nC[d] = plb_quad_sum<PCL_LANES>(nC[d]);
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = fmax(fmin(x[d] + c.dt * nv[d], 1.0 - 3 * c.dx), 0.0); v[d] = nv[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { Cm[d] = nC[d]; F[d] = q.nF[d]; }
    }
    __syncthreads();   // s_key / s_val are rewritten by the next substep
    PCL_STAMP(0, 7);                                            // g2p
  }
  // the buffers go back all-zero: the cells of the last substep, once every part has read them
  if (alive) alive = pcl_barrier(bar, (unsigned)(S + 1), (unsigned)a.W, &s_dead);
  if (alive) {
    double* glast = a.cg[(S - 1) % 3] + (long)bl * a.G * 4;
    const int* kl = s_klist[(S - 1) & 1];
    for (int e = tid; e < nprev; e += PCL_T) {
      double* cell = glast + (long)kl[e] * 4;
      pcl_st(cell, 0.0); pcl_st(cell + 1, 0.0); pcl_st(cell + 2, 0.0); pcl_st(cell + 3, 0.0);
    }
    pcl_drain();
    __syncthreads();
    pcl_exit(bar, (unsigned)a.W);
  }
  if (live && qi == 0) {
    const double bad = alive ? 0.0 : __builtin_nan("");       // a part gave up: the env's outputs are invalid and say so
    const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.xo[o3 + d] = x[d] + bad; a.vo[o3 + d] = v[d] + bad; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { a.Co[o9 + d] = Cm[d] + bad; a.Fo[o9 + d] = F[d] + bad; }
    if (a.keep) {
      double* ho = pcl_hist(a, b, S);
#pragma unroll
      for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = x[d]; ho[(3 + d) * c.Np + p] = v[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { ho[(6 + d) * c.Np + p] = Cm[d]; ho[(15 + d) * c.Np + p] = F[d]; }
    }
  }
  if (w == 0 && tid < c.np * 3) a.prim_o[(long)b * c.np * 3 + tid] = s_pos[S * c.np * 3 + tid];   // copyframe(cur, 0)
  if (tid == 0 && s_dead) atomicAdd(a.timeouts, 1);
  PCL_STAMP(0, 8);                                              // last meeting, zeroing, outputs
  PCL_STAMP_END(0);
}

// ---- backward -----------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PCL_T) pcl_bwd_kernel(const PclArgs a) {
  __shared__ int s_key[PCL_H], s_klist[3][PCL_REC];
  __shared__ unsigned short s_slist[PCL_REC];
  __shared__ double s_mmv[4 * PCL_H];              // (m, mv) of the env's cells; after the grid-op adjoint: cotangents of (mv xyz, m)
  __shared__ double s_vout[3 * PCL_H];             // v_out of the substep being reversed
  __shared__ double s_gv[3 * PCL_H];               // cotangent of v_out: the part's sums, after the read-back the env's
  __shared__ double s_frac[PCL_H];
  __shared__ double s_pos[(PCL_SMAX + 1) * 6];
  __shared__ double s_red[3][PCL_T / 64];
  __shared__ int s_dead, s_poison;
  int bl, w;
  pcl_decode(a.W, bl, w);
  if (bl >= a.Bl) return;
  const PlbConst& c = a.c;
  const int b = a.b0 + bl, tid = threadIdx.x, p = w * PCL_PP + (tid >> 3), qi = tid & 7;
  const bool live = p < c.N;
  const int S = c.S;
  // a time-out nobody has polled yet (ud_plb_poll_timeouts) leaves barrier words and exchange grids of its env dirty: until the poll has
  // reset them, every launch on the handle declines to run -- its parts touch no barrier, no grid, and every output is NaN
  if (tid == 0) { s_dead = 0; s_poison = __hip_atomic_load(a.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
  for (int e = tid; e < (S + 1) * c.np * 3; e += PCL_T) s_pos[e] = a.ck.pos[(long)b * (S + 1) * c.np * 3 + e];
  const int up = live ? a.ck.perm[(long)b * c.Np + p] : 0;
  // cotangent of state f + 1, in registers for the whole launch (every lane of the octet holds the particle's)
  double gx1[3] = {0, 0, 0}, gv1[3] = {0, 0, 0}, gC1[9], gF1[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC1[d] = 0; gF1[d] = 0; }
  if (live) {
    const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx1[d] = a.gx ? a.gx[o3 + d] : 0.0; gv1[d] = a.gv ? a.gv[o3 + d] : 0.0; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC1[d] = a.gC ? a.gC[o9 + d] : 0.0; gF1[d] = a.gF ? a.gF[o9 + d] : 0.0; }
  }
  const double E = a.E[b], nu = a.nu[b], ys = a.ys[b];
  const double* soft = a.softness + b * c.np;
  unsigned* bar = a.bar + (long)bl * PCL_BAR_STRIDE;
  double* gposacc = a.gposacc + (long)bl * (S + 1) * c.np * 3;
  double* gpar = a.gpar + (long)bl * 4;
  double accE = 0, accNu = 0, accYs = 0;
  int nprev = 0;
  unsigned phase = 0;
  __syncthreads();
  bool alive = s_poison == 0;
  PCL_STAMP_BEGIN
  PCL_STAMP(1, 0);
  for (int f = S - 1; f >= 0 && alive; --f) {
    const int ring = (S - 1 - f) % 3;                          // buffers rotate in the order the substeps are reversed
    int* klist = s_klist[ring];
    const int* kprev = s_klist[(ring + 2) % 3];                // cells of substep f + 1
    double* gcur = a.cg[ring] + (long)bl * a.G * 4;
    double* gold = a.cg[(ring + 2) % 3] + (long)bl * a.G * 4;
    // ---- the part's cells of substep f from its records; particle state f and the SVD factors from the checkpoint ----
    if (f == S - 1) {
      for (int s = tid; s < PCL_H; s += PCL_T) { s_key[s] = -1; s_gv[s] = 0.0; s_gv[PCL_H + s] = 0.0; s_gv[2 * PCL_H + s] = 0.0; }
    } else {                                                   // only the slots substep f + 1 used (s_slist still lists them)
      for (int e = tid; e < nprev; e += PCL_T) {
        const int s = s_slist[e];
        s_key[s] = -1; s_gv[s] = 0.0; s_gv[PCL_H + s] = 0.0; s_gv[2 * PCL_H + s] = 0.0;
      }
    }
    const int n = a.ck.rec_cnt[((long)b * S + f) * a.W + w];
    __syncthreads();
    for (int e = tid; e < n; e += PCL_T) {
      const long r = pcl_rec(a, b, f, w) + e;
      const int2 ms = a.ck.rec_meta[r];
      const double* o = a.ck.rec_val + r * 5;
      klist[e] = ms.x; s_slist[e] = (unsigned short)ms.y;
      s_key[ms.y] = ms.x;
      s_mmv[ms.y] = o[0]; s_mmv[PCL_H + ms.y] = o[1]; s_mmv[2 * PCL_H + ms.y] = o[2]; s_mmv[3 * PCL_H + ms.y] = o[3];
      s_frac[ms.y] = o[4];
    }
    double x[3] = {0, 0, 0}, v[3] = {0, 0, 0}, v1[3] = {0, 0, 0}, Cm[9], F[9];
    PlbPre q;
    int base[3] = {0, 0, 0};
    double fx[3], wgt[9], dw[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) { Cm[d] = 0; F[d] = (d % 4 == 0) ? 1.0 : 0.0; }
    if (live) {
      const double* hi_ = pcl_hist(a, b, f);
      const double* ho = pcl_hist(a, b, f + 1);
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = hi_[d * c.Np + p]; v[d] = hi_[(3 + d) * c.Np + p]; v1[d] = ho[(3 + d) * c.Np + p]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { Cm[d] = hi_[(6 + d) * c.Np + p]; F[d] = hi_[(15 + d) * c.Np + p]; }
      const double* o = a.ck.svd + (((long)b * S + f) * 21) * c.Np + p;
#pragma unroll
      for (int i = 0; i < 9; ++i) { q.U[i] = o[i * c.Np]; q.Vh[i] = o[(12 + i) * c.Np]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) q.sig[i] = o[(9 + i) * c.Np];
      plb_weights(c, x, base, fx, wgt, dw);
    }
    __syncthreads();
    PCL_STAMP(1, 1);                                            // restore records, state + SVD loads
    // ---- grid op on the part's cells: v_out ----
    for (int e = tid; e < n; e += PCL_T) {
      const int sl = s_slist[e];
      const double mv[3] = {s_mmv[PCL_H + sl], s_mmv[2 * PCL_H + sl], s_mmv[3 * PCL_H + sl]};
      double vv[3];
      plb_grid_cell(c, klist[e], s_mmv[sl], mv, s_pos + f * c.np * 3, soft, vv);
      s_vout[sl] = vv[0]; s_vout[PCL_H + sl] = vv[1]; s_vout[2 * PCL_H + sl] = vv[2];
    }
    __syncthreads();
    PCL_STAMP(1, 2);                                            // grid op
    // ---- g2p adjoint (:234-253 in reverse): v_out cotangents into the table, the x cotangent that flows through g2p ----
    double gxs[3] = {0, 0, 0};
    const int rot = (p * PCL_LANES) % 27;
    int slot4[4] = {0, 0, 0, 0};
    if (live) {
      double gxp[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {   // x1 = clamp(x + dt v1, 0, 1 - 3 dx): the cotangent passes where the clamp is inactive
        const double xn = x[d] + c.dt * v1[d];
        const double pass = (xn >= 0.0 && xn <= 1.0 - 3 * c.dx) ? 1.0 : 0.0;
        gxp[d] = pass * gx1[d];
        gv1[d] += c.dt * gxp[d];
      }
      double gfx[3] = {0, 0, 0};
      const double k4 = 4 * c.inv_dx;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int it = qi + PCL_LANES * t;
        if (it >= 27) break;
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const double wi = dsel3(wgt, 0, i), wj = dsel3(wgt, 1, j), wk = dsel3(wgt, 2, k);
        const double weight = wi * wj * wk;
        const double dp[3] = {(double)i - fx[0], (double)j - fx[1], (double)k - fx[2]};
        const int sl = pcl_lookup(s_key, (int)plb_stencil_lin(c, base, cidx));
        slot4[t] = sl;
        const double g[3] = {s_vout[sl], s_vout[PCL_H + sl], s_vout[2 * PCL_H + sl]};
        double gw = 0, gdp[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double cd = gC1[r * 3] * dp[0] + gC1[r * 3 + 1] * dp[1] + gC1[r * 3 + 2] * dp[2];
          __hip_atomic_fetch_add(&s_gv[r * PCL_H + sl], weight * (gv1[r] + k4 * cd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          gw += g[r] * (gv1[r] + k4 * cd);
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) gdp[s2] += k4 * weight * gC1[r * 3 + s2] * g[r];
        }
        gfx[0] += gw * dsel3(dw, 0, i) * wj * wk - gdp[0];
        gfx[1] += gw * wi * dsel3(dw, 1, j) * wk - gdp[1];
        gfx[2] += gw * wi * wj * dsel3(dw, 2, k) - gdp[2];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) gxs[d] = gxp[d] + c.inv_dx * plb_quad_sum<PCL_LANES>(gfx[d]);
    }
    __syncthreads();
    PCL_STAMP(1, 3);                                            // g2p adjoint walk
    // ---- flush the v_out cotangents, meet, read the env's sums back; zero the cells of substep f + 1 ----
    {
      const int r = tid & 3;
      if (r < 3)
        for (int e = tid >> 2; e < n; e += PCL_T / 4)
          __hip_atomic_fetch_add(gcur + (long)klist[e] * 4 + r, s_gv[r * PCL_H + s_slist[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PCL_STAMP(1, 4);                                            // flush atomics (drained)
    // the particle pre-pass (stress, return mapping: what the p2g adjoint needs) asks nothing of the siblings: it runs while the arrivals
    // and the generation word travel
    pcl_arrive(bar, ++phase, (unsigned)a.W);
    if (live) plb_prepass(c, E, nu, ys, Cm, F, q, true);
    PCL_STAMP(1, 9);                                            // arrive + pre-pass
    alive = pcl_wait(bar, phase, &s_dead);
    if (!alive) break;
    PCL_STAMP(1, 5);                                            // wait
    // ---- grid-op adjoint per cell (every part that holds the cell: same inputs, same result); its per-env cotangents weighted with the
    // part's share of the cell's mass ----
    for (int e0 = 0; e0 < n; e0 += PCL_T) {
      const int e = e0 + tid;
      double qs[2][3] = {{0, 0, 0}, {0, 0, 0}}, gfric = 0;
      if (e < n) {
        const int k = klist[e], sl = s_slist[e];
        const double* cell = gcur + (long)k * 4;
        const double g[3] = {pcl_ld(cell), pcl_ld(cell + 1), pcl_ld(cell + 2)};
        const double mv[3] = {s_mmv[PCL_H + sl], s_mmv[2 * PCL_H + sl], s_mmv[3 * PCL_H + sl]};
        double ga[4];
        plb_grid_cell_adj(c, k, s_mmv[sl], mv, g, s_pos + f * c.np * 3, soft, ga, qs, gfric);
        s_mmv[sl] = ga[3]; s_mmv[PCL_H + sl] = ga[0]; s_mmv[2 * PCL_H + sl] = ga[1]; s_mmv[3 * PCL_H + sl] = ga[2];   // (gm | gmv xyz)
        const double fr = s_frac[sl];
        gfric *= fr;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) qs[pi][kk] *= fr;
      }
      const bool lead = (tid & 63) == 0;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        if (pi >= c.np) break;
        if (!__any(qs[pi][0] != 0.0 || qs[pi][1] != 0.0 || qs[pi][2] != 0.0)) continue;   // wave-uniform
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const double qq = plb_wave_sum(qs[pi][kk]);
          if (lead && qq != 0.0) {
            __hip_atomic_fetch_add(gposacc + ((f + 1) * c.np + pi) * 3 + kk, qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gposacc + (f * c.np + pi) * 3 + kk, -qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (__any(gfric != 0.0)) {
        const double qq = plb_wave_sum(gfric);
        if (lead && qq != 0.0) __hip_atomic_fetch_add(gpar + 3, qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (int e = tid; e < nprev; e += PCL_T) {
      double* cell = gold + (long)kprev[e] * 4;
      pcl_st(cell, 0.0); pcl_st(cell + 1, 0.0); pcl_st(cell + 2, 0.0);
    }
    nprev = n;
    __syncthreads();
    PCL_STAMP(1, 6);                                            // read-back, grid-op adjoint, per-env atomics, zeroing
    // ---- p2g adjoint (gather from LDS) + particle adjoint: the cotangent of state f replaces that of f + 1 ----
    if (live) {
      double gv[3] = {0, 0, 0}, gaff[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gfx[3] = {0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int it = qi + PCL_LANES * t;
        if (it >= 27) break;
        const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const double wi = dsel3(wgt, 0, i), wj = dsel3(wgt, 1, j), wk = dsel3(wgt, 2, k);
        const double weight = wi * wj * wk;
        const double dp[3] = {((double)i - fx[0]) * c.dx, ((double)j - fx[1]) * c.dx, ((double)k - fx[2]) * c.dx};
        const int sl = slot4[t];
        const double gm = s_mmv[sl], gmv[3] = {s_mmv[PCL_H + sl], s_mmv[2 * PCL_H + sl], s_mmv[3 * PCL_H + sl]};
        double gw = c.p_mass * gm, gdp[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          gw += gmv[r] * (c.p_mass * v[r] + q.aff[r * 3] * dp[0] + q.aff[r * 3 + 1] * dp[1] + q.aff[r * 3 + 2] * dp[2]);
          gv[r] += weight * c.p_mass * gmv[r];
#pragma unroll
          for (int s = 0; s < 3; ++s) { gaff[r * 3 + s] += weight * gmv[r] * dp[s]; gdp[s] += weight * gmv[r] * q.aff[r * 3 + s]; }
        }
        gfx[0] += gw * dsel3(dw, 0, i) * wj * wk - c.dx * gdp[0];
        gfx[1] += gw * wi * dsel3(dw, 1, j) * wk - c.dx * gdp[1];
        gfx[2] += gw * wi * wj * dsel3(dw, 2, k) - c.dx * gdp[2];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) { gv[d] = plb_quad_sum<PCL_LANES>(gv[d]); gfx[d] = plb_quad_sum<PCL_LANES>(gfx[d]); }
#pragma unroll
      for (int d = 0; d < 9; ++d) gaff[d] = plb_quad_sum<PCL_LANES>(gaff[d]);
      double gC0[9], gF0[9], e1, e2, e3;
      plb_particle_adjoint(c, E, nu, ys, q, F, gaff, gF1, gC0, gF0, e1, e2, e3);
      if (qi == 0) { accE += e1; accNu += e2; accYs += e3; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { gx1[d] = gxs[d] + c.inv_dx * gfx[d]; gv1[d] = gv[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { gC1[d] = gC0[d]; gF1[d] = gF0[d]; }
    }
    __syncthreads();   // the tables are rewritten by the next substep
    PCL_STAMP(1, 7);                                            // p2g adjoint gather + particle adjoint
  }
  // E / nu / yield-stress cotangents: one atomic per part and parameter, before the last meeting
  {
    double vals[3] = {accE, accNu, accYs};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double tot = plb_wave_sum(vals[k]);
      if ((tid & 63) == 0) s_red[k][tid >> 6] = tot;
    }
    __syncthreads();
    if (tid < 3) {
      double tot = 0;
      for (int k = 0; k < PCL_T / 64; ++k) tot += s_red[tid][k];
      __hip_atomic_fetch_add(gpar + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (alive) alive = pcl_barrier(bar, ++phase, (unsigned)a.W, &s_dead);
  if (alive) {
    double* glast = a.cg[(S - 1) % 3] + (long)bl * a.G * 4;      // substep 0 used ring (S - 1) % 3
    const int* kl = s_klist[(S - 1) % 3];
    for (int e = tid; e < nprev; e += PCL_T) {
      double* cell = glast + (long)kl[e] * 4;
      pcl_st(cell, 0.0); pcl_st(cell + 1, 0.0); pcl_st(cell + 2, 0.0);
    }
    // part 0: forward_kinematics.grad + set_action in reverse (pos[s + 1] = clamp(pos[s] + pv)), the per-env outputs; the accumulators go back to zero
    if (w == 0) {
      if (tid < c.np * 3) {
        const int pi = tid / 3, d = tid % 3;
        const double raw = (pi == 0) ? a.action[b * 3 + d] : 0.0;
        const double pv = (pi == 0) ? fmin(fmax(raw, -1.0), 1.0) / (double)S : 0.0;
        double gpv = 0.0;
        double tot = pcl_ld(gposacc + (S * c.np + pi) * 3 + d) + (a.gpp ? a.gpp[((long)b * c.np + pi) * 3 + d] : 0.0);
        pcl_st(gposacc + (S * c.np + pi) * 3 + d, 0.0);
        for (int s = S - 1; s >= 0; --s) {
          const double un = s_pos[(s * c.np + pi) * 3 + d] + pv;
          const double pass = (un >= c.lo[d] && un <= c.hi[d]) ? 1.0 : 0.0;
          const double g = pass * tot;
          gpv += g;
          tot = pcl_ld(gposacc + (s * c.np + pi) * 3 + d) + g;
          pcl_st(gposacc + (s * c.np + pi) * 3 + d, 0.0);
        }
        if (a.prim_o) a.prim_o[((long)b * c.np + pi) * 3 + d] = tot;
        if (pi == 0 && a.g_action) a.g_action[b * 3 + d] = (raw >= -1.0 && raw <= 1.0) ? gpv / (double)S : 0.0;
      }
      if (tid >= 64 && tid < 68) {
        const int k = tid - 64;
        const double val = pcl_ld(gpar + k);
        pcl_st(gpar + k, 0.0);
        double* dst = (k == 0) ? a.g_E : ((k == 1) ? a.g_nu : ((k == 2) ? a.g_ys : a.g_fric));
        if (dst) dst[b] = val;
      }
    }
    pcl_drain();
    __syncthreads();
    pcl_exit(bar, (unsigned)a.W);
  }
  if (live && qi == 0) {
    const double bad = alive ? 0.0 : __builtin_nan("");
    const long o3 = ((long)b * c.N + up) * 3, o9 = ((long)b * c.N + up) * 9;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.xo[o3 + d] = gx1[d] + bad; a.vo[o3 + d] = gv1[d] + bad; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { a.Co[o9 + d] = gC1[d] + bad; a.Fo[o9 + d] = gF1[d] + bad; }
  }
  if (tid == 0 && s_dead) atomicAdd(a.timeouts, 1);
  PCL_STAMP(1, 8);
  PCL_STAMP_END(1);
}

}  // namespace ud

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct PclCkOff { size_t hist, pos, perm, svd, rec_cnt, rec_meta, rec_val, total; };
static PclCkOff pcl_ckpt_layout(const ud::PlbConst& c, int W, int B) {
  PclCkOff k{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  k.hist = take((size_t)B * (c.S + 1) * 24 * c.Np * 8);
  k.pos = take((size_t)B * (c.S + 1) * c.np * 3 * 8);
  k.perm = take((size_t)B * c.Np * 4);
  k.svd = take((size_t)B * c.S * 21 * c.Np * 8);
  k.rec_cnt = take((size_t)B * c.S * W * 4);
  k.rec_meta = take((size_t)B * c.S * W * ud::PCL_REC * 8);
  k.rec_val = take((size_t)B * c.S * W * ud::PCL_REC * 40);
  k.total = off;
  return k;
}
static void pcl_bind_ckpt(ud::PclArgs& a, const ud::PlbConst& c, int W, int B, void* ckpt) {
  const PclCkOff k = pcl_ckpt_layout(c, W, B);
  char* base = (char*)ckpt;
  a.ck.hist = (double*)(base + k.hist); a.ck.pos = (double*)(base + k.pos); a.ck.perm = (int*)(base + k.perm); a.ck.svd = (double*)(base + k.svd);
  a.ck.rec_cnt = (int*)(base + k.rec_cnt); a.ck.rec_meta = (int2*)(base + k.rec_meta); a.ck.rec_val = (double*)(base + k.rec_val);
}

size_t plb_cluster_ckpt_bytes(const ud_plb* h, int B) { return pcl_ckpt_layout(h->c, h->cl.W, B).total; }

// Does the persistent path fit this handle?  Every part of a launch must be resident at once: envs per launch = resident workgroups of the
// chip / parts per env, for the kernel with the smaller occupancy (the backward: ~115 KB of LDS per part); and the six rotating exchange
// grids of a launch's envs must fit the budget.  Returns envs per launch (0 = does not fit).
int plb_cluster_plan(ud_plb* h, int max_envs) {
  const ud::PlbConst& c = h->c;
  PlbCluster& cl = h->cl;
  cl.W = (c.N + ud::PCL_PP - 1) / ud::PCL_PP;
  if (c.S > ud::PCL_SMAX) return 0;
  int dev = 0, n_cu = 0, occ_f = 0, occ_b = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_f, (const void*)ud::pcl_fwd_kernel, ud::PCL_T, 0) != hipSuccess) occ_f = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, (const void*)ud::pcl_bwd_kernel, ud::PCL_T, 0) != hipSuccess) occ_b = 0;
  (void)hipGetLastError();
  const int occ = std::min(occ_f, occ_b);
  if (occ < 1 || n_cu < 8) return 0;
  int per = (int)(((long)n_cu * occ) / cl.W);
  if (per >= 8) per = per / 8 * 8;                              // whole XCD rounds (speed only)
  const size_t budget = (size_t)24 << 30;                        // bytes of exchange grids a handle may hold
  const size_t per_env = (size_t)3 * h->G * 32;
  per = (int)std::min<size_t>((size_t)per, budget / per_env);
  return std::min(per, max_envs);
}

int plb_cluster_reserve(ud_plb* h, int per) {
  const ud::PlbConst& c = h->c;
  PlbCluster& cl = h->cl;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  size_t o_cg[3];
  for (int i = 0; i < 3; ++i) o_cg[i] = take((size_t)per * h->G * 32);
  const size_t o_bar = take((size_t)per * ud::PCL_BAR_STRIDE * 4), o_gpos = take((size_t)per * (c.S + 1) * c.np * 3 * 8 + 64), o_gpar = take((size_t)per * 4 * 8);
  const size_t o_to = take(64);
  hipError_t e = hipMalloc(&cl.arena, off);
  if (e != hipSuccess) { ud::set_error("ud_plb_create (persistent path): hipMalloc(%zu MB) failed: %s", off >> 20, hipGetErrorString(e)); return UD_ERR_HIP; }
  e = hipMemset(cl.arena, 0, off);
  if (e != hipSuccess) { ud::set_error("ud_plb_create (persistent path): memset failed"); return UD_ERR_HIP; }
  char* base = (char*)cl.arena;
  for (int i = 0; i < 3; ++i) cl.cg[i] = (double*)(base + o_cg[i]);
  cl.bar = (unsigned*)(base + o_bar); cl.gposacc = (double*)(base + o_gpos); cl.gpar = (double*)(base + o_gpar); cl.timeouts = (int*)(base + o_to);
  cl.bytes = off; cl.per = per;
  return UD_OK;
}

static ud::PclArgs pcl_args(ud_plb* h, const double* softness, const double* action, const double* E, const double* nu, const double* ys) {
  ud::PclArgs a{};
  a.c = h->c; a.G = h->G; a.W = h->cl.W;
  a.softness = softness; a.action = action; a.E = E; a.nu = nu; a.ys = ys;
  for (int i = 0; i < 3; ++i) a.cg[i] = h->cl.cg[i];
  a.bar = h->cl.bar; a.gposacc = h->cl.gposacc; a.gpar = h->cl.gpar; a.timeouts = h->cl.timeouts;
  return a;
}

int plb_cluster_step_fwd(ud_plb* h, int B, const double* x, const double* v, const double* C, const double* F, const double* prim_pos,
                         const double* softness, const double* action, const double* E, const double* nu, const double* ys, double* xo,
                         double* vo, double* Co, double* Fo, double* prim_o, const int* order, void* ckpt, hipStream_t st) {
  ud::PclArgs a = pcl_args(h, softness, action, E, nu, ys);
  a.x = x; a.v = v; a.C = C; a.F = F; a.prim_pos = prim_pos; a.order = order;
  a.xo = xo; a.vo = vo; a.Co = Co; a.Fo = Fo; a.prim_o = prim_o;
  a.keep = ckpt ? 1 : 0;
  if (ckpt) pcl_bind_ckpt(a, h->c, h->cl.W, B, ckpt);
  for (int b0 = 0; b0 < B; b0 += h->cl.per) {
    a.b0 = b0; a.Bl = std::min(h->cl.per, B - b0);
    hipLaunchKernelGGL(ud::pcl_fwd_kernel, dim3(ud::pcl_grid(a.Bl, a.W)), dim3(ud::PCL_T), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_step_fwd (persistent path): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

int plb_cluster_step_bwd(ud_plb* h, int B, const void* ckpt, const double* softness, const double* action, const double* E, const double* nu,
                         const double* ys, const double* g_x, const double* g_v, const double* g_C, const double* g_F, const double* g_prim_pos,
                         double* g_x0, double* g_v0, double* g_C0, double* g_F0, double* g_prim_pos0, double* g_action, double* g_E, double* g_nu,
                         double* g_ys, double* g_fric, hipStream_t st) {
  ud::PclArgs a = pcl_args(h, softness, action, E, nu, ys);
  a.gx = g_x; a.gv = g_v; a.gC = g_C; a.gF = g_F; a.gpp = g_prim_pos;
  a.xo = g_x0; a.vo = g_v0; a.Co = g_C0; a.Fo = g_F0; a.prim_o = g_prim_pos0;
  a.g_action = g_action; a.g_E = g_E; a.g_nu = g_nu; a.g_ys = g_ys; a.g_fric = g_fric;
  pcl_bind_ckpt(a, h->c, h->cl.W, B, const_cast<void*>(ckpt));
  for (int b0 = 0; b0 < B; b0 += h->cl.per) {
    a.b0 = b0; a.Bl = std::min(h->cl.per, B - b0);
    hipLaunchKernelGGL(ud::pcl_bwd_kernel, dim3(ud::pcl_grid(a.Bl, a.W)), dim3(ud::PCL_T), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ud::set_error("ud_plb_step_bwd (persistent path): %s", hipGetErrorString(e)); return UD_ERR_HIP; }
  return UD_OK;
}

#ifdef UD_PCL_STAMPS
extern "C" int ud_debug_pcl_stamps(unsigned long long* out32, int reset) {   // diagnostic builds only: [2][16] counters
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(ud::ud_pcl_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ud::ud_pcl_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
