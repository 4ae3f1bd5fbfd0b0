// Persistent per-env CLUSTER forward of the many-workgroup MPM path: all `steps` substeps of a simulator.step in ONE launch
// (included by mpm_large.hip, which holds the helpers it uses; same C ABI, same checkpoint layout).  The backward of these bodies is
// the multi-kernel one, restoring the grid from the checkpoint this kernel writes (a persistent backward was built in round 3 and
// measured no faster; removed in round 4).
//
// What it replaces: the 4 (forward) / 4-6 (backward) launches per substep of mpm_large.hip for launches that do not fill the chip
// (B x N < 100 k particles: shape_rope 582, the rope at n_grid 128 798, pour_water 702 particles per env, 32 envs).  Those
// kernels sit at 4-14 us each on chains of dependent global round trips (state load -> SVD -> LDS table -> flush atomics ->
// bitmap OR -> list counter -> list store -> next launch: count -> list -> cell -> ...); rocprofv3: 32 + 41 us of kernel time
// per substep pair on shape_rope (profiles/r02j_kernel_stats_shape_rope.csv).  Reference: mpm_simulator.py:413-429 (step =
// fori_loop over substep), :223-330 (substep), :178-221 (p2g / g2p), :332-363 (what the backward differentiates).
//
// Mapping.  An env is cut into W = ceil(N / (T / 4)) PARTS of T / 4 = 16 or 32 consecutive particles (lattice order, or the Morton order of
// lg_sort): one T-lane workgroup per part, 4 lanes per particle (the quad splits the 27 stencil cells 7/7/7/6), the particle
// state in REGISTERS for the whole launch (HBM sees it once per substep, as the checkpoint record), the part's cells in the LDS
// staging table of lg_p2g (8 x 8 x 8 window when the part's base cells span <= 6 per axis, open addressing otherwise).
// What crosses a part boundary is the grid: per substep every part adds its table to a dense HBM grid with float atomics
// (memory-side on gfx950: nothing lives in an L2), the parts of the env meet at ONE barrier, and every part reads the summed
// (m, mv) of ITS OWN cells back, runs the grid op on them (redundantly where parts share cells: same inputs, same result) and
// gathers from LDS.  No active list, no bitmap, no counters: a part only ever asks for the cells in its own table.
//   forward   p2g -> flush -> BARRIER -> read back + grid op -> g2p                                   1 barrier / substep
// Grid buffers rotate by three so that nobody adds into a buffer another part may still be zeroing (flush f, read f after barrier f,
// zero f after barrier f + 1 with the keys kept from substep f, next flush f + 3).
// A part whose particles touch more cells than its 512-slot table holds (32 particles can touch 864; a compact part touches 40-150)
// does not fail: what does not fit goes to the env's HBM grid with atomics directly (clm_scatter), is read back and put through the
// grid op cell by cell in the gather, and is zeroed from a per-part spill list two substeps later; such an env's grid checkpoint is
// incomplete, so it is flagged (status bit 0) and that step's backward recomputes the grid.  Only a spill list that overflows too
// (CLM_SPILL cells per part and substep) is an error (status bit 1).
//
// Hand-off rules (MI355X_MICROARCH.md, inter-workgroup visibility): per-XCD L2s are not coherent and a CU's L1 is never
// refreshed, so EVERY access to data another part may have written inside this launch is an agent-scope operation: float /
// integer atomics (performed at the memory side), sc1 write-through stores for the zeroing, sc1 (L1-bypassing) loads for the
// read-back and the poll.  Each wave drains its own stores / atomics (s_waitcnt vmcnt(0)) before the workgroup barrier in front
// of the arrival; one lane per part arrives with an agent-scope atomic add on the env's monotonic counter, the last arriver
// publishes the phase in a generation word that the others poll with sc1 loads (bounded: ~seconds, then the env is flagged in status[] = 2 and every wave still reaches the end of the kernel); the
// other waves read after a workgroup barrier that lane joins.  Parts of one env get block ids congruent mod 8 (usually one XCD;
// speed only).  Progress needs every part of a launch resident at once: the host cuts a call into launches that fit
// (occupancy query x CUs), back to back on the caller's stream.
#pragma once

namespace ud {

// lanes per part: a template parameter T of the kernels, 4 lanes per particle.  T = 64 (one wave, 16 particles: 16 x 27 = 432
// cells at most, so the 512-slot table can never overflow and the workgroup barriers are free) or 128 (32 particles: half the
// parts per env and less duplicated grid work, but a part whose particles are spread over more than 512 cells is an error).
constexpr int CLM_H = 512, CLM_LOGH = 9;   // staging-table slots per part (= LgTable<4>)
constexpr int CLM_SPILL = 384;             // cells per part and substep that may go past the table (864 - 512 = 352 at the very most)
constexpr unsigned CLM_SPIN = 1u << 22;    // polls (~1 us each) before a part gives up

struct ClusterGrid {
  float4* cg[3];      // [Bl][G] (m, mv), rotating
  int* own[3];        // [Bl][G] smallest part number that touched the cell this substep (the part that books its parameter cotangents)
  unsigned* bar;      // [Bl][CLM_BAR_STRIDE]: arrival counter (word 0) and generation word (word CLM_BAR_GEN) per env, zeroed before every launch
  int W, Bl;          // parts per env; envs of this launch (a.b0 = the first one)
};

// ---- agent-scope accesses -------------------------------------------------------------------------------------------
// (ldc / stc for single floats: mpm_large.hip)
__device__ __forceinline__ int ldci(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 ldc4(const float4* p) {
  const unsigned long long* q = (const unsigned long long*)p;
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__builtin_bit_cast(float, (unsigned)a), __builtin_bit_cast(float, (unsigned)(a >> 32)),
                     __builtin_bit_cast(float, (unsigned)b), __builtin_bit_cast(float, (unsigned)(b >> 32)));
}
__device__ __forceinline__ void stc4_zero(float4* p) {
  unsigned long long* q = (unsigned long long*)p;
  __hip_atomic_store(q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void clm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// blockIdx -> (env inside the launch, part): ids congruent mod 8 share an XCD's L2 under round-robin placement (speed only)
__device__ __forceinline__ void clm_decode(int W, int& bl, int& w) {
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  bl = (j / W) * 8 + xcd;
  w = j % W;
}
__host__ inline int clm_grid(int Bl, int W) { return 8 * W * ((Bl + 7) / 8); }

// The parts of an env meet.  Every wave has drained its global traffic; ONE lane per part arrives with a returning agent-scope
// atomic add on the env's counter; the part whose add completes the count publishes the phase number in the env's generation
// word (sc1 store), the others poll THAT word with sc1 loads.  (First version: every part polled the counter itself.  A word that
// memory-side atomics keep rewriting is never L2-resident, so 800-1600 pollers -- all envs' counters in two cache lines -- became
// a request storm on one memory channel: time per substep doubled with the number of parts, and removing all grid traffic
// changed nothing (profiles/r03b_abl_cluster.txt).  The generation word is written once per barrier and polled out of the L2.)
// Counter and generation word of an env sit on their own 128-byte lines (CLM_BAR_STRIDE words per env).
// Returns false once the env is dead (a part gave up).
constexpr int CLM_BAR_STRIDE = 64, CLM_BAR_GEN = 32;
// idx_out / cnt (forward with a grid checkpoint): the part that completes the count also notes how many grid records the env has
// written so far -- every part's appends precede its arrival, so that is where the next substep's records begin.
__device__ __forceinline__ bool clm_barrier(unsigned* bar, unsigned phase, unsigned W, int* s_dead, int* idx_out = nullptr, const int* cnt = nullptr) {
  clm_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == phase * W) {
      if (idx_out) { *idx_out = ldci(cnt); clm_drain(); }
      __hip_atomic_store(bar + CLM_BAR_GEN, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (unsigned spins = 0; __hip_atomic_load(bar + CLM_BAR_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase; ++spins) {
        if (spins > CLM_SPIN) { *s_dead = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
  return *s_dead == 0;
}

// slot of `cell` in the part's table: the window's arithmetic, or open addressing probed over the WHOLE table (bt_slot gives up
// after 64 probes: fine where a miss falls back to HBM atomics, not here); -1 = the table is full
__device__ __forceinline__ int clm_find(const BlockTable& t, const BlockWin& w, int cell) {
  if (w.on) {
    const int s = bt_win_slot(w, cell);
    if (s >= 0) t.key[s] = cell;
    return s;
  }
  unsigned s = lg_hash<CLM_LOGH>(cell);
  for (int probe = 0; probe < CLM_H; ++probe) {
    const int cur = t.key[s];
    if (cur == cell) return (int)s;
    if (cur == -1) {
      const int old = atomicCAS(&t.key[s], -1, cell);
      if (old == -1 || old == cell) return (int)s;
    }
    s = (s + 1) & (CLM_H - 1);
  }
  return -1;
}
// read-only slot of a cell: clm_lookup_fast for a substep without spills (the walk has put every cell of the part into the table);
// clm_lookup returns -1 for a cell that went past the table
__device__ __forceinline__ int clm_lookup_fast(const int* key, const BlockWin& w, int cell) {
  if (w.on) return max(bt_win_slot(w, cell), 0);
  unsigned s = lg_hash<CLM_LOGH>(cell);
  for (int probe = 0; probe < CLM_H; ++probe) {
    if (key[s] == cell) return (int)s;
    s = (s + 1) & (CLM_H - 1);
  }
  return 0;
}
__device__ __forceinline__ int clm_lookup(const int* key, const BlockWin& w, int cell) {
  if (w.on) return bt_win_slot(w, cell);
  unsigned s = lg_hash<CLM_LOGH>(cell);
  for (int probe = 0; probe < CLM_H; ++probe) {
    const int k = key[s];
    if (k == cell) return (int)s;
    if (k == -1) return -1;
    s = (s + 1) & (CLM_H - 1);
  }
  return -1;
}

// a cell the part's table has no room for: remembered (for the gather's direct read and for the zeroing two substeps later); false = the
// spill list is full too
__device__ __forceinline__ bool clm_spill_note(int* spill, int* s_nsp, int cell) {
  const int e = atomicAdd(s_nsp, 1);
  if (e < CLM_SPILL) { spill[e] = cell; return true; }
  return false;
}

// p2g of one quad lane into the part's table (the walk of lg_p2g<4>).  A cell that finds no slot is added to the env's HBM grid directly
// (gcur) and noted in the part's spill list.  Returns false when the spill list overflowed as well (the only failure left).
__device__ __forceinline__ bool clm_scatter(const MpmConst& c, const BlockTable& bt, const BlockWin& win, const Pre& q, const float* v,
                                            int p, int qi, float4* gcur, int* spill, int* s_nsp) {
  constexpr int TH = CLM_H;
  bool ok = true;
  const bool interior = win.on && q.base[0] >= 0 && q.base[1] >= 0 && q.base[2] >= 0 &&
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
    const int rot9 = (p * 4) % 9;
#pragma unroll 1
    for (int it = qi; it < 9; it += 4) {
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
    return true;
  }
  const int rot = (p * 4) % 27;
#pragma unroll 1
  for (int it = qi; it < 27; it += 4) {
    const int cidx = it + rot >= 27 ? it + rot - 27 : it + rot;
    const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
    const float weight = sel3(q.w, 0, i) * sel3(q.w, 1, j) * sel3(q.w, 2, k);
    const int sc = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
    const int gc = cell_gather(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
    if (sc >= 0) {
      const float dp0 = ((float)i - q.fx[0]) * c.dx, dp1 = ((float)j - q.fx[1]) * c.dx, dp2 = ((float)k - q.fx[2]) * c.dx;
      const int sl = clm_find(bt, win, sc);
      float ad[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) ad[r] = q.affine[r * 3] * dp0 + q.affine[r * 3 + 1] * dp1 + q.affine[r * 3 + 2] * dp2;
      if (sl >= 0) {
        __hip_atomic_fetch_add(&bt.val[sl], (double)(weight * c.p_mass), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 3; ++r)
          __hip_atomic_fetch_add(&bt.val[(1 + r) * TH + sl], (double)(weight * (c.p_mass * v[r] + ad[r])), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {                                    // no room in the table: straight to the env's grid
        float* cell = (float*)(gcur + cell_lin(c, sc));
        atomicAdd(cell, weight * c.p_mass);
#pragma unroll
        for (int r = 0; r < 3; ++r) atomicAdd(cell + 1 + r, weight * (c.p_mass * v[r] + ad[r]));
        ok = clm_spill_note(spill, s_nsp, sc) && ok;
      }
    }
    if (gc != sc && clm_find(bt, win, gc) < 0) ok = clm_spill_note(spill, s_nsp, gc) && ok;   // Q5: a clamped gather cell takes part with m = 0
  }
  return ok;
}

// A window only holds cells inside it: a part whose window is on may still own particles whose stencil leaves it (wrap / clamp
// at the domain edge).  Such cells return slot -1 from bt_find and would be lost -> the part falls back to the hash for the
// substep.  (lg_p2g sends them to HBM atomics instead; here there is no list to put them on.)
__device__ __forceinline__ bool clm_stencil_in_window(const MpmConst& c, const BlockWin& win, const int* base) {
  bool in = true;
#pragma unroll
  for (int i = 0; i < 3; i += 2)
#pragma unroll
    for (int j = 0; j < 3; j += 2)
#pragma unroll
      for (int k = 0; k < 3; k += 2) {
        const int sc = cell_scatter(c, base[0] + i, base[1] + j, base[2] + k), gc = cell_gather(c, base[0] + i, base[1] + j, base[2] + k);
        in = in && (sc < 0 || bt_win_slot(win, sc) >= 0) && bt_win_slot(win, gc) >= 0;
      }
  return in;
}

// the grid op of one cell as lg_grid_cell runs it: (m, mv) -> velocity after gravity, primitives, ground friction, boundary
__device__ __forceinline__ void clm_grid_op(const LargeArgs& a, int b, int f, int key, const float4& mv, float* vo) {
  int ci, cj, ck;
  decode_cell(a.c, key, ci, cj, ck);
  const float mvv[3] = {mv.y, mv.z, mv.w};
  if (a.c.position_control) {
    PrimF pf;
    float pv[3];
    load_prim_f(a, b, f, pf, pv);
    grid_op<false>(a.c, pf, ci, cj, ck, mv.x, mvv, vo, nullptr);
  } else {
    float v0[3], v1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) v0[d] = ((mv.x > 0.f) ? mvv[d] / mv.x : mvv[d]) + a.c.dtg[d];
    const float gp[3] = {(float)ci * a.c.dx, (float)cj * a.c.dx, (float)ck * a.c.dx};
#pragma unroll 1
    for (int ip = 0; ip < a.c.n_prim; ++ip) {
      PrimC pc;
      load_primc_f(a, b, ip, f, pc);
      CollideRec cr;
      collide_cell(pc, a.c.dt, gp, v0, v1, cr);
#pragma unroll
      for (int d = 0; d < 3; ++d) v0[d] = v1[d];
    }
    grid_tail<false>(a.c, a.friction[b], ci, cj, ck, v1, vo, nullptr);
  }
}

// forward kinematics of the whole step, once per launch (the per-substep form of primitives.py:185-194 is a recurrence on rows
// nothing else reads): row 0 = clip(in[0]); row f + 1 = clip(P'[f] + v) with P'[0] = in[0] (read before its clip) and P'[f] =
// row f afterwards; rotation[f + 1] = normalise(qmul(w2quat(w), rotation[f])).  Rows f and f + 1 are final when the grid op of
// substep f reads them -- exactly what lg_clear_fk's per-substep update gives it.  One block per (env, primitive).
__global__ void __launch_bounds__(64) lg_fk_all(LargeArgs a) {
  const int b = blockIdx.x + a.b0, ip = blockIdx.y, S = a.c.steps, tid = threadIdx.x;
  if (a.ls3 && ip == 0 && tid >= 48 && tid < 52) {   // multi-kernel forward without a clear launch: the three list counts, the first record index
    if (tid < 51) a.w.count[(tid - 48) * a.B + b] = 0;
    else if (a.gck_base) gck_idx(a, b)[0] = 0;
  }
  const long bp = (long)b * a.c.n_prim + ip;
  float* pp = a.w.ppos + bp * S * 3;
  float* pr = a.w.prot + bp * S * 4;
  if (tid < 3) {
    const float pva = clipf(a.action[bp * 6 + tid], -1.f, 1.f) * 1.f / (float)S;
    float prev = pp[tid];
    pp[tid] = clipf(prev, -2.f, 2.f);
    for (int f = 0; f + 1 < S; ++f) {
      const float nxt = clipf(prev + pva, -2.f, 2.f);
      pp[(f + 1) * 3 + tid] = nxt;
      prev = nxt;
    }
  }
  if (tid == 32) {
    float pw[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pw[d] = clipf(a.action[bp * 6 + 3 + d], -1.f, 1.f) * 1.f / (float)S;
    const float ang = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) + 1e-12f;
    const float sn = sinf(ang / 2.f);
    const float q[4] = {cosf(ang / 2.f), pw[0] / ang * sn, pw[1] / ang * sn, pw[2] / ang * sn};
    float r[4] = {pr[0], pr[1], pr[2], pr[3]};
    for (int f = 0; f + 1 < S; ++f) {
      const float o0 = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
      const float o1 = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
      const float o2 = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
      const float o3 = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
      const float nn = clipf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3), 1e-12f, INFINITY);
      r[0] = o0 / nn; r[1] = o1 / nn; r[2] = o2 / nn; r[3] = o3 / nn;
      float* w = pr + (f + 1) * 4;
      w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; w[3] = r[3];
    }
  }
}

// ---- the part's occupied cells as a dense list -----------------------------------------------------------------------------
// The table is sparse (a compact part fills 40-150 of its 512 slots) and everything that follows the walk -- flush, read-back + grid op,
// zeroing, the grid-op adjoint -- is per CELL: swept over the 512 slots, each trip of those loops cost its whole body (the grid
// op with its primitive loads, the collide chains) for a handful of live lanes, T = 64 twice as many trips as T = 128; rocprofv3
// counted 3 024 VALU instructions per wave and substep in the forward, half of the kernel's time (a round-3 counter pass; the file was not kept).
// One compaction pass per substep (ballot + mbcnt, one LDS add per wave and trip) leaves (key, slot) pairs; the cell loops then
// make ceil(n / T) trips, normally one.  Every thread of the part calls this; it ends with a workgroup barrier.
template <int T>
__device__ __forceinline__ int clm_compact(const int* key, int* klist, int* slist, int* s_n) {
  for (int s0 = 0; s0 < CLM_H; s0 += T) {
    const int sl = s0 + (int)threadIdx.x, k = key[sl];
    const bool occ = k >= 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
    const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(s_n, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (occ) { klist[base + pre] = k; slist[base + pre] = sl; }
  }
  __syncthreads();
  return *s_n;
}

// ---- forward ------------------------------------------------------------------------------------------------------
// LDS per part: key[512] | klist[2][512] (cell keys of this and the previous substep: the previous ones are needed to zero its
// buffer), slist[512] (their slots) | val[4][512] doubles, reused after the flush as vel[512] float4 (grid velocity, .w = m)
// hist: env b's records at hist + b * a.hist_stride_b; record 0 = the input state (lg_pack).  keep != 0 (a checkpoint): every
// substep's input state is kept, record f + 1 at (f + 1) * rec; keep == 0: only the final state is written, at last_off.
template <int T>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(2))) clm_fwd_kernel(const LargeArgs a, const ClusterGrid g, float* hist, long rec, int keep, long last_off) {
  __shared__ int s_key[CLM_H], s_klist[2][CLM_H], s_slist[CLM_H];
  __shared__ double s_val[4 * CLM_H];
  __shared__ int s_dead, s_ovf, s_hash, s_n, s_no, s_obase, s_nsp, s_spilled;
  __shared__ unsigned short s_olist[CLM_H];
  __shared__ int s_spill[2][CLM_SPILL];            // cells past the table, of this and the previous substep (zeroed two substeps later)
  int bl, w;
  clm_decode(g.W, bl, w);
  if (bl >= g.Bl) return;
  const int b = a.b0 + bl, tid = threadIdx.x, p = w * (T / 4) + (tid >> 2), qi = tid & 3;
  const MpmConst& c = a.c;
  const bool live = p < c.N;
  const int S = c.steps;
  if (tid == 0) { s_dead = 0; s_ovf = 0; s_spilled = 0; }
  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f}, Cm[9], F[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[d] = 0.f; F[d] = (d % 4 == 0) ? 1.f : 0.f; }
  float* hw = hist + (long)b * a.hist_stride_b;
  if (live) load_state(hw, c.Np, p, x, v, Cm, F);          // record 0: written by lg_pack (an earlier launch)
  const int up = live ? user_index(a, b, p) : 0;
  const int material = a.material[up];
  const float hard = a.hard[up], mu_s = a.mu[b], la_s = a.lamda[b];
  float4* vel = (float4*)s_val;
  float4* raw = vel + CLM_H;
  unsigned* bar = g.bar + (long)bl * CLM_BAR_STRIDE;
  // grid checkpoint for the (multi-kernel) backward, in ITS format: one record {key, m, mv, v} per active cell and substep, written by
  // the part that owns the cell (the smallest part number that touched it: an atomicMin beside the flush), at a position drawn
  // from the env's running record counter; idx[f] = first record of substep f (lg_restore, cell_mass_momentum read it).
  const bool recs = keep && a.gck_base != nullptr;
  int* ridx = recs ? gck_idx(a, b) : nullptr;
  float4* rpool = recs ? gck_pool(a, b) : nullptr;
  int* rcnt = a.w.count + b;                              // zeroed by lg_prim_in
  if (recs && w == 0 && tid == 0) ridx[0] = 0;
  int nprev = 0, nsp_prev = 0;
  bool alive = true;
  for (int f = 0; f < S && alive; ++f) {
    int* key = s_key;
    int* klist = s_klist[f & 1];
    const int* kprev = s_klist[(f + 1) & 1];
    int* spill = s_spill[f & 1];
    const int* spill_prev = s_spill[(f + 1) & 1];
    const BlockTable bt{key, s_val};
    float4* gcur = g.cg[f % 3] + (long)bl * a.G;
    float4* gold = g.cg[(f + 2) % 3] + (long)bl * a.G;    // substep f - 1's buffer
    int* ocur = g.own[f % 3] + (long)bl * a.G;
    int* oold = g.own[(f + 2) % 3] + (long)bl * a.G;
    // ---- table clear, pre-pass, window ----
    for (int s = tid; s < CLM_H; s += T) { key[s] = -1; s_val[s] = 0.0; s_val[CLM_H + s] = 0.0; s_val[2 * CLM_H + s] = 0.0; s_val[3 * CLM_H + s] = 0.0; }
    if (tid == 0) { s_hash = 0; s_n = 0; s_no = 0; s_nsp = 0; }
    Pre q;
    q.base[0] = q.base[1] = q.base[2] = 0;
    if (live) {
      particle_pre<false>(c, x, Cm, F, mu_s, la_s, material, hard, q, nullptr,
                          (qi == 0 && keep && a.svd_rows) ? hw + (long)f * rec + (long)24 * c.Np + p : nullptr, nullptr, c.Np);
      if (qi == 0 && keep) {
        float* ho = hw + (long)(f + 1) * rec;
#pragma unroll
        for (int d = 0; d < 9; ++d) ho[(15 + d) * c.Np + p] = q.Fn[d];
      }
    }
    BlockWin win = bt_window(c, live, q.base);              // two workgroup barriers: they also publish the clear
    if (win.on && live && !clm_stencil_in_window(c, win, q.base)) s_hash = 1;
    __syncthreads();
    if (s_hash) win.on = 0;
    // ---- p2g into the table, the occupied cells as a list, flush to the env's grid ----
    if (live && !clm_scatter(c, bt, win, q, v, p, qi, gcur, spill, &s_nsp)) s_ovf = 1;
    __syncthreads();
    const int n = clm_compact<T>(key, klist, s_slist, &s_n);
    if (!(UD_MPM_ABLATE & 16384)) {   // (timing-only diagnostic builds, tools/build_abl.sh: 16384 no flush, 4096 no read-back, 8192 no zeroing, 32768 no barrier)
      const int r = tid & 3;
      for (int e = tid >> 2; e < n; e += T / 4) {
        const long lin = cell_lin(c, klist[e]);
        atomicAdd((float*)(gcur + lin) + r, (float)s_val[r * CLM_H + s_slist[e]]);
        if (recs && r == 0) atomicMin(ocur + lin, w);
      }
    }
    if (UD_MPM_ABLATE & 32768) __syncthreads();
    else alive = clm_barrier(bar, (unsigned)(f + 1), (unsigned)g.W, &s_dead, (recs && f > 0) ? ridx + f : nullptr, rcnt);
    if (!alive) break;
    // ---- read the summed cells back, grid op, zero the cells of substep f - 1 ----
    for (int e0 = 0; e0 < n; e0 += T) {
      const int e = e0 + tid;
      bool mine = false;
      if (e < n) {
        const int k = klist[e], sl = s_slist[e];
        const long lin = cell_lin(c, k);
        const float4 mv = (UD_MPM_ABLATE & 4096) ? make_float4((float)s_val[sl], (float)s_val[CLM_H + sl], (float)s_val[2 * CLM_H + sl], (float)s_val[3 * CLM_H + sl])
                                                  : ldc4(gcur + lin);
        mine = recs && ldci(ocur + lin) == w;
        float vo[3];
        clm_grid_op(a, b, f, k, mv, vo);
        vel[sl] = make_float4(vo[0], vo[1], vo[2], mv.x);
        if (mine) raw[sl] = mv;                           // the record needs (m, mv) once the env's counter has told where it goes
      }
      if (recs) {                                         // the cells this part owns, as a list (wave-uniform trips)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
        const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        int ob = 0;
        if ((tid & 63) == 0 && m) ob = atomicAdd(&s_no, __popcll(m));
        ob = __builtin_amdgcn_readfirstlane(ob);
        if (mine) s_olist[ob + pre] = (unsigned short)e;
      }
    }
    if (!(UD_MPM_ABLATE & 8192))
      for (int e = tid; e < nprev; e += T) {
        const long lin = cell_lin(c, kprev[e]);
        stc4_zero(gold + lin);
        if (recs) __hip_atomic_store(oold + lin, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    for (int e = tid; e < nsp_prev; e += T) stc4_zero(gold + cell_lin(c, spill_prev[e]));   // (several parts may zero the same spilled cell)
    nprev = n;
    nsp_prev = min(s_nsp, CLM_SPILL);
    if (tid == 0 && s_nsp > 0) s_spilled = 1;
    __syncthreads();
    if (recs && tid == 0) s_obase = s_no ? atomicAdd(rcnt, s_no) : 0;    // where this part's records go (its latency runs beside g2p)
    // ---- g2p + advect ----
    const bool nsp_prev_cur = nsp_prev > 0;                 // this substep spilled: a lookup may miss
    if (live) {
      float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int cidx = qi + 4 * t;
        if (cidx >= 27) break;
        const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
        const int gcell = cell_gather(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
        const int sl = nsp_prev_cur ? clm_lookup(key, win, gcell) : clm_lookup_fast(key, win, gcell);
        float4 g4;
        if (sl >= 0) g4 = vel[sl];
        else {                                              // a spilled cell: the env's sum from HBM, the grid op here
          const float4 mvs = ldc4(gcur + cell_lin(c, gcell));
          float vo[3];
          clm_grid_op(a, b, f, gcell, mvs, vo);
          g4 = make_float4(vo[0], vo[1], vo[2], mvs.x);
        }
        const float weight = sel3(q.w, 0, i) * sel3(q.w, 1, j) * sel3(q.w, 2, k);
        const float dp[3] = {(float)i - q.fx[0], (float)j - q.fx[1], (float)k - q.fx[2]};
        const float gv[3] = {g4.x, g4.y, g4.z};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * gv[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (gv[r] * dp[s2]) * c.inv_dx;
        }
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) nv[d] = lg_quad_sum<4>(nv[d]);
#pragma unroll
      for (int d = 0; d < 9; ++d) nC[d] = lg_quad_sum<4>(nC[d]);
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = x[d] + c.dt * nv[d]; v[d] = nv[d]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { Cm[d] = nC[d]; F[d] = q.Fn[d]; }
      if (qi == 0) {
        if (keep) {
          float* ho = hw + (long)(f + 1) * rec;
#pragma unroll
          for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = x[d]; ho[(3 + d) * c.Np + p] = v[d]; }
#pragma unroll
          for (int d = 0; d < 9; ++d) ho[(6 + d) * c.Np + p] = nC[d];
        }
        if (up < 3) {   // Q6: row `up` of the caller's particle `up`
          const float r0 = nC[0] + nC[1] + nC[2], r1 = nC[3] + nC[4] + nC[5], r2 = nC[6] + nC[7] + nC[8];
          atomicAdd(&a.w.trq[(long)b * S + f], (up == 0) ? r0 : ((up == 1) ? r1 : r2));
        }
      }
    }
    __syncthreads();
    if (recs) {
      const int no = s_no, ob = s_obase;
      for (int j = tid; j < no; j += T) {
        const int e = s_olist[j], sl = s_slist[e];
        const float4 mv = raw[sl], vv = vel[sl];
        if (ob + j < a.gck_budget) {
          float4* r = rpool + (long)(ob + j) * 2;
          r[0] = make_float4(__builtin_bit_cast(float, klist[e]), mv.x, mv.y, mv.z);
          r[1] = make_float4(mv.w, vv.x, vv.y, vv.z);
        } else if (a.status) {
          atomicOr(&a.status[b], 1);     // pool exhausted: the backward of this env recomputes the grid (clip bit 1), as on the multi-kernel path
        }
      }
      __syncthreads();  // vel / raw (= val), key and the cell list of substep f - 1 are rewritten by the next substep
    }
  }
  // the buffers go back all-zero: the cells of the last substep, once every part has read them
  if (alive && !(UD_MPM_ABLATE & 32768)) alive = clm_barrier(bar, (unsigned)(S + 1), (unsigned)g.W, &s_dead, recs ? ridx + S : nullptr, rcnt);
  if (alive) {
    float4* glast = g.cg[(S - 1) % 3] + (long)bl * a.G;
    int* olast = g.own[(S - 1) % 3] + (long)bl * a.G;
    const int* kl = s_klist[(S - 1) & 1];
    for (int e = tid; e < nprev; e += T) {
      const long lin = cell_lin(c, kl[e]);
      stc4_zero(glast + lin);
      if (recs) __hip_atomic_store(olast + lin, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int* sp = s_spill[(S - 1) & 1];
    for (int e = tid; e < nsp_prev; e += T) stc4_zero(glast + cell_lin(c, sp[e]));
  }
  if (!keep && live && qi == 0) {
    float* ho = hw + last_off;
#pragma unroll
    for (int d = 0; d < 3; ++d) { ho[d * c.Np + p] = x[d]; ho[(3 + d) * c.Np + p] = v[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { ho[(6 + d) * c.Np + p] = Cm[d]; ho[(15 + d) * c.Np + p] = F[d]; }
  }
  // status: 1 = cells went past a part's table, so the grid checkpoint of this env is incomplete (outputs valid; the backward recomputes the
  // grid: clip bit 1); 2 = a part's spill list overflowed too, 4 = a part gave up waiting for its siblings (outputs invalid either way)
  if (tid == 0 && a.status && recs && s_spilled) atomicOr(&a.status[b], 1);
  if (tid == 0 && a.status && (s_ovf || s_dead)) atomicOr(&a.status[b], s_dead ? 4 : 2);
}

}  // namespace ud
