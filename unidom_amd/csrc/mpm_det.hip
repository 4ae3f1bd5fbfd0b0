// Deterministic MPM forward: kernels around mpm_det.h.  This file alone is compiled with -DUD_MPM_EXACT -ffp-contract=off and hipcc's
// default correctly rounded f32 divide / sqrt (Makefile), so that its arithmetic is the IEEE arithmetic the host compiler gives the
// same source (oracle/csrc/mpm_det_host.cpp) -- the rest of the MPM path keeps its fast-math build.
// Four launches per substep: pre-pass + touched-cell list (one thread per particle), bucket sort (one workgroup per env), cell sums + grid op
// (a group of 32 lanes per touched cell, walking the 27 buckets that reach it), gather (one thread per particle); cost in DESIGN.md 3.2.
// Below them: the pieces of the deterministic BACKWARD that are sums over particles or cells (mpm_large.hip drives it and keeps the arithmetic).
#include <algorithm>

#include "mpm_det.h"
#include "mpm_det_host.h"

namespace ud {

__global__ void __launch_bounds__(64) det_fk_kernel(DetArgs a) {   // one thread per (env, primitive)
  const long bp = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int S = a.c.steps;
  if (bp >= (long)a.B * a.c.n_prim) return;
  det_fk_rows(S, a.action + bp * 6, a.ppos + bp * S * 3, a.prot + bp * S * 4);
}

// pre-pass per particle; its bucket key; its 27 contributions; the cells it touches into the env's list (first toucher of a cell -- whoever
// replaces the stamp -- appends it: the list's ORDER depends on the run, its CONTENT does not, and every cell is summed on its own).
// 32 lanes per particle: all of them run the pre-pass (same inputs, same bits; lane 0 stores), then lane o < 27 forms the particle's
// contribution to its cell o and stamps that cell -- 54 returning atomics in one thread, one after the other, were most of this kernel.
__global__ void __launch_bounds__(256) det_pre_kernel(DetArgs a, int f, int epoch) {
  const int b = blockIdx.y, gid = blockIdx.x * blockDim.x + threadIdx.x, p = gid >> 5, o = gid & 31;
  const MpmConst& c = a.c;
  if (p >= c.N) return;                                   // whole groups of 32 lanes together
  const float* h = a.hist + (long)b * a.stride_b + (long)(a.pingpong ? (f & 1) : f) * a.rec;
  float* hn = a.bwd ? nullptr : a.hist + (long)b * a.stride_b + (long)(a.pingpong ? ((f + 1) & 1) : (f + 1)) * a.rec;
  float* pre = a.pre + (long)b * UD_DET_PRE * c.Np;
  Pre q;
  float v[3];
  const int bkey = det_pre_particle(c, h, hn, p, a.mu[b], a.lamda[b], a.material[p], a.hard[p], pre, o == 0, q, v);
  if (o == 0) a.bkey[(long)b * c.Np + p] = bkey;
  if (o >= 27) return;
  float q4[4];
  det_contrib(c, q, v, o / 9, (o / 3) % 3, o % 3, q4);
  *(float4*)(a.contrib + (((long)b * 27 + o) * c.Np + p) * 4) = make_float4(q4[0], q4[1], q4[2], q4[3]);
  int* flag = a.flag + (long)b * a.G;
  long sl, gl;
  det_touch_base(c, q.base[0], q.base[1], q.base[2], o, sl, gl);
  for (int t = 0; t < 2; ++t) {                             // (the lanes past 26 left above: every lane here has a cell o)
    const long lin = t ? gl : sl;
    const bool want = !(lin < 0 || (t && gl == sl));
    const bool first = want && atomicExch(&flag[lin], epoch) != epoch;
    // first touchers of this wave (two particles of one env) append together: ONE add on the env's counter per wave and pass instead of one per
    // cell -- ~1500 same-address atomics per env and substep otherwise
    const unsigned long long m = __ballot(first);
    if (m == 0) continue;
    const int rank = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    int base = 0;
    if (rank == 0 && first) base = atomicAdd(&a.count[b], __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (first && base + rank < a.cap) a.list[(long)b * a.cap + base + rank] = (int)lin;
  }
}

// particles of one env sorted by (bucket key, index) -- bitonic network in LDS, one workgroup per env -- then the bucket ranges
__global__ void __launch_bounds__(1024) det_sort_kernel(DetArgs a, int npow2, int epoch) {
  extern __shared__ unsigned long long det_sk[];
  const MpmConst& c = a.c;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* bkey = a.bkey + (long)b * c.Np;
  for (int i = tid; i < npow2; i += blockDim.x)
    det_sk[i] = i < c.N ? (((unsigned long long)(unsigned)(bkey[i] + 1)) << 32) | (unsigned)i : ~0ull;   // key -1 (irregular) sorts first
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long ei = det_sk[i], el = det_sk[l];
          const bool up = (i & k) == 0;
          if ((ei > el) == up) { det_sk[i] = el; det_sk[l] = ei; }
        }
      }
      __syncthreads();
    }
  int* order = a.order + (long)b * c.Np;
  DetRange* brange = (DetRange*)a.brange + (long)b * a.G;
  int* bflag = a.bflag + (long)b * a.G;
  if (tid == 0) a.nirr[b] = 0;
  __syncthreads();
  for (int i = tid; i < c.N; i += blockDim.x) {
    const unsigned long long e = det_sk[i];
    const int key = (int)(unsigned)(e >> 32) - 1;
    order[i] = (int)(unsigned)(e & 0xffffffffu);
    const bool first = i == 0 || (int)(unsigned)(det_sk[i - 1] >> 32) - 1 != key;
    const bool last = i == c.N - 1 || (int)(unsigned)(det_sk[i + 1] >> 32) - 1 != key;
    if (key < 0) { if (last) a.nirr[b] = i + 1; continue; }
    if (first) { brange[key].s = i; bflag[key] = epoch; }
    if (last) brange[key].e = i + 1;
  }
}

// One touched cell per group of 32 lanes, lane o < 27 = stencil offset o.  What det_cell (mpm_det.h, the form the host checker runs) does in
// one thread -- 27 bucket look-ups and the contributions behind them, each a dependent global round trip -- the lanes do side by side:
// every lane fetches its own offset's bucket and its particles' contributions (det_pre_kernel left them as one float4 per (offset, particle));
// then the running sums (m, mv) travel through the lanes that have something to add, in offset order, each lane adding its contributions in
// ascending particle index: the SAME additions in the SAME order, so the same bits.  A lane keeps up to DET_K contributions in registers;
// a longer bucket (or irregular particles) is finished with loads inside the ordered pass.
constexpr int DET_K = 8;
// MODE 0: the forward's (m, mv) sums + grid op.  MODE 1 (deterministic backward): the same ordered walk over a.contrib holding the g2p adjoint's
// contributions (x, y, z; w = 0) -- g2p GATHERS with clamped indices, so an irregular particle reaches the cell its clamped index names.
template <int MODE>
__global__ void __launch_bounds__(256) det_cells_kernel(DetArgs a, int f, int epoch) {
  const int b = blockIdx.y, o = threadIdx.x & 31;
  const MpmConst& c = a.c;
  const int ncell = min(a.count[b], a.cap);
  const int Np = c.Np;
  const float* pre = a.pre + (long)b * UD_DET_PRE * Np;
  const int* order = a.order + (long)b * Np;
  const int n_irr = a.nirr[b];
  const int i = o / 9, j = (o / 3) % 3, k = o % 3;
  const float4* contrib = (const float4*)a.contrib + ((long)b * 27 + min(o, 26)) * Np;
  // a bounded grid walks the list (sized for the worst case, the launch would be 170 k mostly empty workgroups at N = 798: dispatch-bound)
  for (int t0 = blockIdx.x * 8; t0 < ncell; t0 += gridDim.x * 8) {
    const int t = t0 + (threadIdx.x >> 5);
    if ((t0 + ((threadIdx.x & ~63) >> 5)) >= ncell) continue;   // (wave-uniform) both cells of this wave lie past the list
    const bool cell_on = t < ncell;                         // whole groups of 32 lanes together
    const long lin = cell_on ? a.list[(long)b * a.cap + t] : 0;
    const int ck = (int)(lin % c.res[2]), cj = (int)((lin / c.res[2]) % c.res[1]), ci = (int)(lin / ((long)c.res[2] * c.res[1]));
    const int key = ci | (cj << 10) | (ck << 20);
    // ---- every lane: its offset's bucket, the first DET_K contributions into registers (while there are no irregular particles to merge with:
    // the common case): index loads together, then the contribution loads together
    int bs = 0, be = 0;
    if (cell_on && o < 27) {
      const int c0 = ci - i, c1 = cj - j, c2 = ck - k;
      if (det_regular(c, c0, c1, c2)) {
        const long bl = det_lin3(c, c0, c1, c2);
        if (a.bflag[(long)b * a.G + bl] == epoch) { const DetRange r = ((const DetRange*)a.brange)[(long)b * a.G + bl]; bs = r.s; be = r.e; }
      }
    }
    const int nreg = n_irr == 0 ? min(be - bs, DET_K) : 0;
    int pu[DET_K];
    float4 cq[DET_K];
#pragma unroll
    for (int u = 0; u < DET_K; ++u) pu[u] = u < nreg ? order[bs + u] : 0;
#pragma unroll
    for (int u = 0; u < DET_K; ++u) cq[u] = u < nreg ? contrib[pu[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- the ordered pass: the running sums visit the lanes that have something to add, in offset order
    const unsigned long long any = __ballot(cell_on && o < 27 && (be > bs || n_irr > 0));
    unsigned mine = (unsigned)(any >> (threadIdx.x & 32));  // this group's 32 lanes
    float m = 0.f, mv[3] = {0.f, 0.f, 0.f};
    while (mine) {
      const int oo = __ffs(mine) - 1;
      mine &= mine - 1;
      float tm = m, tv[3] = {mv[0], mv[1], mv[2]};
      if (o == oo) {
#pragma unroll
        for (int u = 0; u < DET_K; ++u)
          if (u < nreg) { tm += cq[u].x; tv[0] += cq[u].y; tv[1] += cq[u].z; tv[2] += cq[u].w; }
        int q = bs + nreg, ir = 0;                          // what did not fit the registers, merged by index with the irregular particles
        while (q < be || ir < n_irr) {
          const int pb = q < be ? order[q] : 0x7fffffff, pi = ir < n_irr ? order[ir] : 0x7fffffff;
          int p;
          if (pb < pi) { p = pb; ++q; }
          else {
            p = pi; ++ir;
            const int b0 = __builtin_bit_cast(int, pre[p]), b1 = __builtin_bit_cast(int, pre[Np + p]), b2 = __builtin_bit_cast(int, pre[2 * Np + p]);
            if ((MODE ? cell_gather(c, b0 + i, b1 + j, b2 + k) : cell_scatter(c, b0 + i, b1 + j, b2 + k)) != key) continue;
          }
          const float4 x4 = contrib[p];
          tm += x4.x; tv[0] += x4.y; tv[1] += x4.z; tv[2] += x4.w;
        }
      }
      const int src = (threadIdx.x & 32) + oo;              // lane oo of this group of 32
      m = __shfl(tm, src); mv[0] = __shfl(tv[0], src); mv[1] = __shfl(tv[1], src); mv[2] = __shfl(tv[2], src);
    }
    if (!cell_on || o != 0) continue;
    if (MODE) {
      float* og = a.gacc + ((long)b * a.G + lin) * 4;
      og[0] = m; og[1] = mv[0]; og[2] = mv[1]; og[3] = 0.f;
      continue;
    }
    if (a.val_out) {
      float* ov = a.val_out + ((long)b * a.G + lin) * 4;
      ov[0] = m; ov[1] = mv[0]; ov[2] = mv[1]; ov[3] = mv[2];
      a.keylist[(long)b * a.cap + t] = key;
      if (t == 0) a.keycount[b] = ncell;
    }
    const int S = c.steps, P = c.n_prim;
    float vo[3];
    det_grid_cell(c, f, a.ppos + (long)b * P * S * 3, a.prot + (long)b * P * S * 4, a.psize + (long)b * P * 3, a.action + (long)b * P * 6, a.friction[b],
                  ci, cj, ck, m, mv, vo);
    float* ov = a.vel + ((long)b * a.G + lin) * 4;
    ov[0] = vo[0]; ov[1] = vo[1]; ov[2] = vo[2]; ov[3] = 0.f;
  }
}

__global__ void __launch_bounds__(256) det_g2p_kernel(DetArgs a, int f) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const MpmConst& c = a.c;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.count[b] = 0;      // the cells kernel has consumed the list
  if (p >= c.N) return;
  const float* h = a.hist + (long)b * a.stride_b + (long)(a.pingpong ? (f & 1) : f) * a.rec;
  float* hn = a.hist + (long)b * a.stride_b + (long)(a.pingpong ? ((f + 1) & 1) : (f + 1)) * a.rec;
  const float t = det_g2p_particle(c, h, hn, p, a.pre + (long)b * UD_DET_PRE * c.Np, a.vel + (long)b * a.G * 4);
  if (p < 3) a.trq3[((long)b * c.steps + f) * 3 + p] = t;
}

// Q6: the three row sums of a substep, added in a fixed order
__global__ void __launch_bounds__(256) det_trq_kernel(DetArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x, S = a.c.steps;
  if (e >= a.B * S) return;
  const float* t = a.trq3 + (long)e * 3;
  const int n = min(a.c.N, 3);
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += t[k];
  a.trq[e] = s;
}

// ---- deterministic backward: the pieces around mpm_large.hip's g2p-adjoint / grid-op-adjoint / p2g-adjoint kernels ----------------------------
constexpr int DET_SORT_CELLS = 32768;   // touched cells per env the list sort holds (128 KB of LDS); the pre-pass lists them in arrival order
// the env's touched-cell list in ascending order: position t in it is what the per-cell arrays of the backward are indexed by
__global__ void __launch_bounds__(1024) det_sort_cells_kernel(DetArgs a) {
  extern __shared__ int det_sc[];
  const int b = blockIdx.x, tid = threadIdx.x, n = min(min(a.count[b], a.cap), DET_SORT_CELLS);
  int npow2 = 64;
  while (npow2 < n) npow2 <<= 1;
  int* list = a.list + (long)b * a.cap;
  for (int i = tid; i < npow2; i += blockDim.x) det_sc[i] = i < n ? list[i] : 0x7fffffff;
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const int ei = det_sc[i], el = det_sc[l];
          const bool up = (i & k) == 0;
          if ((ei > el) == up) { det_sc[i] = el; det_sc[l] = ei; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += blockDim.x) list[i] = det_sc[i];
}
// cellred -> the env's cotangents of substep f.  Component k: lane t adds cells t, t + 256, ... of the (sorted) list in ascending order, thread 0
// adds the 256 partial sums in lane order -- a fixed order.
__global__ void __launch_bounds__(256) det_reduce_cells_kernel(DetArgs a, int f) {
  __shared__ float part[256];
  const int b = blockIdx.x, tid = threadIdx.x, K = a.K, S = a.c.steps, P = a.c.n_prim;
  const int n = min(min(a.count[b], a.cap), a.capc);
  const float* cr = a.cellred + (long)b * a.capc * K;
  const int f0 = min(max(f, 0), S - 1), f1 = min(max(f + 1, 0), S - 1);
  if (tid == 0 && a.status && min(a.count[b], a.cap) > min(a.capc, DET_SORT_CELLS)) atomicOr(&a.status[b], 1);
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int t = tid; t < n; t += 256) s += cr[(long)t * K + k];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int u = 0; u < 256; ++u) tot += part[u];
      float* dst;
      if (k == 0) dst = a.acc + b * 4;
      else if (a.c.position_control) dst = a.gpv + ((long)b * S + f) * 3 + (k - 1);
      else {
        const int ip = (k - 1) / 18, d = (k - 1) % 18;
        const long bp = (long)b * P + ip;
        dst = (d < 3)    ? a.gppos + bp * S * 3 + f0 * 3 + d
              : (d < 7)  ? a.grot + bp * S * 4 + f0 * 4 + (d - 3)
              : (d < 10) ? a.gppos + bp * S * 3 + f1 * 3 + (d - 7)
              : (d < 14) ? a.grot + bp * S * 4 + f1 * 4 + (d - 10)
                         : a.gpsz + bp * 4 + (d - 14);
      }
      if (tot != 0.f) *dst += tot;
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) det_bwd_clear_kernel(DetArgs a) {
  const int b = blockIdx.y, ncell = min(a.count[b], a.cap);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < ncell; t += gridDim.x * 256)
    ((float4*)a.gacc)[(long)b * a.G + a.list[(long)b * a.cap + t]] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void __launch_bounds__(64) det_count_reset_kernel(DetArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b < a.B) a.count[b] = 0;
}
// pacc -> the env's mu / lamda cotangents, same fixed order over the particles; pacc zeroed again
__global__ void __launch_bounds__(256) det_reduce_particles_kernel(DetArgs a) {
  __shared__ float part[2][256];
  const int b = blockIdx.x, tid = threadIdx.x, Np = a.c.Np;
  float* pa = a.pacc + (long)b * 2 * Np;
  float s0 = 0.f, s1 = 0.f;
  for (int p = tid; p < a.c.N; p += 256) { s0 += pa[p]; s1 += pa[Np + p]; pa[p] = 0.f; pa[Np + p] = 0.f; }
  part[0][tid] = s0; part[1][tid] = s1;
  __syncthreads();
  if (tid < 2) {
    float tot = 0.f;
    for (int k = 0; k < 256; ++k) tot += part[tid][k];
    a.acc[b * 4 + 1 + tid] += tot;
  }
}

int mpm_det_bwd_recompute(const DetArgs& a, int f, int* epoch, hipStream_t st) {
  const MpmConst& c = a.c;
  const dim3 blk(256), gp32((c.N * 32 + 255) / 256, a.B), gc((unsigned)std::min(128, (a.cap + 7) / 8), a.B);
  int npow2 = 64;
  while (npow2 < c.N) npow2 <<= 1;
  (void)hipFuncSetAttribute((const void*)det_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, npow2 * 8);
  const int e = ++*epoch;
  hipLaunchKernelGGL(det_pre_kernel, gp32, blk, 0, st, a, f, e);
  hipLaunchKernelGGL(det_sort_kernel, dim3(a.B), dim3(1024), (size_t)npow2 * 8, st, a, npow2, e);
  (void)hipFuncSetAttribute((const void*)det_sort_cells_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DET_SORT_CELLS * 4);
  hipLaunchKernelGGL(det_sort_cells_kernel, dim3(a.B), dim3(1024), (size_t)DET_SORT_CELLS * 4, st, a);
  hipLaunchKernelGGL(det_cells_kernel<0>, gc, blk, 0, st, a, f, e);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}
int mpm_det_bwd_gcells(const DetArgs& a, int f, int epoch, hipStream_t st) {
  const dim3 gc((unsigned)std::min(128, (a.cap + 7) / 8), a.B);
  hipLaunchKernelGGL(det_cells_kernel<1>, gc, dim3(256), 0, st, a, f, epoch);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}
int mpm_det_bwd_reduce_cells(const DetArgs& a, int f, hipStream_t st) {
  hipLaunchKernelGGL(det_reduce_cells_kernel, dim3(a.B), dim3(256), 0, st, a, f);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}
int mpm_det_bwd_clear(const DetArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(det_bwd_clear_kernel, dim3(16, a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(det_count_reset_kernel, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}
int mpm_det_bwd_reduce_particles(const DetArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(det_reduce_particles_kernel, dim3(a.B), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}

int mpm_det_forward(const DetArgs& a, int* epoch, hipStream_t st) {
  const MpmConst& c = a.c;
  const int S = c.steps;
  const dim3 blk(256), gp((c.N + 255) / 256, a.B), gp32((c.N * 32 + 255) / 256, a.B), gc((unsigned)std::min(128, (a.cap + 7) / 8), a.B);
  int npow2 = 64;
  while (npow2 < c.N) npow2 <<= 1;
  (void)hipFuncSetAttribute((const void*)det_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, npow2 * 8);
  hipLaunchKernelGGL(det_fk_kernel, dim3((a.B * c.n_prim + 63) / 64), dim3(64), 0, st, a);
  (void)hipMemsetAsync(a.count, 0, (size_t)a.B * sizeof(int), st);
  for (int f = 0; f < S; ++f) {
    const int e = ++*epoch;
    hipLaunchKernelGGL(det_pre_kernel, gp32, blk, 0, st, a, f, e);
    hipLaunchKernelGGL(det_sort_kernel, dim3(a.B), dim3(1024), (size_t)npow2 * 8, st, a, npow2, e);
    hipLaunchKernelGGL(det_cells_kernel<0>, gc, blk, 0, st, a, f, e);
    hipLaunchKernelGGL(det_g2p_kernel, gp, blk, 0, st, a, f);
  }
  hipLaunchKernelGGL(det_trq_kernel, dim3((a.B * S + 255) / 256), blk, 0, st, a);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}

}  // namespace ud
