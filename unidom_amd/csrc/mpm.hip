// MLS-MPM `step` (conf.steps substeps) for gfx950: one workgroup per environment, 4 lanes (a DPP quad) per
// particle, the touched part of the `res` grid in an LDS open-addressing cell table.
//
// What it replaces (reference, /root/reference/DaXBench/daxbench/core/engine/):
//   mpm_simulator.py  substep :223-330, p2g_micro :178-194, g2p_micro :196-221, step :413-429,
//                     copy_frame :365-373, norm_grad_state / norm_grad :375-411, bwd loss leaves :343-354
//   svd_safe_batch.py svd :19-51, _svd_bwd :65-102
//   primitives/primitives.py forward_kinematics :185-194, set_action :212-229, position_control_batch :232-239,
//                     sdf_batch :112-114, inv_trans_batch :105-109, qrot_batch :95-102, qmul :73-81, w2quat :84-92
//   primitives/box.py _sdf_batch :6-18
//
// Why a cell table instead of the dense res grid: g2p only reads cells inside some particle's 3x3x3 support,
// and every grid-op input of such a cell is (m, mv) scattered by particles, so cells no particle touches can
// never influence x, v, C, F or any cotangent (their cotangent is 0, or NaN that nothing gathers -- Q7).
// The reference spends its time on ~10 dense 32^3 temporaries per substep; here only the few hundred touched
// cells exist, in LDS.  The whole `steps`-substep loop runs in one launch; the backward streams per-substep
// particle checkpoints (24 floats/particle, SoA) from HBM in reverse.
//
// Lane mapping: lane = 4*particle + q.  The four lanes of a quad compute the particle pre-pass (F update, 3x3
// Jacobi SVD, stress) redundantly -- there are far fewer particles than lanes on a CU -- and split the 27
// stencil cells (7/7/7/6); quad sums go through DPP quad_perm.  No MFMA: stencil / scatter work.
#include "mpm_device.h"
#include "mpm_large.h"

namespace ud {

struct MpmFwdArgs {
  MpmConst c;
  const int* material;
  const float* hard;
  int B;
  const float *x, *v, *C, *F, *J, *ppos, *prot, *psize, *friction, *mu, *lamda, *action;
  float *xo, *vo, *Co, *Fo, *Jo, *ppos_o, *prot_o, *pv_o, *pw_o;
  float* ckpt;
  int* status;
};

struct MpmBwdArgs {
  MpmConst c;
  const int* material;
  const float* hard;
  int B;
  const float* ckpt;
  const float *psize, *friction, *mu, *lamda, *action;
  const float *gx, *gv, *gC, *gF, *gppos;
  int clip;
  float *gx0, *gv0, *gC0, *gF0, *gppos0, *gfric, *gmu, *glam, *gaction;
  int* status;
};

// ---- LDS cell table --------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cell_hash(int cell, int logH) {
  unsigned h = (unsigned)cell;
  h ^= h >> 9; h *= 2654435761u; h ^= h >> 15;
  return h >> (32 - logH);
}

// Keys persist for the whole launch (cells change slowly from substep to substep), so the common case is a plain
// LDS read that finds the key; only a cell seen for the first time pays a ds_cmpst and is appended to the
// occupied-slot list that the per-substep value clear and the grid op iterate over.
__device__ __forceinline__ int table_insert(int* key, int* list, int* count, int H, int logH, int cell) {
  unsigned s = cell_hash(cell, logH);
  for (int probe = 0; probe < H; ++probe) {
    int cur = key[s];
    if (cur == cell) return (int)s;
    if (cur == -1) {
      const int old = atomicCAS(&key[s], -1, cell);
      if (old == -1) { list[atomicAdd(count, 1)] = (int)s; return (int)s; }
      if (old == cell) return (int)s;
    }
    s = (s + 1) & (unsigned)(H - 1);
  }
  return -1;
}

// LDS accumulation is in float64: on gfx950 ds_add_f32 retires one wave-instruction per ~190 cycles per CU even
// without conflicts, ds_add_f64 one per ~9 (tools/ubench_lds_atomic.hip) -- the f32 scatter was >50 % of a substep.
__device__ __forceinline__ void lds_add(double* p, float v) {
  __hip_atomic_fetch_add(p, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// quad (4-lane) all-reduce through DPP quad_perm
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_f<0xB1>(v);  // [1,0,3,2]
  v += dpp_f<0x4E>(v);  // [2,3,0,1]
  return v;
}

// Consecutive particles are spatial neighbours (lattice seeding order), so the same stencil lane of adjacent quads very
// often targets the same cell (2.7 particles per cell along the whip_rope rope).  Before the LDS atomics the up-to-four
// quads of a 16-lane DPP row are therefore reduced by runs of equal slot (lanes 4 apart: row_shr:4 / row_shr:8, a
// two-step segmented scan) and only the last lane of a run issues the atomic: ~2.7x fewer atomic lane-operations and
// no same-address serialisation inside a wave-instruction.  Disabled / out-of-row source lanes read as "different slot".
template <int CTRL>
__device__ __forceinline__ int dpp_i_old(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f_old0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int NV>
__device__ __forceinline__ bool quadrun_reduce(int slot, float (&val)[NV]) {   // returns true on the lane that scatters
  const int prev = dpp_i_old<0x114>(slot, -2);                    // row_shr:4: the same stencil lane of the previous quad
  const int f1 = (prev == slot && slot >= 0) ? 1 : 0;
  const int f2 = f1 & dpp_i_old<0x114>(f1, 0);
  const int cont = dpp_i_old<0x104>(f1, 0);                       // row_shl:4: the next quad continues this run
  const float f1f = (float)f1, f2f = (float)f2;
#pragma unroll
  for (int n = 0; n < NV; ++n) val[n] += f1f * dpp_f_old0<0x114>(val[n]);
#pragma unroll
  for (int n = 0; n < NV; ++n) val[n] += f2f * dpp_f_old0<0x118>(val[n]);   // row_shr:8
  return slot >= 0 && cont == 0;
}

// lane q of a quad owns stencil cells cidx = q, q+4, ..., < 27
#define UD_NCELL 7
__device__ __forceinline__ bool cell_of(int q, int t, int& i, int& j, int& k) {
  const int cidx = q + 4 * t;
  i = cidx / 9; j = (cidx / 3) % 3; k = cidx % 3;
  return cidx < 27;
}

// LDS layout helpers
// Cell slot s owns four doubles acc[a][s]: while p2g scatters they are (m, mv[3]); the grid op reads them and
// rewrites the same bytes as floats {m, mv[3], vel[3], -} (accf = float view, 8 per slot).  In the adjoint gacc[0..2][s]
// are doubles while the g2p adjoint scatters; the grid-op adjoint rewrites them as floats {g_mv[3], g_m, -, -} (6 per slot).
// Component-major inside the table (acc[4][H], gacc[3][H]): with slot-major rows of 32 bytes the lanes of one ds_add_f64 (all
// on the same component) land on 8 of the 64 banks.  Measured on whip_rope: 1 % (0.691 + 1.186 -> 0.684 + 1.175 ms per step) --
// kept for consistency with mpm_large.hip, not because it matters.  The float overlay of double (a, s) is floats 2 (a H + s) + {0, 1}.
#define ACC_D(H, s, a) ((a) * (H) + (s))
#define ACC_F(H, s, k) ((((k) >> 1) * (H) + (s)) * 2 + ((k) & 1))
struct Lds {
  int* key; double* acc;             // [H], [4][H]
  double* gacc;                      // bwd only: [3H]
  float* ppos; float* prot;          // [S*3], [S*4]
  float* ppin; float* gppos; double* gpv;  // bwd only: [S*3] each (gpv accumulates atomically -> double)
  float* scr;                        // [64] scratch
  int* list; int* count;             // occupied slots [H], their number
};

__device__ __forceinline__ void prim_at(const Lds& L, int f, int S, const float* psize, const float* pv, float friction, PrimF& pf) {
  const int fc = min(max(f, 0), S - 1);
#pragma unroll
  for (int a = 0; a < 3; ++a) { pf.pos[a] = L.ppos[fc * 3 + a]; pf.size[a] = psize[a]; pf.pv[a] = pv[a]; }
  float r0 = L.prot[fc * 4], r1 = -L.prot[fc * 4 + 1], r2 = -L.prot[fc * 4 + 2], r3 = -L.prot[fc * 4 + 3];
  float n = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + 1e-12f;   // :105-109
  pf.iq[0] = r0 / n; pf.iq[1] = r1 / n; pf.iq[2] = r2 / n; pf.iq[3] = r3 / n;
  pf.friction = friction;
}

// forward kinematics writes (:185-194): called by every thread between two barriers
__device__ __forceinline__ void fk_read(const Lds& L, int f, int S, const float* pv, int tid, float& pending) {
  // entry e = tid: value the array holds after `set(f+1, pos[f]+v[f])`, before the whole-array clip
  if (tid < S * 3) {
    const int row = tid / 3, a = tid - row * 3;
    const int fc = min(max(f, 0), S - 1);
    const float pva = (a == 0) ? pv[0] : ((a == 1) ? pv[1] : pv[2]);
    pending = (row == f + 1) ? (L.ppos[fc * 3 + a] + pva) : L.ppos[tid];
  }
}
__device__ __forceinline__ void fk_write(const Lds& L, int f, int S, const float* pw, int tid, float pending) {
  if (tid < S * 3) L.ppos[tid] = clipf(pending, -2.f, 2.f);
  if (tid == 0 && f + 1 < S) {   // rotation[f+1] = qmul(w2quat(w[f]), rotation[f])  (:73-92)
    float ang = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) + 1e-12f;
    float sn = sinf(ang / 2.f);
    float q[4] = {cosf(ang / 2.f), pw[0] / ang * sn, pw[1] / ang * sn, pw[2] / ang * sn};
    const float* r = L.prot + f * 4;
    float o0 = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    float o1 = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    float o2 = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    float o3 = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
    float nn = clipf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3), 1e-12f, INFINITY);
    float* w = L.prot + (f + 1) * 4;
    w[0] = o0 / nn; w[1] = o1 / nn; w[2] = o2 / nn; w[3] = o3 / nn;
  }
}

// p2g of one quad lane: find / insert its cells, scatter mass / momentum.
// slots[t] = (scatter_slot+1) << 16 | gather_slot; pcell[t] = the cell key slots[t] was resolved for (a particle
// crosses a cell boundary only every ~100 substeps, so the table probe is skipped almost always).
__device__ __forceinline__ bool p2g_lane(const MpmConst& c, const Lds& L, const Pre& q, const float* v, int qi, int* slots,
                                         int* pcell) {
  bool ok = true;
#pragma unroll
  for (int t = 0; t < UD_NCELL; ++t) {
    int i, j, k;
    if (!cell_of(qi, t, i, j, k)) continue;
    const float weight = sel3(q.w, 0, i) * sel3(q.w, 1, j) * sel3(q.w, 2, k);
    const int sc = cell_scatter(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
    const int gc = cell_gather(c, q.base[0] + i, q.base[1] + j, q.base[2] + k);
    int ss;
    if (sc == gc && sc == pcell[t]) {
      ss = (slots[t] >> 16) - 1;                       // cached
    } else {
      ss = -1;
      if (sc >= 0 && !(UD_MPM_ABLATE & 32)) { ss = table_insert(L.key, L.list, L.count, c.H, c.logH, sc); ok = ok && (ss >= 0); }
      int gs = ss;
      if (gc != sc && !(UD_MPM_ABLATE & 32)) { gs = table_insert(L.key, L.list, L.count, c.H, c.logH, gc); ok = ok && (gs >= 0); }
      slots[t] = ((ss + 1) << 16) | (max(gs, 0) & 0xffff);
      pcell[t] = (sc == gc && ss >= 0) ? sc : -2;      // only in-range cells are cacheable
    }
    if (!(UD_MPM_ABLATE & 2)) {
      const float dp0 = ((float)i - q.fx[0]) * c.dx, dp1 = ((float)j - q.fx[1]) * c.dx, dp2 = ((float)k - q.fx[2]) * c.dx;
      float contrib[4];
      contrib[0] = weight * c.p_mass;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float ad = q.affine[a * 3] * dp0 + q.affine[a * 3 + 1] * dp1 + q.affine[a * 3 + 2] * dp2;
        contrib[1 + a] = weight * (c.p_mass * v[a] + ad);
      }
      if (quadrun_reduce<4>(ss, contrib)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) lds_add(&L.acc[ACC_D(c.H, ss, a)], contrib[a]);
      }
    }
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Two lane mappings share the workgroup:
//   compact  lane = particle (tid < N, the first ceil(N/64) waves): owns the particle state x, v, C, F, J and runs the
//            particle pre-pass (F update, Jacobi SVD, stress) ONCE per particle; the other waves skip it (wave-uniform
//            branch), so its ~800 instructions are no longer issued four times per particle on every SIMD;
//   quad     lane = 4*particle + q: the 27-cell stencil work (p2g scatter, g2p gather), 7/7/7/6 cells per lane.
// The pre-pass hands (base, fx, w, affine, v) to the quads through LDS (7 float4 per particle, SoA) under the barrier that
// already separates the table clear from p2g; g2p returns (v_new, C_new) the same way under the end-of-substep barrier.
constexpr int UD_STG = 7, UD_RET = 3;   // float4 per particle: pre-pass -> quads, quads -> state

// forward kinematics of the whole step, once per launch (the per-substep form of :185-194 is a recurrence on rows that
// nothing else reads): P[0] = clip(in[0]); P[f+1] = clip(P'[f] + v) with P'[0] = in[0] (read before the first clip) and
// P'[f] = P[f] after that; rotation[f+1] = normalise(qmul(w2quat(w), rotation[f])); iq[f] = inverse rotation for the SDF.
__device__ __forceinline__ void fk_prologue(const Lds& L, float* iq, int S, const float* pv, const float* pw, int tid) {
  if (tid < 3) {
    const float pva = (tid == 0) ? pv[0] : ((tid == 1) ? pv[1] : pv[2]);
    float prev = L.ppos[tid];
    L.ppos[tid] = clipf(prev, -2.f, 2.f);
    for (int f = 0; f + 1 < S; ++f) {
      const float nxt = clipf(prev + pva, -2.f, 2.f);
      L.ppos[(f + 1) * 3 + tid] = nxt;
      prev = nxt;
    }
  }
  if (tid == 64 || (blockDim.x <= 64 && tid == 3)) {   // another wave when there is one
    const float ang = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) + 1e-12f;   // w2quat :84-92
    const float sn = sinf(ang / 2.f);
    const float q[4] = {cosf(ang / 2.f), pw[0] / ang * sn, pw[1] / ang * sn, pw[2] / ang * sn};
    float r[4] = {L.prot[0], L.prot[1], L.prot[2], L.prot[3]};
    for (int f = 0; f + 1 < S; ++f) {                   // qmul :73-81, normalize :66-70
      const float o0 = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
      const float o1 = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
      const float o2 = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
      const float o3 = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
      const float nn = clipf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3), 1e-12f, INFINITY);
      r[0] = o0 / nn; r[1] = o1 / nn; r[2] = o2 / nn; r[3] = o3 / nn;
      float* w = L.prot + (f + 1) * 4;
      w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; w[3] = r[3];
    }
  }
  __syncthreads();
  for (int f = tid; f < S; f += blockDim.x) {           // inv_trans :105-109
    const float r0 = L.prot[f * 4], r1 = -L.prot[f * 4 + 1], r2 = -L.prot[f * 4 + 2], r3 = -L.prot[f * 4 + 3];
    const float n = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + 1e-12f;
    iq[f * 4] = r0 / n; iq[f * 4 + 1] = r1 / n; iq[f * 4 + 2] = r2 / n; iq[f * 4 + 3] = r3 / n;
  }
}

__device__ __forceinline__ void prim_from_lds(const Lds& L, const float* iq, int f, const float* psize, const float* pv,
                                              float friction, PrimF& pf) {
#pragma unroll
  for (int a = 0; a < 3; ++a) { pf.pos[a] = L.ppos[f * 3 + a]; pf.size[a] = psize[a]; pf.pv[a] = pv[a]; }
#pragma unroll
  for (int a = 0; a < 4; ++a) pf.iq[a] = iq[f * 4 + a];
  pf.friction = friction;
}

__global__ void __launch_bounds__(512) mpm_step_fwd_kernel(MpmFwdArgs a) {
  extern __shared__ float smem[];
  const MpmConst c = a.c;
  const int tid = threadIdx.x, b = blockIdx.x, nt = blockDim.x;
  const int N = c.N, S = c.steps, H = c.H, Np = c.Np;
  Lds L;
  L.key = (int*)smem; L.acc = (double*)(smem + H);
  float4* stage = (float4*)(smem + 9 * H);              // [UD_STG][Np]
  float4* ret = stage + UD_STG * Np;                    // [UD_RET][Np]
  L.ppos = (float*)(ret + UD_RET * Np); L.prot = L.ppos + S * 3;
  float* iq = L.prot + S * 4;
  L.scr = iq + S * 4;
  L.list = (int*)(L.scr + 64); L.count = L.list + H;
  float* accf = (float*)L.acc;
  // quad mapping
  const int p = tid >> 2, qi = tid & 3;
  const bool live = p < N;
  // compact mapping
  const bool clive = tid < N;
  const bool cwave = (tid & ~63) < N;                   // wave-uniform: this wave holds compact lanes
  const int cp = clive ? tid : 0;
  float x[3], v[3], Cm[9], F[9], Fn[9], Jp;
#pragma unroll
  for (int d = 0; d < 3; ++d) { x[d] = nan_to_num(a.x[((size_t)b * N + cp) * 3 + d]); v[d] = nan_to_num(a.v[((size_t)b * N + cp) * 3 + d]); }
#pragma unroll
  for (int d = 0; d < 9; ++d) { Cm[d] = nan_to_num(a.C[((size_t)b * N + cp) * 9 + d]); F[d] = nan_to_num(a.F[((size_t)b * N + cp) * 9 + d]); Fn[d] = F[d]; }
  Jp = nan_to_num(a.J[(size_t)b * N + cp]);                      // norm_grad_state fwd (:377-381)
  const int material = a.material[cp];
  const float hard = a.hard[cp];
  for (int e = tid; e < S * 3; e += nt) L.ppos[e] = a.ppos[(size_t)b * S * 3 + e];
  for (int e = tid; e < S * 4; e += nt) L.prot[e] = a.prot[(size_t)b * S * 4 + e];
  for (int s = tid; s < H; s += nt) { L.key[s] = -1; L.acc[ACC_D(c.H, s, 0)] = 0.0; L.acc[ACC_D(c.H, s, 1)] = 0.0; L.acc[ACC_D(c.H, s, 2)] = 0.0; L.acc[ACC_D(c.H, s, 3)] = 0.0; }
  if (tid == 0) *L.count = 0;
  float pv[3], pw[3], psize[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {                                   // clip + set_action (:419-423, primitives.py:212-229)
    pv[d] = clipf(a.action[b * 6 + d], -1.f, 1.f) * 1.f / (float)S;
    pw[d] = clipf(a.action[b * 6 + 3 + d], -1.f, 1.f) * 1.f / (float)S;
    psize[d] = a.psize[b * 3 + d];
  }
  const float friction = a.friction[b], mu_s = a.mu[b], la_s = a.lamda[b];
  bool ok = true;
  int slots[UD_NCELL], pcell[UD_NCELL];
#pragma unroll
  for (int t = 0; t < UD_NCELL; ++t) { slots[t] = 0; pcell[t] = -2; }
  const size_t ck_env = ((size_t)S * 24 * c.Np + (size_t)S * 10);
  float* ck = a.ckpt ? a.ckpt + (size_t)b * ck_env : nullptr;
  __syncthreads();
  fk_prologue(L, iq, S, pv, pw, tid);
  __syncthreads();
  for (int f = 0; f < S; ++f) {
    // ---- A: clear the cell table; compact lanes: land the previous g2p, checkpoint, particle pre-pass -> stage ----
    for (int e = tid, n = *L.count; e < n; e += nt) {   // clear the values of the slots seen so far
      const int s = L.list[e];
      L.acc[ACC_D(c.H, s, 0)] = 0.0; L.acc[ACC_D(c.H, s, 1)] = 0.0; L.acc[ACC_D(c.H, s, 2)] = 0.0; L.acc[ACC_D(c.H, s, 3)] = 0.0;
    }
    if (cwave) {
      if (f > 0) {
        const float4 r0 = ret[cp], r1 = ret[Np + cp], r2 = ret[2 * Np + cp];
        const float nv[3] = {r0.x, r0.y, r0.z};
        const float nC[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
        for (int d = 0; d < 3; ++d) { v[d] = nv[d]; x[d] = x[d] + c.dt * nv[d]; }
#pragma unroll
        for (int d = 0; d < 9; ++d) { Cm[d] = nC[d]; F[d] = Fn[d]; }
        const float trq = L.scr[0] + L.scr[1] + L.scr[2];   // J of the previous substep (:327, Q6): one scalar for all particles
        Jp = Jp * (1.f + c.dt * trq);
      }
      if (ck && clive) {
        float* r = ck + (size_t)f * 24 * c.Np + cp;
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[d * c.Np] = x[d]; r[(3 + d) * c.Np] = v[d]; }
#pragma unroll
        for (int d = 0; d < 9; ++d) { r[(6 + d) * c.Np] = Cm[d]; r[(15 + d) * c.Np] = F[d]; }
      }
      Pre q;
      particle_pre<false>(c, x, Cm, F, mu_s, la_s, material, hard, q, nullptr);
#pragma unroll
      for (int d = 0; d < 9; ++d) Fn[d] = q.Fn[d];
      if (clive) {
        stage[cp] = make_float4(__int_as_float(q.base[0]), __int_as_float(q.base[1]), __int_as_float(q.base[2]), q.fx[0]);
        stage[Np + cp] = make_float4(q.fx[1], q.fx[2], q.w[0], q.w[1]);
        stage[2 * Np + cp] = make_float4(q.w[2], q.w[3], q.w[4], q.w[5]);
        stage[3 * Np + cp] = make_float4(q.w[6], q.w[7], q.w[8], q.affine[0]);
        stage[4 * Np + cp] = make_float4(q.affine[1], q.affine[2], q.affine[3], q.affine[4]);
        stage[5 * Np + cp] = make_float4(q.affine[5], q.affine[6], q.affine[7], q.affine[8]);
        stage[6 * Np + cp] = make_float4(v[0], v[1], v[2], 0.f);
      }
    }
    __syncthreads();
    // ---- B: quads read their particle's pre-pass, p2g ----
    Pre q;
    float vq[3] = {0.f, 0.f, 0.f};
    if (live) {
      const float4 s0 = stage[p], s1 = stage[Np + p], s2 = stage[2 * Np + p], s3 = stage[3 * Np + p], s4 = stage[4 * Np + p],
                   s5 = stage[5 * Np + p], s6 = stage[6 * Np + p];
      q.base[0] = __float_as_int(s0.x); q.base[1] = __float_as_int(s0.y); q.base[2] = __float_as_int(s0.z);
      q.fx[0] = s0.w; q.fx[1] = s1.x; q.fx[2] = s1.y;
      q.w[0] = s1.z; q.w[1] = s1.w; q.w[2] = s2.x; q.w[3] = s2.y; q.w[4] = s2.z; q.w[5] = s2.w; q.w[6] = s3.x; q.w[7] = s3.y; q.w[8] = s3.z;
      q.affine[0] = s3.w; q.affine[1] = s4.x; q.affine[2] = s4.y; q.affine[3] = s4.z; q.affine[4] = s4.w;
      q.affine[5] = s5.x; q.affine[6] = s5.y; q.affine[7] = s5.z; q.affine[8] = s5.w;
      vq[0] = s6.x; vq[1] = s6.y; vq[2] = s6.z;
      ok = p2g_lane(c, L, q, vq, qi, slots, pcell) && ok;
    }
    __syncthreads();
    // ---- C: grid op on the occupied slots ----
    if (!(UD_MPM_ABLATE & 4)) {
      PrimF pf;
      prim_from_lds(L, iq, f, psize, pv, friction, pf);
      for (int e = tid, n = *L.count; e < n; e += nt) {
        const int s = L.list[e];
        int ci, cj, ckk;
        decode_cell(c, L.key[s], ci, cj, ckk);
        const float mm = (float)L.acc[ACC_D(c.H, s, 0)];
        float mvv[3] = {(float)L.acc[ACC_D(c.H, s, 1)], (float)L.acc[ACC_D(c.H, s, 2)], (float)L.acc[ACC_D(c.H, s, 3)]}, vo[3];
        grid_op<false>(c, pf, ci, cj, ckk, mm, mvv, vo, nullptr);
        accf[ACC_F(c.H, s, 4)] = vo[0]; accf[ACC_F(c.H, s, 5)] = vo[1]; accf[ACC_F(c.H, s, 6)] = vo[2];   // same thread read the doubles above
      }
    }
    __syncthreads();
    // ---- D: g2p (:196-221): quads gather, lane 0 of the quad returns (v_new, C_new) to the particle's compact lane ----
    float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live && !(UD_MPM_ABLATE & 8)) {
#pragma unroll
      for (int t = 0; t < UD_NCELL; ++t) {
        int i, j, k;
        if (!cell_of(qi, t, i, j, k)) continue;
        const int gs = slots[t] & 0xffff;
        const float weight = sel3(q.w, 0, i) * sel3(q.w, 1, j) * sel3(q.w, 2, k);
        const float dp[3] = {(float)i - q.fx[0], (float)j - q.fx[1], (float)k - q.fx[2]};
        const float g[3] = {accf[ACC_F(c.H, gs, 4)], accf[ACC_F(c.H, gs, 5)], accf[ACC_F(c.H, gs, 6)]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g[r];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) nv[d] = quad_sum(nv[d]);
#pragma unroll
    for (int d = 0; d < 9; ++d) nC[d] = quad_sum(nC[d]);
    if (live && qi == 0) {
      ret[p] = make_float4(nv[0], nv[1], nv[2], nC[0]);
      ret[Np + p] = make_float4(nC[1], nC[2], nC[3], nC[4]);
      ret[2 * Np + p] = make_float4(nC[5], nC[6], nC[7], nC[8]);
      if (p < 3) {   // row p of particle p (Q6)
        const float r0 = nC[0] + nC[1] + nC[2], r1 = nC[3] + nC[4] + nC[5], r2 = nC[6] + nC[7] + nC[8];
        L.scr[p] = (p == 0) ? r0 : ((p == 1) ? r1 : r2);
      }
    }
    if (tid == 0 && N < 3) { for (int e = N; e < 3; ++e) L.scr[e] = 0.f; }
    __syncthreads();
  }
  if (cwave) {   // land the last g2p
    const float4 r0 = ret[cp], r1 = ret[Np + cp], r2 = ret[2 * Np + cp];
    const float nv[3] = {r0.x, r0.y, r0.z};
    const float nC[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
    for (int d = 0; d < 3; ++d) { v[d] = nv[d]; x[d] = x[d] + c.dt * nv[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { Cm[d] = nC[d]; F[d] = Fn[d]; }
    const float trq = L.scr[0] + L.scr[1] + L.scr[2];
    Jp = Jp * (1.f + c.dt * trq);
  }
  if (clive) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.xo[((size_t)b * N + cp) * 3 + d] = x[d]; a.vo[((size_t)b * N + cp) * 3 + d] = v[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { a.Co[((size_t)b * N + cp) * 9 + d] = Cm[d]; a.Fo[((size_t)b * N + cp) * 9 + d] = F[d]; }
    a.Jo[(size_t)b * N + cp] = Jp;
  }
  // checkpoint tail: primitive arrays before copy_frame (position | rotation)
  if (ck) {
    float* tail = ck + (size_t)S * 24 * c.Np;
    for (int e = tid; e < S * 3; e += nt) tail[e] = L.ppos[e];
    for (int e = tid; e < S * 4; e += nt) tail[S * 3 + e] = L.prot[e];
    for (int e = tid; e < S * 3; e += nt) tail[S * 7 + e] = a.ppos[(size_t)b * S * 3 + e];
  }
  // copy_frame(steps, 0): source index clamps to steps-1 (Q5)
  for (int e = tid; e < S * 3; e += nt) {
    const int row = e / 3, d = e - row * 3;
    a.ppos_o[(size_t)b * S * 3 + e] = (row == 0) ? L.ppos[(S - 1) * 3 + d] : L.ppos[e];
    a.pv_o[(size_t)b * S * 3 + e] = (d == 0) ? pv[0] : ((d == 1) ? pv[1] : pv[2]);
    a.pw_o[(size_t)b * S * 3 + e] = (d == 0) ? pw[0] : ((d == 1) ? pw[1] : pw[2]);
  }
  for (int e = tid; e < S * 4; e += nt) {
    const int row = e / 4, d = e - row * 4;
    a.prot_o[(size_t)b * S * 4 + e] = (row == 0) ? L.prot[(S - 1) * 4 + d] : L.prot[e];
  }
  const int bad = __syncthreads_or(ok ? 0 : 1);
  if (tid == 0 && a.status) a.status[b] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum1(float v, float* red, int nw) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int q = 0; q < nw; ++q) t += red[q];
  return t;
}

// pre-clip value of primitive position entry (row j, component a) at substep f (see forward_kinematics :185-187)
__device__ __forceinline__ float ppos_preclip(const Lds& L, int f, int S, int j, int a, float pva) {
  if (j == f + 1) return L.ppos[f * 3 + a] + pva;
  if (j <= f) return (f == 0) ? L.ppin[j * 3 + a] : L.ppos[j * 3 + a];
  return (f == 0) ? L.ppin[j * 3 + a] : clipf(L.ppin[j * 3 + a], -2.f, 2.f);
}

constexpr int UD_PARK = 12;   // float4 per particle parked across the stencil phases of the adjoint

__global__ void __launch_bounds__(512) mpm_step_bwd_kernel(MpmBwdArgs a) {
  extern __shared__ float smem[];
  const MpmConst c = a.c;
  const int tid = threadIdx.x, b = blockIdx.x, nt = blockDim.x;
  const int N = c.N, S = c.steps, H = c.H;
  const int nw = nt >> 6;
  Lds L;
  L.key = (int*)smem; L.acc = (double*)(smem + H); L.gacc = (double*)(smem + 9 * H);
  L.gpv = (double*)(smem + 15 * H);
  // `park`: what only the particle adjoint at the end of the iteration needs from the pre-pass (Fn, U, Vh, sigma, A:
  // 45 floats per particle) waits in LDS while the stencil phases run, instead of holding them live in every lane
  float4* park = (float4*)(smem + 15 * H + 6 * S + (S & 1) * 2);
  L.ppos = (float*)(park + UD_PARK * c.Np); L.prot = L.ppos + S * 3; L.ppin = L.prot + S * 4; L.gppos = L.ppin + S * 3;
  L.scr = L.gppos + S * 3;
  L.list = (int*)(L.scr + 64); L.count = L.list + H;
  float* accf = (float*)L.acc;
  float* gaccf = (float*)L.gacc;
  const int p = tid >> 2, qi = tid & 3;
  const bool live = p < N;
  const int pc = live ? p : 0;
  const int material = a.material[pc];
  const float hard = a.hard[pc];
  const size_t ck_env = ((size_t)S * 24 * c.Np + (size_t)S * 10);
  const float* ck = a.ckpt + (size_t)b * ck_env;
  {
    const float* tail = ck + (size_t)S * 24 * c.Np;
    for (int e = tid; e < S * 3; e += nt) { L.ppos[e] = tail[e]; L.ppin[e] = tail[S * 7 + e]; L.gpv[e] = 0.0; }
    for (int e = tid; e < S * 4; e += nt) L.prot[e] = tail[S * 3 + e];
    for (int s = tid; s < H; s += nt) {
      L.key[s] = -1; L.acc[ACC_D(c.H, s, 0)] = 0.0;
      for (int d = 0; d < 3; ++d) { L.acc[ACC_D(c.H, s, 1 + d)] = 0.0; L.gacc[ACC_D(c.H, s, d)] = 0.0; }
    }
    if (tid == 0) *L.count = 0;
    // copy_frame adjoint: position[0] <- position[steps-1]
    for (int e = tid; e < S * 3; e += nt) {
      const int row = e / 3, d = e - row * 3;
      float g = a.gppos[(size_t)b * S * 3 + e];
      if (S > 1) {
        if (row == 0) g = 0.f;
        if (row == S - 1) g += a.gppos[(size_t)b * S * 3 + d];
      }
      L.gppos[e] = g;
    }
  }
  float gx[3], gv[3], gC[9], gF[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) { gx[d] = a.gx[((size_t)b * N + pc) * 3 + d]; gv[d] = a.gv[((size_t)b * N + pc) * 3 + d]; }
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[d] = a.gC[((size_t)b * N + pc) * 9 + d]; gF[d] = a.gF[((size_t)b * N + pc) * 9 + d]; }
  float ac[6], pv[3], pw[3], psize[3];
#pragma unroll
  for (int d = 0; d < 6; ++d) ac[d] = clipf(a.action[b * 6 + d], -1.f, 1.f);
#pragma unroll
  for (int d = 0; d < 3; ++d) { pv[d] = ac[d] * 1.f / (float)S; pw[d] = ac[3 + d] * 1.f / (float)S; psize[d] = a.psize[b * 3 + d]; }
  const float friction = a.friction[b], mu_s = a.mu[b], la_s = a.lamda[b];
  float acc_fric = 0.f, acc_mu = 0.f, acc_la = 0.f;
  bool ok = true;
  int slots[UD_NCELL], pcell[UD_NCELL];
#pragma unroll
  for (int t = 0; t < UD_NCELL; ++t) { slots[t] = 0; pcell[t] = -2; }
  float pend_val = 0.f, pend_pv = 0.f;   // FK-adjoint values computed in phase F, written at the top of the next iteration
  bool pend = false;
  __syncthreads();
  for (int f = S - 1; f >= 0; --f) {
    float x[3], v[3], Cm[9], F[9];
    {
      const float* r = ck + (size_t)f * 24 * c.Np + pc;
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = r[d * c.Np]; v[d] = r[(3 + d) * c.Np]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { Cm[d] = r[(6 + d) * c.Np]; F[d] = r[(15 + d) * c.Np]; }
    }
    // ---- A: clear the table; land the FK-adjoint writes of the previous iteration ----
    for (int e = tid, n = *L.count; e < n; e += nt) {
      const int s = L.list[e];
      L.acc[ACC_D(c.H, s, 0)] = 0.0;
#pragma unroll
      for (int d = 0; d < 3; ++d) { L.acc[ACC_D(c.H, s, 1 + d)] = 0.0; L.gacc[ACC_D(c.H, s, d)] = 0.0; }
    }
    if (pend && tid < S * 3) { L.gppos[tid] = pend_val; L.gpv[tid] += (double)pend_pv; }
    __syncthreads();
    // ---- B: particle pre-pass, p2g ----
    Pre q;
    PreB kb;
    particle_pre<true>(c, x, Cm, F, mu_s, la_s, material, hard, q, &kb);
    if (live && qi == 0) {
      float pk[UD_PARK * 4];
#pragma unroll
      for (int d = 0; d < 3; ++d) { pk[d] = kb.sig_raw[d]; pk[3 + d] = kb.sig[d]; }
      pk[6] = kb.Jd; pk[7] = kb.mu; pk[8] = kb.la;
#pragma unroll
      for (int d = 0; d < 9; ++d) { pk[9 + d] = q.Fn[d]; pk[18 + d] = kb.U[d]; pk[27 + d] = kb.Vh[d]; pk[36 + d] = kb.A[d]; }
      pk[45] = 0.f; pk[46] = 0.f; pk[47] = 0.f;
#pragma unroll
      for (int k = 0; k < UD_PARK; ++k) park[k * c.Np + p] = make_float4(pk[k * 4], pk[k * 4 + 1], pk[k * 4 + 2], pk[k * 4 + 3]);
    }
    if (live) ok = p2g_lane(c, L, q, v, qi, slots, pcell) && ok;
    __syncthreads();
    // ---- C: grid op forward -> vel ----
    PrimF pf;
    prim_at(L, f, S, psize, pv, friction, pf);
    for (int e = tid, n = *L.count; e < n; e += nt) {
      const int s = L.list[e];
      int ci, cj, ckk;
      decode_cell(c, L.key[s], ci, cj, ckk);
      const float mm = (float)L.acc[ACC_D(c.H, s, 0)];
      float mvv[3] = {(float)L.acc[ACC_D(c.H, s, 1)], (float)L.acc[ACC_D(c.H, s, 2)], (float)L.acc[ACC_D(c.H, s, 3)]}, vo[3];
      grid_op<false>(c, pf, ci, cj, ckk, mm, mvv, vo, nullptr);
      accf[ACC_F(c.H, s, 0)] = mm; accf[ACC_F(c.H, s, 1)] = mvv[0]; accf[ACC_F(c.H, s, 2)] = mvv[1]; accf[ACC_F(c.H, s, 3)] = mvv[2];   // raw values for the adjoint
      accf[ACC_F(c.H, s, 4)] = vo[0]; accf[ACC_F(c.H, s, 5)] = vo[1]; accf[ACC_F(c.H, s, 6)] = vo[2];
    }
    __syncthreads();
    // ---- D: g2p adjoint (scatter g onto grid velocities; weight / fx cotangents) ----
    float gnv[3], gw[9], gfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) gnv[d] = gv[d] + c.dt * gx[d];       // x_out = x + dt*v_new
#pragma unroll
    for (int d = 0; d < 9; ++d) gw[d] = 0.f;
    if (live) {
#pragma unroll
      for (int t = 0; t < UD_NCELL; ++t) {
        int i, j, k;
        if (!cell_of(qi, t, i, j, k)) continue;
        const int gs = slots[t] & 0xffff;
        const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
        const float weight = wi * wj * wk;
        const float dp[3] = {(float)i - q.fx[0], (float)j - q.fx[1], (float)k - q.fx[2]};
        float gwt = 0.f, gsc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gCd = gC[r * 3] * dp[0] + gC[r * 3 + 1] * dp[1] + gC[r * 3 + 2] * dp[2];
          const float vel = accf[ACC_F(c.H, gs, 4 + r)];
          gsc[r] = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
          gwt += vel * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gC[r * 3 + s2] * vel;
        }
        if (quadrun_reduce<3>(gs, gsc)) {
#pragma unroll
          for (int r = 0; r < 3; ++r) lds_add(&L.gacc[ACC_D(c.H, gs, r)], gsc[r]);
        }
        // gw[k*3+d]: select-accumulate (i, j, k are compile-time after unrolling only through cidx = q + 4t)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
          gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
          gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- E: grid-op adjoint per occupied slot ----
    for (int e = tid, n = *L.count; e < n; e += nt) {
      const int s = L.list[e];
      int ci, cj, ckk;
      decode_cell(c, L.key[s], ci, cj, ckk);
      const float m = accf[ACC_F(c.H, s, 0)];
      float mvv[3] = {accf[ACC_F(c.H, s, 1)], accf[ACC_F(c.H, s, 2)], accf[ACC_F(c.H, s, 3)]};
      float g[3] = {(float)L.gacc[ACC_D(c.H, s, 0)], (float)L.gacc[ACC_D(c.H, s, 1)], (float)L.gacc[ACC_D(c.H, s, 2)]}, gmm, dfric, dpv[3];
      const bool ctrl = grid_op_adjoint(c, pf, ci, cj, ckk, m, mvv, g, gmm, dfric, dpv);
      acc_fric += dfric;
      if (ctrl) {
#pragma unroll
        for (int d = 0; d < 3; ++d) lds_add(&L.gpv[f * 3 + d], dpv[d]);
      }
      gaccf[ACC_F(c.H, s, 0)] = g[0]; gaccf[ACC_F(c.H, s, 1)] = g[1]; gaccf[ACC_F(c.H, s, 2)] = g[2]; gaccf[ACC_F(c.H, s, 3)] = gmm;   // same thread read the doubles
    }
    __syncthreads();
    // ---- F: p2g adjoint (gather) + particle pre-pass adjoint + FK adjoint ----
    float gaff[9], gvp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 9; ++d) gaff[d] = 0.f;
    if (live) {
#pragma unroll
      for (int t = 0; t < UD_NCELL; ++t) {
        int i, j, k;
        if (!cell_of(qi, t, i, j, k)) continue;
        const int ss = (slots[t] >> 16) - 1;
        if (ss < 0) continue;
        const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
        const float weight = wi * wj * wk;
        const float dpos[3] = {((float)i - q.fx[0]) * c.dx, ((float)j - q.fx[1]) * c.dx, ((float)k - q.fx[2]) * c.dx};
        float gwt = c.p_mass * gaccf[ACC_F(c.H, ss, 3)];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gc = gaccf[ACC_F(c.H, ss, r)];
          const float ad = q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2];
          gwt += gc * (c.p_mass * v[r] + ad);
          gvp[r] += weight * c.p_mass * gc;
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) {
            gaff[r * 3 + s2] += weight * gc * dpos[s2];
            gfx[s2] -= c.dx * weight * gc * q.affine[r * 3 + s2];
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
    for (int d = 0; d < 9; ++d) { gw[d] = quad_sum(gw[d]); gaff[d] = quad_sum(gaff[d]); }
#pragma unroll
    for (int d = 0; d < 3; ++d) { gfx[d] = quad_sum(gfx[d]); gvp[d] = quad_sum(gvp[d]); }
    {
      Pre qa; PreB ka;
      float Ca[9], Fa[9], pk[UD_PARK * 4];
#pragma unroll
      for (int k = 0; k < UD_PARK; ++k) { const float4 t = park[k * c.Np + pc]; pk[k * 4] = t.x; pk[k * 4 + 1] = t.y; pk[k * 4 + 2] = t.z; pk[k * 4 + 3] = t.w; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { qa.fx[d] = q.fx[d]; ka.sig_raw[d] = pk[d]; ka.sig[d] = pk[3 + d]; }
      ka.Jd = pk[6]; ka.mu = pk[7]; ka.la = pk[8];
#pragma unroll
      for (int d = 0; d < 9; ++d) { qa.Fn[d] = pk[9 + d]; ka.U[d] = pk[18 + d]; ka.Vh[d] = pk[27 + d]; ka.A[d] = pk[36 + d]; }
      {  // C and F of this substep come back from the checkpoint (L2-resident) rather than from LDS
        const float* r = ck + (size_t)f * 24 * c.Np + pc;
#pragma unroll
        for (int d = 0; d < 9; ++d) { Ca[d] = r[(6 + d) * c.Np]; Fa[d] = r[(15 + d) * c.Np]; }
      }
      float gmu_p, gla_p;
      particle_adjoint(c, qa, ka, Ca, Fa, material, gw, gfx, gaff, gvp, gx, gv, gC, gF, gmu_p, gla_p);
      const float h = clipf(hard, 0.1f, 5.f);
      if (live && qi == 0 && material != 0) { acc_mu += gmu_p * h; acc_la += gla_p * h; }
    }
    // FK adjoint (:185-187): position' = clip(set(position, f+1, position[f] + v[f])); reads now, writes in A
    pend = true;
    pend_val = 0.f; pend_pv = 0.f;
    if (tid < S * 3) {
      const int row = tid / 3, d = tid - row * 3;
      const float pva = (d == 0) ? pv[0] : ((d == 1) ? pv[1] : pv[2]);
      const float mine = L.gppos[tid] * clip_grad(ppos_preclip(L, f, S, row, d, pva), -2.f, 2.f);
      float val = mine;
      if (f + 1 < S) {
        if (row == f + 1) val = 0.f;
        if (row == f) {
          const float t = L.gppos[tid + 3] * clip_grad(ppos_preclip(L, f, S, f + 1, d, pva), -2.f, 2.f);
          val += t;
          pend_pv = t;
        }
      }
      pend_val = val;
    }
    __syncthreads();
  }
  if (pend && tid < S * 3) { L.gppos[tid] = pend_val; L.gpv[tid] += (double)pend_pv; }
  __syncthreads();
  // ---- step boundary: set_action adjoint, action clip, norm_grad(_state) (:375-411, :419-423) ----
  float* red = L.scr;
  float tot_fric = block_sum1(acc_fric, red, nw);
  float tot_mu = block_sum1(acc_mu, red, nw);
  float tot_la = block_sum1(acc_la, red, nw);
  float ga[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gscale[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < S; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float t = (float)L.gpv[j * 3 + d]; ga[d] += t * 1.f / (float)S; gscale[d] += t * ac[d] / (float)S; }
  // rotation path (action[3:6]): the reference's d|w|/dw at w = 0 is NaN and nan_to_num zeroes it here -> 0
#pragma unroll
  for (int d = 0; d < 6; ++d) ga[d] *= clip_grad(a.action[b * 6 + d], -1.f, 1.f);
  if (a.clip) {
    float n2 = 0.f;
#pragma unroll
    for (int d = 0; d < 6; ++d) { ga[d] = nan_to_num(ga[d] + 0.f); n2 += ga[d] * ga[d]; }
    const float nrm = sqrtf(n2);
    if (!(nrm < 1.f)) {
#pragma unroll
      for (int d = 0; d < 6; ++d) ga[d] = ga[d] / nrm;
    }
    float s2 = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = nan_to_num(gx[d] + 0.f); gv[d] = nan_to_num(gv[d] + 0.f); }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = nan_to_num(gC[d] + 0.f); gF[d] = nan_to_num(gF[d] + 0.f); }
    if (live && qi == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) s2 += gx[d] * gx[d] + gv[d] * gv[d];
#pragma unroll
      for (int d = 0; d < 9; ++d) s2 += gC[d] * gC[d] + gF[d] * gF[d];
    }
    if (tid < S * 3) { const float t = nan_to_num(L.gppos[tid] + 0.f); L.gppos[tid] = t; s2 += t * t; }
    tot_fric = nan_to_num(tot_fric); tot_mu = nan_to_num(tot_mu); tot_la = nan_to_num(tot_la);
    if (tid == 0) {
      s2 += tot_fric * tot_fric + tot_mu * tot_mu + tot_la * tot_la;
#pragma unroll
      for (int d = 0; d < 3; ++d) { const float t = nan_to_num(gscale[d]); s2 += t * t; }
    }
    const float sn = sqrtf(block_sum1(s2, red, nw));
    if (!(sn < 1.f)) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { gx[d] = gx[d] / sn; gv[d] = gv[d] / sn; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { gC[d] = gC[d] / sn; gF[d] = gF[d] / sn; }
      if (tid < S * 3) L.gppos[tid] = L.gppos[tid] / sn;
      tot_fric = tot_fric / sn; tot_mu = tot_mu / sn; tot_la = tot_la / sn;
    }
  }
  if (live && qi == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.gx0[((size_t)b * N + p) * 3 + d] = gx[d]; a.gv0[((size_t)b * N + p) * 3 + d] = gv[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { a.gC0[((size_t)b * N + p) * 9 + d] = gC[d]; a.gF0[((size_t)b * N + p) * 9 + d] = gF[d]; }
  }
  if (tid < S * 3) a.gppos0[(size_t)b * S * 3 + tid] = L.gppos[tid];
  if (tid == 0) {
    a.gfric[b] = tot_fric; a.gmu[b] = tot_mu; a.glam[b] = tot_la;
#pragma unroll
    for (int d = 0; d < 6; ++d) a.gaction[b * 6 + d] = ga[d];
  }
  const int bad = __syncthreads_or(ok ? 0 : 1);
  if (tid == 0 && a.status) a.status[b] = bad ? 1 : 0;
}


// ------------------------------------------------------------------------------------------------
// backward, wave-specialised (N <= 96)
// ------------------------------------------------------------------------------------------------
// The one-mapping adjoint above keeps the particle pre-pass / SVD-VJP state and the stencil state live in the same
// lanes (256 VGPRs + scratch traffic inside the per-cell loops).  Here the first ceil(N/64) waves are PARTICLE waves
// (lane = particle: checkpoint load, pre-pass with adjoint extras, particle adjoint) and the remaining waves are STENCIL
// waves (lane = 4*particle + q: p2g recompute, grid op, g2p adjoint scatter, grid-op adjoint, p2g adjoint gather).  The
// two roles are separate loop nests behind one wave-uniform branch, so each gets its own register allocation, and
// they overlap in time: while the stencil waves run substep f, the particle waves finish the particle adjoint of
// substep f+1 (ready before the g2p adjoint needs its result) and then pre-compute the pre-pass of substep f-1 (it
// only needs the checkpoint).  Hand-offs go through LDS (stage: particle -> stencil, ret: stencil -> particle); both
// roles execute the same six workgroup barriers per substep.
constexpr int UD_WS_STG = 10, UD_WS_RET = 6;   // float4 per particle

struct PartSet { float fx[3], Fn[9], U[9], Vh[9], A[9], sig_raw[3], sig[3], Jd, mu, la, Cm[9], F[9]; };

__global__ void __launch_bounds__(512) mpm_step_bwd_ws_kernel(MpmBwdArgs a) {
  extern __shared__ float smem[];
  const MpmConst c = a.c;
  const int tid = threadIdx.x, b = blockIdx.x, nt = blockDim.x;
  const int N = c.N, S = c.steps, H = c.H, Np = c.Np;
  const int nw = nt >> 6;
  const int nP = (N + 63) / 64 * 64;                  // particle threads first
  const int nS = nt - nP;                             // stencil threads
  const bool prole = tid < nP;
  Lds L;
  L.key = (int*)smem; L.acc = (double*)(smem + H); L.gacc = (double*)(smem + 9 * H);
  L.gpv = (double*)(smem + 15 * H);
  float4* stage = (float4*)(smem + 15 * H + 6 * S + (S & 1) * 2);
  float4* ret = stage + UD_WS_STG * Np;
  L.ppos = (float*)(ret + UD_WS_RET * Np); L.prot = L.ppos + S * 3; L.ppin = L.prot + S * 4; L.gppos = L.ppin + S * 3;
  L.scr = L.gppos + S * 3;
  L.list = (int*)(L.scr + 64); L.count = L.list + H;
  float* accf = (float*)L.acc;
  float* gaccf = (float*)L.gacc;
  // particle role
  const bool clive = prole && tid < N;
  const int cp = clive ? tid : 0;
  // stencil role
  const int ft = tid - nP;                            // stencil thread id (>= 0 in the stencil role)
  const int p = ft >> 2, qi = ft & 3;
  const bool live = !prole && p < N;
  const bool fkl = !prole && ft < S * 3;              // lanes that own the primitive-position cotangent entries
  const size_t ck_env = ((size_t)S * 24 * c.Np + (size_t)S * 10);
  const float* ck = a.ckpt + (size_t)b * ck_env;
  {
    const float* tail = ck + (size_t)S * 24 * c.Np;
    for (int e = tid; e < S * 3; e += nt) { L.ppos[e] = tail[e]; L.ppin[e] = tail[S * 7 + e]; L.gpv[e] = 0.0; }
    for (int e = tid; e < S * 4; e += nt) L.prot[e] = tail[S * 3 + e];
    for (int s = tid; s < H; s += nt) {
      L.key[s] = -1; L.acc[ACC_D(c.H, s, 0)] = 0.0;
      for (int d = 0; d < 3; ++d) { L.acc[ACC_D(c.H, s, 1 + d)] = 0.0; L.gacc[ACC_D(c.H, s, d)] = 0.0; }
    }
    if (tid == 0) *L.count = 0;
    for (int e = tid; e < S * 3; e += nt) {           // copy_frame adjoint: position[0] <- position[steps-1]
      const int row = e / 3, d = e - row * 3;
      float g = a.gppos[(size_t)b * S * 3 + e];
      if (S > 1) {
        if (row == 0) g = 0.f;
        if (row == S - 1) g += a.gppos[(size_t)b * S * 3 + d];
      }
      L.gppos[e] = g;
    }
  }
  float ac[6], pv[3], pw[3], psize[3];
#pragma unroll
  for (int d = 0; d < 6; ++d) ac[d] = clipf(a.action[b * 6 + d], -1.f, 1.f);
#pragma unroll
  for (int d = 0; d < 3; ++d) { pv[d] = ac[d] * 1.f / (float)S; pw[d] = ac[3 + d] * 1.f / (float)S; psize[d] = a.psize[b * 3 + d]; }
  const float friction = a.friction[b], mu_s = a.mu[b], la_s = a.lamda[b];
  float acc_fric = 0.f, acc_mu = 0.f, acc_la = 0.f;
  bool ok = true;
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f}, gC[9], gF[9];   // particle role: cotangents of the particle state
#pragma unroll
  for (int d = 0; d < 9; ++d) { gC[d] = 0.f; gF[d] = 0.f; }
  __syncthreads();

  if (prole) {
    // ================================ particle waves ================================
    const int material = a.material[cp];
    const float hard = a.hard[cp];
    const float hcl = clipf(hard, 0.1f, 5.f);
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = a.gx[((size_t)b * N + cp) * 3 + d]; gv[d] = a.gv[((size_t)b * N + cp) * 3 + d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = a.gC[((size_t)b * N + cp) * 9 + d]; gF[d] = a.gF[((size_t)b * N + cp) * 9 + d]; }
    // pre-pass of substep f: keeps what the particle adjoint needs in `out`, hands the stencil part to the quads
    auto prepass = [&](int f, PartSet& out) {
      float x[3], v[3];
      const float* r = ck + (size_t)f * 24 * c.Np + cp;
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = r[d * c.Np]; v[d] = r[(3 + d) * c.Np]; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { out.Cm[d] = r[(6 + d) * c.Np]; out.F[d] = r[(15 + d) * c.Np]; }
      Pre q; PreB kb;
      particle_pre<true>(c, x, out.Cm, out.F, mu_s, la_s, material, hard, q, &kb);
#pragma unroll
      for (int d = 0; d < 3; ++d) { out.fx[d] = q.fx[d]; out.sig_raw[d] = kb.sig_raw[d]; out.sig[d] = kb.sig[d]; }
      out.Jd = kb.Jd; out.mu = kb.mu; out.la = kb.la;
#pragma unroll
      for (int d = 0; d < 9; ++d) { out.Fn[d] = q.Fn[d]; out.U[d] = kb.U[d]; out.Vh[d] = kb.Vh[d]; out.A[d] = kb.A[d]; }
      if (clive) {
        stage[cp] = make_float4(__int_as_float(q.base[0]), __int_as_float(q.base[1]), __int_as_float(q.base[2]), q.fx[0]);
        stage[Np + cp] = make_float4(q.fx[1], q.fx[2], q.w[0], q.w[1]);
        stage[2 * Np + cp] = make_float4(q.w[2], q.w[3], q.w[4], q.w[5]);
        stage[3 * Np + cp] = make_float4(q.w[6], q.w[7], q.w[8], q.affine[0]);
        stage[4 * Np + cp] = make_float4(q.affine[1], q.affine[2], q.affine[3], q.affine[4]);
        stage[5 * Np + cp] = make_float4(q.affine[5], q.affine[6], q.affine[7], q.affine[8]);
        stage[6 * Np + cp] = make_float4(v[0], v[1], v[2], 0.f);
      }
    };
    PartSet cur, prevset;
    prepass(S - 1, cur);
    prevset = cur;
    for (int f = S - 1; f >= 0; --f) {
      __syncthreads();   // b1: table cleared, stage[0..6] of substep f visible
      if (f != S - 1) {  // finish the particle adjoint of substep f+1 (its stencil cotangents arrived under b6)
        Pre q; PreB kb;
        float gw[9], gfx[3], gaff[9], gvp[3], rt[UD_WS_RET * 4];
#pragma unroll
        for (int k = 0; k < UD_WS_RET; ++k) { const float4 t = ret[k * Np + cp]; rt[k * 4] = t.x; rt[k * 4 + 1] = t.y; rt[k * 4 + 2] = t.z; rt[k * 4 + 3] = t.w; }
#pragma unroll
        for (int d = 0; d < 9; ++d) { gw[d] = rt[d]; gaff[d] = rt[12 + d]; }
#pragma unroll
        for (int d = 0; d < 3; ++d) { gfx[d] = rt[9 + d]; gvp[d] = rt[21 + d]; }
        // NOTE: `prev` (substep f+1) was saved before `cur` was overwritten with substep f -- see the end of the loop body
#pragma unroll
        for (int d = 0; d < 3; ++d) { q.fx[d] = prevset.fx[d]; kb.sig_raw[d] = prevset.sig_raw[d]; kb.sig[d] = prevset.sig[d]; }
        kb.Jd = prevset.Jd; kb.mu = prevset.mu; kb.la = prevset.la;
#pragma unroll
        for (int d = 0; d < 9; ++d) { q.Fn[d] = prevset.Fn[d]; kb.U[d] = prevset.U[d]; kb.Vh[d] = prevset.Vh[d]; kb.A[d] = prevset.A[d]; }
        float gmu_p, gla_p;
        particle_adjoint(c, q, kb, prevset.Cm, prevset.F, material, gw, gfx, gaff, gvp, gx, gv, gC, gF, gmu_p, gla_p);
        if (clive && material != 0) { acc_mu += gmu_p * hcl; acc_la += gla_p * hcl; }
      }
      if (clive) {       // cotangents entering substep f's g2p adjoint
        stage[7 * Np + cp] = make_float4(gv[0] + c.dt * gx[0], gv[1] + c.dt * gx[1], gv[2] + c.dt * gx[2], gC[0]);   // x_out = x + dt*v_new
        stage[8 * Np + cp] = make_float4(gC[1], gC[2], gC[3], gC[4]);
        stage[9 * Np + cp] = make_float4(gC[5], gC[6], gC[7], gC[8]);
      }
      __syncthreads();   // b2
      __syncthreads();   // b3: stage[7..9] visible to the g2p adjoint
      __syncthreads();   // b4: the quads are done reading stage
      prevset = cur;     // substep f's set waits for its stencil cotangents
      if (f > 0) prepass(f - 1, cur);   // overlaps the stencil waves' grid-op adjoint and gather
      __syncthreads();   // b5
      __syncthreads();   // b6: ret of substep f visible
    }
    {  // particle adjoint of substep 0
      Pre q; PreB kb;
      float gw[9], gfx[3], gaff[9], gvp[3], rt[UD_WS_RET * 4];
#pragma unroll
      for (int k = 0; k < UD_WS_RET; ++k) { const float4 t = ret[k * Np + cp]; rt[k * 4] = t.x; rt[k * 4 + 1] = t.y; rt[k * 4 + 2] = t.z; rt[k * 4 + 3] = t.w; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { gw[d] = rt[d]; gaff[d] = rt[12 + d]; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { gfx[d] = rt[9 + d]; gvp[d] = rt[21 + d]; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { q.fx[d] = prevset.fx[d]; kb.sig_raw[d] = prevset.sig_raw[d]; kb.sig[d] = prevset.sig[d]; }
      kb.Jd = prevset.Jd; kb.mu = prevset.mu; kb.la = prevset.la;
#pragma unroll
      for (int d = 0; d < 9; ++d) { q.Fn[d] = prevset.Fn[d]; kb.U[d] = prevset.U[d]; kb.Vh[d] = prevset.Vh[d]; kb.A[d] = prevset.A[d]; }
      float gmu_p, gla_p;
      particle_adjoint(c, q, kb, prevset.Cm, prevset.F, material, gw, gfx, gaff, gvp, gx, gv, gC, gF, gmu_p, gla_p);
      if (clive && material != 0) { acc_mu += gmu_p * hcl; acc_la += gla_p * hcl; }
    }
  } else {
    // ================================ stencil waves ================================
    int slots[UD_NCELL], pcell[UD_NCELL];
#pragma unroll
    for (int t = 0; t < UD_NCELL; ++t) { slots[t] = 0; pcell[t] = -2; }
    float pend_val = 0.f, pend_pv = 0.f;   // FK-adjoint values computed in phase F, written at the top of the next iteration
    bool pend = false;
    for (int f = S - 1; f >= 0; --f) {
      // ---- A: clear the table; land the FK-adjoint writes of the previous iteration ----
      for (int e = ft, n = *L.count; e < n; e += nS) {
        const int s = L.list[e];
        L.acc[ACC_D(c.H, s, 0)] = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) { L.acc[ACC_D(c.H, s, 1 + d)] = 0.0; L.gacc[ACC_D(c.H, s, d)] = 0.0; }
      }
      if (pend && fkl) { L.gppos[ft] = pend_val; L.gpv[ft] += (double)pend_pv; }
      __syncthreads();   // b1
      // ---- B: read the particle's pre-pass, p2g ----
      Pre q;
      float vq[3] = {0.f, 0.f, 0.f};
      if (live) {
        const float4 s0 = stage[p], s1 = stage[Np + p], s2 = stage[2 * Np + p], s3 = stage[3 * Np + p], s4 = stage[4 * Np + p],
                     s5 = stage[5 * Np + p], s6 = stage[6 * Np + p];
        q.base[0] = __float_as_int(s0.x); q.base[1] = __float_as_int(s0.y); q.base[2] = __float_as_int(s0.z);
        q.fx[0] = s0.w; q.fx[1] = s1.x; q.fx[2] = s1.y;
        q.w[0] = s1.z; q.w[1] = s1.w; q.w[2] = s2.x; q.w[3] = s2.y; q.w[4] = s2.z; q.w[5] = s2.w; q.w[6] = s3.x; q.w[7] = s3.y; q.w[8] = s3.z;
        q.affine[0] = s3.w; q.affine[1] = s4.x; q.affine[2] = s4.y; q.affine[3] = s4.z; q.affine[4] = s4.w;
        q.affine[5] = s5.x; q.affine[6] = s5.y; q.affine[7] = s5.z; q.affine[8] = s5.w;
        vq[0] = s6.x; vq[1] = s6.y; vq[2] = s6.z;
        ok = p2g_lane(c, L, q, vq, qi, slots, pcell) && ok;
      }
      __syncthreads();   // b2
      // ---- C: grid op forward -> vel ----
      PrimF pf;
      prim_at(L, f, S, psize, pv, friction, pf);
      for (int e = ft, n = *L.count; e < n; e += nS) {
        const int s = L.list[e];
        int ci, cj, ckk;
        decode_cell(c, L.key[s], ci, cj, ckk);
        const float mm = (float)L.acc[ACC_D(c.H, s, 0)];
        float mvv[3] = {(float)L.acc[ACC_D(c.H, s, 1)], (float)L.acc[ACC_D(c.H, s, 2)], (float)L.acc[ACC_D(c.H, s, 3)]}, vo[3];
        grid_op<false>(c, pf, ci, cj, ckk, mm, mvv, vo, nullptr);
        accf[ACC_F(c.H, s, 0)] = mm; accf[ACC_F(c.H, s, 1)] = mvv[0]; accf[ACC_F(c.H, s, 2)] = mvv[1]; accf[ACC_F(c.H, s, 3)] = mvv[2];   // raw values for the adjoint
        accf[ACC_F(c.H, s, 4)] = vo[0]; accf[ACC_F(c.H, s, 5)] = vo[1]; accf[ACC_F(c.H, s, 6)] = vo[2];
      }
      __syncthreads();   // b3
      // ---- D: g2p adjoint (scatter g onto grid velocities; weight / fx cotangents) ----
      float gw[9], gfx[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 9; ++d) gw[d] = 0.f;
      if (live) {
        const float4 s7 = stage[7 * Np + p], s8 = stage[8 * Np + p], s9 = stage[9 * Np + p];
        const float gnv[3] = {s7.x, s7.y, s7.z};
        const float gCq[9] = {s7.w, s8.x, s8.y, s8.z, s8.w, s9.x, s9.y, s9.z, s9.w};
#pragma unroll
        for (int t = 0; t < UD_NCELL; ++t) {
          int i, j, k;
          if (!cell_of(qi, t, i, j, k)) continue;
          const int gs = slots[t] & 0xffff;
          const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
          const float weight = wi * wj * wk;
          const float dp[3] = {(float)i - q.fx[0], (float)j - q.fx[1], (float)k - q.fx[2]};
          float gwt = 0.f, gsc[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float gCd = gCq[r * 3] * dp[0] + gCq[r * 3 + 1] * dp[1] + gCq[r * 3 + 2] * dp[2];
            const float vel = accf[ACC_F(c.H, gs, 4 + r)];
            gsc[r] = weight * gnv[r] + 4.f * c.inv_dx * weight * gCd;
            gwt += vel * (gnv[r] + 4.f * c.inv_dx * gCd);
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) gfx[s2] -= 4.f * c.inv_dx * weight * gCq[r * 3 + s2] * vel;
          }
          if (quadrun_reduce<3>(gs, gsc)) {
#pragma unroll
            for (int r = 0; r < 3; ++r) lds_add(&L.gacc[ACC_D(c.H, gs, r)], gsc[r]);
          }
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            gw[kk * 3 + 0] += (i == kk) ? gwt * wj * wk : 0.f;
            gw[kk * 3 + 1] += (j == kk) ? gwt * wi * wk : 0.f;
            gw[kk * 3 + 2] += (k == kk) ? gwt * wi * wj : 0.f;
          }
        }
      }
      __syncthreads();   // b4
      // ---- E: grid-op adjoint per occupied slot ----
      for (int e = ft, n = *L.count; e < n; e += nS) {
        const int s = L.list[e];
        int ci, cj, ckk;
        decode_cell(c, L.key[s], ci, cj, ckk);
        const float m = accf[ACC_F(c.H, s, 0)];
        float mvv[3] = {accf[ACC_F(c.H, s, 1)], accf[ACC_F(c.H, s, 2)], accf[ACC_F(c.H, s, 3)]};
        float g[3] = {(float)L.gacc[ACC_D(c.H, s, 0)], (float)L.gacc[ACC_D(c.H, s, 1)], (float)L.gacc[ACC_D(c.H, s, 2)]}, gmm, dfric, dpv[3];
        const bool ctrl = grid_op_adjoint(c, pf, ci, cj, ckk, m, mvv, g, gmm, dfric, dpv);
        acc_fric += dfric;
        if (ctrl) {
#pragma unroll
          for (int d = 0; d < 3; ++d) lds_add(&L.gpv[f * 3 + d], dpv[d]);
        }
        gaccf[ACC_F(c.H, s, 0)] = g[0]; gaccf[ACC_F(c.H, s, 1)] = g[1]; gaccf[ACC_F(c.H, s, 2)] = g[2]; gaccf[ACC_F(c.H, s, 3)] = gmm;   // same thread read the doubles
      }
      __syncthreads();   // b5
      // ---- F: p2g adjoint (gather) -> ret; FK adjoint ----
      float gaff[9], gvp[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 9; ++d) gaff[d] = 0.f;
      if (live) {
#pragma unroll
        for (int t = 0; t < UD_NCELL; ++t) {
          int i, j, k;
          if (!cell_of(qi, t, i, j, k)) continue;
          const int ss = (slots[t] >> 16) - 1;
          if (ss < 0) continue;
          const float wi = sel3(q.w, 0, i), wj = sel3(q.w, 1, j), wk = sel3(q.w, 2, k);
          const float weight = wi * wj * wk;
          const float dpos[3] = {((float)i - q.fx[0]) * c.dx, ((float)j - q.fx[1]) * c.dx, ((float)k - q.fx[2]) * c.dx};
          float gwt = c.p_mass * gaccf[ACC_F(c.H, ss, 3)];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float gc = gaccf[ACC_F(c.H, ss, r)];
            const float ad = q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2];
            gwt += gc * (c.p_mass * vq[r] + ad);
            gvp[r] += weight * c.p_mass * gc;
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
              gaff[r * 3 + s2] += weight * gc * dpos[s2];
              gfx[s2] -= c.dx * weight * gc * q.affine[r * 3 + s2];
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
      for (int d = 0; d < 9; ++d) { gw[d] = quad_sum(gw[d]); gaff[d] = quad_sum(gaff[d]); }
#pragma unroll
      for (int d = 0; d < 3; ++d) { gfx[d] = quad_sum(gfx[d]); gvp[d] = quad_sum(gvp[d]); }
      if (live && qi == 0) {
        ret[p] = make_float4(gw[0], gw[1], gw[2], gw[3]);
        ret[Np + p] = make_float4(gw[4], gw[5], gw[6], gw[7]);
        ret[2 * Np + p] = make_float4(gw[8], gfx[0], gfx[1], gfx[2]);
        ret[3 * Np + p] = make_float4(gaff[0], gaff[1], gaff[2], gaff[3]);
        ret[4 * Np + p] = make_float4(gaff[4], gaff[5], gaff[6], gaff[7]);
        ret[5 * Np + p] = make_float4(gaff[8], gvp[0], gvp[1], gvp[2]);
      }
      // FK adjoint (:185-187): position' = clip(set(position, f+1, position[f] + v[f])); reads now, writes in A
      pend = true;
      pend_val = 0.f; pend_pv = 0.f;
      if (fkl) {
        const int row = ft / 3, d = ft - row * 3;
        const float pva = (d == 0) ? pv[0] : ((d == 1) ? pv[1] : pv[2]);
        const float mine = L.gppos[ft] * clip_grad(ppos_preclip(L, f, S, row, d, pva), -2.f, 2.f);
        float val = mine;
        if (f + 1 < S) {
          if (row == f + 1) val = 0.f;
          if (row == f) {
            const float t = L.gppos[ft + 3] * clip_grad(ppos_preclip(L, f, S, f + 1, d, pva), -2.f, 2.f);
            val += t;
            pend_pv = t;
          }
        }
        pend_val = val;
      }
      __syncthreads();   // b6
    }
    if (pend && fkl) { L.gppos[ft] = pend_val; L.gpv[ft] += (double)pend_pv; }
  }
  __syncthreads();
  // ---- step boundary: set_action adjoint, action clip, norm_grad(_state) (:375-411, :419-423) ----
  float* red = L.scr;
  float tot_fric = block_sum1(acc_fric, red, nw);
  float tot_mu = block_sum1(acc_mu, red, nw);
  float tot_la = block_sum1(acc_la, red, nw);
  float ga[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gscale[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < S; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float t = (float)L.gpv[j * 3 + d]; ga[d] += t * 1.f / (float)S; gscale[d] += t * ac[d] / (float)S; }
  // rotation path (action[3:6]): the reference's d|w|/dw at w = 0 is NaN and nan_to_num zeroes it here -> 0
#pragma unroll
  for (int d = 0; d < 6; ++d) ga[d] *= clip_grad(a.action[b * 6 + d], -1.f, 1.f);
  if (a.clip) {
    float n2 = 0.f;
#pragma unroll
    for (int d = 0; d < 6; ++d) { ga[d] = nan_to_num(ga[d] + 0.f); n2 += ga[d] * ga[d]; }
    const float nrm = sqrtf(n2);
    if (!(nrm < 1.f)) {
#pragma unroll
      for (int d = 0; d < 6; ++d) ga[d] = ga[d] / nrm;
    }
    float s2 = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) { gx[d] = nan_to_num(gx[d] + 0.f); gv[d] = nan_to_num(gv[d] + 0.f); }
#pragma unroll
    for (int d = 0; d < 9; ++d) { gC[d] = nan_to_num(gC[d] + 0.f); gF[d] = nan_to_num(gF[d] + 0.f); }
    if (prole && clive) {
#pragma unroll
      for (int d = 0; d < 3; ++d) s2 += gx[d] * gx[d] + gv[d] * gv[d];
#pragma unroll
      for (int d = 0; d < 9; ++d) s2 += gC[d] * gC[d] + gF[d] * gF[d];
    }
    if (fkl) { const float t = nan_to_num(L.gppos[ft] + 0.f); L.gppos[ft] = t; s2 += t * t; }
    tot_fric = nan_to_num(tot_fric); tot_mu = nan_to_num(tot_mu); tot_la = nan_to_num(tot_la);
    if (tid == 0) {
      s2 += tot_fric * tot_fric + tot_mu * tot_mu + tot_la * tot_la;
#pragma unroll
      for (int d = 0; d < 3; ++d) { const float t = nan_to_num(gscale[d]); s2 += t * t; }
    }
    const float sn = sqrtf(block_sum1(s2, red, nw));
    if (!(sn < 1.f)) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { gx[d] = gx[d] / sn; gv[d] = gv[d] / sn; }
#pragma unroll
      for (int d = 0; d < 9; ++d) { gC[d] = gC[d] / sn; gF[d] = gF[d] / sn; }
      if (fkl) L.gppos[ft] = L.gppos[ft] / sn;
      tot_fric = tot_fric / sn; tot_mu = tot_mu / sn; tot_la = tot_la / sn;
    }
  }
  if (prole && clive) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.gx0[((size_t)b * N + cp) * 3 + d] = gx[d]; a.gv0[((size_t)b * N + cp) * 3 + d] = gv[d]; }
#pragma unroll
    for (int d = 0; d < 9; ++d) { a.gC0[((size_t)b * N + cp) * 9 + d] = gC[d]; a.gF0[((size_t)b * N + cp) * 9 + d] = gF[d]; }
  }
  if (fkl) a.gppos0[(size_t)b * S * 3 + ft] = L.gppos[ft];
  if (tid == 0) {
    a.gfric[b] = tot_fric; a.gmu[b] = tot_mu; a.glam[b] = tot_la;
#pragma unroll
    for (int d = 0; d < 6; ++d) a.gaction[b * 6 + d] = ga[d];
  }
  const int bad = __syncthreads_or(ok ? 0 : 1);
  if (tid == 0 && a.status) a.status[b] = bad ? 1 : 0;
}

}  // namespace ud

// ------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ------------------------------------------------------------------------------------------------
#include <vector>

struct ud_mpm {
  ud::MpmConst c;
  int device = 0;
  int* d_material = nullptr;
  float* d_hard = nullptr;
  size_t lds_fwd = 0, lds_bwd = 0;
  int nthreads_bwd_ws = 0;         // > 0: the wave-specialised adjoint kernel (N <= 96) with this many threads
  ud::MpmLarge* large = nullptr;   // N > 128: many-workgroup path (mpm_large.hip)
  int max_envs = 0;                // ud_mpm_conf.max_envs
};

extern "C" {

int ud_mpm_create(const ud_mpm_conf* conf, const int* material, const float* hardness, ud_mpm** out) {
  if (!conf || !material || !hardness || !out) { ud::set_error("ud_mpm_create: null argument"); return UD_ERR_INVALID; }
  const int N = conf->n_particles, S = conf->steps;
  if (N < 1 || S < 1 || conf->n_grid < 4 || conf->res[0] < 4 || conf->res[1] < 4 || conf->res[2] < 4) {
    ud::set_error("ud_mpm_create: bad sizes (N=%d steps=%d n_grid=%d)", N, S, conf->n_grid); return UD_ERR_INVALID;
  }
  if (conf->max_envs < 1) { ud::set_error("ud_mpm_create: max_envs = %d (every arena is sized at create: give the largest B any call will pass)", conf->max_envs); return UD_ERR_INVALID; }
  if (!(conf->tune_lanes == 0 || conf->tune_lanes == 1 || conf->tune_lanes == 4) || !(conf->tune_cluster_part_lanes == 0 || conf->tune_cluster_part_lanes == 64 || conf->tune_cluster_part_lanes == 128)) {
    ud::set_error("ud_mpm_create: tune_lanes = %d (0, 1, 4), tune_cluster_part_lanes = %d (0, 64, 128)", conf->tune_lanes, conf->tune_cluster_part_lanes); return UD_ERR_INVALID;
  }
  const int n_prim = conf->n_primitive > 0 ? conf->n_primitive : 1;
  if (n_prim > UD_MAX_PRIM || (conf->use_position_control && n_prim != 1) || conf->sdf_kind < 0 || conf->sdf_kind > 1 ||
      (conf->use_position_control && conf->sdf_kind != 0)) {
    ud::set_error("ud_mpm_create: n_primitive=%d sdf_kind=%d unsupported (position control: one box primitive; soft contact: 1..%d box or container primitives)",
                  n_prim, conf->sdf_kind, UD_MAX_PRIM);
    return UD_ERR_UNSUPPORTED;
  }
  // soft contact (collide_batch) runs on the many-workgroup path whatever N is: every reference env that uses it has N > 128
  // the deterministic forward lives beside the many-workgroup kernels (dense grid in HBM), whatever N is
  const bool large = N > 128 || !conf->use_position_control || conf->deterministic;
  if (!large && S * 3 > 256) { ud::set_error("ud_mpm_create: steps=%d too large for the in-LDS primitive arrays", S); return UD_ERR_UNSUPPORTED; }
  if (conf->res[0] > 1024 || conf->res[1] > 1024 || conf->res[2] > 1024) { ud::set_error("ud_mpm_create: res > 1024"); return UD_ERR_UNSUPPORTED; }
  auto* h = new ud_mpm();
  ud::MpmConst& c = h->c;
  c.N = N; c.Np = (N + 15) / 16 * 16; c.n_grid = conf->n_grid; c.steps = S;
  for (int d = 0; d < 3; ++d) c.res[d] = conf->res[d];
  const double dx = 1.0 / conf->n_grid;
  c.dt = conf->dt; c.dx = (float)dx; c.inv_dx = (float)(double)conf->n_grid;
  c.p_mass = conf->p_mass; c.p_vol = conf->p_vol;
  c.stress_c = (float)(-(double)conf->dt * (double)conf->p_vol * 4.0);   // :267
  c.dx2 = (float)(dx * dx);
  for (int d = 0; d < 3; ++d) c.dtg[d] = conf->dt * conf->gravity[d];    // :285
  c.position_control = conf->use_position_control ? 1 : 0;
  c.prim_friction = conf->prim_friction; c.prim_softness = conf->prim_softness;
  for (int i = 0; i < 4; ++i) {
    const bool each = conf->prim_softness_each[i] > 0.f;
    c.prim_friction_each[i] = each ? conf->prim_friction_each[i] : conf->prim_friction;
    c.prim_softness_each[i] = each ? conf->prim_softness_each[i] : conf->prim_softness;
  }
  c.n_prim = conf->n_primitive > 0 ? conf->n_primitive : 1;   // 0 = unset = 1
  c.sdf_kind = conf->sdf_kind;
  c.gck = conf->grid_ckpt_cells > 0 ? conf->grid_ckpt_cells : 0;
  c.sort = conf->sort_particles ? 1 : 0;
  c.det = conf->deterministic ? 1 : 0;
  if (c.det) { c.gck = 0; c.sort = 0; }   // particle order = the caller's; the backward recomputes the grid
  int Hh = 1024, lg = 10;
  while (Hh < 16 * N) { Hh *= 2; ++lg; }                                 // load factor <= ~0.3 for a compact body
  const size_t per_particle = (N <= 96) ? 64 : 48;   // floats of LDS hand-off per particle in the adjoint (stage+ret / park)
  while (Hh > 1024 && ((size_t)16 * Hh + per_particle * c.Np + (size_t)S * 19 + 72) * sizeof(float) > 160 * 1024) { Hh /= 2; --lg; }   // LDS budget of the adjoint
  c.H = Hh; c.logH = lg;
  c.nthreads = std::max(256, (4 * std::min(N, 128) + 63) / 64 * 64);
  h->lds_fwd = ((size_t)10 * Hh + (size_t)4 * 10 * h->c.Np + (size_t)S * 11 + 64 + 4) * sizeof(float);   // key, acc (4 doubles), list, stage/ret (10 float4 per particle), primitives + inverse rotations, scratch, count
  h->lds_bwd = ((size_t)16 * Hh + per_particle * h->c.Np + (size_t)S * 19 + 64 + 8) * sizeof(float);  // + gacc (3 doubles), gpv (doubles), stage+ret (16 float4 per particle, N <= 96) or park (12), adjoint primitive arrays
  if (!large && h->lds_bwd > 160 * 1024) { ud::set_error("ud_mpm_create: LDS cell table too large"); delete h; return UD_ERR_UNSUPPORTED; }
  hipError_t e = hipGetDevice(&h->device);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_material, N * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_hard, N * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->d_material, material, N * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_hard, hardness, N * sizeof(float), hipMemcpyHostToDevice);
  if (!large && e == hipSuccess) e = hipFuncSetAttribute((const void*)ud::mpm_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_fwd);
  if (!large && e == hipSuccess) e = hipFuncSetAttribute((const void*)ud::mpm_step_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bwd);
  if (!large && e == hipSuccess) e = hipFuncSetAttribute((const void*)ud::mpm_step_bwd_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bwd);
  if (!large && N <= 96) h->nthreads_bwd_ws = (N + 63) / 64 * 64 + (4 * N + 63) / 64 * 64;   // particle waves + stencil waves <= 512
  if (e != hipSuccess) {
    ud::set_error("ud_mpm_create: %s", hipGetErrorString(e));
    if (h->d_material) (void)hipFree(h->d_material);
    if (h->d_hard) (void)hipFree(h->d_hard);
    delete h;
    return UD_ERR_HIP;
  }
  bool has_liquid = false;
  for (int i = 0; i < N; ++i) has_liquid = has_liquid || material[i] == 0;
  h->max_envs = conf->max_envs;
  if (large) {
    const ud::LgTune tune{conf->max_envs, conf->tune_lanes, conf->tune_cluster, conf->tune_cluster_part_lanes, conf->tune_cluster_envs,
                          conf->tune_env_groups, conf->tune_bwd_two_launch, conf->tune_collide_records};
    h->large = ud::mpm_large_create(h->c, h->d_material, h->d_hard, has_liquid, tune);
    if (!h->large) { (void)hipFree(h->d_material); (void)hipFree(h->d_hard); delete h; return UD_ERR_HIP; }
  }
  *out = h;
  return UD_OK;
}

int ud_mpm_reset(ud_mpm* h, void* stream) {
  if (!h) { ud::set_error("ud_mpm_reset: null handle"); return UD_ERR_INVALID; }
  return h->large ? ud::mpm_large_reset(h->large, (hipStream_t)stream) : UD_OK;   // the one-workgroup path owns no scratch
}

void ud_mpm_destroy(ud_mpm* h) {
  if (!h) return;
  ud::mpm_large_destroy(h->large);
  (void)hipFree(h->d_material);
  (void)hipFree(h->d_hard);
  delete h;
}

size_t ud_mpm_ckpt_bytes(const ud_mpm* h, int B) {
  if (!h || B < 0) return 0;
  if (h->large) return ud::mpm_large_ckpt_bytes(h->large, B);
  return (size_t)B * ((size_t)h->c.steps * 24 * h->c.Np + (size_t)h->c.steps * 10) * sizeof(float);
}

int ud_mpm_ckpt_cells(const ud_mpm* h, int B, const void* ckpt, int* cells, void* stream) {
  if (!h || !ckpt || !cells || B < 1) { ud::set_error("ud_mpm_ckpt_cells: bad argument"); return UD_ERR_INVALID; }
  if (h->large) return ud::mpm_large_ckpt_cells(h->large, B, (const float*)ckpt, cells, (hipStream_t)stream);
  return hipMemsetAsync(cells, 0, (size_t)B * sizeof(int), (hipStream_t)stream) == hipSuccess ? UD_OK : UD_ERR_HIP;
}

int ud_mpm_launch_plan(const ud_mpm* h, int B) {
  if (!h || B < 1) return -1;
  return h->large ? ud::mpm_large_plan(h->large, B) : 0;
}

int ud_mpm_step_fwd(ud_mpm* h, int B, const float* x, const float* v, const float* C, const float* F, const float* J,
                    const float* prim_position, const float* prim_rotation, const float* prim_size,
                    const float* friction, const float* mu, const float* lamda, const float* action, float* x_out,
                    float* v_out, float* C_out, float* F_out, float* J_out, float* prim_position_out,
                    float* prim_rotation_out, float* prim_v_out, float* prim_w_out, void* ckpt, int* status,
                    void* stream) {
  if (!h || !x || !v || !C || !F || !J || !prim_position || !prim_rotation || !prim_size || !friction || !mu || !lamda ||
      !action || !x_out || !v_out || !C_out || !F_out || !J_out || !prim_position_out || !prim_rotation_out ||
      !prim_v_out || !prim_w_out) {
    ud::set_error("ud_mpm_step_fwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1 || B > h->max_envs) { ud::set_error("ud_mpm_step_fwd: B=%d (max_envs=%d)", B, h->max_envs); return UD_ERR_INVALID; }
  if (h->large)
    return ud::mpm_large_step_fwd(h->large, B, x, v, C, F, J, prim_position, prim_rotation, prim_size, friction, mu, lamda, action,
                                  x_out, v_out, C_out, F_out, J_out, prim_position_out, prim_rotation_out, prim_v_out, prim_w_out,
                                  (float*)ckpt, status, (hipStream_t)stream);
  ud::MpmFwdArgs a{};
  a.c = h->c; a.material = h->d_material; a.hard = h->d_hard; a.B = B;
  a.x = x; a.v = v; a.C = C; a.F = F; a.J = J; a.ppos = prim_position; a.prot = prim_rotation; a.psize = prim_size;
  a.friction = friction; a.mu = mu; a.lamda = lamda; a.action = action;
  a.xo = x_out; a.vo = v_out; a.Co = C_out; a.Fo = F_out; a.Jo = J_out; a.ppos_o = prim_position_out;
  a.prot_o = prim_rotation_out; a.pv_o = prim_v_out; a.pw_o = prim_w_out; a.ckpt = (float*)ckpt; a.status = status;
  hipLaunchKernelGGL(ud::mpm_step_fwd_kernel, dim3(B), dim3(h->c.nthreads), h->lds_fwd, (hipStream_t)stream, a);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_mpm_step_bwd(ud_mpm* h, int B, const void* ckpt, const float* prim_size, const float* friction,
                    const float* mu, const float* lamda, const float* action, const float* g_x, const float* g_v,
                    const float* g_C, const float* g_F, const float* g_prim_position, const float* g_prim_rotation, int clip,
                    float* g_x0, float* g_v0, float* g_C0, float* g_F0, float* g_prim_position0, float* g_prim_rotation0,
                    float* g_friction, float* g_mu, float* g_lamda, float* g_action, int* status, void* stream) {
  if (!h || !ckpt || !prim_size || !friction || !mu || !lamda || !action || !g_x || !g_v || !g_C || !g_F ||
      !g_prim_position || !g_x0 || !g_v0 || !g_C0 || !g_F0 || !g_prim_position0 || !g_friction || !g_mu || !g_lamda ||
      !g_action) {
    ud::set_error("ud_mpm_step_bwd: null argument"); return UD_ERR_INVALID;
  }
  if (B < 1 || B > h->max_envs) { ud::set_error("ud_mpm_step_bwd: B=%d (max_envs=%d)", B, h->max_envs); return UD_ERR_INVALID; }
  if (h->large)
    return ud::mpm_large_step_bwd(h->large, B, (const float*)ckpt, prim_size, friction, mu, lamda, action, g_x, g_v, g_C, g_F,
                                  g_prim_position, g_prim_rotation, clip, g_x0, g_v0, g_C0, g_F0, g_prim_position0, g_prim_rotation0,
                                  g_friction, g_mu, g_lamda, g_action, status, (hipStream_t)stream);
  // one-workgroup path = position control: no cotangent reaches the rotation array
  if (g_prim_rotation0) UD_HIP_CHECK(hipMemsetAsync(g_prim_rotation0, 0, (size_t)B * h->c.n_prim * h->c.steps * 4 * sizeof(float), (hipStream_t)stream));
  ud::MpmBwdArgs a{};
  a.c = h->c; a.material = h->d_material; a.hard = h->d_hard; a.B = B; a.ckpt = (const float*)ckpt;
  a.psize = prim_size; a.friction = friction; a.mu = mu; a.lamda = lamda; a.action = action;
  a.gx = g_x; a.gv = g_v; a.gC = g_C; a.gF = g_F; a.gppos = g_prim_position; a.clip = clip;
  a.gx0 = g_x0; a.gv0 = g_v0; a.gC0 = g_C0; a.gF0 = g_F0; a.gppos0 = g_prim_position0; a.gfric = g_friction;
  a.gmu = g_mu; a.glam = g_lamda; a.gaction = g_action; a.status = status;
  if (h->nthreads_bwd_ws > 0)
    hipLaunchKernelGGL(ud::mpm_step_bwd_ws_kernel, dim3(B), dim3(h->nthreads_bwd_ws), h->lds_bwd, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ud::mpm_step_bwd_kernel, dim3(B), dim3(h->c.nthreads), h->lds_bwd, (hipStream_t)stream, a);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

}  // extern "C"
