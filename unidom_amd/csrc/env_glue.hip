// Cloth-env arithmetic either side of the rollout, fused: the reference jit-compiles step_diff, so XLA turns the
// chamfer reward, the contact distance and the 6-vector -> 40 x 8 macro-action expansion into a handful of fused
// kernels; done op by op on the device they are ~180 tiny launches (and a dense [B,P,Q] autograd graph) per
// step_diff.  Here each piece is one forward and one backward kernel, one workgroup per environment.
// The MPM envs' counterpart (focus shift before the step; un-shift, nan_to_num, reward and observation after it) is the
// second half of this file.
//   chamfer        core/utils/util.py:138-153     d(p,q) = sqrt(mean_xyz((x_p - y_q)^2)); mean_p min_q + mean_q min_p
//   pnp / contact  core/envs/basic/cloth_env.py:134-173 (get_pnp_actions), :206-209 (contact_distance)
// The backward kernels return what jax.grad / torch.autograd return for the same expressions: the minimum passes its
// cotangent to one argmin (the first on ties), sqrt's derivative at 0 is left as the 0/0 it is in the reference.
#include "common.h"

namespace ud {

constexpr int GLUE_T = 256;       // threads per workgroup
constexpr int GLUE_MAXPTS = 4096; // points per cloud staged in LDS (12 floats/point budget: 2 clouds = 96 KB)

__device__ __forceinline__ float mean_sq3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy + dz * dz) / 3.0f;   // ((a - b) ** 2).mean(-1)
}

// (value, index) minimum as jnp.min / torch.min take it: a NaN candidate wins over any number (the reduction propagates
// NaN), equal values keep the smaller index.  The index starts at 0, so it is in range whatever the values are: with
// every distance +inf or NaN the backward kernels still address element 0 and the non-finite value flows through sqrt
// exactly as it does in the reference's expression (no sentinel index can reach an address computation).
__device__ __forceinline__ bool min_takes(float ov, int oi, float best, int bi) {
  const bool on = ov != ov, bn = best != best;
  if (on || bn) return on && (!bn || oi < bi);
  return ov < best || (ov == best && oi < bi);
}

// block-wide sum of one float per thread (GLUE_T threads); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wv] = v;
  __syncthreads();
  float t = 0.f;
  for (int q = 0; q < GLUE_T / 64; ++q) t += scratch[q];
  return t;
}

// ---- chamfer -------------------------------------------------------------------------------------
// grid (B, tiles, 2): z = 0 scans, for 64 particles per workgroup, all goal points; z = 1 the other way round.
// Four lanes share a row (each takes every fourth column), the quad then reduces (value, first index) with DPP.
// The two means are accumulated into out[b] (zeroed by the caller) with one atomic per workgroup.
constexpr int CH_ROWS = GLUE_T / 4;

__global__ void __launch_bounds__(GLUE_T) chamfer_fwd_kernel(int P, int Q, const float* __restrict__ x, const float* __restrict__ y,
                                                             float* __restrict__ out, int* __restrict__ ixy, int* __restrict__ iyx) {
  extern __shared__ float lds[];   // cols[3][nc] | scratch[8]
  const int b = blockIdx.x, tid = threadIdx.x, dir = blockIdx.z;
  const float* xb = x + (size_t)b * P * 3;
  const float* rows = dir == 0 ? xb : y;
  const float* cols = dir == 0 ? y : xb;
  const int nr = dir == 0 ? P : Q, nc = dir == 0 ? Q : P;
  const int r0 = blockIdx.y * CH_ROWS;
  if (r0 >= nr) return;   // uniform: the grid is sized for the larger cloud
  float* cs = lds;
  float* scratch = lds + 3 * nc;
  for (int i = tid; i < nc; i += GLUE_T) { cs[i] = cols[i * 3]; cs[nc + i] = cols[i * 3 + 1]; cs[2 * nc + i] = cols[i * 3 + 2]; }
  __syncthreads();
  const int r = r0 + (tid >> 2), c0 = tid & 3;
  const bool live = r < nr;
  const int rr = live ? r : nr - 1;
  const float ax = rows[rr * 3], ay = rows[rr * 3 + 1], az = rows[rr * 3 + 2];
  float best = INFINITY; int bi = 0;
  for (int c = c0; c < nc; c += 4) {
    const float m = dir == 0 ? mean_sq3(ax, ay, az, cs[c], cs[nc + c], cs[2 * nc + c])
                             : mean_sq3(cs[c], cs[nc + c], cs[2 * nc + c], ax, ay, az);   // (x - y) in both directions
    if (m < best || (m != m && best == best)) { best = m; bi = c; }   // first minimum; the first NaN sticks
  }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {   // the four lanes of a row: smallest value, then smallest index
    const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
    if (min_takes(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  float s = 0.f;
  if (live && c0 == 0) {
    (dir == 0 ? ixy + (size_t)b * P : iyx + (size_t)b * Q)[r] = bi;
    s = sqrtf(best);   // min of sqrt == sqrt of min (sqrtf is monotone)
  }
  const float t = block_sum(s, scratch);
  if (tid == 0) atomicAdd(&out[b], t / (float)nr);
}

__global__ void __launch_bounds__(GLUE_T) chamfer_bwd_kernel(int P, int Q, const float* __restrict__ x, const float* __restrict__ y,
                                                             const int* __restrict__ ixy, const int* __restrict__ iyx,
                                                             const float* __restrict__ g_out, float* __restrict__ g_x) {
  extern __shared__ float lds[];   // gx[P][3]
  float* gx = lds;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * P * 3;
  const float g = g_out[b];
  const float gp = g / (float)P, gq = g / (float)Q;   // mean's backward
  for (int p = tid; p < P; p += GLUE_T) {
    const int q = ixy[(size_t)b * P + p];
    const float dx = xb[p * 3] - y[q * 3], dy = xb[p * 3 + 1] - y[q * 3 + 1], dz = xb[p * 3 + 2] - y[q * 3 + 2];
    const float d = sqrtf((dx * dx + dy * dy + dz * dz) / 3.0f);
    const float c = gp / (3.0f * d);                  // d sqrt(mean((a-b)^2)) / da = (a - b) / (3 d)
    gx[p * 3] = c * dx; gx[p * 3 + 1] = c * dy; gx[p * 3 + 2] = c * dz;
  }
  __syncthreads();
  for (int q = tid; q < Q; q += GLUE_T) {
    const int p = iyx[(size_t)b * Q + q];
    const float dx = xb[p * 3] - y[q * 3], dy = xb[p * 3 + 1] - y[q * 3 + 1], dz = xb[p * 3 + 2] - y[q * 3 + 2];
    const float d = sqrtf((dx * dx + dy * dy + dz * dz) / 3.0f);
    const float c = gq / (3.0f * d);
    atomicAdd(&gx[p * 3], c * dx); atomicAdd(&gx[p * 3 + 1], c * dy); atomicAdd(&gx[p * 3 + 2], c * dz);
  }
  __syncthreads();
  float* o = g_x + (size_t)b * P * 3;
  for (int i = tid; i < P * 3; i += GLUE_T) o[i] = gx[i];
}

// ---- pick-and-place expansion + contact distance -------------------------------------------------
// Divisions by the literals 3 and 20 are multiplications by the f32 reciprocal, as XLA's AlgebraicSimplifier rewrites them under
// jit (A / Const => A * (1 / Const)): the reference's recorded primitive trajectories tell the two forms apart for / 3 (and
// for robot_step's / 50); 0.06 / 10 is constant / constant and is folded as a true division.
constexpr float R3 = 1.0f / 3.0f, R20 = 1.0f / 20.0f;
// macro rows (cloth_env.py:148-171): 3 x down ((pick_xz - primitive0.xyz)/3, 1) | 10 x up (0, 0.06/10, 0, 0) |
// 20 x move ((place_xz - pick_xz)/20, 0) | 7 x release (0,0,0,1); columns 4-7 (second gripper) are zero.
__global__ void __launch_bounds__(GLUE_T) pnp_fwd_kernel(int B, int P, const float* __restrict__ actions, const float* __restrict__ prim0,
                                                         const float* __restrict__ x, float* __restrict__ macro,
                                                         float* __restrict__ contact, int* __restrict__ contact_idx) {
  __shared__ float sv[GLUE_T / 64];
  __shared__ int si[GLUE_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* a = actions + b * 6;
  const float* xb = x + (size_t)b * P * 3;
  const float px = a[0], py = a[1], pz = a[2];
  float best = INFINITY; int bi = 0;
  for (int p = tid; p < P; p += GLUE_T) {   // contact_distance = min_p |pick - x_p|  (:206-209)
    const float dx = px - xb[p * 3], dy = py - xb[p * 3 + 1], dz = pz - xb[p * 3 + 2];
    const float s = dx * dx + dy * dy + dz * dz;
    if (s < best || (s != s && best == best)) { best = s; bi = p; }
  }
  for (int off = 32; off >= 1; off >>= 1) {   // (value, first index) minimum over the wave
    const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
    if (min_takes(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < GLUE_T / 64; ++q)
      if (min_takes(sv[q], si[q], best, bi)) { best = sv[q]; bi = si[q]; }
    contact[b] = sqrtf(best);
    contact_idx[b] = bi;
  }
  if (tid < 40) {
    const int t = tid;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    if (t < 3) {
      r0 = (px - prim0[b * 4]) * R3; r1 = (0.f - prim0[b * 4 + 1]) * R3; r2 = (pz - prim0[b * 4 + 2]) * R3; r3 = 1.f;
    } else if (t < 13) {
      r0 = 0.f / 10.0f; r1 = 0.06f / 10.0f; r2 = 0.f / 10.0f;
    } else if (t < 33) {
      r0 = (a[3] - px) * R20; r1 = 0.f * R20; r2 = (a[5] - pz) * R20;
    } else {
      r3 = 1.f;
    }
    float* m = macro + ((size_t)t * B + b) * 8;
    m[0] = r0; m[1] = r1; m[2] = r2; m[3] = r3; m[4] = 0.f; m[5] = 0.f; m[6] = 0.f; m[7] = 0.f;
  }
}

__global__ void __launch_bounds__(GLUE_T) pnp_bwd_kernel(int B, int P, const float* __restrict__ actions, const float* __restrict__ x,
                                                         const float* __restrict__ contact, const int* __restrict__ contact_idx,
                                                         const float* __restrict__ g_macro, const float* __restrict__ g_contact,
                                                         float* __restrict__ g_actions, float* __restrict__ g_prim0, float* __restrict__ g_x) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float* gxb = g_x + (size_t)b * P * 3;
  for (int i = tid; i < P * 3; i += GLUE_T) gxb[i] = 0.f;
  __syncthreads();
  if (tid == 0) {
    float down[3] = {0.f, 0.f, 0.f}, move[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < 3; ++t)
      for (int d = 0; d < 3; ++d) down[d] += g_macro[((size_t)t * B + b) * 8 + d];
    for (int t = 13; t < 33; ++t)
      for (int d = 0; d < 3; ++d) move[d] += g_macro[((size_t)t * B + b) * 8 + d];
    for (int d = 0; d < 3; ++d) { down[d] *= R3; move[d] *= R20; }
    // contact = |pick - x_p*|: d/dpick = (pick - x_p*) / contact, d/dx_p* = -(...)
    const int p = contact_idx[b];
    const float gc = g_contact ? g_contact[b] : 0.f;
    const float* a = actions + b * 6;
    float cg[3];
    for (int d = 0; d < 3; ++d) cg[d] = gc * ((a[d] - x[((size_t)b * P + p) * 3 + d]) / contact[b]);
    g_actions[b * 6 + 0] = down[0] - move[0] + cg[0];
    g_actions[b * 6 + 1] = cg[1];                       // pick.y is zeroed before the expansion (:145)
    g_actions[b * 6 + 2] = down[2] - move[2] + cg[2];
    g_actions[b * 6 + 3] = move[0];
    g_actions[b * 6 + 4] = 0.f;                          // place.y zeroed (:146)
    g_actions[b * 6 + 5] = move[2];
    g_prim0[b * 4 + 0] = -down[0]; g_prim0[b * 4 + 1] = -down[1]; g_prim0[b * 4 + 2] = -down[2]; g_prim0[b * 4 + 3] = 0.f;
    for (int d = 0; d < 3; ++d) gxb[p * 3 + d] = -cg[d];
  }
}


// ---- MPM envs: focus shift, and the tail of step_diff ----------------------------------------------
//   focus   core/envs/basic/mpm_env.py:99-114  shift = (res / 2 / n_grid - mean_n x) * (1,0,1); x and every primitive
//           trajectory move by it before the step, and back afterwards (:116-125)
//   finish  :116-125 (x, primitive positions - shift), :150-154 (nan_to_num on x v C F J), :90-94 reward
//           e ** (-10 * mean_n sqrt(mean_xyz((x - goal)^2))), :57-76 obs = x | v | primitive 0 trajectory
// One workgroup per env.  Cotangent pointers may be null (that output was not used).
constexpr int MG_MAXPRIM = 4;   // = UD_MAX_PRIM of the simulator
struct PrimIn { const float* p[MG_MAXPRIM]; };
struct PrimOut { float* p[MG_MAXPRIM]; };

__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float (*red)[GLUE_T / 64]) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = c; }
  __syncthreads();
  a = b = c = 0.f;
  for (int q = 0; q < GLUE_T / 64; ++q) { a += red[0][q]; b += red[1][q]; c += red[2][q]; }
}

__device__ __forceinline__ bool finitef(float x) { return fabsf(x) < INFINITY; }   // false for NaN and +-inf

__global__ void __launch_bounds__(GLUE_T) mpm_focus_fwd_kernel(int N, int S, int n_prim, float cx, float cz, const float* __restrict__ x,
                                                               PrimIn pin, float* __restrict__ xo, PrimOut pout, float* __restrict__ shift) {
  __shared__ float red[3][GLUE_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * N * 3;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int n = tid; n < N; n += GLUE_T) { sx += xb[n * 3]; sz += xb[n * 3 + 2]; }
  block_sum3(sx, sy, sz, red);
  const float sh[3] = {cx - sx / (float)N, 0.f, cz - sz / (float)N};
  float* xob = xo + (size_t)b * N * 3;
  for (int i = tid; i < N * 3; i += GLUE_T) xob[i] = xb[i] + sh[i % 3];
  for (int p = 0; p < n_prim; ++p) {
    const float* src = pin.p[p] + (size_t)b * S * 3;
    float* dst = pout.p[p] + (size_t)b * S * 3;
    for (int i = tid; i < S * 3; i += GLUE_T) dst[i] = src[i] + sh[i % 3];
  }
  if (tid < 3) shift[b * 3 + tid] = sh[tid];
}

// x_out = x + shift(x), shift = c - mean(x) on the x and z axes: g_x = g_xo - (1/N) (g_shift + sum_n g_xo + sum g_pos_out)
__global__ void __launch_bounds__(GLUE_T) mpm_focus_bwd_kernel(int N, int S, int n_prim, const float* __restrict__ g_xo, PrimIn g_pout,
                                                               const float* __restrict__ g_shift, float* __restrict__ g_x) {
  __shared__ float red[3][GLUE_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* gb = g_xo ? g_xo + (size_t)b * N * 3 : nullptr;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (gb)
    for (int n = tid; n < N; n += GLUE_T) { sx += gb[n * 3]; sz += gb[n * 3 + 2]; }
  for (int p = 0; p < n_prim; ++p) {
    if (!g_pout.p[p]) continue;
    const float* g = g_pout.p[p] + (size_t)b * S * 3;
    for (int s = tid; s < S; s += GLUE_T) { sx += g[s * 3]; sz += g[s * 3 + 2]; }
  }
  block_sum3(sx, sy, sz, red);
  if (g_shift) { sx += g_shift[b * 3]; sz += g_shift[b * 3 + 2]; }
  const float t[3] = {sx / (float)N, 0.f, sz / (float)N};
  float* o = g_x + (size_t)b * N * 3;
  for (int i = tid; i < N * 3; i += GLUE_T) o[i] = (gb ? gb[i] : 0.f) - t[i % 3];
}

__global__ void __launch_bounds__(GLUE_T) mpm_finish_fwd_kernel(int N, int S, int n_prim, int Q, const float* __restrict__ x, const float* __restrict__ v,
                                                                const float* __restrict__ Cm, const float* __restrict__ F,
                                                                const float* __restrict__ J, const float* __restrict__ shift, PrimIn pin,
                                                                const float* __restrict__ goal, float* __restrict__ xo, float* __restrict__ vo,
                                                                float* __restrict__ Co, float* __restrict__ Fo, float* __restrict__ Jo,
                                                                PrimOut pout, float* __restrict__ reward, float* __restrict__ obs) {
  __shared__ float scratch[GLUE_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t o3 = (size_t)b * N * 3, o9 = (size_t)b * N * 9;
  float sh[3] = {0.f, 0.f, 0.f};
  if (shift) { sh[0] = shift[b * 3]; sh[1] = shift[b * 3 + 1]; sh[2] = shift[b * 3 + 2]; }
  float* ob = obs + (size_t)b * (6 * N + 3 * S);
  float acc = 0.f;
  for (int n = tid; n < N; n += GLUE_T) {
    float m = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float xv = nan_to_num(x[o3 + n * 3 + d] - sh[d]);
      xo[o3 + n * 3 + d] = xv; ob[n * 3 + d] = xv;
      const float df = xv - goal[(Q == 1 ? 0 : n) * 3 + d];   // calc_l2 broadcasts a one-row goal
      m += df * df;
      const float vv = nan_to_num(v[o3 + n * 3 + d]);
      vo[o3 + n * 3 + d] = vv; ob[3 * N + n * 3 + d] = vv;
    }
    acc += sqrtf(m / 3.0f);
    Jo[(size_t)b * N + n] = nan_to_num(J[(size_t)b * N + n]);
  }
  for (int i = tid; i < N * 9; i += GLUE_T) { Co[o9 + i] = nan_to_num(Cm[o9 + i]); Fo[o9 + i] = nan_to_num(F[o9 + i]); }
  for (int p = 0; p < n_prim; ++p) {
    const float* src = pin.p[p] + (size_t)b * S * 3;
    float* dst = pout.p[p] + (size_t)b * S * 3;
    for (int i = tid; i < S * 3; i += GLUE_T) {
      const float q = src[i] - sh[i % 3];
      dst[i] = q;
      if (p == 0) ob[6 * N + i] = q;
    }
  }
  const float tot = block_sum(acc, scratch);
  if (tid == 0) reward[b] = powf(2.718281828459045f, -(tot / (float)N) * 10.0f);
}

// nan_to_num passes the cotangent where its argument was finite and nothing elsewhere (jnp.where selection: a NaN
// cotangent does not leak through a replaced entry)
__global__ void __launch_bounds__(GLUE_T) mpm_finish_bwd_kernel(int N, int S, int n_prim, int Q, const float* __restrict__ x, const float* __restrict__ v,
                                                                const float* __restrict__ Cm, const float* __restrict__ F,
                                                                const float* __restrict__ shift, const float* __restrict__ goal,
                                                                const float* __restrict__ reward, const float* __restrict__ g_xo,
                                                                const float* __restrict__ g_vo, const float* __restrict__ g_Co,
                                                                const float* __restrict__ g_Fo, PrimIn g_pout, const float* __restrict__ g_reward,
                                                                const float* __restrict__ g_obs, float* __restrict__ g_x, float* __restrict__ g_v,
                                                                float* __restrict__ g_C, float* __restrict__ g_F, PrimOut g_pin,
                                                                float* __restrict__ g_shift) {
  // grid (blocks of GLUE_T particles, B): one workgroup per env took 0.38 ms at 7631 particles; g_shift is summed with one atomic
  // per block and axis (the host zeroes it), block 0 takes the primitive arrays along
  __shared__ float red[3][GLUE_T / 64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const size_t o3 = (size_t)b * N * 3, o9 = (size_t)b * N * 9;
  float sh[3] = {0.f, 0.f, 0.f};
  if (shift) { sh[0] = shift[b * 3]; sh[1] = shift[b * 3 + 1]; sh[2] = shift[b * 3 + 2]; }
  const float* gob = g_obs ? g_obs + (size_t)b * (6 * N + 3 * S) : nullptr;
  // reward = e^t, t = -10 * l2, l2 = (1/N) sum_n sqrt(m_n), m_n = |x - goal|^2 / 3
  const float gr = g_reward ? g_reward[b] * reward[b] * (-10.0f) / (float)N : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int n = blockIdx.x * GLUE_T + tid;
  if (n < N) {
    float xs[3], df[3], m = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      xs[d] = x[o3 + n * 3 + d] - sh[d];
      df[d] = nan_to_num(xs[d]) - goal[(Q == 1 ? 0 : n) * 3 + d];
      m += df[d] * df[d];
    }
    const float dn = sqrtf(m / 3.0f);
    float g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float t = (g_xo ? g_xo[o3 + n * 3 + d] : 0.f) + (gob ? gob[n * 3 + d] : 0.f);
      if (g_reward) t += gr * (df[d] / (3.0f * dn));     // 0/0 at a particle sitting on its goal, as in the reference
      g[d] = finitef(xs[d]) ? t : 0.f;
      g_x[o3 + n * 3 + d] = g[d];
      const float tv = (g_vo ? g_vo[o3 + n * 3 + d] : 0.f) + (gob ? gob[3 * N + n * 3 + d] : 0.f);
      g_v[o3 + n * 3 + d] = finitef(v[o3 + n * 3 + d]) ? tv : 0.f;
    }
    s0 += g[0]; s1 += g[1]; s2 += g[2];
  }
  for (int i = blockIdx.x * GLUE_T * 9 + tid; i < min(N, (int)(blockIdx.x + 1) * GLUE_T) * 9; i += GLUE_T) {
    g_C[o9 + i] = (g_Co && finitef(Cm[o9 + i])) ? g_Co[o9 + i] : 0.f;
    g_F[o9 + i] = (g_Fo && finitef(F[o9 + i])) ? g_Fo[o9 + i] : 0.f;
  }
  for (int p = 0; p < (blockIdx.x == 0 ? n_prim : 0); ++p) {
    const float* src = g_pout.p[p] ? g_pout.p[p] + (size_t)b * S * 3 : nullptr;
    float* dst = g_pin.p[p] + (size_t)b * S * 3;
    for (int i = tid; i < S * 3; i += GLUE_T) {
      const float q = (src ? src[i] : 0.f) + ((p == 0 && gob) ? gob[6 * N + i] : 0.f);
      dst[i] = q;
      const int d = i % 3;
      if (d == 0) s0 += q; else if (d == 1) s1 += q; else s2 += q;
    }
  }
  if (g_shift) {    // every shifted quantity is (value - shift)
    block_sum3(s0, s1, s2, red);
    if (tid == 0) { atomicAdd(&g_shift[b * 3], -s0); atomicAdd(&g_shift[b * 3 + 1], -s1); atomicAdd(&g_shift[b * 3 + 2], -s2); }
  }
}

}  // namespace ud

using namespace ud;

extern "C" {

int ud_chamfer_fwd(int B, int P, int Q, const float* x, const float* y, float* out, int* idx_xy, int* idx_yx, void* stream) {
  if (B <= 0 || P <= 0 || Q <= 0 || !x || !y || !out || !idx_xy || !idx_yx) { set_error("ud_chamfer_fwd: bad argument"); return UD_ERR_INVALID; }
  if (P > GLUE_MAXPTS || Q > GLUE_MAXPTS) { set_error("ud_chamfer_fwd: P=%d / Q=%d above the %d points staged in LDS", P, Q, GLUE_MAXPTS); return UD_ERR_UNSUPPORTED; }
  const int big = P > Q ? P : Q;
  const size_t shmem = (size_t)(3 * big + 8) * sizeof(float);
  UD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)B * sizeof(float), (hipStream_t)stream));
  hipLaunchKernelGGL(chamfer_fwd_kernel, dim3(B, (big + CH_ROWS - 1) / CH_ROWS, 2), dim3(GLUE_T), shmem, (hipStream_t)stream, P, Q, x, y,
                     out, idx_xy, idx_yx);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_chamfer_bwd(int B, int P, int Q, const float* x, const float* y, const int* idx_xy, const int* idx_yx, const float* g_out,
                   float* g_x, void* stream) {
  if (B <= 0 || P <= 0 || Q <= 0 || !x || !y || !idx_xy || !idx_yx || !g_out || !g_x) { set_error("ud_chamfer_bwd: bad argument"); return UD_ERR_INVALID; }
  if (P > GLUE_MAXPTS) { set_error("ud_chamfer_bwd: P=%d above %d", P, GLUE_MAXPTS); return UD_ERR_UNSUPPORTED; }
  const size_t shmem = (size_t)3 * P * sizeof(float);
  hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(B), dim3(GLUE_T), shmem, (hipStream_t)stream, P, Q, x, y, idx_xy, idx_yx, g_out, g_x);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_cloth_pnp_fwd(int B, int P, const float* actions, const float* primitive0, const float* x, float* macro_actions,
                     float* contact_distance, int* contact_idx, void* stream) {
  if (B <= 0 || P <= 0 || !actions || !primitive0 || !x || !macro_actions || !contact_distance || !contact_idx) {
    set_error("ud_cloth_pnp_fwd: bad argument"); return UD_ERR_INVALID;
  }
  hipLaunchKernelGGL(pnp_fwd_kernel, dim3(B), dim3(GLUE_T), 0, (hipStream_t)stream, B, P, actions, primitive0, x, macro_actions,
                     contact_distance, contact_idx);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_cloth_pnp_bwd(int B, int P, const float* actions, const float* x, const float* contact_distance, const int* contact_idx,
                     const float* g_macro_actions, const float* g_contact_distance, float* g_actions, float* g_primitive0, float* g_x,
                     void* stream) {
  if (B <= 0 || P <= 0 || !actions || !x || !contact_distance || !contact_idx || !g_macro_actions || !g_actions || !g_primitive0 || !g_x) {
    set_error("ud_cloth_pnp_bwd: bad argument"); return UD_ERR_INVALID;
  }
  hipLaunchKernelGGL(pnp_bwd_kernel, dim3(B), dim3(GLUE_T), 0, (hipStream_t)stream, B, P, actions, x, contact_distance, contact_idx,
                     g_macro_actions, g_contact_distance, g_actions, g_primitive0, g_x);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}


static bool prim_args_ok(int n_prim, const float* const* a, bool allow_null_entries) {
  if (n_prim < 0 || n_prim > MG_MAXPRIM) return false;
  if (n_prim > 0 && !a) return false;
  if (!allow_null_entries)
    for (int p = 0; p < n_prim; ++p)
      if (!a[p]) return false;
  return true;
}

int ud_mpm_focus_fwd(int B, int N, int n_prim, int S, float cx, float cz, const float* x, const float* const* prim_pos, float* x_out,
                     float* const* prim_pos_out, float* shift, void* stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !x || !x_out || !shift || !prim_args_ok(n_prim, prim_pos, false) ||
      !prim_args_ok(n_prim, (const float* const*)prim_pos_out, false)) {
    set_error("ud_mpm_focus_fwd: bad argument"); return UD_ERR_INVALID;
  }
  PrimIn pi{}; PrimOut po{};
  for (int p = 0; p < n_prim; ++p) { pi.p[p] = prim_pos[p]; po.p[p] = prim_pos_out[p]; }
  hipLaunchKernelGGL(mpm_focus_fwd_kernel, dim3(B), dim3(GLUE_T), 0, (hipStream_t)stream, N, S, n_prim, cx, cz, x, pi, x_out, po, shift);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_mpm_focus_bwd(int B, int N, int n_prim, int S, const float* g_x_out, const float* const* g_prim_pos_out, const float* g_shift,
                     float* g_x, void* stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !g_x || !prim_args_ok(n_prim, g_prim_pos_out, true)) {
    set_error("ud_mpm_focus_bwd: bad argument"); return UD_ERR_INVALID;
  }
  PrimIn gi{};
  for (int p = 0; p < n_prim; ++p) gi.p[p] = g_prim_pos_out[p];
  hipLaunchKernelGGL(mpm_focus_bwd_kernel, dim3(B), dim3(GLUE_T), 0, (hipStream_t)stream, N, S, n_prim, g_x_out, gi, g_shift, g_x);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_mpm_finish_fwd(int B, int N, int n_prim, int S, int Q, const float* x, const float* v, const float* C, const float* F, const float* J,
                      const float* shift, const float* const* prim_pos, const float* goal, float* x_out, float* v_out, float* C_out,
                      float* F_out, float* J_out, float* const* prim_pos_out, float* reward, float* obs, void* stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !x || !v || !C || !F || !J || !goal || !x_out || !v_out || !C_out || !F_out || !J_out || !reward ||
      !obs || n_prim < 1 || (Q != N && Q != 1) || !prim_args_ok(n_prim, prim_pos, false) || !prim_args_ok(n_prim, (const float* const*)prim_pos_out, false)) {
    set_error("ud_mpm_finish_fwd: bad argument"); return UD_ERR_INVALID;
  }
  PrimIn pi{}; PrimOut po{};
  for (int p = 0; p < n_prim; ++p) { pi.p[p] = prim_pos[p]; po.p[p] = prim_pos_out[p]; }
  hipLaunchKernelGGL(mpm_finish_fwd_kernel, dim3(B), dim3(GLUE_T), 0, (hipStream_t)stream, N, S, n_prim, Q, x, v, C, F, J, shift, pi, goal,
                     x_out, v_out, C_out, F_out, J_out, po, reward, obs);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

int ud_mpm_finish_bwd(int B, int N, int n_prim, int S, int Q, const float* x, const float* v, const float* C, const float* F, const float* shift,
                      const float* goal, const float* reward, const float* g_x_out, const float* g_v_out, const float* g_C_out,
                      const float* g_F_out, const float* const* g_prim_pos_out, const float* g_reward, const float* g_obs, float* g_x,
                      float* g_v, float* g_C, float* g_F, float* const* g_prim_pos, float* g_shift, void* stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !x || !v || !C || !F || !goal || !reward || !g_x || !g_v || !g_C || !g_F || n_prim < 1 || (Q != N && Q != 1) ||
      !prim_args_ok(n_prim, g_prim_pos_out, true) || !prim_args_ok(n_prim, (const float* const*)g_prim_pos, false) ||
      ((shift == nullptr) != (g_shift == nullptr))) {
    set_error("ud_mpm_finish_bwd: bad argument"); return UD_ERR_INVALID;
  }
  PrimIn gi{}; PrimOut go{};
  for (int p = 0; p < n_prim; ++p) { gi.p[p] = g_prim_pos_out[p]; go.p[p] = g_prim_pos[p]; }
  if (g_shift) UD_HIP_CHECK(hipMemsetAsync(g_shift, 0, (size_t)B * 3 * sizeof(float), (hipStream_t)stream));
  hipLaunchKernelGGL(mpm_finish_bwd_kernel, dim3((N + GLUE_T - 1) / GLUE_T, B), dim3(GLUE_T), 0, (hipStream_t)stream, N, S, n_prim, Q, x, v, C, F, shift, goal, reward,
                     g_x_out, g_v_out, g_C_out, g_F_out, gi, g_reward, g_obs, g_x, g_v, g_C, g_F, go, g_shift);
  UD_HIP_CHECK(hipGetLastError());
  return UD_OK;
}

}  // extern "C"
