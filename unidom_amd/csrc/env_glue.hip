// Cloth-env arithmetic either side of the rollout, fused: the reference jit-compiles step_diff, so XLA turns the
// chamfer reward, the contact distance and the 6-vector -> 40 x 8 macro-action expansion into a handful of fused
// kernels; done op by op on the device they are ~180 tiny launches (and a dense [B,P,Q] autograd graph) per
// step_diff.  Here each piece is one forward and one backward kernel, one workgroup per environment.
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
  float best = INFINITY; int bi = 0x7fffffff;
  for (int c = c0; c < nc; c += 4) {
    const float m = dir == 0 ? mean_sq3(ax, ay, az, cs[c], cs[nc + c], cs[2 * nc + c])
                             : mean_sq3(cs[c], cs[nc + c], cs[2 * nc + c], ax, ay, az);   // (x - y) in both directions
    if (m < best) { best = m; bi = c; }
  }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {   // the four lanes of a row: smallest value, then smallest index
    const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
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
  float best = INFINITY; int bi = 0x7fffffff;
  for (int p = tid; p < P; p += GLUE_T) {   // contact_distance = min_p |pick - x_p|  (:206-209)
    const float dx = px - xb[p * 3], dy = py - xb[p * 3 + 1], dz = pz - xb[p * 3 + 2];
    const float s = dx * dx + dy * dy + dz * dz;
    if (s < best) { best = s; bi = p; }
  }
  for (int off = 32; off >= 1; off >>= 1) {   // (value, first index) minimum over the wave
    const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < GLUE_T / 64; ++q)
      if (sv[q] < best || (sv[q] == best && si[q] < bi)) { best = sv[q]; bi = si[q]; }
    contact[b] = sqrtf(best);
    contact_idx[b] = bi;
  }
  if (tid < 40) {
    const int t = tid;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    if (t < 3) {
      r0 = (px - prim0[b * 4]) / 3.0f; r1 = (0.f - prim0[b * 4 + 1]) / 3.0f; r2 = (pz - prim0[b * 4 + 2]) / 3.0f; r3 = 1.f;
    } else if (t < 13) {
      r0 = 0.f / 10.0f; r1 = 0.06f / 10.0f; r2 = 0.f / 10.0f;
    } else if (t < 33) {
      r0 = (a[3] - px) / 20.0f; r1 = 0.f / 20.0f; r2 = (a[5] - pz) / 20.0f;
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
    for (int d = 0; d < 3; ++d) { down[d] /= 3.0f; move[d] /= 20.0f; }
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

}  // extern "C"
