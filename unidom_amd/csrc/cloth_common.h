// Shared argument structs of the cloth kernels (exact-order kernels: cloth.hip, fast kernels: cloth_fast.hip).
#pragma once
#include "common.h"

namespace ud {

struct ClothConst {
  float gdt;      // float(gravity*dt)        :259
  float g;        // gravity                  :278
  float dt;
  float damp;     // exp(-damping*dt) in f32  :309
  float max_v;
  float eps;      // small_num
  float n_mask;   // cloth_mask.sum()         :192
  int P, Pp, S;
  float cell;     // 1/N
  float Ls, Ld;   // rest lengths of the straight (links 0-3) / diagonal (4-7) springs, f32 as cloth_simulator.py:61-63
};

struct ClothFwdArgs {
  ClothConst c;
  const int* nbr;      // [8][Pp]
  const float* L0;     // [8][Pp]
  int B, T;
  const float *x, *v, *prim, *k, *mu, *actions;
  float *x_out, *v_out, *prim_out, *x_list, *v_list, *prim_list;
  float* ckpt;
  uint8_t* grasp;
};

struct ClothBwdArgs {
  ClothConst c;
  const int* nbr;
  const float* L0;
  int B, T;
  const float* ckpt;
  const float *k, *mu, *actions;
  const float *g_x, *g_v, *g_prim, *g_x_list, *g_v_list, *g_prim_list;
  int normalize;
  float *g_x0, *g_v0, *g_prim0, *g_actions, *g_k, *g_mu;
};


// Checkpoint arena written by the forward kernels, read by the backward kernels:
//   per env b: (T*S + 1) records of (6*Pp + 8) floats = x[3][Pp] | v[3][Pp] | primitive0[4] primitive1[4];
//   record t*S+s is the INPUT state of substep (t,s); the last record is the final state.
__host__ __device__ inline size_t cloth_rec_floats(int Pp) { return (size_t)6 * Pp + 8; }
__host__ __device__ inline size_t cloth_env_records(int T, int S) { return (size_t)T * S + 1; }

// The grasp test |x - pos| <= radius (cloth_simulator.py:201,213) without a per-particle square root.  IEEE sqrtf
// is monotone, so  sqrtf(s) <= r  <=>  s <= T(r)  where T(r) is the largest float whose correctly rounded root is
// <= r: the booleans are identical to the sqrt form's (this is what keeps cloth_v2.hip bit-exact to the oracle).
// T(r) lies within [-1, +3] ulp of RN(r*r) (|(r + ulp/2)^2 - r^2| < 2.5 ulp(r^2)), so a short walk finds it; the
// result is verified and a miss traps instead of returning a wrong set (unreachable: tools/check_exact_math.hip runs
// the same walk for all 2 139 095 040 finite non-negative radii on the GPU and finds every one tight).
__device__ inline float grasp_thr(float r) {
  if (!(r >= 0.f)) return -1.f;                         // negative / NaN radius never grasps (s >= 0)
  if (r == INFINITY) return r;
  unsigned t = __builtin_bit_cast(unsigned, r * r);
  for (int it = 0; it < 4; ++it) if (sqrtf(__builtin_bit_cast(float, t)) > r) --t;
  for (int it = 0; it < 4; ++it) if (sqrtf(__builtin_bit_cast(float, t + 1u)) <= r) ++t;
  if (!(sqrtf(__builtin_bit_cast(float, t)) <= r && !(sqrtf(__builtin_bit_cast(float, t + 1u)) <= r))) __builtin_trap();
  return __builtin_bit_cast(float, t);
}

// The radius is component 3 of a primitive; the substep maps it to clip(radius + 0, 0, 1) (:322-323), which is
// idempotent: only the very first substep of a rollout can see an unclipped radius.  Two thresholds per gripper
// (first substep / every later one), computed once per launch, uniform.
struct GraspThr {
  float first, rest;
  __device__ __forceinline__ void init(float radius0) {
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, radius0)));
    first = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, grasp_thr(r0))));
    rest = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, grasp_thr(clipf(r0 + 0.f, 0.f, 1.f)))));
  }
  __device__ __forceinline__ float at(bool first_substep) const { return first_substep ? first : rest; }
};

void cloth_launch_fwd_v2(const ClothFwdArgs& a, hipStream_t stream);
bool cloth_ref_fast_ok(const ClothConst& c);                                   // cloth_ref.hip: the mode-3 forward's per-launch checks
void cloth_launch_fwd_ref(const ClothFwdArgs& a, hipStream_t stream);
void cloth_launch_fwd_fast(const ClothFwdArgs& a, hipStream_t stream);
void cloth_launch_bwd_fast(const ClothBwdArgs& a, hipStream_t stream);

}  // namespace ud
