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

void cloth_launch_fwd_v2(const ClothFwdArgs& a, hipStream_t stream);
void cloth_launch_fwd_fast(const ClothFwdArgs& a, hipStream_t stream);
void cloth_launch_bwd_fast(const ClothBwdArgs& a, hipStream_t stream);

}  // namespace ud
