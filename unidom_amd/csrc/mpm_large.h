// Host interface of the many-workgroup MPM path (mpm_large.hip), used by the C ABI in mpm.hip when N > 128.
#pragma once
#include "mpm_device.h"

namespace ud {

struct MpmLarge;
// kernel selection of a handle, fixed at create (ud_mpm_conf: max_envs and the `tune_*` fields; 0 = the library's choice by measurement)
struct LgTune {
  int max_envs;            // every arena is sized for this many envs at create
  int lanes;               // lanes per particle of the multi-kernel particle kernels: 0 by launch size, 1, 4
  int cluster;             // persistent forward (mpm_cluster.h): 0 where the library's rule takes it, 1 wherever it fits, -1 never
  int cluster_part_lanes;  // lanes per part of the persistent forward: 0 = 128 (32 particles), 64 (16 particles)
  int cluster_envs;        // > 0: at most this many envs per persistent launch (tests: several launches per call)
  int env_groups;          // > 0: this many stream groups on the multi-kernel path
  int bwd_two_launch;      // backward with the grid checkpoint: 0 two launches per substep where they apply, -1 always the four-kernel sequence
  int collide_records;     // soft contact + grid checkpoint: 0 the grid op records (e, n) per cell and primitive for the adjoint, -1 the adjoint evaluates the SDFs again
};
MpmLarge* mpm_large_create(const MpmConst& c, const int* d_material, const float* d_hard, bool has_liquid, const LgTune& tune);   // has_liquid: some particle has material 0; nullptr = an allocation failed (ud_last_error)
void mpm_large_destroy(MpmLarge* L);
int mpm_large_reset(MpmLarge* L, hipStream_t st);   // every arena back to its rest state (after a device-side time-out)
size_t mpm_large_ckpt_bytes(const MpmLarge* L, int B);
int mpm_large_plan(MpmLarge* L, int B);   // ud_mpm_launch_plan's bits for a call with B envs
int mpm_large_ckpt_cells(const MpmLarge* L, int B, const float* ckpt, int* cells, hipStream_t st);   // ud_mpm_ckpt_cells
int mpm_large_step_fwd(MpmLarge* L, int B, const float* x, const float* v, const float* C, const float* F, const float* J,
                       const float* ppos, const float* prot, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, float* xo, float* vo, float* Co, float* Fo, float* Jo, float* ppos_o,
                       float* prot_o, float* pv_o, float* pw_o, float* ckpt, int* status, hipStream_t st);
int mpm_large_step_bwd(MpmLarge* L, int B, const float* ckpt, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, const float* gx, const float* gv, const float* gC, const float* gF,
                       const float* gppos, const float* gprot, int clip, float* gx0, float* gv0, float* gC0, float* gF0, float* gppos0,
                       float* grot0, float* gfric, float* gmu, float* glam, float* gaction, int* status, hipStream_t st);

}  // namespace ud
