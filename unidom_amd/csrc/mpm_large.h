// Host interface of the many-workgroup MPM path (mpm_large.hip), used by the C ABI in mpm.hip when N > 128.
#pragma once
#include "mpm_device.h"

namespace ud {

struct MpmLarge;
MpmLarge* mpm_large_create(const MpmConst& c, const int* d_material, const float* d_hard, bool has_liquid);   // has_liquid: some particle has material 0
void mpm_large_destroy(MpmLarge* L);
size_t mpm_large_ckpt_bytes(const MpmLarge* L, int B);
int mpm_large_plan(MpmLarge* L, int B);   // ud_mpm_launch_plan's bits for a call with B envs
int mpm_large_step_fwd(MpmLarge* L, int B, const float* x, const float* v, const float* C, const float* F, const float* J,
                       const float* ppos, const float* prot, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, float* xo, float* vo, float* Co, float* Fo, float* Jo, float* ppos_o,
                       float* prot_o, float* pv_o, float* pw_o, float* ckpt, int* status, hipStream_t st);
int mpm_large_step_bwd(MpmLarge* L, int B, const float* ckpt, const float* psize, const float* friction, const float* mu,
                       const float* lamda, const float* action, const float* gx, const float* gv, const float* gC, const float* gF,
                       const float* gppos, const float* gprot, int clip, float* gx0, float* gv0, float* gC0, float* gF0, float* gppos0,
                       float* grot0, float* gfric, float* gmu, float* glam, float* gaction, int* status, hipStream_t st);

}  // namespace ud
