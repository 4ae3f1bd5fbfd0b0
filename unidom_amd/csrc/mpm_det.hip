// Deterministic MPM forward: kernels around mpm_det.h.  This file alone is compiled with -DUD_MPM_EXACT -ffp-contract=off and hipcc's
// default correctly rounded f32 divide / sqrt (Makefile), so that its arithmetic is the IEEE arithmetic the host compiler gives the
// same source (oracle/csrc/mpm_det_host.cpp) -- the rest of the MPM path keeps its fast-math build.
// Test mode, not a fast path: three launches per substep, one thread per particle / per grid cell, the dense grid swept for the
// epoch stamps; cost in DESIGN.md 3.2.
#include "mpm_det.h"
#include "mpm_det_host.h"

namespace ud {

__global__ void __launch_bounds__(64) det_fk_kernel(DetArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, S = a.c.steps;
  if (b >= a.B) return;
  det_fk_rows(S, a.action + (long)b * 6, a.ppos + (long)b * S * 3, a.prot + (long)b * S * 4);
}

__global__ void __launch_bounds__(256) det_pre_kernel(DetArgs a, int f, int epoch) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const MpmConst& c = a.c;
  if (p >= c.N) return;
  const float* h = a.hist + (long)b * a.stride_b + (long)(a.pingpong ? (f & 1) : f) * a.rec;
  float* hn = a.hist + (long)b * a.stride_b + (long)(a.pingpong ? ((f + 1) & 1) : (f + 1)) * a.rec;
  det_pre_particle(c, h, hn, p, a.mu[b], a.lamda[b], a.material[p], a.hard[p], a.pre + (long)b * UD_DET_PRE * c.Np, a.flag + (long)b * a.G, epoch);
}

__global__ void __launch_bounds__(256) det_cells_kernel(DetArgs a, int f, int epoch) {
  const int b = blockIdx.y;
  const long lin = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const MpmConst& c = a.c;
  if (lin >= a.G || a.flag[(long)b * a.G + lin] != epoch) return;
  const int ck = (int)(lin % c.res[2]), cj = (int)((lin / c.res[2]) % c.res[1]), ci = (int)(lin / ((long)c.res[2] * c.res[1]));
  PrimF pf;
  const int S = c.steps;
  det_prim(S, f, a.ppos + (long)b * S * 3, a.prot + (long)b * S * 4, a.psize + (long)b * 3, a.action + (long)b * 6, a.friction[b], pf);
  float vo[3];
  det_cell(c, ci, cj, ck, a.pre + (long)b * UD_DET_PRE * c.Np, pf, vo);
  float* o = a.vel + ((long)b * a.G + lin) * 4;
  o[0] = vo[0]; o[1] = vo[1]; o[2] = vo[2]; o[3] = 0.f;
}

__global__ void __launch_bounds__(256) det_g2p_kernel(DetArgs a, int f) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const MpmConst& c = a.c;
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

int mpm_det_forward(const DetArgs& a, int* epoch, hipStream_t st) {
  const MpmConst& c = a.c;
  const int S = c.steps;
  const dim3 blk(256), gp((c.N + 255) / 256, a.B), gc((unsigned)((a.G + 255) / 256), a.B);
  hipLaunchKernelGGL(det_fk_kernel, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
  for (int f = 0; f < S; ++f) {
    const int e = ++*epoch;
    hipLaunchKernelGGL(det_pre_kernel, gp, blk, 0, st, a, f, e);
    hipLaunchKernelGGL(det_cells_kernel, gc, blk, 0, st, a, f, e);
    hipLaunchKernelGGL(det_g2p_kernel, gp, blk, 0, st, a, f);
  }
  hipLaunchKernelGGL(det_trq_kernel, dim3((a.B * S + 255) / 256), blk, 0, st, a);
  return hipGetLastError() == hipSuccess ? UD_OK : UD_ERR_HIP;
}

}  // namespace ud
