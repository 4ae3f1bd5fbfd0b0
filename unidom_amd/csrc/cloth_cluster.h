// Cloth bodies of more than 1024 particles on SEVERAL workgroups per environment (fold_tshirt: 3573 particles = 7 parts).
//
// The reference sizes run one workgroup per env (cloth_v2.hip / cloth_fast.hip: one particle per lane, everything in
// registers, 2 waves per SIMD).  A 3573-particle body on one workgroup is 4 particles per lane on ONE CU: 23 us (forward) /
// 92 us (literal adjoint) per substep while 252 CUs idle.  Here an env is cut into W = ceil(P / 512) parts of 512
// consecutive particles; each part is a 512-lane workgroup of the SAME per-particle code as the one-workgroup kernels,
// on its own CU, and what crosses a part boundary per substep goes through HBM:
//   forward   the positions of the H particles either side of the part (H >= the widest index distance of a spring);
//   adjoint   the nine block sums of norm_grad (W x 9 floats, every part adds them in the same order, so all parts hold
//             bit-identical totals) and the force cotangents of the same halo.
// Hand-off protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): every
// exchanged float travels as ONE naturally aligned 8-byte granule {value, tag} written by one agent-scope (sc1,
// write-through) store and read by agent-scope (sc1, L1-bypassing) loads; tag = the substep counter.  A consumer polls
// the granules it needs until their tags match -- no flag, no fence, no atomic read-modify-write, nothing to order, and no
// dependence on where a workgroup runs (blocks b and b + 8 usually share an XCD; parts of one env are spread that way,
// for speed only).  Buffers that are rewritten while a slow part may still read the previous value are double-buffered by
// step parity (positions, sums); see the reuse arguments at each buffer.
// Progress: a part waits for other parts of its env, so all W parts must be resident together.  The host cuts a call
// into launches of at most floor(CUs / W) envs (one 512-lane workgroup per CU: the adjoint kernel uses the whole register
// file of a CU), back to back on the caller's stream, so every workgroup of a launch is resident at once.  Every poll is bounded: after CL_SPIN_LIMIT polls (seconds) a part gives up, its siblings time out
// in turn, every output of the env is filled with NaN and each part that gave up adds 1 to the handle's time-out counter
// (ClusterArgs::timeouts -> ud_cloth_poll_timeouts -> ClothSimulator.check_status) -- loud, and every wave reaches the end of the kernel.
#pragma once
#include "cloth_common.h"

namespace ud {

constexpr int CL_T = 512;                            // lanes (particles) per part
constexpr int CL_HMAX = 256;                         // widest halo supported (T-shirt: 79 -> 128)
constexpr int CL_STRIDE = CL_T + 2 * CL_HMAX + 1;    // LDS plane stride in floats (odd: see UD_CLOTH_MAXP in cloth_fast.hip)
constexpr unsigned CL_SPIN_LIMIT = 1u << 22;         // polls of ~1 us each before a part gives up
constexpr int CL_SLOT = 16;                          // granules per sum slot (9 block sums / 8 action sums + 2)

typedef unsigned long long cl_granule;               // {value bits (low), tag (high)}

struct ClusterArgs {
  int W, H;                // parts per env; halo width (multiple of 64, <= CL_HMAX)
  int b0, Bl;              // this launch covers envs b0 .. b0 + Bl - 1 of the call's B (the host cuts a call into launches
                           // whose parts all fit on the chip at once); the arena is indexed by the env's number in the launch
  cl_granule* arena;       // [Bl][cl_env_granules(Pp, W)], zeroed before every launch (tags start at 1)
  int* timeouts;           // handle-owned device counter: + 1 for every part that gave up a poll (ud_cloth_poll_timeouts)
};

// per-env arena, in granules: XE[2][3][Pp] positions (forward) | GE[2][3][Pp] force cotangents (adjoint) |
// SE[2][W][CL_SLOT] block sums | SA[2][W][CL_SLOT] action / parameter sums; every [2] is the step (or macro-step) parity
__host__ __device__ inline size_t cl_env_granules(int Pp, int W) { return (size_t)12 * Pp + (size_t)4 * W * CL_SLOT; }

__device__ __forceinline__ void cl_put(cl_granule* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((cl_granule)tag << 32) | (cl_granule)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cl_get(const cl_granule* p, unsigned tag, float& v) {
  const cl_granule g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __builtin_bit_cast(float, (unsigned)g);
  return (unsigned)(g >> 32) == tag;
}

// blockIdx -> (env, part).  Block ids are dealt round-robin over the 8 XCDs, so ids congruent mod 8 share an XCD's L2:
// the parts of env b get ids b % 8 + 8 * (W * (b / 8) + w).  Placement is a speed matter only (see above).
__device__ __forceinline__ void cl_decode(int W, int& bl, int& w) {   // bl = the env's number inside this launch
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  bl = (j / W) * 8 + xcd;
  w = j % W;
}
__host__ inline int cl_grid(int B, int W) { return 8 * W * ((B + 7) / 8); }

// Poll up to three granules (one per component plane, `plane` granules apart) until their tags match; wave-uniform
// loop, bounded.  Returns false when the wave gave up.
__device__ __forceinline__ bool cl_poll3(const cl_granule* p, size_t plane, unsigned tag, bool need, float (&out)[3]) {
  bool ok0 = !need, ok1 = !need, ok2 = !need;
  out[0] = out[1] = out[2] = 0.f;
  for (unsigned spins = 0;; ++spins) {
    if (!ok0) ok0 = cl_get(p, tag, out[0]);
    if (!ok1) ok1 = cl_get(p + plane, tag, out[1]);
    if (!ok2) ok2 = cl_get(p + 2 * plane, tag, out[2]);
    if (__builtin_amdgcn_ballot_w64(!(ok0 && ok1 && ok2)) == 0) return true;
    if (spins > CL_SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}

void cloth_launch_fwd_cluster(const ClothFwdArgs& a, const ClusterArgs& q, hipStream_t stream);
void cloth_launch_bwd_cluster(const ClothBwdArgs& a, const ClusterArgs& q, hipStream_t stream);

}  // namespace ud
