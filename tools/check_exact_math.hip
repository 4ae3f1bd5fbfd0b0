// Exhaustive hardware check of unidom_amd/csrc/exact_math.h against the compiler's IEEE sqrtf and division:
// every float in [2^-96, FLT_MAX] for the sqrt, every float in [2^-64, 2^64] for the reciprocal, and the
// grasp-threshold walk of cloth_common.h for every finite non-negative radius.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/check_exact_math.hip -o tools/check_exact_math
#include "../unidom_amd/csrc/exact_math.h"
#include <cstdio>
#include <cstring>

__global__ void check_sqrt(unsigned lo, unsigned hi, unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (unsigned)u);
    if (__builtin_bit_cast(unsigned, ud::sqrt_rn_inrange(x)) != __builtin_bit_cast(unsigned, sqrtf(x))) ++n;
  }
  if (n) atomicAdd(bad, n);
}
__global__ void check_rcp(unsigned lo, unsigned hi, unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (unsigned)u);
    if (__builtin_bit_cast(unsigned, ud::rcp_rn_inrange(x)) != __builtin_bit_cast(unsigned, 1.0f / x)) ++n;
  }
  if (n) atomicAdd(bad, n);
}
// grasp_thr (cloth_common.h): for every non-negative float radius the 4 + 4 step walk must end on the largest t with
// sqrtf(t) <= r (same code as the kernel's, with the trap replaced by a counter)
__global__ void check_grasp(unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < 0x7f800000ull; u += (unsigned long long)gridDim.x * blockDim.x) {
    const float r = __builtin_bit_cast(float, (unsigned)u);
    unsigned t = __builtin_bit_cast(unsigned, r * r);
    for (int it = 0; it < 4; ++it) if (sqrtf(__builtin_bit_cast(float, t)) > r) --t;
    for (int it = 0; it < 4; ++it) if (sqrtf(__builtin_bit_cast(float, t + 1u)) <= r) ++t;
    if (!(sqrtf(__builtin_bit_cast(float, t)) <= r && !(sqrtf(__builtin_bit_cast(float, t + 1u)) <= r))) ++n;
  }
  if (n) atomicAdd(bad, n);
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
  unsigned long long *d, h[3] = {0, 0, 0};
  if (hipMalloc(&d, 24) != hipSuccess) return 2;
  hipMemset(d, 0, 24);
  const unsigned s_lo = bits(0x1p-96f), s_hi = 0x7f7fffffu, r_lo = bits(0x1p-64f), r_hi = bits(0x1p64f);
  hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, s_lo, s_hi, d);
  hipLaunchKernelGGL(check_rcp, dim3(4096), dim3(256), 0, 0, r_lo, r_hi, d + 1);
  hipLaunchKernelGGL(check_grasp, dim3(4096), dim3(256), 0, 0, d + 2);
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("grasp_thr walk: %llu radii of %llu not tight\n", h[2], 0x7f800000ull);
  printf("sqrt_rn_inrange: %llu mismatches of %llu   rcp_rn_inrange: %llu mismatches of %llu\n", h[0],
         (unsigned long long)s_hi - s_lo + 1, h[1], (unsigned long long)r_hi - r_lo + 1);
  // special values of the sqrt
  return (h[0] || h[1] || h[2]) ? 1 : 0;
}
