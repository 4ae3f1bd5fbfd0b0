// Hardware check of exact_math.h's division and guarded sqrt against the compiler's IEEE f32 division / sqrtf (hipcc's default:
// correctly rounded) on an MI355X.  Operand pairs per thread from a counter-based generator, in four families:
//   0  random sign / mantissa, exponents uniform over the stated in-range window (d: 2^-40 .. 2^40, a: 2^-60 .. 2^60, upper ends exclusive)
//   1  near-halfway quotients: a = RN(q * d) for a random q, then a's bits moved by -2 .. +2 -- the cases where a one-ulp error of
//      the quotient shows (a / d lands next to a float or next to the midpoint of two)
//   2  edge operands: mantissa all ones / all zeros / one bit, a = +-0, d at the window's ends
//   3  ANY bit patterns (denormals, inf, NaN, huge, tiny): div_rn / sqrt_rn must take their fallback and still agree (NaN == NaN)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/check_exact_div.hip -o tools/check_exact_div ; run: tools/check_exact_div [log2 cases per family, default 33] [AE DE]
#include "../unidom_amd/csrc/exact_math.h"
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {   // splitmix64
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float mk(unsigned sign, int e, unsigned man) { return __builtin_bit_cast(float, (sign << 31) | ((unsigned)(e + 127) << 23) | (man & 0x7fffffu)); }
__device__ __forceinline__ bool same(float x, float y) { return __builtin_bit_cast(unsigned, x) == __builtin_bit_cast(unsigned, y) || (x != x && y != y); }

__global__ void check(int family, int AE, int DE, unsigned long long n, unsigned long long seed, unsigned long long* bad, float* first) {
  unsigned long long nb = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long r0 = mix(seed + 2 * i), r1 = mix(seed + 2 * i + 1);
    float a, d;
    if (family == 0) {
      d = mk((unsigned)(r0 >> 63), (int)((r0 >> 32) % (2 * DE)) - DE, (unsigned)r0);
      a = mk((unsigned)(r1 >> 63), (int)((r1 >> 32) % (2 * AE)) - AE, (unsigned)r1);
    } else if (family == 1) {
      d = mk((unsigned)(r0 >> 63), (int)((r0 >> 32) % 61) - 30, (unsigned)r0);
      const float q = mk((unsigned)(r1 >> 63), (int)((r1 >> 32) % 41) - 20, (unsigned)r1);
      const int k = (int)((r1 >> 40) % 5) - 2;
      a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, q * d) + (unsigned)k);
    } else if (family == 2) {
      const unsigned pick[6] = {0x7fffffu, 0u, 1u, 0x400000u, 0x7ffffeu, 0x555555u};
      d = mk((unsigned)(r0 >> 63), (r0 & 1) ? ((r0 & 2) ? DE - 1 : -DE) : (int)((r0 >> 32) % (2 * DE)) - DE, pick[(r0 >> 8) % 6]);
      a = ((r1 & 7) == 0) ? ((r1 & 8) ? -0.0f : 0.0f) : mk((unsigned)(r1 >> 63), (r1 & 16) ? ((r1 & 32) ? AE - 1 : -AE) : (int)((r1 >> 32) % (2 * AE)) - AE, pick[(r1 >> 8) % 6]);
    } else {
      d = __builtin_bit_cast(float, (unsigned)r0);
      a = __builtin_bit_cast(float, (unsigned)r1);
    }
    const float ref = a / d;
    bool ok = same(ud::div_rn(a, d), ref);
    if (family < 3) {
      if (AE == 60 && DE == 40 && !(ud::div_den_inrange(d) && ud::div_num_inrange(a))) ok = false;   // the generator must stay inside div_rn's window
      ok = ok && same(ud::div_rn_prepped(a, d, ud::div_prep(d)), ref);
      const float nz = ud::div_rn_prepped_nz(a, d, ud::div_prep(d));                       // the variant without the signed-zero select
      ok = ok && (same(nz, ref) || (a == 0.0f && nz == 0.0f));
    }
    const float x = __builtin_bit_cast(float, (unsigned)(r0 >> 16));                          // any pattern
    ok = ok && same(ud::sqrt_rn(x), sqrtf(x));
    if (!ok) { if (nb == 0 && atomicAdd(bad + 4, 1ull) == 0) { first[0] = a; first[1] = d; first[2] = x; } ++nb; }
  }
  if (nb) atomicAdd(bad + family, nb);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 33;
  // operand windows: |a| in [2^-AE, 2^AE), |d| in [2^-DE, 2^DE).  60 / 40 = div_rn's guarded window (mpm_collide.h); 100 / 24 = the
  // cloth's reference-order forward (cloth.hip), whose denominators are spring lengths and friction speeds
  const int AE = argc > 2 ? atoi(argv[2]) : 60, DE = argc > 3 ? atoi(argv[3]) : 40;
  unsigned long long *bad, h[5];
  float *first, hf[3];
  if (hipMalloc(&bad, 40) != hipSuccess || hipMalloc(&first, 12) != hipSuccess) return 2;
  (void)hipMemset(bad, 0, 40);
  (void)hipMemset(first, 0, 12);
  const unsigned long long n = 1ull << lg;
  for (int f = 0; f < 4; ++f) hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, f, AE, DE, n, 0x1234567ull * (f + 1), bad, first);
  if (hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  (void)hipMemcpy(hf, first, 12, hipMemcpyDeviceToHost);
  const char* name[4] = {"random in-range", "near-halfway", "edge operands", "any bit patterns (fallback)"};
  printf("windows: |a| in [2^-%d, 2^%d), |d| in [2^-%d, 2^%d)\n", AE, AE, DE, DE);
  for (int f = 0; f < 4; ++f) printf("div_rn / div_rn_prepped(_nz) / sqrt_rn, %-28s: %llu mismatches of %llu pairs\n", name[f], h[f], n);
  if (h[4]) printf("first mismatch: a = %a  d = %a  x = %a\n", hf[0], hf[1], hf[2]);
  return (h[0] || h[1] || h[2] || h[3]) ? 1 : 0;
}
