// Per-instruction issue cost on one CU at the cloth kernels' occupancy: 64-instruction unrolled bodies so that the
// loop overhead (s_add/s_cmp/s_cbranch) is amortised.  hipcc --offload-arch=gfx950 -O3 ubench_issue.hip -o ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define R8(x) x x x x x x x x
#define FMA8 "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %9\n"
#define MUL8 "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
#define PKFMA4 "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
#define PKMUL4 "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
#define MULNOP8 "v_mul_f32 %0, %0, %8\n s_nop 1\n v_mul_f32 %1, %1, %8\n s_nop 1\n v_mul_f32 %2, %2, %8\n s_nop 1\n v_mul_f32 %3, %3, %8\n s_nop 1\n v_mul_f32 %4, %4, %8\n s_nop 1\n v_mul_f32 %5, %5, %8\n s_nop 1\n v_mul_f32 %6, %6, %8\n s_nop 1\n v_mul_f32 %7, %7, %8\n s_nop 1\n"
#define MULSALU8 "v_mul_f32 %0, %0, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 %11, %11, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 %11, %11, 1\n v_mul_f32 %4, %4, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %5, %5, %8\n s_add_u32 %11, %11, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %7, %7, %8\n s_add_u32 %11, %11, 1\n"
#define SALU8 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
#define CND8 "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define RSQ8 "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
#define CNDS8 "v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
#define CMPS8 "v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n v_cmp_lt_f32_e64 s[24:25], %2, %8\n v_cmp_lt_f32_e64 s[26:27], %3, %8\n v_cmp_lt_f32_e64 s[20:21], %4, %8\n v_cmp_lt_f32_e64 s[22:23], %5, %8\n v_cmp_lt_f32_e64 s[24:25], %6, %8\n v_cmp_lt_f32_e64 s[26:27], %7, %8\n"
#define CMPCND8 "v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n v_cmp_lt_f32_e64 s[24:25], %2, %8\n v_cmp_lt_f32_e64 s[26:27], %3, %8\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e64 %6, %6, %8, s[24:25]\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]\n"
#define CMPVCC8 "v_cmp_lt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_f32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
#define DPP8 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define MOV8 "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
#define MULS8 "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n"
#define DEPMUL8 "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float s, float s2) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 sv = {s, s}, sw = {s2, s2};
  const unsigned long long m64 = 0x5555aaaa5555aaaaull + (unsigned)iters;
  const float sf = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
  int s0 = iters, s1 = iters + 1, s2i = iters + 2, s3 = iters + 3;
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) asm volatile(R8(FMA8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(s2));
    if (KIND == 1) asm volatile(R8(MUL8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (KIND == 2) asm volatile(R8(PKFMA4 PKFMA4) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv), "v"(sw));
    if (KIND == 3) asm volatile(R8(PKMUL4 PKMUL4) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
    if (KIND == 4) asm volatile(R8(MULNOP8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (KIND == 5) asm volatile(R8(MULSALU8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(s2), "s"(s0), "s"(s1) : "scc");
    if (KIND == 6) asm volatile(R8(SALU8) : "+s"(s0), "+s"(s1), "+s"(s2i), "+s"(s3) :: "scc");
    if (KIND == 7) asm volatile(R8(CND8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (KIND == 8) asm volatile(R8(RSQ8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (KIND == 10) asm volatile(R8(CNDS8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "s"(m64));
    if (KIND == 11) asm volatile(R8(CMPS8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s20","s21","s22","s23","s24","s25","s26","s27");
    if (KIND == 12) asm volatile(R8(CMPCND8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s20","s21","s22","s23","s24","s25","s26","s27");
    if (KIND == 13) asm volatile(R8(CMPVCC8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    if (KIND == 14) asm volatile(R8(DPP8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (KIND == 15) asm volatile(R8(MOV8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (KIND == 16) asm volatile(R8(MULS8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "s"(sf));
    if (KIND == 9) asm volatile(R8(DEPMUL8) : "+v"(a0) : "v"(s));
  }
  long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(s0 + s1 + s2i + s3);
  if (threadIdx.x == 0) ((long*)out)[1024] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads, int per_iter) {
  float* d; hipMalloc(&d, 16384);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 0.999f, 0.001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long cyc; hipMemcpy(&cyc, (char*)d + 8192, 8, hipMemcpyDeviceToHost);
  const double wall = ms * 1e-3 * 2.4e9 / iters / per_iter;   // cycles at 2.4 GHz per instruction of one wave's stream, kernel wall time
  const int wps = threads > 256 ? threads / 256 : 1;
  printf("%-26s threads=%4d  wave0 cycles/instr %6.2f   wall cycles/instr %6.2f   wall SIMD-cycles per wave-instr %6.2f\n", name, threads,
         (double)cyc / iters / per_iter, wall, wall / wps);
  hipFree(d);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  for (int th : {64, 512, 1024}) {
    run<0>("v_fma_f32 x64", th, 64);
    run<1>("v_mul_f32 x64", th, 64);
    run<2>("v_pk_fma_f32 x64", th, 64);
    run<3>("v_pk_mul_f32 x64", th, 64);
    run<4>("(v_mul; s_nop 1) x64", th, 128);
    run<5>("(v_mul; s_add) x64", th, 128);
    run<6>("s_add_u32 x64", th, 64);
    run<7>("v_cndmask x64", th, 64);
    run<8>("v_rsq_f32 x64", th, 64);
    run<9>("dependent v_mul x64", th, 64);
    run<10>("v_cndmask_e64 (sgpr cond)", th, 64);
    run<11>("v_cmp_e64 -> sgpr pair", th, 64);
    run<12>("4 v_cmp_e64 + 4 cndmask_e64", th, 64);
    run<13>("(v_cmp vcc; v_cndmask vcc)", th, 64);
    run<14>("v_add_f32_dpp x64", th, 64);
    run<15>("v_mov_b32 x64", th, 64);
    run<16>("v_mul_f32 v, v, sgpr", th, 64);
  }
  return 0;
}
