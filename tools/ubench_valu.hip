// Micro-benchmark: issue cost of scalar vs packed f32 VALU on one CU (1 workgroup, 8 waves = 2 per SIMD),
// the occupancy the cloth kernels run at.  Build: hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 sv = {s, s};
  int s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  __shared__ float ldsbuf[512 * 4 + 64];
  ldsbuf[threadIdx.x] = a0;
  const unsigned ldsaddr = (unsigned)(threadIdx.x & 511) * 16;
  __syncthreads();
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // 8 scalar fma
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    } else if (KIND == 1) {  // 4 packed fma = same flops
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
    } else if (KIND == 2) {  // 8 scalar mul
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    } else if (KIND == 3) {  // 4 packed mul
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
    } else if (KIND == 4) {  // 4 packed add
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
    } else if (KIND == 5) {  // 8 cndmask
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    } else if (KIND == 6) {  // 8 dpp adds (row_shr:1)
      asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                   "v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == 7) {  // 8 rsq
      asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == 8) {  // dependent chain of 8 scalar fma
      asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                   "v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                   : "+v"(a0) : "v"(s));
    } else if (KIND == 9) {  // dependent chain of 4 packed fma
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n"
                   : "+v"(p0) : "v"(sv));
    } else if (KIND == 10) {  // 8 salu
      asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                   "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
    } else if (KIND == 12) {  // 8 x (v_fma ; s_nop 1)
      asm volatile("v_fma_f32 %0, %0, %8, %8\n s_nop 1\n v_fma_f32 %1, %1, %8, %8\n s_nop 1\n v_fma_f32 %2, %2, %8, %8\n s_nop 1\n v_fma_f32 %3, %3, %8, %8\n s_nop 1\n"
                   "v_fma_f32 %4, %4, %8, %8\n s_nop 1\n v_fma_f32 %5, %5, %8, %8\n s_nop 1\n v_fma_f32 %6, %6, %8, %8\n s_nop 1\n v_fma_f32 %7, %7, %8, %8\n s_nop 1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    } else if (KIND == 13) {  // 8 x (v_fma ; s_add)
      asm volatile("v_fma_f32 %0, %0, %8, %8\n s_add_u32 %9, %9, 1\n v_fma_f32 %1, %1, %8, %8\n s_add_u32 %10, %10, 1\n v_fma_f32 %2, %2, %8, %8\n s_add_u32 %9, %9, 1\n v_fma_f32 %3, %3, %8, %8\n s_add_u32 %10, %10, 1\n"
                   "v_fma_f32 %4, %4, %8, %8\n s_add_u32 %9, %9, 1\n v_fma_f32 %5, %5, %8, %8\n s_add_u32 %10, %10, 1\n v_fma_f32 %6, %6, %8, %8\n s_add_u32 %9, %9, 1\n v_fma_f32 %7, %7, %8, %8\n s_add_u32 %10, %10, 1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "s"(s0), "s"(s1) : "scc");
    } else if (KIND == 14) {  // 8 independent ds_read_b128 + wait
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                   "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(ldsaddr));
    } else if (KIND == 11) {  // 8 v_cmp + cndmask pairs -> 4 pairs
      asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    }
  }
  long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(s0 + s1 + s2 + s3) + q0.x + q1.y + q2.z + q3.w;
  if (threadIdx.x == 0) ((long*)out)[512] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads, int per_iter) {
  float* d; hipMalloc(&d, 8192);
  const int iters = 5000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 0.999f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 0.999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long cyc; hipMemcpy(&cyc, (char*)d + 4096, 8, hipMemcpyDeviceToHost);
  printf("%-28s threads=%4d  %8.3f ms  %7.2f ns/iter  %6.2f ns/instr  (memtime ticks/iter %.1f)\n", name, threads, ms, ms * 1e6 / iters, ms * 1e6 / iters / per_iter, (double)cyc / iters);
  hipFree(d);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  printf("start\n");
  int ndev = 0; hipGetDeviceCount(&ndev); printf("devices %d\n", ndev);
  for (int th : {64, 256, 512, 1024}) {
    run<0>("8 v_fma_f32", th, 8);
    run<1>("4 v_pk_fma_f32", th, 4);
    run<2>("8 v_mul_f32", th, 8);
    run<3>("4 v_pk_mul_f32", th, 4);
    run<4>("4 v_pk_add_f32", th, 4);
    run<5>("8 v_cndmask", th, 8);
    run<6>("8 v_add_f32_dpp", th, 8);
    run<7>("8 v_rsq_f32", th, 8);
    run<8>("8 dep v_fma_f32", th, 8);
    run<9>("4 dep v_pk_fma_f32", th, 4);
    run<10>("8 s_add_u32", th, 8);
    run<12>("8 (v_fma; s_nop 1)", th, 16);
    run<13>("8 (v_fma; s_add)", th, 16);
    run<14>("8 ds_read_b128 + wait", th, 8);
    run<11>("4 (v_cmp+v_cndmask)", th, 8);
  }
  return 0;
}
