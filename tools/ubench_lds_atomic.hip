// LDS float atomic throughput on one CU: ds_add_f32 with distinct addresses, k-way same-address conflicts, and the
// same traffic as plain ds_write_b32.  hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_atomic.hip -o tools/ubench_lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, int iters, int share, int stride) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  __shared__ unsigned long long lds64[2048];
  __shared__ double ldsd[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) { lds64[i] = 0; ldsd[i] = 0.0; }
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int idx = ((threadIdx.x / share) * stride) & 8191;   // `share` consecutive lanes hit one address
  float v = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) __hip_atomic_fetch_add(&lds[(idx + u * 64) & 8191], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 1) ((volatile float*)lds)[(idx + u * 64) & 8191] = v;
      if (KIND == 3) __hip_atomic_fetch_add((unsigned*)&lds[(idx + u * 64) & 8191], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 4) __hip_atomic_fetch_add(&lds64[(idx + u * 64) & 2047], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 5) __hip_atomic_fetch_add(&ldsd[(idx + u * 64) & 2047], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 6) __hip_atomic_fetch_max((int*)&lds[(idx + u * 64) & 8191], (int)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 2) v += ((volatile float*)lds)[(idx + u * 64) & 8191];
    }
  }
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x] + v + (float)lds64[threadIdx.x] + (float)ldsd[threadIdx.x];
}

template <int KIND>
void run(const char* name, int threads, int share, int stride) {
  float* d; hipMalloc(&d, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, share, stride);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, share, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 * 2.4e9 / iters / 16 / (threads / 64);
  printf("%-14s threads=%4d share=%2d stride=%2d : %7.1f cycles (2.4 GHz) per wave-instruction, %5.2f per lane\n", name, threads, share, stride, per, per / 64);
  hipFree(d);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  for (int th : {64, 320}) {
    run<0>("ds_add_f32", th, 1, 1);
    run<0>("ds_add_f32", th, 2, 1);
    run<0>("ds_add_f32", th, 4, 1);
    run<0>("ds_add_f32", th, 64, 1);
    run<0>("ds_add_f32", th, 1, 3);
    run<0>("ds_add_f32", th, 1, 33);
    run<3>("ds_add_u32", th, 1, 1);
    run<3>("ds_add_u32", th, 4, 1);
    run<3>("ds_add_u32", th, 64, 1);
    run<4>("ds_add_u64", th, 1, 1);
    run<4>("ds_add_u64", th, 4, 1);
    run<5>("ds_add_f64", th, 1, 1);
    run<6>("ds_max_i32", th, 1, 1);
    run<1>("ds_write_b32", th, 1, 1);
    run<1>("ds_write_b32", th, 4, 1);
    run<2>("ds_read_b32", th, 1, 1);
  }
  return 0;
}
