// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the MPM kernels (MI355X_MICROARCH.md calibrates only
// the 16-B-per-lane streaming read: FETCH_SIZE reports half its bytes; "other access widths are uncalibrated").  Each kernel reads (or writes) a
// known number of bytes ONCE from a buffer larger than L2 + Infinity Cache (1 GiB), so the memory-side counters see every byte:
//   read16   16 B per lane, consecutive lanes consecutive addresses            (the calibrated case: expect FETCH_SIZE = bytes / 2)
//   read4     4 B per lane, consecutive lanes consecutive addresses            (SoA rows, one lane per particle)
//   read4q    4 B per lane, the four lanes of a quad on ONE address            (SoA rows, four lanes per particle: 64 B per wave-instruction)
//   gather16 16 B per lane at pseudo-random 16-B cells, 1 cell in 4 touched    (grid gathers: 64-B lines fetched for 16 useful bytes)
//   write16 / write4q / atomic4   the store-side twins (atomic4: float atomicAdd, one dword per lane, consecutive)
//   scatter16  16-B stores, one cell per 128-B line at pseudo-random lines (grid scatters)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
// Run (GPU box): for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --kernel-trace -d out_$C -o p -f csv -- tools/ubench_fetch; done   (tools/ubench_fetch.sh)
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t BYTES = 1ull << 30;   // per kernel

__global__ void read16(const float4* p, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 1.2345e30f) *sink = acc;
}
__global__ void read4(const float* p, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 1.2345e30f) *sink = acc;
}
__global__ void read4q(const float* p, size_t n, float* sink) {      // n = floats read; thread t reads element t / 4 of its stripe
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n; i += (size_t)gridDim.x * blockDim.x) acc += p[i >> 2];
  if (acc == 1.2345e30f) *sink = acc;
}
__global__ void gather16(const float4* p, size_t ncell, size_t n, float* sink) {   // n gathers; cell = 4 * hash: one 16-B cell per 64-B line
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t h = (i * 2654435761ull) % (ncell / 4);
    const float4 v = p[4 * h];
    acc += v.x + v.w;
  }
  if (acc == 1.2345e30f) *sink = acc;
}
__global__ void write16(float4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write4q(float* p, size_t n) {                         // one lane of each quad stores 4 B: 64 B per wave-instruction
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n; i += (size_t)gridDim.x * blockDim.x) if ((i & 3) == 0) p[i >> 2] = 1.f;
}
__global__ void scatter16(float4* p, size_t ncell, size_t n) {        // n stores of one 16-B cell per 128-B line, pseudo-random lines
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t h = (i * 2654435761ull) % (ncell / 8);
    p[8 * h] = make_float4(1.f, 2.f, 3.f, 4.f);
  }
}
__global__ void atomic4(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(p + i, 1.f);
}

// What a LAUNCH costs by itself: a kernel that takes a by-value argument struct the size of the MPM kernels' (LargeArgs: ~1 KB) and whose
// blocks read one field and leave -- FETCH_SIZE / WRITE_SIZE against the number of blocks (kernel arguments and code are fetched per
// workgroup / per XCD; the completion signal and the dispatch packet are written)
struct BigArgs { int v[256]; float* out; };
__global__ void __launch_bounds__(256) args1536(BigArgs a) { if (a.v[blockIdx.x & 255] == 123456789 && threadIdx.x == 0) *a.out = 1.f; }
__global__ void __launch_bounds__(256) args96(BigArgs a) { if (a.v[blockIdx.x & 255] == 123456789 && threadIdx.x == 0) *a.out = 1.f; }

int main() {
  void *buf, *sink;
  if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 2;
  (void)hipMemset(buf, 0, BYTES);
  const dim3 g(8192), b(256);
  hipLaunchKernelGGL(read16, g, b, 0, 0, (const float4*)buf, BYTES / 16, (float*)sink);
  hipLaunchKernelGGL(read4, g, b, 0, 0, (const float*)buf, BYTES / 4, (float*)sink);
  hipLaunchKernelGGL(read4q, g, b, 0, 0, (const float*)buf, BYTES / 4, (float*)sink);
  hipLaunchKernelGGL(gather16, g, b, 0, 0, (const float4*)buf, BYTES / 16, BYTES / 64, (float*)sink);
  hipLaunchKernelGGL(write16, g, b, 0, 0, (float4*)buf, BYTES / 16);
  hipLaunchKernelGGL(write4q, g, b, 0, 0, (float*)buf, BYTES / 4);
  hipLaunchKernelGGL(atomic4, g, b, 0, 0, (float*)buf, BYTES / 4);
  hipLaunchKernelGGL(scatter16, g, b, 0, 0, (float4*)buf, BYTES / 16, BYTES / 128);
  BigArgs ba{};
  ba.out = (float*)sink;
  for (int r = 0; r < 100; ++r) {       // 100 launches each: the counters are per launch, the mean is what tools/ubench_fetch.sh reports
    hipLaunchKernelGGL(args1536, dim3(1536), dim3(256), 0, 0, ba);
    hipLaunchKernelGGL(args96, dim3(96), dim3(256), 0, 0, ba);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  printf("useful bytes per kernel: read16 / read4 / read4q %zu, gather16 %zu (in %zu bytes of 64-B lines), write16 / write4q / atomic4 %zu\n", BYTES, BYTES / 4, BYTES, BYTES);
  return 0;
}
