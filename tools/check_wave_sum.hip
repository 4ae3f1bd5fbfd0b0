// Hardware check of ud::wave_sum8_t (transposed butterfly over DPP + gfx950 permlane swaps): every lane of every
// wave must return the wave total of value j = 4*bit4(lane) + 2*bit1(lane) + bit0(lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/check_wave_sum.hip -o tools/check_wave_sum
#include "../unidom_amd/csrc/common.h"
#include <cmath>
#include <vector>
namespace ud { void set_error(const char*, ...) {} }

__global__ void k(const float* in, float* out) {
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = in[q * blockDim.x + threadIdx.x];
  out[threadIdx.x] = ud::wave_sum8_t(v, threadIdx.x & 63);
}

int main() {
  const int n = 512;
  std::vector<float> h(8 * n), o(n);
  for (int q = 0; q < 8; ++q) for (int i = 0; i < n; ++i) h[q * n + i] = (float)((i * 7 + q * 13) % 17 - 8) + 0.25f * q;   // exact in f32
  float *di, *dout;
  if (hipMalloc(&di, h.size() * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) return 2;
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, di, dout);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const int lane = i & 63, wv = i >> 6, j = ((lane >> 4) & 1) * 4 + (lane & 3);
    double want = 0;
    for (int l = 0; l < 64; ++l) want += h[j * n + wv * 64 + l];
    if (std::fabs(o[i] - want) > 1e-3) { if (bad < 8) printf("lane %d: got %g want %g\n", i, o[i], want); ++bad; }
  }
  printf("wave_sum8_t: %s (%d bad of %d)\n", bad ? "FAIL" : "ok", bad, n);
  return bad != 0;
}
