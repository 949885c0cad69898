// How does the scalar-cache weight stream of the real mlp_fwd behave when the weight working set grows past the
// scalar cache (16 KB?) - and does a deeper prefetch help?  Uses the product's own device code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../opf-graph-neural-solver_amd/csrc/gns_device.h"

template <int NB>
__global__ void __launch_bounds__(256) k_stream(float* out, const float* __restrict__ W, int rows, float seed, int blkstride) {
  f2 x[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = f2{seed + 0.01f * i + 1e-4f * threadIdx.x, seed - 0.02f * i};
  float total = 0.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int r = 0; r < rows; ++r) {
    f2 a1[5], a2[5], y[10];
    const int b = (r + wave) % NB;
    mlp_fwd<25, 10, 20>((cfp)W + b * blkstride, x, a1, a2, y);
#pragma unroll
    for (int j = 0; j < 10; ++j) x[j] += y[j] * 1e-3f;
    total += x[0].x;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int stride = 592 + 16;   // 608 floats, 64-byte aligned blocks
  const int maxb = 64;
  std::vector<float> h(stride * maxb + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.05f * ((i * 37 % 101) / 101.f - 0.5f);
  float *W, *out;
  (void)hipMalloc(&W, h.size() * 4); (void)hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 4 * 256 * 8 * 256);
  printf("weight blocks (x2.4 KB) | waves/SIMD=2 TF | waves/SIMD=4 TF\n");
  const int rows = 2000;
  auto run = [&](auto nb_, int wps) {
    constexpr int NB = decltype(nb_)::value;
    int blocks = 256 * wps;
    float t = timeit([&] { hipLaunchKernelGGL(k_stream<NB>, dim3(blocks), dim3(256), 0, 0, out, W, rows, 0.1f, stride); });
    return 2.0 * blocks * 256 * rows * 590 / t * 1e-9;
  };
#define ROW(NBV) printf("%23d | %15.1f | %15.1f\n", NBV, run(std::integral_constant<int, NBV>{}, 2), run(std::integral_constant<int, NBV>{}, 4));
  ROW(1) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12) ROW(16) ROW(32) ROW(64)
  return 0;
}
