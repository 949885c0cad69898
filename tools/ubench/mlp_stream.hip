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


// variant: 32-float steps (two s_load_dwordx16 in flight per wait)
template <int NF, class F>
__device__ __forceinline__ void stream_pairs32(cfp p, F&& f) {
  constexpr int NST = (NF + 31) / 32;
  p += opaque_zero();
  f16v a0, a1, b0, b1;
  a0 = *(cf16p)(p); a1 = *(cf16p)(p + 16);
  static_for<0, NST>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    f16v& c0 = (c & 1) ? b0 : a0; f16v& c1 = (c & 1) ? b1 : a1;
    f16v& n0 = (c & 1) ? a0 : b0; f16v& n1 = (c & 1) ? a1 : b1;
    touch(c0); touch(c1);
    if constexpr (c + 1 < NST) { n0 = *(cf16p)(p + 32 * (c + 1)); n1 = *(cf16p)(p + 32 * (c + 1) + 16); }
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int w = c * 32 + 2 * t;
      if constexpr (w < NF) { if constexpr (t < 8) f(std::integral_constant<int, w>{}, f2{c0[2 * t], c0[2 * t + 1]}); else f(std::integral_constant<int, w>{}, f2{c1[2 * (t - 8)], c1[2 * (t - 8) + 1]}); }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}
template <int IN, int H, int OUTP>
__device__ __forceinline__ void mlp_fwd32(cfp blk, const f2 (&x)[(IN + 1) / 2], f2 (&a1)[H / 2], f2 (&a2)[H / 2], f2 (&y)[OUTP / 2]) {
  using B = TLay<IN, H, OUTP>;
  stream_pairs32<B::total>(blk, [&](auto w_, f2 s) {
    constexpr int w = decltype(w_)::value;
    if constexpr (w < B::ob1) { constexpr int i = w / H, j = (w % H) / 2; const f2 xi = splat(lane_of<i>(x)); a1[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a1[j]); }
    else if constexpr (w < B::oW2) { constexpr int j = (w - B::ob1) / 2; a1[j] = lrelu2(a1[j] + s); }
    else if constexpr (w < B::ob2) { constexpr int q = w - B::oW2, i = q / H, j = (q % H) / 2; const f2 xi = splat(lane_of<i>(a1)); a2[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a2[j]); }
    else if constexpr (w < B::oW4) { constexpr int j = (w - B::ob2) / 2; a2[j] = lrelu2(a2[j] + s); }
    else if constexpr (w < B::ob4) { constexpr int q = w - B::oW4, i = q / OUTP, j = (q % OUTP) / 2; const f2 xi = splat(lane_of<i>(a2)); y[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, y[j]); }
    else { constexpr int j = (w - B::ob4) / 2; y[j] += s; }
  });
  pin_all(y);
}
template <int NB>
__global__ void __launch_bounds__(256) k_stream32(float* out, const float* __restrict__ W, int rows, float seed, int blkstride) {
  f2 x[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = f2{seed + 0.01f * i + 1e-4f * threadIdx.x, seed - 0.02f * i};
  float total = 0.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int r = 0; r < rows; ++r) {
    f2 a1[5], a2[5], y[10];
    const int b = (r + wave) % NB;
    mlp_fwd32<25, 10, 20>((cfp)W + b * blkstride, x, a1, a2, y);
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
  auto run32 = [&](auto nb_, int wps) {
    constexpr int NB = decltype(nb_)::value;
    int blocks = 256 * wps;
    float t = timeit([&] { hipLaunchKernelGGL(k_stream32<NB>, dim3(blocks), dim3(256), 0, 0, out, W, rows, 0.1f, stride); });
    return 2.0 * blocks * 256 * rows * 590 / t * 1e-9;
  };
#define ROW(NBV) printf("%23d | %15.1f | %15.1f | 32-float steps: %7.1f %7.1f\n", NBV, run(std::integral_constant<int, NBV>{}, 2), run(std::integral_constant<int, NBV>{}, 4), run32(std::integral_constant<int, NBV>{}, 2), run32(std::integral_constant<int, NBV>{}, 4));
  ROW(1) ROW(4) ROW(12)
  return 0;
}
