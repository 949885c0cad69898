// Micro-benchmark for the weight-gradient contraction: how many cycles does a wave need per 256 outputs contracted over 64 rows
//   (a) as ONE pass of 16 x v_mfma_f32_16x16x4_f32 with operands read by ds_read2_b32 from a [row][RS] window (what ships), and
//   (b) as ONE set of 64 x v_mfma_f32_4x4x1_16b_f32 (16 exact 4x4 tiles) with operands read four rows at a time by
//       ds_read_b128 from a TRANSPOSED window [col][68] (row = lane),
// at 1 and 2 waves per SIMD, alone and with a packed-FMA stream running in the partner wave?  Prints cycles per pass / set.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../opf-graph-neural-solver_amd/csrc/gns_dw.h"      // gws_pass: the shipped pass (operand ring)
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int MODE>   // 0: 16x16x4 passes   1: 4x4x1 sets   2: packed FMA only
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, int nsets_rt) {
  constexpr int nsets = 5;
  __shared__ __attribute__((aligned(16))) float win[8][64 * 68];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* w = win[wave];
  for (int i = lane; i < 64 * 68; i += 64) w[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[6];
  for (int s = 0; s < 6; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  f2 pk[8];
  for (int j = 0; j < 8; ++j) pk[j] = f2{1.f + lane, 2.f + j};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < nsets; ++s) {
        gws_pass<GwSubWide, 0>(w, lane, acc[s % 6]);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < nsets; ++s) {
        const int ca = (lane >> 4) * 4 + (lane & 3) + 16 * (s & 1), cb = ((lane >> 2) & 3) * 4 + (lane & 3) + 16 + 4 * s;   // some tile table
        const f32x4* pa = reinterpret_cast<const f32x4*>(w + (ca % 60) * 68);
        const f32x4* pb = reinterpret_cast<const f32x4*>(w + (cb % 60) * 68);
        f32x4 p0 = acc[s % 6], p1 = {0.f, 0.f, 0.f, 0.f}, p2 = p1, p3 = p1;      // four independent chains: a 4x4x1 result is not ready for the next instruction
        f32x4 ra[3], rb[3];                                                        // operand ring: the quads of step r4 + 2 are read behind the MFMAs of step r4
        ra[0] = pa[0]; rb[0] = pb[0]; ra[1] = pa[1]; rb[1] = pb[1];
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 16>([&](auto r_) {
          constexpr int r4 = decltype(r_)::value;
          if constexpr (r4 + 2 < 16) { ra[(r4 + 2) % 3] = pa[r4 + 2]; rb[(r4 + 2) % 3] = pb[r4 + 2]; }
          const f32x4 a = ra[r4 % 3], b = rb[r4 % 3];
          p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], p1, 0, 0, 0);
          p2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], p2, 0, 0, 0);
          p3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], p3, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        acc[s % 6] = (p0 + p1) + (p2 + p3);
      }
    } else {
#pragma unroll 4
      for (int r = 0; r < 64 * nsets; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = __builtin_elementwise_fma(pk[j], f2{1.0001f, 0.9999f}, f2{0.5f, 0.25f});
    }
  }
  const long long t1 = clock64();
  float sacc = 0.f;
  for (int s = 0; s < 6; ++s) sacc += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
  for (int j = 0; j < 8; ++j) sacc += pk[j].x + pk[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static double run(int waves, int iters, int nsets) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 2, nsets);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, nsets);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("   [mode %d waves %d: kernel %.3f ms = %.0f ns per pass/set per wave]\n", MODE, waves, ms, ms * 1e6 / iters / nsets);
  long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { tot += (double)h[b * 8 + w]; ++n; }
  hipFree(out); hipFree(cyc);
  return tot / n / iters / nsets;
}

int main() {
  for (int waves : {4, 8}) {
    printf("waves per CU %d (%d per SIMD):  16x16x4 pass %.0f cycles   4x4x1 set %.0f cycles   (both = 256 outputs over 64 rows);  512 packed FMAs %.0f cycles\n",
           waves, waves / 4, run<0>(waves, 200, 5), run<1>(waves, 200, 5), run<2>(waves, 50, 1));
  }
  return 0;
}
