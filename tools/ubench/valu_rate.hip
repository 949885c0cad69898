// Micro-benchmark: what FP32 FMA rate does one CU sustain on gfx950 for the
// shapes the GNS kernels use?  (a) register-only v_fma_f32 chains, (b) v_pk_fma_f32,
// (c) a 25->10->10->20 LearningBlock with wave-uniform weights fetched through the
// scalar cache (s_load), as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

typedef float float2v __attribute__((ext_vector_type(2)));

template<int NACC>
__global__ void k_fma(float* out, int iters, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = threadIdx.x * 1e-3f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_fmaf(acc[j], a, b);
  }
  float s = 0; 
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int NACC>
__global__ void k_pkfma(float* out, int iters, float a, float b) {
  float2v acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = float2v{threadIdx.x * 1e-3f + j, 1.f + j};
  float2v av{a, a * 1.0001f}, bv{b, b * 0.999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_elementwise_fma(acc[j], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ float lrelu(float x) { return fmaxf(x, 0.01f * x); }

// one LearningBlock 25->10->10->20 per "row", weights wave-uniform from global (scalar loads)
__global__ void k_mlp(float* out, const float* __restrict__ W, int rows, float seed, int zero) {
  const float* W1 = W;            // [10][25]
  const float* b1 = W1 + 250;
  const float* W2 = b1 + 10;      // [10][10]
  const float* b2 = W2 + 100;
  const float* W4 = b2 + 10;      // [20][10]
  const float* b4 = W4 + 200;
  float x[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) x[i] = seed + 0.01f * i + 1e-4f * threadIdx.x;
  float total = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float* Wo = W + (r & zero);
    const float* W1 = Wo; const float* b1 = W1 + 250; const float* W2 = b1 + 10;
    const float* b2 = W2 + 100; const float* W4 = b2 + 10; const float* b4 = W4 + 200;
    float h1[10], h2[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      float a = b1[j];
#pragma unroll
      for (int i = 0; i < 25; ++i) a = __builtin_fmaf(W1[j * 25 + i], x[i], a);
      h1[j] = lrelu(a);
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      float a = b2[j];
#pragma unroll
      for (int i = 0; i < 10; ++i) a = __builtin_fmaf(W2[j * 10 + i], h1[i], a);
      h2[j] = lrelu(a);
    }
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      float a = b4[j];
#pragma unroll
      for (int i = 0; i < 10; ++i) a = __builtin_fmaf(W4[j * 10 + i], h2[i], a);
      x[j] += 1e-3f * a;   // feed back so nothing is hoisted
    }
    total += x[0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

// same, weights staged in LDS and read with broadcast ds_read
__global__ void k_mlp_lds(float* out, const float* __restrict__ W, int rows, float seed) {
  __shared__ float sW[592];
  for (int i = threadIdx.x; i < 590; i += blockDim.x) sW[i] = W[i];
  __syncthreads();
  const float* W1 = sW; const float* b1 = W1 + 250; const float* W2 = b1 + 10;
  const float* b2 = W2 + 100; const float* W4 = b2 + 10; const float* b4 = W4 + 200;
  float x[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) x[i] = seed + 0.01f * i + 1e-4f * threadIdx.x;
  float total = 0.f;
  for (int r = 0; r < rows; ++r) {
    float h1[10], h2[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      float a = b1[j];
#pragma unroll
      for (int i = 0; i < 25; ++i) a = __builtin_fmaf(W1[j * 25 + i], x[i], a);
      h1[j] = lrelu(a);
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      float a = b2[j];
#pragma unroll
      for (int i = 0; i < 10; ++i) a = __builtin_fmaf(W2[j * 10 + i], h1[i], a);
      h2[j] = lrelu(a);
    }
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      float a = b4[j];
#pragma unroll
      for (int i = 0; i < 10; ++i) a = __builtin_fmaf(W4[j * 10 + i], h2[i], a);
      x[j] += 1e-3f * a;
    }
    total += x[0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}



typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f16v* cptr16;

template<int IN, int H, int OUT>
struct Blk {
  static constexpr int oW1 = 0, ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H, total = ob4 + OUT;
  static constexpr int nch = (total + 15) / 16;
};


template<int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template<int IN, int H, int OUT>
__device__ __forceinline__ void mlp_stream(const float* blk, const float (&x)[IN], float (&y)[OUT]) {
  using B = Blk<IN, H, OUT>;
  float a1[H], a2[H];
  f16v bufA, bufB;
  bufA = *(const f16v*)(blk);
  static_for<0, B::nch>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    f16v& cur = (c & 1) ? bufB : bufA;
    f16v& nxt = (c & 1) ? bufA : bufB;
    if constexpr (c + 1 < B::nch) nxt = *(const f16v*)(blk + 16 * (c + 1));
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int w = c * 16 + t;
      const float s = cur[t];
      if constexpr (w < B::ob1) { constexpr int j = w / IN, i = w % IN; a1[j] = (i == 0) ? s * x[0] : __builtin_fmaf(s, x[i], a1[j]); }
      else if constexpr (w < B::oW2) { constexpr int j = w - B::ob1; float z = a1[j] + s; a1[j] = fmaxf(z, 0.01f * z); }
      else if constexpr (w < B::ob2) { constexpr int q = w - B::oW2; constexpr int j = q / H, i = q % H; a2[j] = (i == 0) ? s * a1[0] : __builtin_fmaf(s, a1[i], a2[j]); }
      else if constexpr (w < B::oW4) { constexpr int j = w - B::ob2; float z = a2[j] + s; a2[j] = fmaxf(z, 0.01f * z); }
      else if constexpr (w < B::ob4) { constexpr int q = w - B::oW4; constexpr int j = q / H, i = q % H; y[j] = (i == 0) ? s * a2[0] : __builtin_fmaf(s, a2[i], y[j]); }
      else if constexpr (w < B::total) { constexpr int j = w - B::ob4; y[j] += s; }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

__global__ void k_mlp_asm(float* out, const float* __restrict__ W, int rows, float seed, int zero) {
  float x[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) x[i] = seed + 0.01f * i + 1e-4f * threadIdx.x;
  float total = 0.f;
  for (int r = 0; r < rows; ++r) {
    float y[20];
    mlp_stream<25, 10, 20>(W + (r & zero), x, y);
#pragma unroll
    for (int j = 0; j < 20; ++j) x[j] += 1e-3f * y[j];
    total += x[0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}


// ---- packed (v_pk_fma_f32) variant: weights stored [in][out] so that two output neurons of the
// same input are one aligned SGPR pair; accumulators are float2; x_i is broadcast by op_sel.
typedef float f2 __attribute__((ext_vector_type(2)));
template<int IN, int H, int OUT>
struct BlkT {  // stream: W1t[IN][H], b1[H], W2t[H][H], b2[H], W4t[H][OUT], b4[OUT]
  static constexpr int oW1 = 0, ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + OUT * H, total = ob4 + OUT;
  static constexpr int nch = (total + 15) / 16;
};
__device__ __forceinline__ f2 lrelu2(f2 z) { return __builtin_elementwise_max(z, z * 0.01f); }

template<int IN, int H, int OUT>
__device__ __forceinline__ void mlp_stream_pk(const float* blk, const float (&x)[IN], f2 (&y)[OUT / 2]) {
  using B = BlkT<IN, H, OUT>;
  f2 a1[H / 2], a2[H / 2];
  f16v bufA, bufB;
  bufA = *(const f16v*)(blk);
  static_for<0, B::nch>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    f16v& cur = (c & 1) ? bufB : bufA;
    f16v& nxt = (c & 1) ? bufA : bufB;
    if constexpr (c + 1 < B::nch) nxt = *(const f16v*)(blk + 16 * (c + 1));
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 8>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int w = c * 16 + 2 * t;       // even element of a pair
      const f2 s = f2{cur[2 * t], cur[2 * t + 1]};
      if constexpr (w < B::ob1) { constexpr int i = w / H, j = (w % H) / 2; f2 xi = f2{x[i], x[i]}; a1[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a1[j]); }
      else if constexpr (w < B::oW2) { constexpr int j = (w - B::ob1) / 2; a1[j] = lrelu2(a1[j] + s); }
      else if constexpr (w < B::ob2) { constexpr int q = w - B::oW2; constexpr int i = q / H, j = (q % H) / 2;
        const float xs = (i & 1) ? a1[i / 2].y : a1[i / 2].x; f2 xi = f2{xs, xs}; a2[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, a2[j]); }
      else if constexpr (w < B::oW4) { constexpr int j = (w - B::ob2) / 2; a2[j] = lrelu2(a2[j] + s); }
      else if constexpr (w < B::ob4) { constexpr int q = w - B::oW4; constexpr int i = q / OUT, j = (q % OUT) / 2;
        const float xs = (i & 1) ? a2[i / 2].y : a2[i / 2].x; f2 xi = f2{xs, xs}; y[j] = (i == 0) ? s * xi : __builtin_elementwise_fma(s, xi, y[j]); }
      else if constexpr (w < B::total) { constexpr int j = (w - B::ob4) / 2; y[j] += s; }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

__global__ void k_mlp_pk(float* out, const float* __restrict__ W, int rows, float seed, int zero) {
  float x[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) x[i] = seed + 0.01f * i + 1e-4f * threadIdx.x;
  float total = 0.f;
  for (int r = 0; r < rows; ++r) {
    f2 y[10];
    mlp_stream_pk<25, 10, 20>(W + (r & zero), x, y);
#pragma unroll
    for (int j = 0; j < 10; ++j) { x[2 * j] += 1e-3f * y[j].x; x[2 * j + 1] += 1e-3f * y[j].y; }
    total += x[0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

__global__ void k_trig(float* out, int iters, float seed) {
  float a = seed + 1e-3f * threadIdx.x, s = 0.f;
  for (int it = 0; it < iters; ++it) { s += sinf(a) + cosf(a * 1.3f); a += 0.37f; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<class F> float timeit(F f, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  int ncu = 256;
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 64 * 1024));
  std::vector<float> hW(592);
  for (int i = 0; i < 592; ++i) hW[i] = 0.05f * ((i * 37 % 101) / 101.f - 0.5f);
  float* W; CK(hipMalloc(&W, 592 * 4)); CK(hipMemcpy(W, hW.data(), 592 * 4, hipMemcpyHostToDevice));
  printf("waves/SIMD | v_fma(8 acc) TF | v_fma(2 acc) TF | v_pk_fma(8) TF | mlp s_load TF | mlp lds TF | trig Gcalls/s\n");
  for (int wps : {1, 2, 3, 4, 8}) {
    int threads = 256;           // 4 waves = one per SIMD
    int blocks = ncu * wps;      // wps blocks per CU -> wps waves per SIMD (if resident)
    int iters = 20000;
    double fl = 2.0 * blocks * threads;
    float t8 = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-4f); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k_fma<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-4f); });
    float tp = timeit([&] { hipLaunchKernelGGL(k_pkfma<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-4f); });
    int rows = 2000;
    float tm = timeit([&] { hipLaunchKernelGGL(k_mlp, dim3(blocks), dim3(threads), 0, 0, out, W, rows, 0.1f, 0); });
    float tl = timeit([&] { hipLaunchKernelGGL(k_mlp_lds, dim3(blocks), dim3(threads), 0, 0, out, W, rows, 0.1f); });
    float ta = timeit([&] { hipLaunchKernelGGL(k_mlp_asm, dim3(blocks), dim3(threads), 0, 0, out, W, rows, 0.1f, 0); });
    float tk = timeit([&] { hipLaunchKernelGGL(k_mlp_pk, dim3(blocks), dim3(threads), 0, 0, out, W, rows, 0.1f, 0); });
    float tt = timeit([&] { hipLaunchKernelGGL(k_trig, dim3(blocks), dim3(threads), 0, 0, out, 4000, 0.1f); });
    printf("%10d | %15.1f | %15.1f | %14.1f | %13.1f | %10.1f | %8.1f | asm %8.1f | pk %8.1f\n", wps,
           fl * iters * 8 / t8 * 1e-9, fl * iters * 2 / t2 * 1e-9, fl * iters * 16 / tp * 1e-9,
           fl * rows * 550 / tm * 1e-9, fl * rows * 550 / tl * 1e-9,
           (double)blocks * threads * 4000 * 2 / tt * 1e-6, fl * rows * 590 / ta * 1e-9, fl * rows * 590 / tk * 1e-9);
  }
  return 0;
}
