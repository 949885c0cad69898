// Probe: operand/result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer products per instruction).
// Expectation checked here: lane l feeds A[blk = l/4][i = l%4] and B[blk][j = l%4]; result register r of lane l is
// D[blk][i = r][j = l%4] = A[blk][r] * B[blk][l%4].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256], *a, *b, *d;
  for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 100.f + 3.f * i; }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[4 * (l / 4) + r] * hb[l]) ++bad;
  printf("mfma_4x4x1 layout as expected: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3]);
  return bad != 0;
}
