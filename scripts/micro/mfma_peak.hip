// Calibration: sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic (random-ish operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = seed + threadIdx.x * 0.001f, y = seed * 0.5f - threadIdx.x * 0.002f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    x = -x; y = y * 0.999f;
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 768}) {
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      k<4><<<blocks, 256>>>(out, iters, 1.37f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
      printf("blocks %4d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s\n", blocks, blocks / 256, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
