// Calibration: sustained v_mfma_f32_32x32x16_bf16 rate with no memory traffic, NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
// build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o mfma_bf16_peak
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  u32x4 xs[4], ys[4];
  for (int r = 0; r < 4; ++r) for (int e = 0; e < 4; ++e) {
    h = h * 1664525u + 1013904223u; xs[r][e] = (h & 0x807f807fu) | 0x3f003f00u;
    h = h * 1664525u + 1013904223u; ys[r][e] = (h & 0x807f807fu) | 0x3f003f00u;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, xs[r]), __builtin_bit_cast(bf16x8_t, ys[(r + a) & 3]), acc[a], 0, 0, 0);
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1234u + rep);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * NACC * (32.0 * 32 * 16 * 2);
    printf("bf16 32x32x16: NACC %d blocks %4d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, ms, flops / ms / 1e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  run<4>(out, 256); run<4>(out, 512); run<1>(out, 512); run<2>(out, 512);
  return 0;
}
