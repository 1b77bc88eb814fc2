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
// Same loop with operands that differ per lane, per instruction and per iteration (hashed bits): data-dependent
// switching power lowers the sustained clock, so this is the attainable MFMA rate on real activations.
template <int NACC>
__global__ __launch_bounds__(256) void krand(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float xs[8], ys[8];
  for (int r = 0; r < 8; ++r) {
    h = h * 1664525u + 1013904223u; xs[r] = __uint_as_float(0x3f800000u | (h >> 9)) - 1.5f;
    h = h * 1664525u + 1013904223u; ys[r] = __uint_as_float(0x3f800000u | (h >> 9)) - 1.5f;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[r], ys[(r + a) & 7], acc[a], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 8; ++r) { xs[r] = __uint_as_float(__float_as_uint(xs[r]) ^ ((i * 0x9E3779B1u) & 0x807fffffu)); }
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
  for (int blocks : {512, 768}) {
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      krand<4><<<blocks, 256>>>(out, iters, 12345u + rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
      printf("random operands: blocks %4d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
