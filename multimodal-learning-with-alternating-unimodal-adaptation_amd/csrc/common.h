// Shared helpers for the gfx950 kernels of libmla_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mla_hip.h"

#define MLA_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void mla_set_error(const char* fmt, ...);

#define MLA_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      mla_set_error(__VA_ARGS__);           \
      return MLA_ERR_INVALID_ARG;           \
    }                                       \
  } while (0)

#define MLA_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      mla_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return MLA_ERR_LAUNCH;                                               \
    }                                                                      \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// the one BatchNorm expression (explicit fma) every kernel that forms or re-forms bn(x) uses: identical rounding everywhere
// (bn.hip: bn_val; the convolution kernels that fold relu(bn(.)) into their operand staging)
__device__ __forceinline__ float bn_val1(float x, float mu, float is, float ga, float be) { return fmaf((x - mu) * is, ga, be); }

// Logical tile id -> (tm, tn).  Narrow outputs (gridN <= 8: every ResNet conv) keep the row-major order.  Wide outputs
// (transformer Linears: N = 768..3072, up to 48 column tiles) are walked in column PANELS of 8 tiles: all row tiles of a
// panel before the next panel, so the panel's B operand (8 x BN x K x 4 B <= 3 MB) stays in the 4 MB per-XCD L2 while A
// streams through once per panel.  Row-major order re-read the whole weight matrix (7-9 MB) from the fabric for every row
// of tiles: 1.8 GB per 16448 x 2304 x 768 GEMM, which held the Linear layers at 76 TFLOP/s (profiles/r02_m3ae_*).
__device__ __forceinline__ void tile_coords(int wg, int gridM, int gridN, int& tm, int& tn) {
  if (gridN <= 8) {
    tm = wg / gridN;
    tn = wg - tm * gridN;
    return;
  }
  const int per_panel = gridM * 8;
  int panel = wg / per_panel;
  const int full = gridN >> 3;                     // panels of width 8; a narrower one follows if gridN % 8 != 0
  if (panel > full) panel = full;
  const int rem = wg - panel * per_panel;
  const int pw = panel < full ? 8 : gridN - full * 8;
  tm = rem / pw;
  tn = panel * 8 + (rem - tm * pw);
}

// Bijective XCD-aware remap of a flat workgroup id: consecutive logical ids land on the same XCD
// (blocks b and b+8 share an XCD/L2 under round-robin dispatch).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, l = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}
