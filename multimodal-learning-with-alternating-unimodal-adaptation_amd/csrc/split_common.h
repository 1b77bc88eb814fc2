// Pieces shared by the split-bf16 kernels (conv_igemm_split.hip, stem_split.hip): fp32 -> three bf16 terms, product set.
#pragma once
#include "igemm_common.h"

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // low half = bf16(a), high half = bf16(b), RNE
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// a - b as ONE v_sub_f32: the SLP vectoriser otherwise pairs the two residual subtractions of split_pair into v_pk_add_f32,
// which costs more issue time beside MFMAs than the two scalar subtractions it replaces (gather-GEMM forward / input gradient:
// -0.5...1 % time in a same-box A/B; weight gradient -3.5 %: there the packing even needs v_mov pairs to line the registers up.
// Residuals by v_dot2c_f32_bf16 (one instruction instead of expand + subtract) measured +4...6 % slower and not bit-identical)
__device__ __forceinline__ float sub_scalar(float a, float b) {
  float r;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// two fp32 values -> three packed bf16 pairs with x = hi + mid + lo exactly
template <bool NOPK = false>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  float r0, r1;
  if constexpr (NOPK) {
    r0 = sub_scalar(x0, __uint_as_float(hi << 16));
    r1 = sub_scalar(x1, __uint_as_float(hi & 0xffff0000u));
  } else {
    r0 = x0 - __uint_as_float(hi << 16);
    r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  }
  mid = cvt_pk_bf16(r0, r1);
  if constexpr (NOPK) {
    r0 = sub_scalar(r0, __uint_as_float(mid << 16));
    r1 = sub_scalar(r1, __uint_as_float(mid & 0xffff0000u));
  } else {
    r0 -= __uint_as_float(mid << 16);
    r1 -= __uint_as_float(mid & 0xffff0000u);
  }
  lo = cvt_pk_bf16(r0, r1);
}

__device__ __forceinline__ u32x4 buf_load4u(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}

// Products kept, largest first; TERMS = 6 is the fp32-equivalent set (i + j <= 2), 8 adds the 2^-24 terms,
// 3 is the "bf16x3" set (relative error ~2^-16 per product: NOT fp32-equivalent, kept for measurements only).
__device__ constexpr int TERM_A[8] = {0, 0, 1, 1, 0, 2, 1, 2};
__device__ constexpr int TERM_B[8] = {0, 1, 0, 1, 2, 0, 2, 1};

