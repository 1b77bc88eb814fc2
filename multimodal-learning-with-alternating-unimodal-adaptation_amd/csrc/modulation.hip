// OGM / OGM-GE gradient modulation (the --modulation OGM | OGM_GE surface of main.py:312-410).
//
//   coefficients  main.py:373-384 (two modalities) / 314-337 (three): score_m = sum_i softmax(out_m)[i][label_i];
//                 ratio_v = score_v / score_a (three: score_m / sum of the others); the dominant modality gets
//                 coeff = 1 - tanh(alpha * relu(ratio)), the others 1.
//   modulation    main.py:394-408 / 347-369: for every 4-D (conv) gradient of the encoder of modality m:
//                 OGM:     grad *= coeff_m
//                 OGM_GE:  grad = grad * coeff_m + N(0, grad.std() + 1e-8)        (std: unbiased, of the UNSCALED gradient)
//
// The reference walks named_parameters() and issues ~4 ATen kernels + one .item() host sync per tensor (20 per ResNet
// encoder).  Here the coefficients stay on the device and one launch pair handles every conv gradient of an encoder in its
// flat gradient buffer: a deterministic two-level fp64 reduction for the per-tensor statistics, then scale (+ noise) with a
// counter-based generator (Philox4x32-10 + Box-Muller: element i of the buffer always draws from counter i / 4, so the
// noise does not depend on the launch geometry).
#include "common.h"

namespace {

#define OGM_MAXM 3
struct OgmArgs {
  const float* out[OGM_MAXM];
  int M, B, C;
  float alpha;
};

// one workgroup; thread m < M accumulates score_m over the rows IN ROW ORDER (the reference's python sum())
__global__ __launch_bounds__(64) void ogm_coeff_kernel(const OgmArgs a, const int64_t* __restrict__ labels, float* __restrict__ coeff,
                                                        float* __restrict__ info) {
  __shared__ float score[OGM_MAXM];
  const int m = threadIdx.x;
  if (m < a.M) {
    float s = 0.f;
    for (int r = 0; r < a.B; ++r) {
      const float* l = a.out[m] + (size_t)r * a.C;
      float mx = -INFINITY;
      for (int c = 0; c < a.C; ++c) mx = fmaxf(mx, l[c]);
      float den = 0.f;
      for (int c = 0; c < a.C; ++c) den += expf(l[c] - mx);
      const long lab = (long)labels[r];
      s += (lab >= 0 && lab < a.C) ? expf(l[lab] - mx) / den : NAN;      // bad label: poison (the reference would raise)
    }
    score[m] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ratio[OGM_MAXM], cf[OGM_MAXM] = {1.f, 1.f, 1.f};
    if (a.M == 2) {                                          // index 0 = audio, 1 = visual (main.py:376-384)
      ratio[1] = score[1] / score[0];
      ratio[0] = 1.f / ratio[1];
      if (ratio[1] > 1.f) cf[1] = 1.f - tanhf(a.alpha * fmaxf(ratio[1], 0.f));
      else cf[0] = 1.f - tanhf(a.alpha * fmaxf(ratio[0], 0.f));
    } else {                                                 // index 0 = audio, 1 = visual, 2 = text (main.py:319-337)
      ratio[0] = score[0] / (score[1] + score[2]);
      ratio[1] = score[1] / (score[0] + score[2]);
      ratio[2] = score[2] / (score[1] + score[0]);
      if (ratio[1] > 1.f) cf[1] = 1.f - tanhf(a.alpha * fmaxf(ratio[1], 0.f));
      else if (ratio[2] > 1.f) cf[2] = 1.f - tanhf(a.alpha * fmaxf(ratio[2], 0.f));
      else cf[0] = 1.f - tanhf(a.alpha * fmaxf(ratio[0], 0.f));
    }
    for (int m2 = 0; m2 < a.M; ++m2) {
      coeff[m2] = cf[m2];
      if (info) {
        info[m2] = score[m2];
        info[OGM_MAXM + m2] = ratio[m2];
      }
    }
  }
}

// ---- per-segment statistics: partial (sum, sum of squares) in fp64 -------------------------------------------------------
#define OGM_CHUNK 16384       // elements per partial
__global__ __launch_bounds__(256) void ogm_stats_partial_kernel(const float* __restrict__ g, const int64_t* __restrict__ seg,
                                                                 const int* __restrict__ first_chunk, int n_seg,
                                                                 double* __restrict__ partial) {
  // chunk id -> segment by scanning the (short) table
  const int chunk = blockIdx.x;
  int s = 0;
  while (s + 1 < n_seg && first_chunk[s + 1] <= chunk) ++s;
  const int64_t off = seg[2 * s], n = seg[2 * s + 1];
  const int64_t lo = (int64_t)(chunk - first_chunk[s]) * OGM_CHUNK, hi = lo + OGM_CHUNK < n ? lo + OGM_CHUNK : n;
  double sum = 0.0, sq = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)g[off + i];
    sum += v;
    sq += v * v;
  }
  __shared__ double red[2][4];
  sum = wave_sum_d(sum);
  sq = wave_sum_d(sq);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sum;
    red[1][threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * (size_t)chunk] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * (size_t)chunk + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// one wave per segment: ordered sum of its partials -> unbiased std (torch.Tensor.std default)
__global__ __launch_bounds__(64) void ogm_stats_final_kernel(const double* __restrict__ partial, const int64_t* __restrict__ seg,
                                                              const int* __restrict__ first_chunk, int n_seg, float* __restrict__ stdv) {
  const int s = blockIdx.x;
  const int c0 = first_chunk[s], c1 = first_chunk[s + 1];
  double sum = 0.0, sq = 0.0;
  for (int c = c0 + threadIdx.x; c < c1; c += 64) {
    sum += partial[2 * (size_t)c];
    sq += partial[2 * (size_t)c + 1];
  }
  sum = wave_sum_d(sum);
  sq = wave_sum_d(sq);
  if (threadIdx.x == 0) {
    const double n = (double)seg[2 * s + 1];
    const double var = n > 1.0 ? (sq - sum * sum / n) / (n - 1.0) : NAN;        // torch: std of one element is nan
    stdv[s] = (float)sqrt(var > 0.0 ? var : (var == var ? 0.0 : NAN));
  }
}

// ---- Philox4x32-10 ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}
__device__ __forceinline__ void philox4x32(uint64_t counter, uint64_t stream_id, uint64_t seed, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

// grad[off + i] = grad[off + i] * coeff (+ std_s * z_i): one thread per 4 consecutive elements of a segment
__global__ __launch_bounds__(256) void ogm_apply_kernel(float* __restrict__ g, const int64_t* __restrict__ seg,
                                                         const int* __restrict__ first_chunk, int n_seg, const float* __restrict__ coeff,
                                                         const float* __restrict__ stdv, int ge, uint64_t seed, uint64_t step) {
  const int chunk = blockIdx.x;
  int s = 0;
  while (s + 1 < n_seg && first_chunk[s + 1] <= chunk) ++s;
  const int64_t off = seg[2 * s], n = seg[2 * s + 1];
  const int64_t lo = (int64_t)(chunk - first_chunk[s]) * OGM_CHUNK, hi = lo + OGM_CHUNK < n ? lo + OGM_CHUNK : n;
  const float cf = *coeff;
  const float sd = ge ? stdv[s] + 1e-8f : 0.f;                                     // main.py:399: std().item() + 1e-8
  for (int64_t i4 = lo / 4 + threadIdx.x; i4 * 4 < hi; i4 += 256) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (ge) {
      uint32_t r[4];
      philox4x32((uint64_t)i4, ((uint64_t)step << 16) | (uint64_t)s, seed, r);     // stream = (call counter, tensor)
      const float r0 = sqrtf(-2.f * logf(u01(r[0]))), r1 = sqrtf(-2.f * logf(u01(r[2])));
      float s0, c0, s1, c1;
      sincosf(6.28318530717958647692f * u01(r[1]), &s0, &c0);
      sincosf(6.28318530717958647692f * u01(r[3]), &s1, &c1);
      z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i4 * 4 + k;
      if (i >= lo && i < hi) g[off + i] = g[off + i] * cf + sd * z[k];
    }
  }
}

}  // namespace

extern "C" int mla_ogm_coeff(const float* out0, const float* out1, const float* out2, const int64_t* labels, int M, int B, int C,
                             float alpha, float* coeff, float* info, void* stream) {
  MLA_REQUIRE(out0 && out1 && labels && coeff && (M == 2 || (M == 3 && out2)) && B > 0 && C > 0, "mla_ogm_coeff: bad argument");
  OgmArgs a;
  a.out[0] = out0; a.out[1] = out1; a.out[2] = out2;
  a.M = M; a.B = B; a.C = C; a.alpha = alpha;
  ogm_coeff_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, labels, coeff, info);
  MLA_CHECK_LAUNCH("ogm_coeff_kernel");
  return MLA_OK;
}

extern "C" size_t mla_ogm_ws_bytes(int total_chunks, int n_seg) { return (size_t)total_chunks * 2 * sizeof(double) + (size_t)n_seg * sizeof(float) + 16; }
extern "C" int mla_ogm_chunk_elems(void) { return OGM_CHUNK; }

extern "C" int mla_ogm_modulate(float* grad, const int64_t* seg_desc, const int* first_chunk, int n_seg, int total_chunks,
                                const float* coeff, int ge, uint64_t seed, uint64_t step, void* ws, size_t ws_bytes, void* stream) {
  MLA_REQUIRE(grad && seg_desc && first_chunk && coeff && n_seg > 0 && total_chunks > 0, "mla_ogm_modulate: bad argument");
  MLA_REQUIRE(!ge || (ws && ws_bytes >= mla_ogm_ws_bytes(total_chunks, n_seg)), "mla_ogm_modulate: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  float* stdv = ge ? (float*)(partial + 2 * (size_t)total_chunks) : nullptr;
  if (ge) {
    ogm_stats_partial_kernel<<<total_chunks, 256, 0, st>>>(grad, seg_desc, first_chunk, n_seg, partial);
    MLA_CHECK_LAUNCH("ogm_stats_partial_kernel");
    ogm_stats_final_kernel<<<n_seg, 64, 0, st>>>(partial, seg_desc, first_chunk, n_seg, stdv);
    MLA_CHECK_LAUNCH("ogm_stats_final_kernel");
  }
  ogm_apply_kernel<<<total_chunks, 256, 0, st>>>(grad, seg_desc, first_chunk, n_seg, coeff, stdv, ge, seed, step);
  MLA_CHECK_LAUNCH("ogm_apply_kernel");
  return MLA_OK;
}
