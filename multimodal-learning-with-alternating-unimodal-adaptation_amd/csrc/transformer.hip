// Transformer-encoder kernels for the M3AE / CAV-MAE modality encoders (fp32, gfx950).
//
//   LayerNorm fwd/bwd             nn.LayerNorm (models/m3ae.py:138, 142, 176; eps 1e-5)
//   attention                     Attention.forward (models/m3ae.py:102-125): QK^T*scale, where(mask>0,-1e7),
//                                 softmax, PV -- strided batched GEMMs on v_mfma_f32_32x32x2_f32 that read q/k/v
//                                 straight out of the (B,n,3,H,64) qkv buffer and write (B,n,D), plus a
//                                 wave-per-row masked softmax.  The probabilities are materialised (288 GB HBM:
//                                 2.4 GB per encoder at B=64) and reused by the backward instead of recomputed.
//   token assembly / embedding    MaskedMultimodalAutoencoder.forward_representation (models/m3ae.py:342-370):
//                                 [cls] + (patch-linear | text-embedding) + sin-cos position + type embedding
//   patchify                      einops 'b c (h p1) (w p2) -> b (h w) (c p1 p2)' (models/basic_model.py:184-186)
//   column sums                   bias gradients of the Linear layers
// The Linear layers themselves run on the gather-GEMM of conv_igemm.hip (mla_linear_*).
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, D <= 2048, D % 64 == 0
// ---------------------------------------------------------------------------------------------------
template <int PER>   // elements per lane = D / 64 (compile-time so the row lives in registers, not scratch)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                      float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    v[k] = x[(size_t)row * D + k * 64 + lane];
    s += v[k];
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const float d = v[k] - mu;
    q += d * d;
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = k * 64 + lane;
    y[(size_t)row * D + c] = (v[k] - mu) * rs * w[c] + b[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;  dx += add (residual-branch gradient).
// A workgroup owns LN_TR = 16 rows (4 per wave) and also forms the column sums of the affine gradients over them --
// sum dy (bias) and sum dy * xhat (weight) -- from the values it holds anyway: partial[tile][2][D], finalized by
// colreduce_finalize_kernel.  (A separate column-reduction pass re-read dy and x: 19 us per LayerNorm at 16448 x 768.)
#define LN_TR 16
template <int PER>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* add, float* dx,
                                                         float* __restrict__ partial, int M, int D) {
  __shared__ float red[2][4][PER * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float cb[PER], cw[PER], wv[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    cb[k] = cw[k] = 0.f;
    wv[k] = w[k * 64 + lane];
  }
  for (int rr = 0; rr < LN_TR / 4; ++rr) {
    const int row = blockIdx.x * LN_TR + rr * 4 + wave;       // wave-uniform
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    float g[PER], xh[PER];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const size_t i = (size_t)row * D + k * 64 + lane;
      const float dv = dy[i];
      g[k] = dv * wv[k];
      xh[k] = (x[i] - mu) * rs;
      s1 += g[k];
      s2 += g[k] * xh[k];
      cb[k] += dv;
      cw[k] += dv * xh[k];
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const size_t i = (size_t)row * D + k * 64 + lane;
      float r = rs * (g[k] - s1 - xh[k] * s2);
      if (add) r += add[i];
      dx[i] = r;
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    red[0][wave][k * 64 + lane] = cb[k];
    red[1][wave][k * 64 + lane] = cw[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    partial[((size_t)blockIdx.x * 2 + 0) * D + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    partial[((size_t)blockIdx.x * 2 + 1) * D + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

// Column reductions over rows.  MODE 0: s0 = sum x.  MODE 1 (LayerNorm affine grads): s0 = sum dy, s1 = sum dy*xhat.
// grid (C/64, row tiles); 256 threads = 16 float4 column groups x 16 row lanes.
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ partial, int M, int C, int tile_rows) {
  __shared__ f32x4 red[2][256];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c4 = blockIdx.x * 16 + cg, c4n = C >> 2;
  const int r0 = blockIdx.y * tile_rows, r1 = min(M, r0 + tile_rows);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rl; r < r1; r += 16) {
    const f32x4 av = reinterpret_cast<const f32x4*>(a)[(size_t)r * c4n + c4];
    s0 += av;
    if (MODE == 1) {
      const f32x4 xv = reinterpret_cast<const f32x4*>(x)[(size_t)r * c4n + c4];
      s1 += av * ((xv - mean[r]) * rstd[r]);
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < 16; ++k) {
      s0 += red[0][k * 16 + cg];
      s1 += red[1][k * 16 + cg];
    }
    reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 0) * C)[c4] = s0;
    reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 1) * C)[c4] = s1;
  }
}

// out0[c] = sum_t partial[t][0][c] (fp64), out1[c] = sum_t partial[t][1][c]; one wave per column
__global__ __launch_bounds__(256) void colreduce_finalize_kernel(const float* __restrict__ partial, int tiles, int C,
                                                                  float* __restrict__ out0, float* __restrict__ out1) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int t = lane; t < tiles; t += 64) {
    s += (double)partial[((size_t)t * 2 + 0) * C + c];
    q += (double)partial[((size_t)t * 2 + 1) * C + c];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (lane == 0) {
    out0[c] = (float)s;
    if (out1) out1[c] = (float)q;
  }
}

static int colreduce_tile_rows(int M) {
  long t = ((long)M + 255) / 256;
  t = ((t + 15) / 16) * 16;
  if (t < 64) t = 64;
  return (int)t;
}
extern "C" size_t mla_colreduce_ws_elems(int M, int C) {   // column sums: row tiles of colreduce_tile_rows; LayerNorm backward: tiles of 16 rows
  const size_t a = (size_t)cdiv(M, colreduce_tile_rows(M)) * 2 * C, b = (size_t)cdiv(M, 16) * 2 * C;
  return a > b ? a : b;
}

extern "C" int mla_colsum_rows(const float* x, float* out, float* ws, int M, int C, void* stream) {
  MLA_REQUIRE(x && out && ws && M > 0 && C > 0 && C % 64 == 0, "mla_colsum_rows: bad argument (C %% 64 == 0)");
  hipStream_t st = (hipStream_t)stream;
  const int tr = colreduce_tile_rows(M), nt = cdiv(M, tr);
  colreduce_kernel<0><<<dim3(C / 64, nt), 256, 0, st>>>(x, nullptr, nullptr, nullptr, ws, M, C, tr);
  MLA_CHECK_LAUNCH("colreduce_kernel<0>");
  colreduce_finalize_kernel<<<cdiv(C, 4), 256, 0, st>>>(ws, nt, C, out, nullptr);
  MLA_CHECK_LAUNCH("colreduce_finalize_kernel");
  return MLA_OK;
}

extern "C" int mla_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                                 int M, int D, float eps, void* stream) {
  MLA_REQUIRE(x && w && b && y && mean && rstd && M > 0, "mla_layernorm_fwd: bad argument");
  MLA_REQUIRE(D == 512 || D == 768 || D == 1024, "mla_layernorm_fwd: D=%d unsupported (512, 768, 1024)", D);
  hipStream_t st = (hipStream_t)stream;
  if (D == 512) ln_fwd_kernel<8><<<cdiv(M, 4), 256, 0, st>>>(x, w, b, y, mean, rstd, M, D, eps);
  else if (D == 768) ln_fwd_kernel<12><<<cdiv(M, 4), 256, 0, st>>>(x, w, b, y, mean, rstd, M, D, eps);
  else ln_fwd_kernel<16><<<cdiv(M, 4), 256, 0, st>>>(x, w, b, y, mean, rstd, M, D, eps);
  MLA_CHECK_LAUNCH("ln_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                                 const float* add, float* dx, float* dw, float* db, float* ws, int M, int D, void* stream) {
  MLA_REQUIRE(dy && x && w && mean && rstd && dx && dw && db && ws && M > 0, "mla_layernorm_bwd: bad argument");
  MLA_REQUIRE(D == 512 || D == 768 || D == 1024, "mla_layernorm_bwd: D=%d unsupported (512, 768, 1024)", D);
  hipStream_t st = (hipStream_t)stream;
  const int nt = cdiv(M, LN_TR);
  // dx may alias dy or add (each row is fully read into registers before it is written)
  if (D == 512) ln_bwd_dx_kernel<8><<<nt, 256, 0, st>>>(dy, x, w, mean, rstd, add, dx, ws, M, D);
  else if (D == 768) ln_bwd_dx_kernel<12><<<nt, 256, 0, st>>>(dy, x, w, mean, rstd, add, dx, ws, M, D);
  else ln_bwd_dx_kernel<16><<<nt, 256, 0, st>>>(dy, x, w, mean, rstd, add, dx, ws, M, D);
  MLA_CHECK_LAUNCH("ln_bwd_dx_kernel");
  colreduce_finalize_kernel<<<cdiv(D, 4), 256, 0, st>>>(ws, nt, D, db, dw);
  MLA_CHECK_LAUNCH("colreduce_finalize_kernel");
  return MLA_OK;
}

// ---------------------------------------------------------------------------------------------------
// Strided batched GEMM (attention): C[z][i][j] = alpha * sum_k A[z][i][k] * B[z][k][j], z = (b, h).
// 64x64 tile, 4 waves x one 32x32 MFMA tile, K step 32 through LDS; arbitrary element strides, so q/k/v
// are read in place from the (B,n,3,H,hd) qkv buffer and results land directly in their final layout.
// ---------------------------------------------------------------------------------------------------
struct BGemmDesc {
  int M, N, K, H;                       // per-batch dims, heads per batch (z = b*H + h)
  long a_b, a_h, a_i, a_k;              // element strides of A: batch, head, row i, reduction k
  long b_b, b_h, b_k, b_j;
  long c_b, c_h, c_i, c_j;
  float alpha;
};

// Thread -> element maps of the staging loads follow the contiguous dimension of each operand so a wave reads whole
// 128/256-B pieces whatever the layout: AK = A is k-contiguous (a_k == 1: rows of Q / P / dS), else lanes run along the
// rows (a_i == 1: the transposed operands P^T, dS^T); BJ = B is column-contiguous (b_j == 1: V, K, Q, dO as [K][N]),
// else lanes run along k (b_k == 1: K^T, V^T).  The next K step's loads are issued before the MFMAs of the current one.
template <bool AK, bool BJ>
__global__ __launch_bounds__(256) void bgemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, const BGemmDesc d) {
  constexpr int LDB = 65;   // odd stride: conflict-free for both store maps and for the MFMA operand reads
  __shared__ __attribute__((aligned(16))) float As[64 * 36];
  __shared__ float Bs[32 * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z, b = z / d.H, h = z - b * d.H;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const float* Ab = A + b * d.a_b + h * d.a_h;
  const float* Bb = B + b * d.b_b + h * d.b_h;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // A tile element p of this thread: AK: (r = (tid >> 5) + 8 p, kk = tid & 31); else (r = tid & 63, kk = (tid >> 6) + 4 p)
  // B tile element p:                BJ: (kb = (tid >> 6) + 4 p, c = tid & 63);  else (kb = tid & 31, c = (tid >> 5) + 8 p)
  float av[8], bv[8];
  auto load = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int r = i0 + (AK ? (tid >> 5) + 8 * p : (tid & 63)), kk = k0 + (AK ? (tid & 31) : (tid >> 6) + 4 * p);
      av[p] = (r < d.M && kk < d.K) ? Ab[r * d.a_i + kk * d.a_k] : 0.f;
      const int kb = k0 + (BJ ? (tid >> 6) + 4 * p : (tid & 31)), c = j0 + (BJ ? (tid & 63) : (tid >> 5) + 8 * p);
      bv[p] = (kb < d.K && c < d.N) ? Bb[kb * d.b_k + c * d.b_j] : 0.f;
    }
  };
  load(0);
  for (int k0 = 0; k0 < d.K; k0 += 32) {
    __syncthreads();            // the previous step's MFMA operand reads are done
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      As[(AK ? (tid >> 5) + 8 * p : (tid & 63)) * 36 + (AK ? (tid & 31) : (tid >> 6) + 4 * p)] = av[p];
      Bs[(BJ ? (tid >> 6) + 4 * p : (tid & 31)) * LDB + (BJ ? (tid & 63) : (tid >> 5) + 8 * p)] = bv[p];
    }
    __syncthreads();
    if (k0 + 32 < d.K) load(k0 + 32);
    const int i = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + i) * 36 + kk * 8 + 4 * hh]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj], Bs[(kk * 8 + 4 * hh + jj) * LDB + wn * 32 + i], acc, 0, 0, 0);
    }
  }
  float* Cb = C + b * d.c_b + h * d.c_h;
  const int hh = lane >> 5, j = j0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = i0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (i < d.M && j < d.N) Cb[i * d.c_i + j * d.c_j] = acc[e] * d.alpha;
  }
}

extern "C" int mla_bgemm(const float* A, const float* B, float* C, int batches, int heads, int M, int N, int K,
                         const long* a_strides, const long* b_strides, const long* c_strides, size_t a_extent, size_t b_extent,
                         size_t c_extent, float alpha, void* stream) {
  MLA_REQUIRE(A && B && C && a_strides && b_strides && c_strides, "mla_bgemm: null pointer");
  MLA_REQUIRE(batches > 0 && heads > 0 && M > 0 && N > 0 && K > 0 && (long)batches * heads < 65536, "mla_bgemm: bad dims");
  // The descriptor is all the kernel knows about the buffers: check that the largest element it can address lies inside the
  // extent (in elements, counted from the pointer) the caller vouches for, so a wrong stride table is an error code here
  // and not a memory fault on the device.
  const long dims[3][4] = {{batches, heads, M, K}, {batches, heads, K, N}, {batches, heads, M, N}};
  const long* strides[3] = {a_strides, b_strides, c_strides};
  const size_t extents[3] = {a_extent, b_extent, c_extent};
  for (int o = 0; o < 3; ++o) {
    unsigned long long last = 0;
    for (int k = 0; k < 4; ++k) {
      MLA_REQUIRE(strides[o][k] >= 0, "mla_bgemm: negative stride (operand %d, axis %d)", o, k);
      last += (unsigned long long)strides[o][k] * (unsigned long long)(dims[o][k] - 1);
    }
    MLA_REQUIRE(last < extents[o], "mla_bgemm: operand %d reaches element %llu but its extent is %zu", o, last, extents[o]);
  }
  BGemmDesc d;
  d.M = M; d.N = N; d.K = K; d.H = heads; d.alpha = alpha;
  d.a_b = a_strides[0]; d.a_h = a_strides[1]; d.a_i = a_strides[2]; d.a_k = a_strides[3];
  d.b_b = b_strides[0]; d.b_h = b_strides[1]; d.b_k = b_strides[2]; d.b_j = b_strides[3];
  d.c_b = c_strides[0]; d.c_h = c_strides[1]; d.c_i = c_strides[2]; d.c_j = c_strides[3];
  const dim3 grid(cdiv(N, 64), cdiv(M, 64), batches * heads);
  hipStream_t st = (hipStream_t)stream;
  const bool ak = d.a_k == 1 || d.a_i != 1, bj = d.b_j == 1 || d.b_k != 1;   // generic strides use the default maps
  if (ak && bj) bgemm_kernel<true, true><<<grid, 256, 0, st>>>(A, B, C, d);
  else if (ak) bgemm_kernel<true, false><<<grid, 256, 0, st>>>(A, B, C, d);
  else if (bj) bgemm_kernel<false, true><<<grid, 256, 0, st>>>(A, B, C, d);
  else bgemm_kernel<false, false><<<grid, 256, 0, st>>>(A, B, C, d);
  MLA_CHECK_LAUNCH("bgemm_kernel");
  return MLA_OK;
}

// ---------------------------------------------------------------------------------------------------
// masked softmax over the last axis of S (B, H, n, n), in place; wave per row; n <= 1024
// mask: pm (B, n) float, pm > 0 -> score replaced by -1e7 (models/m3ae.py:111-117)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ pm, int rows,
                                                           int n, int rows_per_batch) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* s = S + (size_t)row * n;
  const float* m = pm ? pm + (size_t)(row / rows_per_batch) * n : nullptr;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int j = k * 64 + lane;
    float t = -INFINITY;
    if (j < n) {
      t = s[j];
      if (m && m[j] > 0.f) t = -1e7f;
    }
    v[k] = t;
    mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = (k * 64 + lane < n) ? expf(v[k] - mx) : 0.f;
    sum += v[k];
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k * 64 + lane < n) s[k * 64 + lane] = v[k] * inv;
}

// dS = P * (dP - sum_j dP*P), in place in dP
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int rows, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = P + (size_t)row * n;
  float* g = dP + (size_t)row * n;
  float pv[16], gv[16];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int j = k * 64 + lane;
    pv[k] = j < n ? p[j] : 0.f;
    gv[k] = j < n ? g[j] : 0.f;
    dot += pv[k] * gv[k];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k * 64 + lane < n) g[k * 64 + lane] = pv[k] * (gv[k] - dot);
}

extern "C" int mla_softmax_fwd(float* S, const float* pad_mask, int B, int H, int n, void* stream) {
  MLA_REQUIRE(S && B > 0 && H > 0 && n > 0 && n <= 1024, "mla_softmax_fwd: bad argument (n <= 1024)");
  const long rows = (long)B * H * n;
  MLA_REQUIRE(rows < (1L << 31), "mla_softmax_fwd: too many rows");
  softmax_fwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(S, pad_mask, (int)rows, n, H * n);
  MLA_CHECK_LAUNCH("softmax_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_softmax_bwd(const float* P, float* dP, int B, int H, int n, void* stream) {
  MLA_REQUIRE(P && dP && B > 0 && H > 0 && n > 0 && n <= 1024, "mla_softmax_bwd: bad argument (n <= 1024)");
  const long rows = (long)B * H * n;
  softmax_bwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(P, dP, (int)rows, n);
  MLA_CHECK_LAUNCH("softmax_bwd_kernel");
  return MLA_OK;
}

// ---------------------------------------------------------------------------------------------------
// token assembly:  x0[b][0] = cls;  x0[b][1+i] = (table[ids[b][i]] | x0[b][1+i] (already holds the patch Linear))
//                                                 + pos[i] + type
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_kernel(float* __restrict__ x0, const float* __restrict__ table,
                                                        const int64_t* __restrict__ ids, const float* __restrict__ pos,
                                                        const float* __restrict__ type, const float* __restrict__ cls,
                                                        int B, int L, int D, int V) {
  const int d4n = D >> 2;
  const int hc = cls ? 1 : 0;                       // CAV-MAE has no [cls] token (cav_mae.py:337-343)
  const size_t total = (size_t)B * (L + hc) * d4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d4n);
    const size_t r = i / d4n;
    const int t = (int)(r % (L + hc)) + (1 - hc);   // t == 0 only for the cls row
    const size_t b = r / (L + hc);
    f32x4 v;
    if (t == 0) {
      v = reinterpret_cast<const f32x4*>(cls)[c];
    } else {
      if (table) {
        const int64_t id = ids[b * L + t - 1];     // nn.Embedding asserts on an id outside [0, V): poison the row instead of
        v = (id >= 0 && id < V) ? reinterpret_cast<const f32x4*>(table)[(size_t)id * d4n + c]      // reading out of bounds
                                : f32x4{NAN, NAN, NAN, NAN};
      } else {
        v = reinterpret_cast<const f32x4*>(x0)[i];
      }
      v += reinterpret_cast<const f32x4*>(pos)[(size_t)(t - 1) * d4n + c] + reinterpret_cast<const f32x4*>(type)[c];
    }
    reinterpret_cast<f32x4*>(x0)[i] = v;
  }
}

// Gradients of the assembly: dcls[d] = sum_b dx0[b][0][d]; dtype[d] = sum_{b,t>=1} dx0[b][t][d] (given the
// all-row column sum `tot`: dtype = tot - dcls).
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx0, const float* __restrict__ tot,
                                                            float* __restrict__ dcls, float* __restrict__ dtype, int B, int L, int D) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < (size_t)D) {
    float s = 0.f;
    if (dcls) {
      for (int b = 0; b < B; ++b) s += dx0[(size_t)b * (L + 1) * D + gid];
      dcls[gid] = s;
    }
    dtype[gid] = tot[gid] - s;
  }
}

// ---------------------------------------------------------------------------------------------------
// Embedding gradient dtable[ids[r]] += dx0 row of token r (nn.Embedding backward, m3ae.py:306, 360), DETERMINISTIC: the rows
// of one id are summed in ascending token order whatever the launch geometry, so two runs agree bit for bit (round 2 used
// fp32 atomics: the one order-dependent sum of the library, and 14x slower on the [PAD] id that half of a padded batch
// shares -- MI355X_MICROARCH.md, Global float atomics, contention row).  Per chunk of <= EMB_CHUNK tokens:
//   emb_sort_kernel     one workgroup: keys id * NPOS + r (u32) bitonic-sorted in LDS -> sorted (id, r) pairs; ids outside
//                       [0, V) sort to the end as 0xFFFFFFFF and are skipped (the forward poisoned their rows with NaN)
//   emb_segment_kernel  one wave per block of 32 sorted positions: runs of equal id are summed in order; a run that is a
//                       whole segment goes straight to dtable, a run cut by the block boundary into part[block][0 | 1]
//   emb_combine_kernel  one wave per block whose LAST run starts a segment that continues to the right: adds the partials
//                       of the following blocks in order (a [PAD] segment of 8192 tokens = 256 partial rows), then dtable
// Chunks (B * L > EMB_CHUNK) are processed one after the other on the stream and accumulate into dtable in chunk order.
// ---------------------------------------------------------------------------------------------------
#define EMB_CHUNK 16384
#define EMB_BLK 32
#define EMB_MAXD4 4          // float4 chunks per lane: D <= 1024

__global__ __launch_bounds__(1024) void emb_sort_kernel(const int64_t* __restrict__ ids, int n, int npos, int V,
                                                         unsigned* __restrict__ sorted) {
  __shared__ unsigned keys[EMB_CHUNK];
  for (int r = threadIdx.x; r < npos; r += 1024) {
    unsigned k = 0xFFFFFFFFu;
    if (r < n) {
      const int64_t id = ids[r];
      if (id >= 0 && id < V) k = (unsigned)id * (unsigned)npos + (unsigned)r;
    }
    keys[r] = k;
  }
  __syncthreads();
  for (int size = 2; size <= npos; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npos >> 1); t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));          // index with the `stride` bit clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  for (int r = threadIdx.x; r < npos; r += 1024) sorted[r] = keys[r];
}

__device__ __forceinline__ unsigned emb_id(unsigned key, int sh) { return key == 0xFFFFFFFFu ? 0xFFFFFFFFu : key >> sh; }

// dx0 row of chunk-local token r: tokens are numbered b * L + t over the chunk's token range starting at r0
__device__ __forceinline__ const f32x4* emb_row(const float* __restrict__ dx0, unsigned r, int L, int D) {
  const unsigned b = r / (unsigned)L, t = r - b * (unsigned)L;
  return reinterpret_cast<const f32x4*>(dx0 + ((size_t)b * (L + 1) + t + 1) * D);
}

__global__ __launch_bounds__(256) void emb_segment_kernel(const float* __restrict__ dx0, const unsigned* __restrict__ sorted,
                                                           int npos, int sh, int r0, int L, int D, float* __restrict__ dtable,
                                                           float* __restrict__ part) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = w * EMB_BLK;
  if (p0 >= npos) return;
  const int d4n = D >> 2;
  const unsigned mykey = lane < EMB_BLK ? sorted[p0 + lane] : 0xFFFFFFFFu;
  const unsigned left = p0 > 0 ? emb_id(sorted[p0 - 1], sh) : 0xFFFFFFFEu;                 // 0xFFFFFFFE: no such id
  const unsigned right = p0 + EMB_BLK < npos ? emb_id(sorted[p0 + EMB_BLK], sh) : 0xFFFFFFFEu;
  f32x4 acc[EMB_MAXD4];
  unsigned cur = 0xFFFFFFFFu;
  int run_first = 0;                                                                       // position (0..31) the current run began at
  auto flush = [&](int end) {                                                              // run [run_first, end) of id `cur`
    if (cur == 0xFFFFFFFFu) return;
    const bool open_l = run_first == 0 && left == cur, open_r = end == EMB_BLK && right == cur;
    float* dst = (open_l || open_r) ? part + ((size_t)w * 2 + (run_first == 0 ? 0 : 1)) * D : dtable + (size_t)cur * D;
#pragma unroll
    for (int j = 0; j < EMB_MAXD4; ++j) {
      const int c = lane + 64 * j;
      if (c < d4n) {
        f32x4* q = reinterpret_cast<f32x4*>(dst) + c;
        *q = (open_l || open_r) ? acc[j] : *q + acc[j];
      }
    }
  };
  for (int i = 0; i < EMB_BLK; ++i) {
    const unsigned key = __shfl(mykey, i, 64);
    const unsigned id = emb_id(key, sh);
    if (id != cur || id == 0xFFFFFFFFu) {
      flush(i);
      cur = id;
      run_first = i;
#pragma unroll
      for (int j = 0; j < EMB_MAXD4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (id != 0xFFFFFFFFu) {
      const f32x4* row = emb_row(dx0, (unsigned)r0 + (key & ((1u << sh) - 1u)), L, D);
#pragma unroll
      for (int j = 0; j < EMB_MAXD4; ++j) {
        const int c = lane + 64 * j;
        if (c < d4n) acc[j] += row[c];
      }
    }
  }
  flush(EMB_BLK);
}

__global__ __launch_bounds__(256) void emb_combine_kernel(const unsigned* __restrict__ sorted, int npos, int sh, int D,
                                                           const float* __restrict__ part, float* __restrict__ dtable) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = w * EMB_BLK, nblk = npos / EMB_BLK;
  if (p0 + EMB_BLK >= npos) return;                                     // the last block has nothing to its right
  const unsigned id = emb_id(sorted[p0 + EMB_BLK - 1], sh);             // id of this block's last run
  if (id == 0xFFFFFFFFu || emb_id(sorted[p0 + EMB_BLK], sh) != id) return;      // not open to the right
  const bool single = emb_id(sorted[p0], sh) == id;                     // the block is one run
  if (single && p0 > 0 && emb_id(sorted[p0 - 1], sh) == id) return;     // ... that continues a segment started further left
  const int d4n = D >> 2;
  f32x4 acc[EMB_MAXD4];
  const f32x4* first = reinterpret_cast<const f32x4*>(part + ((size_t)w * 2 + (single ? 0 : 1)) * D);
#pragma unroll
  for (int j = 0; j < EMB_MAXD4; ++j) {
    const int c = lane + 64 * j;
    acc[j] = c < d4n ? first[c] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int w2 = w + 1; w2 < nblk; ++w2) {                               // every following block that begins with this id
    const int q0 = w2 * EMB_BLK;
    const f32x4* pr = reinterpret_cast<const f32x4*>(part + (size_t)w2 * 2 * D);
#pragma unroll
    for (int j = 0; j < EMB_MAXD4; ++j) {
      const int c = lane + 64 * j;
      if (c < d4n) acc[j] += pr[c];
    }
    const bool whole = emb_id(sorted[q0 + EMB_BLK - 1], sh) == id;
    if (!(whole && q0 + EMB_BLK < npos && emb_id(sorted[q0 + EMB_BLK], sh) == id)) break;
  }
  f32x4* dst = reinterpret_cast<f32x4*>(dtable + (size_t)id * D);
#pragma unroll
  for (int j = 0; j < EMB_MAXD4; ++j) {
    const int c = lane + 64 * j;
    if (c < d4n) dst[c] += acc[j];
  }
}

extern "C" int mla_tokens_assemble(float* x0, const float* table, const int64_t* ids, const float* pos, const float* type,
                                   const float* cls, int B, int L, int D, int V, void* stream) {
  MLA_REQUIRE(x0 && pos && type && B > 0 && L > 0 && D % 4 == 0 && ((table == nullptr) == (ids == nullptr)) && (!table || V > 0),
              "mla_tokens_assemble: bad argument");
  MLA_REQUIRE(cls || !table, "mla_tokens_assemble: the text path always has a [cls] token");
  size_t blocks = ((size_t)B * (L + 1) * (D / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  assemble_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x0, table, ids, pos, type, cls, B, L, D, V);
  MLA_CHECK_LAUNCH("assemble_kernel");
  return MLA_OK;
}

extern "C" size_t mla_tokens_assemble_bwd_ws_bytes(int B, int L, int D) {
  (void)B; (void)L;
  return (size_t)EMB_CHUNK * sizeof(unsigned) + (size_t)(EMB_CHUNK / EMB_BLK) * 2 * D * sizeof(float);
}

extern "C" int mla_tokens_assemble_bwd(const float* dx0, const float* colsum_all, const int64_t* ids, float* dcls,
                                       float* dtype, float* dtable, int B, int L, int D, int V, void* ws, size_t ws_bytes,
                                       void* stream) {
  MLA_REQUIRE(dx0 && colsum_all && dtype && B > 0 && L > 0 && D > 0 && ((dtable == nullptr) == (ids == nullptr)) && (!dtable || V > 0),
              "mla_tokens_assemble_bwd: bad argument");
  MLA_REQUIRE(dcls || !dtable, "mla_tokens_assemble_bwd: the text path always has a [cls] token");
  hipStream_t st = (hipStream_t)stream;
  assemble_bwd_kernel<<<cdiv(D, 256), 256, 0, st>>>(dx0, colsum_all, dcls, dtype, B, L, D);
  MLA_CHECK_LAUNCH("assemble_bwd_kernel");
  if (!dtable) return MLA_OK;
  MLA_REQUIRE(D % 4 == 0 && D <= 256 * EMB_MAXD4, "mla_tokens_assemble_bwd: D=%d must be a multiple of 4, <= %d", D, 256 * EMB_MAXD4);
  MLA_REQUIRE(L <= EMB_CHUNK && (long)V * EMB_CHUNK <= 0xFFFFFFFFL, "mla_tokens_assemble_bwd: L=%d / V=%d out of range", L, V);
  MLA_REQUIRE(ws && ws_bytes >= mla_tokens_assemble_bwd_ws_bytes(B, L, D), "mla_tokens_assemble_bwd: workspace too small");
  unsigned* sorted = (unsigned*)ws;
  float* part = (float*)(sorted + EMB_CHUNK);
  const int rows_per_chunk = EMB_CHUNK / L;                         // whole sequences per chunk
  for (int b0 = 0; b0 < B; b0 += rows_per_chunk) {
    const int nb = min(rows_per_chunk, B - b0), n = nb * L;
    int npos = 64, sh = 6;
    while (npos < n) { npos <<= 1; ++sh; }
    emb_sort_kernel<<<1, 1024, 0, st>>>(ids + (size_t)b0 * L, n, npos, V, sorted);
    MLA_CHECK_LAUNCH("emb_sort_kernel");
    const float* dx_chunk = dx0 + (size_t)b0 * (L + 1) * D;
    const int waves = npos / EMB_BLK;
    emb_segment_kernel<<<cdiv(waves, 4), 256, 0, st>>>(dx_chunk, sorted, npos, sh, 0, L, D, dtable, part);
    MLA_CHECK_LAUNCH("emb_segment_kernel");
    emb_combine_kernel<<<cdiv(waves, 4), 256, 0, st>>>(sorted, npos, sh, D, part, dtable);
    MLA_CHECK_LAUNCH("emb_combine_kernel");
  }
  return MLA_OK;
}

// einops 'b c (h p1) (w p2) -> b (h w) (c p1 p2)': out[b][h*GW+w][c*P*P + p1*P + p2] = img[b][c][h*P+p1][w*P+p2]
// transposed != 0: the image is stored (B, C, W, H) -- the spectrogram (B, time, freq) that cav_mae.py:339-340 views as
// (B, 1, freq, time) by unsqueeze + transpose, folded into the gather instead of a transposed copy.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C,
                                                        int Himg, int Wimg, int P, int transposed) {
  const int GW = Wimg / P, GH = Himg / P, F = C * P * P;
  const size_t total = (size_t)B * GH * GW * F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    size_t r = i / F;
    const int w = (int)(r % GW); r /= GW;
    const int h = (int)(r % GH);
    const size_t b = r / GH;
    const int p2 = f % P, p1 = (f / P) % P, c = f / (P * P);
    const int y = h * P + p1, x = w * P + p2;
    out[i] = transposed ? img[((b * C + c) * Wimg + x) * Himg + y] : img[((b * C + c) * Himg + y) * Wimg + x];
  }
}

extern "C" int mla_patchify(const float* img, float* out, int B, int C, int H, int W, int P, int transposed, void* stream) {
  MLA_REQUIRE(img && out && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "mla_patchify: bad argument");
  size_t blocks = ((size_t)B * C * H * W + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  patchify_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(img, out, B, C, H, W, P, transposed);
  MLA_CHECK_LAUNCH("patchify_kernel");
  return MLA_OK;
}
