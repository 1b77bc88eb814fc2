// BatchNorm2d (training mode) over pixel-major [M][C] fp32 activations, gfx950.
//
// Replaces the ATen batch_norm forward/backward that nn.BatchNorm2d triggers in the reference
// (models/backbone.py:29, 32, 86, 128; momentum 0.1, eps 1e-5, biased batch variance for the
// normalisation, unbiased for running_var) together with the in-place ReLU and the residual add
// of BasicBlock.forward (backbone.py:41, 49-50), which are fused into the apply pass.
//
// HBM-bound: every pass is float4-vectorised with channels on the fast axis, so a wave reads
// whole 256-B..1-KiB pixel rows; per-channel reductions are thread-private over a row stripe,
// then LDS across row lanes, then fp64 across tiles (mla_bn_finalize) -- no atomics, results are
// bitwise reproducible.  Statistics normally arrive pre-reduced from the conv epilogue
// (conv_igemm.hip), saving one full read of the activation.
#include "common.h"

static size_t bn_red_scratch_floats(int C);
// ---- tiling shared by the stats and backward-reduce kernels --------------------------------
static int bn_tile_rows(int M) {
  // aim at ~2048 workgroups for the big activations (stem, layer1: > 256 K rows), ~1024 below (then the statistics are
  // finalized by ONE launch, bn_finalize_tiles_kernel)
  long t = M > (1 << 18) ? ((long)M + 2047) / 2048 : ((long)M + 1023) / 1024;
  t = ((t + 15) / 16) * 16;
  if (t < 32) t = 32;
  if (t > 1024) t = 1024;
  return (int)t;
}

// Every partial buffer carries a scratch tail for the two-stage reduction (declared in bn.hip, used by the conv too).
extern "C" size_t mla_bn_partial_scratch_elems(int C) { return bn_red_scratch_floats(C); }
extern "C" size_t mla_bn_stats_partial_elems(int M, int C) {
  return (size_t)cdiv(M, bn_tile_rows(M)) * 2 * C * 2 + bn_red_scratch_floats(C);     // forward statistics: fp64 partials
}
extern "C" size_t mla_bn_bwd_ws_elems(int M, int C) { return (size_t)cdiv(M, bn_tile_rows(M)) * 2 * C + bn_red_scratch_floats(C); }

// Reduce two per-channel quantities over a tile of rows.  MODE 0: (x, x^2).  MODE 1: (g, g*xhat).
// Block: 256 threads = (C/4 column groups) x (row lanes); C/4 <= 256 and divides 256.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                         const float* __restrict__ relu_out,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         float* __restrict__ partial, int M, int C, int tile_rows) {
  __shared__ f32x4 red[2][256];
  const int c4n = C >> 2;
  const int cg = threadIdx.x % c4n, rl = threadIdx.x / c4n, nrl = 256 / c4n;
  const int r0 = blockIdx.x * tile_rows, r1 = min(M, r0 + tile_rows);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
  if (MODE == 1) {
    mu = reinterpret_cast<const f32x4*>(mean)[cg];
    is = reinterpret_cast<const f32x4*>(invstd)[cg];
  }
  if (MODE == 0) {   // forward statistics: everything in fp64 (x * x is exact there), see the conv epilogue in igemm_common.h
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = r0 + rl; r < r1; r += nrl) {
      const f32x4 xv = reinterpret_cast<const f32x4*>(x)[(size_t)r * c4n + cg];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double v = (double)xv[e];
        d0[e] += v;
        d1[e] += v * v;
      }
    }
    __shared__ double redd[2][4][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      redd[0][e][threadIdx.x] = d0[e];
      redd[1][e][threadIdx.x] = d1[e];
    }
    __syncthreads();
    if (rl == 0) {
      double* pd = reinterpret_cast<double*>(partial);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double a = d0[e], q = d1[e];
        for (int k = 1; k < nrl; ++k) {
          a += redd[0][e][k * c4n + cg];
          q += redd[1][e][k * c4n + cg];
        }
        pd[((size_t)blockIdx.x * 2 + 0) * C + cg * 4 + e] = a;
        pd[((size_t)blockIdx.x * 2 + 1) * C + cg * 4 + e] = q;
      }
    }
    return;
  }
  for (int r = r0 + rl; r < r1; r += nrl) {
    const size_t idx = (size_t)r * c4n + cg;
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[idx];
    if (MODE == 0) {
      s0 += xv;
      s1 += xv * xv;
    } else {
      f32x4 g = reinterpret_cast<const f32x4*>(dout)[idx];
      if (relu_out) {
        const f32x4 o = reinterpret_cast<const f32x4*>(relu_out)[idx];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
      }
      s0 += g;
      s1 += g * ((xv - mu) * is);
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < nrl; ++k) {
      s0 += red[0][k * c4n + cg];
      s1 += red[1][k * c4n + cg];
    }
    reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * C)[cg] = s0;
    reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * C)[cg] = s1;
  }
}

// Two-stage fp64 reduction of the per-tile partials partial[t][0|1][c] (up to 9408 tiles at the 64x64 conv tile).
// Stage 1: grid (C/64, S <= 64): a block = 64 channels (lanes: coalesced 256-B rows) x 4 waves striding over its chunk
// of >= 64 tiles -> fp64 chunk sums in `scratch[s][0|1][c]`.  Stage 2: one block per 64 channels = 64 channels x 4
// chunk lanes, each lane loads its <= 16 chunk sums at once (no serial load chain) and the lanes are combined through
// LDS in a fixed order.  Deterministic.
#define BN_RED_MAXS 64
static int bn_red_chunks(int tiles) { int s = cdiv(tiles, 64); return s < 1 ? 1 : (s > BN_RED_MAXS ? BN_RED_MAXS : s); }
static size_t bn_red_scratch_floats(int C) { return (size_t)BN_RED_MAXS * 2 * C * 2 + 2; }   // doubles, as floats (+ alignment)

template <typename PT>   // PT = double: forward statistics partials; float: backward (sum g, sum g * xhat) partials
__global__ __launch_bounds__(256) void bn_tiles_stage1_kernel(const PT* __restrict__ partial, int tiles, int C,
                                                               double* __restrict__ scratch) {
  __shared__ double red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int S = gridDim.y, per = (tiles + S - 1) / S;
  const int t0 = blockIdx.y * per, t1 = min(tiles, t0 + per);
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int t = t0 + wave; t < t1; t += 4) {
      s += (double)partial[((size_t)t * 2 + 0) * C + c];
      q += (double)partial[((size_t)t * 2 + 1) * C + c];
    }
  red[0][wave][lane] = s;
  red[1][wave][lane] = q;
  __syncthreads();
  if (wave == 0 && c < C) {
    for (int w = 1; w < 4; ++w) {
      s += red[0][w][lane];
      q += red[1][w][lane];
    }
    scratch[((size_t)blockIdx.y * 2 + 0) * C + c] = s;
    scratch[((size_t)blockIdx.y * 2 + 1) * C + c] = q;
  }
}

// sums of the S chunk sums for channel c = blockIdx.x*64 + lane; valid in wave 0
__device__ __forceinline__ void chunk_sums(const double* __restrict__ scratch, int S, int C, double& s, double& q) {
  __shared__ double red2[2][4][64];
  const int lane = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double sv[BN_RED_MAXS / 4], qv[BN_RED_MAXS / 4];
#pragma unroll
  for (int j = 0; j < BN_RED_MAXS / 4; ++j) {        // all loads of this lane in flight at once
    const int k = kl + 4 * j;
    const bool ok = k < S && c < C;
    sv[j] = ok ? scratch[((size_t)k * 2 + 0) * C + c] : 0.0;
    qv[j] = ok ? scratch[((size_t)k * 2 + 1) * C + c] : 0.0;
  }
  s = 0.0;
  q = 0.0;
#pragma unroll
  for (int j = 0; j < BN_RED_MAXS / 4; ++j) {
    s += sv[j];
    q += qv[j];
  }
  red2[0][kl][lane] = s;
  red2[1][kl][lane] = q;
  __syncthreads();
  if (kl == 0)
    for (int w = 1; w < 4; ++w) {
      s += red2[0][w][lane];
      q += red2[1][w][lane];
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ scratch, int S, int M, int C, float eps,
                                                           float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                           float* running_mean, float* running_var) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, q;
  chunk_sums(scratch, S, C, s, q);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  const double m = s / M;
  double var = q / M - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ scratch, int S, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, q;
  chunk_sums(scratch, S, C, s, q);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

// One-launch finalize for tile counts <= BN_ONE_MAXT (every layer but the stem and layer1 forward): a block owns 16
// channels x 16 tile lanes; lane k sums tiles k, k+16, ... (four loads in flight), the 16 lanes are combined through LDS
// in a fixed order, then the statistics are formed exactly as in bn_finalize_kernel / bn_bwd_finalize_kernel.  Replaces
// the stage-1 + finalize pair (two dependent 5-8 us launches on the serial conv -> statistics -> apply chain) by one.
#define BN_ONE_MAXT 1024
template <typename PT, int FWD>
__global__ __launch_bounds__(256) void bn_finalize_tiles_kernel(const PT* __restrict__ partial, int tiles, int M, int C, float eps,
                                                                 float momentum, float* __restrict__ out0, float* __restrict__ out1,
                                                                 float* running_mean, float* running_var) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    int t = kl;
    for (; t + 48 < tiles; t += 64) {
      const double a0 = (double)partial[((size_t)t * 2 + 0) * C + c], b0 = (double)partial[((size_t)t * 2 + 1) * C + c];
      const double a1 = (double)partial[((size_t)(t + 16) * 2 + 0) * C + c], b1 = (double)partial[((size_t)(t + 16) * 2 + 1) * C + c];
      const double a2 = (double)partial[((size_t)(t + 32) * 2 + 0) * C + c], b2 = (double)partial[((size_t)(t + 32) * 2 + 1) * C + c];
      const double a3 = (double)partial[((size_t)(t + 48) * 2 + 0) * C + c], b3 = (double)partial[((size_t)(t + 48) * 2 + 1) * C + c];
      s += (a0 + a1) + (a2 + a3);
      q += (b0 + b1) + (b2 + b3);
    }
    for (; t < tiles; t += 16) {
      s += (double)partial[((size_t)t * 2 + 0) * C + c];
      q += (double)partial[((size_t)t * 2 + 1) * C + c];
    }
  }
  red[0][kl][cl] = s;
  red[1][kl][cl] = q;
  __syncthreads();
  if (kl != 0 || c >= C) return;
  for (int k = 1; k < 16; ++k) {
    s += red[0][k][cl];
    q += red[1][k][cl];
  }
  if (FWD) {
    const double m = s / M;
    double var = q / M - m * m;
    if (var < 0.0) var = 0.0;
    out0[c] = (float)m;                                           // mean
    out1[c] = (float)(1.0 / sqrt(var + (double)eps));             // invstd
    if (running_mean) {
      const double unb = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  } else {
    out1[c] = (float)s;                                           // dbeta
    out0[c] = (float)q;                                           // dgamma
  }
}

// ... and for MORE than BN_ONE_MAXT tiles (layer1: 2048 / 2352 conv tiles, the pooled stem backward): a block owns 4 channels x 64
// tile lanes (lane k sums tiles k, k + 64, ...: <= 37 values at 2352 tiles, four loads in flight), combined through LDS in a fixed
// order.  One launch instead of the stage-1 + finalize pair of round 2 (two dependent launches on the serial conv ->
// statistics -> apply chain); bn_tiles_stage1_kernel / bn_finalize_kernel stay for tile counts beyond BN_WIDE_MAXT.
#define BN_WIDE_MAXT 16384
template <typename PT, int FWD>
__global__ __launch_bounds__(256) void bn_finalize_tiles_wide_kernel(const PT* __restrict__ partial, int tiles, int M, int C, float eps,
                                                                      float momentum, float* __restrict__ out0, float* __restrict__ out1,
                                                                      float* running_mean, float* running_var) {
  __shared__ double red[2][64][4];
  const int cl = threadIdx.x & 3, kl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    int t = kl;
    for (; t + 192 < tiles; t += 256) {
      const double a0 = (double)partial[((size_t)t * 2 + 0) * C + c], b0 = (double)partial[((size_t)t * 2 + 1) * C + c];
      const double a1 = (double)partial[((size_t)(t + 64) * 2 + 0) * C + c], b1 = (double)partial[((size_t)(t + 64) * 2 + 1) * C + c];
      const double a2 = (double)partial[((size_t)(t + 128) * 2 + 0) * C + c], b2 = (double)partial[((size_t)(t + 128) * 2 + 1) * C + c];
      const double a3 = (double)partial[((size_t)(t + 192) * 2 + 0) * C + c], b3 = (double)partial[((size_t)(t + 192) * 2 + 1) * C + c];
      s += (a0 + a1) + (a2 + a3);
      q += (b0 + b1) + (b2 + b3);
    }
    for (; t < tiles; t += 64) {
      s += (double)partial[((size_t)t * 2 + 0) * C + c];
      q += (double)partial[((size_t)t * 2 + 1) * C + c];
    }
  }
  red[0][kl][cl] = s;
  red[1][kl][cl] = q;
  __syncthreads();
  if (kl != 0 || c >= C) return;
  for (int k = 1; k < 64; ++k) {
    s += red[0][k][cl];
    q += red[1][k][cl];
  }
  if (FWD) {
    const double m = s / M;
    double var = q / M - m * m;
    if (var < 0.0) var = 0.0;
    out0[c] = (float)m;                                           // mean
    out1[c] = (float)(1.0 / sqrt(var + (double)eps));             // invstd
    if (running_mean) {
      const double unb = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  } else {
    out1[c] = (float)s;                                           // dbeta
    out0[c] = (float)q;                                           // dgamma
  }
}

// the one BN expression (explicit fma) every kernel that forms or re-forms bn(x) uses: identical rounding everywhere
__device__ __forceinline__ f32x4 bn_val(const f32x4 x, const f32x4 mu, const f32x4 is, const f32x4 ga, const f32x4 be) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = bn_val1(x[e], mu[e], is[e], ga[e], be[e]);
  return r;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* residual, float* out, size_t n4, int c4n, int relu) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c4n);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cg], is = reinterpret_cast<const f32x4*>(invstd)[cg];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cg], be = reinterpret_cast<const f32x4*>(beta)[cg];
    f32x4 v = bn_val(reinterpret_cast<const f32x4*>(x)[idx], mu, is, ga, be);
    if (residual) v += reinterpret_cast<const f32x4*>(residual)[idx];
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    reinterpret_cast<f32x4*>(out)[idx] = v;
  }
}

// dx may alias dout (in place): every element is read before it is written by the same thread.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dout, const float* __restrict__ relu_out,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ dgamma,
                                                            const float* __restrict__ dbeta, float* dx, float* g_out,
                                                            size_t n4, int c4n, float invM) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c4n);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cg], is = reinterpret_cast<const f32x4*>(invstd)[cg];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cg];
    const f32x4 dg = reinterpret_cast<const f32x4*>(dgamma)[cg] * invM, db = reinterpret_cast<const f32x4*>(dbeta)[cg] * invM;
    f32x4 g = reinterpret_cast<const f32x4*>(dout)[idx];
    if (relu_out) {
      const f32x4 o = reinterpret_cast<const f32x4*>(relu_out)[idx];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    const f32x4 xhat = (reinterpret_cast<const f32x4*>(x)[idx] - mu) * is;
    if (g_out) reinterpret_cast<f32x4*>(g_out)[idx] = g;
    reinterpret_cast<f32x4*>(dx)[idx] = ga * is * (g - db - xhat * dg);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem: conv1 -> bn1 -> relu -> maxpool(3, 2, 1) (backbone.py:149-152).  The stem activations are the largest tensors of the
// network (N x 112 x 112 x 64 resp. N x 512 x 64 x 64: 1.15 GB per pass at batch 64), and everything around them is
// HBM-bound, so the ReLU output is never materialised:
//   forward   mla_bn_relu_maxpool_fwd: the max-pool reads y (the conv output) and applies BN + ReLU on the fly
//             (saves writing and re-reading a_stem: 2.3 GB per step);
//   backward  mla_bn_bwd_pooled: the BatchNorm backward takes its upstream gradient straight from the POOLED gradient --
//             g[pixel] = [bn(y) > 0] * sum over the <= 4 windows that selected this pixel -- in its reduction pass (as a sum
//             over pooled outputs) and in its apply pass (saves writing and twice re-reading the scattered gradient and reading a_stem: 4.2 GB per step).
// bn_val (above) is the one expression both directions use, so the recomputed ReLU mask and the max-pool decisions agree
// bit for bit.
// ---------------------------------------------------------------------------------------------------------------------

// idx = kh*3+kw of the FIRST maximum in row-major window order (ATen: `val > maxval || isnan(val)`), as maxpool_fwd_kernel.
// A thread forms a 2x2 block of pooled outputs from the 5x5 input pixels they cover (25 loads and BN evaluations instead of
// 36), walking the input rows top to bottom so that every output still sees its window in row-major order.
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ out,
                                                                   uint8_t* __restrict__ idx, int N, int H, int W, int C, int OH, int OW) {
  const int c4n = C >> 2, QH = (OH + 1) >> 1, QW = (OW + 1) >> 1;
  const size_t total = (size_t)N * QH * QW * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    size_t p = i / c4n;
    const int oxq = (int)(p % QW); p /= QW;
    const int oyq = (int)(p % QH);
    const int n = (int)(p / QH);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cg], is = reinterpret_cast<const f32x4*>(invstd)[cg];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cg], be = reinterpret_cast<const f32x4*>(beta)[cg];
    f32x4 best[2][2];
    int bi[2][2][4];
    bool first[2][2];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        best[qy][qx] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        first[qy][qx] = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[qy][qx][e] = 0;
      }
    const int iy0 = oyq * 4 - 1, ix0 = oxq * 4 - 1;          // first input row / column of the 5x5 patch
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int iy = iy0 + r;
      if ((unsigned)iy >= (unsigned)H) continue;
      f32x4 v[5];
      bool ok[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int ix = ix0 + c;
        ok[c] = (unsigned)ix < (unsigned)W;
        v[c] = ok[c] ? reinterpret_cast<const f32x4*>(y)[((size_t)(n * H + iy) * W + ix) * c4n + cg] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        v[c] = bn_val(v[c], mu, is, ga, be);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] = v[c][e] != v[c][e] ? v[c][e] : fmaxf(v[c][e], 0.f);   // relu, NaN kept
      }
#pragma unroll
      for (int qy = 0; qy < 2; ++qy) {
        const int kh = r - 2 * qy;                            // row of output qy's window
        if (kh < 0 || kh > 2) continue;
#pragma unroll
        for (int qx = 0; qx < 2; ++qx)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int c = 2 * qx + kw;
            if (!ok[c]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ve = v[c][e];
              if (first[qy][qx] || ve > best[qy][qx][e] || ve != ve) {
                best[qy][qx][e] = ve;
                bi[qy][qx][e] = kh * 3 + kw;
              }
            }
            first[qy][qx] = false;
          }
      }
    }
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        const int oy = oyq * 2 + qy, ox = oxq * 2 + qx;
        if (oy >= OH || ox >= OW) continue;
        const size_t o = ((size_t)(n * OH + oy) * OW + ox) * c4n + cg;
        reinterpret_cast<f32x4*>(out)[o] = best[qy][qx];
        reinterpret_cast<uchar4*>(idx)[o] = make_uchar4(bi[qy][qx][0], bi[qy][qx][1], bi[qy][qx][2], bi[qy][qx][3]);
      }
  }
}

// The reduction pass in scatter form: sum_pixels g = sum over POOLED outputs o of dpool[o] * [bn(y[sel(o)]) > 0] (every pooled output
// selected exactly one pixel per channel), likewise sum g * xhat -- a quarter of the iterations of the per-pixel gather form and
// 6 loads per thread instead of 9.  Tiles are rows of pooled pixels; partial layout as bn_reduce_kernel<1>.
__global__ __launch_bounds__(256) void bn_bwd_pooled_reduce_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                    const float* __restrict__ y, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ out, int N, int H,
                                                                    int W, int C, int OH, int OW, int tile_rows) {
  __shared__ f32x4 red[2][256];
  const int c4n = C >> 2, MP = N * OH * OW;
  const int cg = threadIdx.x % c4n, rl = threadIdx.x / c4n, nrl = 256 / c4n;
  const int r0 = blockIdx.x * tile_rows, r1 = min(MP, r0 + tile_rows);
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cg], is = reinterpret_cast<const f32x4*>(invstd)[cg];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cg], be = reinterpret_cast<const f32x4*>(beta)[cg];
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rl; r < r1; r += nrl) {
    const int ox = r % OW, t = r / OW, oy = t % OH, n = t / OH;
    const size_t o = (size_t)r * c4n + cg;
    const f32x4 g = reinterpret_cast<const f32x4*>(dpool)[o];
    const uchar4 sel = reinterpret_cast<const uchar4*>(idx)[o];
    const int code[4] = {sel.x, sel.y, sel.z, sel.w};
    f32x4 xv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kh = (code[e] * 11) >> 5, kw = code[e] - 3 * kh;      // code = kh * 3 + kw, 0..8
      xv[e] = y[((size_t)(n * H + oy * 2 - 1 + kh) * W + (ox * 2 - 1 + kw)) * C + cg * 4 + e];
    }
    const f32x4 a = bn_val(xv, mu, is, ga, be);
    const f32x4 xhat = (xv - mu) * is;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = a[e] > 0.f ? g[e] : 0.f;
      s0[e] += ge;
      s1[e] += ge * xhat[e];
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < nrl; ++k) {
      s0 += red[0][k * c4n + cg];
      s1 += red[1][k * c4n + cg];
    }
    reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 2 + 0) * C)[cg] = s0;
    reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 2 + 1) * C)[cg] = s1;
  }
}

// The apply pass: dy = gamma * invstd * (g - db/M - xhat * dg/M).  A thread owns the 2x2 pixel quad (2a + py, 2b + px): its
// pixels lie in the four windows (a | a+1, b | b+1) only, at window positions that are compile-time constants --
//   (0,0): (a,b) code 4;   (0,1): (a,b) 5, (a,b+1) 3;   (1,0): (a,b) 7, (a+1,b) 1;   (1,1): (a,b) 8, (a,b+1) 6, (a+1,b) 2, (a+1,b+1) 0
// -- so the pooled gradient and the index bytes of each window are loaded once per quad (4 + 4 loads for 4 pixels instead of
// up to 8 per pixel in the per-pixel gather form).
__global__ __launch_bounds__(256) void bn_bwd_pooled_apply_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                   const float* __restrict__ y, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                                   const float* __restrict__ dbeta, float* __restrict__ out, int N, int H,
                                                                   int W, int C, int OH, int OW) {
  const int c4n = C >> 2, QH = (H + 1) >> 1, QW = (W + 1) >> 1;
  const size_t total = (size_t)N * QH * QW * c4n;
  const float invM = 1.0f / ((float)N * (float)H * (float)W);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    size_t p = i / c4n;
    const int b = (int)(p % QW); p /= QW;
    const int a = (int)(p % QH);
    const int n = (int)(p / QH);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cg], is = reinterpret_cast<const f32x4*>(invstd)[cg];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[cg], be = reinterpret_cast<const f32x4*>(beta)[cg];
    const f32x4 dg = reinterpret_cast<const f32x4*>(dgamma)[cg] * invM, db = reinterpret_cast<const f32x4*>(dbeta)[cg] * invM;
    f32x4 wg[2][2];                    // pooled gradient of window (a + wy, b + wx); zero outside the pooled grid
    int ws[2][2][4];                   // its selected positions; -1 outside
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int oy = a + wy, ox = b + wx;
        if (oy < OH && ox < OW) {
          const size_t o = ((size_t)(n * OH + oy) * OW + ox) * c4n + cg;
          wg[wy][wx] = reinterpret_cast<const f32x4*>(dpool)[o];
          const uchar4 s = reinterpret_cast<const uchar4*>(idx)[o];
          ws[wy][wx][0] = s.x; ws[wy][wx][1] = s.y; ws[wy][wx][2] = s.z; ws[wy][wx][3] = s.w;
        } else {
          wg[wy][wx] = f32x4{0.f, 0.f, 0.f, 0.f};
          ws[wy][wx][0] = ws[wy][wx][1] = ws[wy][wx][2] = ws[wy][wx][3] = -1;
        }
      }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int iy = 2 * a + py, ix = 2 * b + px;
        if (iy >= H || ix >= W) continue;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wy = 0; wy <= py; ++wy)             // window row a + wy covers pixel row 2a + py at kh = py + 1 - 2 wy
#pragma unroll
          for (int wx = 0; wx <= px; ++wx) {
            const int code = (py + 1 - 2 * wy) * 3 + (px + 1 - 2 * wx);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += ws[wy][wx][e] == code ? wg[wy][wx][e] : 0.f;
          }
        const size_t o = ((size_t)(n * H + iy) * W + ix) * c4n + cg;
        const f32x4 xv = reinterpret_cast<const f32x4*>(y)[o];
        const f32x4 av = bn_val(xv, mu, is, ga, be);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = av[e] > 0.f ? g[e] : 0.f;
        const f32x4 xhat = (xv - mu) * is;
        reinterpret_cast<f32x4*>(out)[o] = ga * is * (g - db - xhat * dg);
      }
  }
}

static int bn_check(const char* who, int M, int C) {
  MLA_REQUIRE(M > 0 && C > 0, "%s: non-positive dims", who);
  MLA_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "%s: C=%d must be 4*2^k, <= 1024", who, C);
  return MLA_OK;
}

static int ew_grid(size_t n4) {
  size_t b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int mla_bn_stats_partial(const float* x, int M, int C, float* partial, int* tiles, void* stream) {
  if (int rc = bn_check("mla_bn_stats_partial", M, C)) return rc;
  MLA_REQUIRE(x && partial, "mla_bn_stats_partial: null pointer");
  const int tr = bn_tile_rows(M), nt = cdiv(M, tr);
  bn_reduce_kernel<0><<<nt, 256, 0, (hipStream_t)stream>>>(x, nullptr, nullptr, nullptr, nullptr, partial, M, C, tr);
  MLA_CHECK_LAUNCH("bn_reduce_kernel<0>");
  if (tiles) *tiles = nt;
  return MLA_OK;
}

extern "C" int mla_bn_finalize(const float* partial, int tiles, int M, int C, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, void* stream) {
  MLA_REQUIRE(partial && mean && invstd && tiles > 0 && M > 0 && C > 0, "mla_bn_finalize: bad argument");
  MLA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mla_bn_finalize: running stats must come in pairs");
  MLA_REQUIRE(((uintptr_t)partial % 8) == 0, "mla_bn_finalize: partial must be 8-byte aligned");
  // scratch tail: right after the tiles
  hipStream_t st = (hipStream_t)stream;
  const double* pd = reinterpret_cast<const double*>(partial);                     // fp64 [tiles][2][C] (conv epilogue / bn_stats_partial)
  if (tiles <= BN_ONE_MAXT) {
    bn_finalize_tiles_kernel<double, 1><<<cdiv(C, 16), 256, 0, st>>>(pd, tiles, M, C, eps, momentum, mean, invstd, running_mean, running_var);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_kernel");
    return MLA_OK;
  }
  if (tiles <= BN_WIDE_MAXT) {
    bn_finalize_tiles_wide_kernel<double, 1><<<cdiv(C, 4), 256, 0, st>>>(pd, tiles, M, C, eps, momentum, mean, invstd, running_mean, running_var);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_wide_kernel");
    return MLA_OK;
  }
  double* scratch = const_cast<double*>(pd) + (size_t)tiles * 2 * C;
  const int S = bn_red_chunks(tiles);
  bn_tiles_stage1_kernel<double><<<dim3(cdiv(C, 64), S), 256, 0, st>>>(pd, tiles, C, scratch);
  MLA_CHECK_LAUNCH("bn_tiles_stage1_kernel");
  bn_finalize_kernel<<<cdiv(C, 64), 256, 0, st>>>(scratch, S, M, C, eps, momentum, mean, invstd, running_mean, running_var);
  MLA_CHECK_LAUNCH("bn_finalize_kernel");
  return MLA_OK;
}

extern "C" int mla_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* residual, float* out, int M, int C, int relu, void* stream) {
  if (int rc = bn_check("mla_bn_apply", M, C)) return rc;
  MLA_REQUIRE(x && mean && invstd && gamma && beta && out, "mla_bn_apply: null pointer");
  const size_t n4 = (size_t)M * C / 4;
  bn_apply_kernel<<<ew_grid(n4), 256, 0, (hipStream_t)stream>>>(x, mean, invstd, gamma, beta, residual, out, n4, C / 4, relu);
  MLA_CHECK_LAUNCH("bn_apply_kernel");
  return MLA_OK;
}

extern "C" int mla_bn_bwd(const float* dout, const float* relu_out, const float* x, const float* mean,
                          const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta, float* g_out,
                          float* ws, int M, int C, void* stream) {
  if (int rc = bn_check("mla_bn_bwd", M, C)) return rc;
  MLA_REQUIRE(dout && x && mean && invstd && gamma && dx && dgamma && dbeta && ws, "mla_bn_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int tr = bn_tile_rows(M), nt = cdiv(M, tr);
  bn_reduce_kernel<1><<<nt, 256, 0, st>>>(x, dout, relu_out, mean, invstd, ws, M, C, tr);
  MLA_CHECK_LAUNCH("bn_reduce_kernel<1>");
  if (nt <= BN_ONE_MAXT) {
    bn_finalize_tiles_kernel<float, 0><<<cdiv(C, 16), 256, 0, st>>>(ws, nt, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_kernel");
  } else if (nt <= BN_WIDE_MAXT) {
    bn_finalize_tiles_wide_kernel<float, 0><<<cdiv(C, 4), 256, 0, st>>>(ws, nt, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_wide_kernel");
  } else {
    double* scratch = reinterpret_cast<double*>(ws + (size_t)nt * 2 * C);
    const int S = bn_red_chunks(nt);
    bn_tiles_stage1_kernel<float><<<dim3(cdiv(C, 64), S), 256, 0, st>>>(ws, nt, C, scratch);
    MLA_CHECK_LAUNCH("bn_tiles_stage1_kernel");
    bn_bwd_finalize_kernel<<<cdiv(C, 64), 256, 0, st>>>(scratch, S, C, dgamma, dbeta);
    MLA_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  }
  const size_t n4 = (size_t)M * C / 4;
  bn_bwd_apply_kernel<<<ew_grid(n4), 256, 0, st>>>(dout, relu_out, x, mean, invstd, gamma, dgamma, dbeta, dx, g_out, n4,
                                                   C / 4, 1.0f / (float)M);
  MLA_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return MLA_OK;
}

// BatchNorm backward whose reduction pass was done by the producer of dout (mla_conv2d_dgrad[_split]_bn): finalize the
// `tiles` per-tile sums of `partial` (its scratch tail follows the tiles) into dgamma / dbeta, then the apply pass.
extern "C" int mla_bn_bwd_from_partial(const float* dout, const float* x, const float* mean, const float* invstd,
                                       const float* gamma, float* dx, float* dgamma, float* dbeta, float* partial, int tiles,
                                       int M, int C, void* stream) {
  if (int rc = bn_check("mla_bn_bwd_from_partial", M, C)) return rc;
  MLA_REQUIRE(dout && x && mean && invstd && gamma && dx && dgamma && dbeta && partial && tiles > 0,
              "mla_bn_bwd_from_partial: null pointer or no tiles");
  hipStream_t st = (hipStream_t)stream;
  if (tiles <= BN_ONE_MAXT) {
    bn_finalize_tiles_kernel<float, 0><<<cdiv(C, 16), 256, 0, st>>>(partial, tiles, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_kernel");
  } else if (tiles <= BN_WIDE_MAXT) {
    bn_finalize_tiles_wide_kernel<float, 0><<<cdiv(C, 4), 256, 0, st>>>(partial, tiles, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_wide_kernel");
  } else {
    double* scratch = reinterpret_cast<double*>(partial + (size_t)tiles * 2 * C);
    const int S = bn_red_chunks(tiles);
    bn_tiles_stage1_kernel<float><<<dim3(cdiv(C, 64), S), 256, 0, st>>>(partial, tiles, C, scratch);
    MLA_CHECK_LAUNCH("bn_tiles_stage1_kernel");
    bn_bwd_finalize_kernel<<<cdiv(C, 64), 256, 0, st>>>(scratch, S, C, dgamma, dbeta);
    MLA_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  }
  const size_t n4 = (size_t)M * C / 4;
  bn_bwd_apply_kernel<<<ew_grid(n4), 256, 0, st>>>(dout, nullptr, x, mean, invstd, gamma, dgamma, dbeta, dx, nullptr, n4, C / 4,
                                                   1.0f / (float)M);
  MLA_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return MLA_OK;
}

extern "C" int mla_bn_relu_maxpool_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                       float* out, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  if (int rc = bn_check("mla_bn_relu_maxpool_fwd", N * H * W, C)) return rc;
  MLA_REQUIRE(y && mean && invstd && gamma && beta && out && idx && N > 0 && H > 0 && W > 0, "mla_bn_relu_maxpool_fwd: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t n4 = (size_t)N * ((OH + 1) / 2) * ((OW + 1) / 2) * (C / 4);     // one thread per 2x2 block of pooled outputs and 4 channels
  bn_relu_maxpool_fwd_kernel<<<ew_grid(n4), 256, 0, (hipStream_t)stream>>>(y, mean, invstd, gamma, beta, out, idx, N, H, W, C, OH, OW);
  MLA_CHECK_LAUNCH("bn_relu_maxpool_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_bn_bwd_pooled(const float* dpool, const uint8_t* idx, const float* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta, float* ws, int N,
                                 int H, int W, int C, void* stream) {
  const long Ml = (long)N * H * W;
  MLA_REQUIRE(Ml > 0 && Ml < (1L << 31), "mla_bn_bwd_pooled: bad dims");
  const int M = (int)Ml;
  if (int rc = bn_check("mla_bn_bwd_pooled", M, C)) return rc;
  MLA_REQUIRE(dpool && idx && y && mean && invstd && gamma && beta && dy && dgamma && dbeta && ws, "mla_bn_bwd_pooled: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const int MP = N * OH * OW;
  const int trp = bn_tile_rows(MP), nt = cdiv(MP, trp);     // reduction: tiles of pooled pixels (<= the tiles of M the workspace holds)
  bn_bwd_pooled_reduce_kernel<<<nt, 256, 0, st>>>(dpool, idx, y, mean, invstd, gamma, beta, ws, N, H, W, C, OH, OW, trp);
  MLA_CHECK_LAUNCH("bn_bwd_pooled_reduce_kernel");
  if (nt <= BN_ONE_MAXT) {
    bn_finalize_tiles_kernel<float, 0><<<cdiv(C, 16), 256, 0, st>>>(ws, nt, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_kernel");
  } else if (nt <= BN_WIDE_MAXT) {
    bn_finalize_tiles_wide_kernel<float, 0><<<cdiv(C, 4), 256, 0, st>>>(ws, nt, M, C, 0.f, 0.f, dgamma, dbeta, nullptr, nullptr);
    MLA_CHECK_LAUNCH("bn_finalize_tiles_wide_kernel");
  } else {
    double* scratch = reinterpret_cast<double*>(ws + (size_t)nt * 2 * C);
    const int S = bn_red_chunks(nt);
    bn_tiles_stage1_kernel<float><<<dim3(cdiv(C, 64), S), 256, 0, st>>>(ws, nt, C, scratch);
    MLA_CHECK_LAUNCH("bn_tiles_stage1_kernel");
    bn_bwd_finalize_kernel<<<cdiv(C, 64), 256, 0, st>>>(scratch, S, C, dgamma, dbeta);
    MLA_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  }
  const size_t nq = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);     // one thread per 2x2 pixel quad and 4 channels
  bn_bwd_pooled_apply_kernel<<<ew_grid(nq), 256, 0, st>>>(dpool, idx, y, mean, invstd, gamma, beta, dgamma, dbeta, dy, N, H, W, C, OH, OW);
  MLA_CHECK_LAUNCH("bn_bwd_pooled_apply_kernel");
  return MLA_OK;
}
