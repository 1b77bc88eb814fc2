// Pooling + layout kernels (pixel-major NHWC fp32), gfx950.  All HBM-bound, float4 on the channel axis.
//
//   max-pool 3x3 s2 p1      nn.MaxPool2d (models/backbone.py:88, 152) + its autograd scatter
//   global average pool     F.adaptive_avg_pool2d / adaptive_avg_pool3d(.,1) + flatten
//                           (models/basic_model.py:56-65) + its autograd broadcast
//   video -> frames         x.permute(0,2,1,3,4).contiguous().view(B*T,C,H,W) (backbone.py:144-147),
//                           fused with the NCHW->NHWC change of layout
#include "common.h"

// idx = kh*3+kw of the FIRST maximum in row-major window order (ATen: `val > maxval || isnan(val)`).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                                           int OH, int OW) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * OH * OW * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    size_t p = i / c4n;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * 2 - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox * 2 - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[((size_t)(n * H + iy) * W + ix) * c4n + cg];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * 3 + kw;
          }
        first = false;
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(idx)[i] = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}

// Gather form of the scatter: every input pixel inspects the <= 4 windows that contain it.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                           const float* __restrict__ relu_src, float* __restrict__ dx,
                                                           int N, int H, int W, int C, int OH, int OW) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * H * W * c4n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    size_t p = i / c4n;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int n = (int)(p / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= OH) continue;
      const int kh = iy - (oy * 2 - 1);
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= OW) continue;
        const int kw = ix - (ox * 2 - 1);
        const size_t o = ((size_t)(n * OH + oy) * OW + ox) * c4n + cg;
        const uchar4 b = reinterpret_cast<const uchar4*>(idx)[o];
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
        const int code = kh * 3 + kw;
        if (b.x == code) acc[0] += g[0];
        if (b.y == code) acc[1] += g[1];
        if (b.z == code) acc[2] += g[2];
        if (b.w == code) acc[3] += g[3];
      }
    }
    if (relu_src) {
      const f32x4 r = reinterpret_cast<const f32x4*>(relu_src)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = r[e] > 0.f ? acc[e] : 0.f;
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// x [NB][P][C] -> y [NB][C]; block = (C/4 column groups) x (row lanes), LDS across row lanes.
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C,
                                                           int cgpb) {  // cgpb: float4 column groups per block (divides 256 and C/4)
  __shared__ f32x4 red[256];
  const int c4n = C >> 2;
  const int nrl = 256 / cgpb;
  const int cg = blockIdx.y * cgpb + threadIdx.x % cgpb, rl = threadIdx.x / cgpb;
  const int nb = blockIdx.x;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int p = rl; p < P; p += nrl) s += reinterpret_cast<const f32x4*>(x)[((size_t)nb * P + p) * c4n + cg];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < nrl; ++k) s += red[k * cgpb + threadIdx.x];
    reinterpret_cast<f32x4*>(y)[(size_t)nb * c4n + cg] = s * (1.0f / (float)P);
  }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ relu_src,
                                                           float* __restrict__ dx, size_t n4, int P, int c4n) {
  const float invP = 1.0f / (float)P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    const size_t nb = i / ((size_t)P * c4n);
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[nb * c4n + cg] * invP;
    if (relu_src) {
      const f32x4 r = reinterpret_cast<const f32x4*>(relu_src)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = r[e] > 0.f ? g[e] : 0.f;
    }
    reinterpret_cast<f32x4*>(dx)[i] = g;
  }
}

// (B,C,T,H,W) -> (B*T,H,W,C), small C (frames: 3).
__global__ __launch_bounds__(256) void video_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int B, int C, int T, int H, int W) {
  const size_t hw = (size_t)H * W, total = (size_t)B * T * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % hw;
    const size_t bt = i / hw;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    for (int c = 0; c < C; ++c) dst[i * C + c] = src[((b * C + c) * T + t) * hw + pix];
  }
}

// generic per-image transpose [R][S] -> [S][R] (NCHW<->NHWC with R=C,S=HW or R=HW,S=C)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int S) {
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * R * S;
  const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, s = s0 + tx;
    if (r < R && s < S) tile[k][tx] = src[base + (size_t)r * S + s];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int s = s0 + k, r = r0 + tx;
    if (r < R && s < S) dst[base + (size_t)s * R + r] = tile[tx][k];
  }
}

static int ew_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

extern "C" int mla_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  MLA_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "mla_maxpool3x3s2_fwd: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  maxpool_fwd_kernel<<<ew_grid((size_t)N * OH * OW * (C / 4)), 256, 0, (hipStream_t)stream>>>(x, y, idx, N, H, W, C, OH, OW);
  MLA_CHECK_LAUNCH("maxpool_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, const float* relu_src, float* dx, int N, int H,
                                    int W, int C, void* stream) {
  MLA_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "mla_maxpool3x3s2_bwd: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  maxpool_bwd_kernel<<<ew_grid((size_t)N * H * W * (C / 4)), 256, 0, (hipStream_t)stream>>>(dy, idx, relu_src, dx, N, H, W, C, OH, OW);
  MLA_CHECK_LAUNCH("maxpool_bwd_kernel");
  return MLA_OK;
}

extern "C" int mla_avgpool_fwd(const float* x, float* y, int NB, int P, int C, void* stream) {
  MLA_REQUIRE(x && y && NB > 0 && P > 0 && C > 0 && C % 4 == 0, "mla_avgpool_fwd: bad argument");
  const int c4n = C / 4;
  int cgpb = 256;
  while (cgpb > 1 && c4n % cgpb != 0) cgpb >>= 1;     // largest power of two <= 256 dividing C/4 (C=768 -> 64)
  MLA_REQUIRE(cgpb >= 4, "mla_avgpool_fwd: C=%d unsupported", C);
  dim3 grid(NB, c4n / cgpb);
  avgpool_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, P, C, cgpb);
  MLA_CHECK_LAUNCH("avgpool_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_avgpool_bwd(const float* dy, const float* relu_src, float* dx, int NB, int P, int C, void* stream) {
  MLA_REQUIRE(dy && dx && NB > 0 && P > 0 && C > 0 && C % 4 == 0, "mla_avgpool_bwd: bad argument");
  const size_t n4 = (size_t)NB * P * (C / 4);
  avgpool_bwd_kernel<<<ew_grid(n4), 256, 0, (hipStream_t)stream>>>(dy, relu_src, dx, n4, P, C / 4);
  MLA_CHECK_LAUNCH("avgpool_bwd_kernel");
  return MLA_OK;
}

extern "C" int mla_video_to_nhwc(const float* src, float* dst, int B, int C, int T, int H, int W, void* stream) {
  MLA_REQUIRE(src && dst && B > 0 && C > 0 && C <= 8 && T > 0 && H > 0 && W > 0, "mla_video_to_nhwc: bad argument");
  video_to_nhwc_kernel<<<ew_grid((size_t)B * T * H * W), 256, 0, (hipStream_t)stream>>>(src, dst, B, C, T, H, W);
  MLA_CHECK_LAUNCH("video_to_nhwc_kernel");
  return MLA_OK;
}

static int transpose_launch(const float* src, float* dst, int batch, int R, int S, hipStream_t st) {
  MLA_REQUIRE(src && dst && batch > 0 && batch < 65536 && R > 0 && S > 0, "transpose: bad argument");
  MLA_REQUIRE(cdiv(R, 32) < 65536, "transpose: too many rows");
  transpose_kernel<<<dim3(cdiv(S, 32), cdiv(R, 32), batch), 256, 0, st>>>(src, dst, R, S);
  MLA_CHECK_LAUNCH("transpose_kernel");
  return MLA_OK;
}

extern "C" int mla_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
  return transpose_launch(src, dst, N, C, H * W, (hipStream_t)stream);
}
extern "C" int mla_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
  return transpose_launch(src, dst, N, H * W, C, (hipStream_t)stream);
}
