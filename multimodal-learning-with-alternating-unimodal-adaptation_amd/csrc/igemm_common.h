// Shared pieces of the gather-GEMM convolution kernels (conv_igemm.hip: exact-fp32 MFMA;
// conv_igemm_split.hip: fp32 operands split into three bf16 terms on the bf16 MFMA).
#pragma once
#include "common.h"

#define MAX_TAPS 49
#define BK 32

struct IGemmGeom {
  int N, H, W, C;          // input tensor (NHWC); C = GEMM-K per tap
  int OH, OW;              // logical output grid
  int CO;                  // GEMM-N
  int sy, sx;              // input coordinate stride
  int OHF, OWF;            // spatial dims of the output tensor
  int osy, osx, ooy, oox;  // output pixel = (oy*osy+ooy, ox*osx+oox)
  int T;                   // number of taps
  int M;                   // N*OH*OW
  int K;                   // scalar-gather mode: T*C (un-padded flattened K)
  unsigned x_bytes, w_bytes, y_bytes;   // byte sizes of the gathered tensor, the weights and the second operand (wgrad: dY)
  int epi;                 // 0: MASK = ReLU mask (v = MASK > 0 ? v : 0);  1: MASK = GELU pre-activation u (v *= gelu'(u))
  int tap[MAX_TAPS];       // (dy & 0xff) | (dx & 0xff) << 8 | wt << 16   (int32: read with s_load_dword)
  // Input-gradient launches only: up to two BatchNorm layers whose backward consumes the tensor this launch writes (the
  // BatchNorm that follows in the backward chain, and the downsample BatchNorm beside it).  For each, the epilogue also forms the
  // per-tile column sums  sum v  and  sum v * xhat,  xhat = (bn_x - bn_mean) * bn_invstd  (v = the value it stores), which is the
  // whole reduction pass of that BatchNorm backward: part[bn_tile0 + tile][2][CO] floats, the layout of bn_reduce_kernel<1>.
  const float* bn_x[2];
  const float* bn_mean[2];
  const float* bn_invstd[2];
  float* bn_part[2];
  int bn_tile0;
  int m0;                  // first output row of this launch (split kernels: a launch may cover the row window [m0, M) of the problem)
  // BatchNorm folded into this launch's operands (the 64 -> 64 persistent patch kernel; conv1 -> bn1 -> relu -> conv2 of a BasicBlock,
  // models/backbone.py:38-46, without materialising relu(bn1(.))):
  const float* in_bn[4];   // forward: the gathered tensor is y and the convolution runs over relu(bn(y)); {mean, invstd, gamma, beta} per channel
  const float* mask_gb[2]; // input gradient: the ReLU mask is bn(bn_x[0]) > 0 with {gamma, beta} (mean / invstd: bn_mean[0] / bn_invstd[0])
};


// Gathers use raw buffer loads: the hardware range check of the buffer descriptor returns 0 for any
// offset >= num_records, so padding / out-of-range rows are "loaded" as zeros by pointing them at
// OOB_OFF -- no zero page, no pointer select, no branch, and no select on the loaded value (which would
// make the compiler wait for the prefetch before the MFMA block).  Offsets are 32-bit: tensors < 4 GiB.
#define OOB_OFF 0xFFFFFFFFu
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// F.gelu default (erf form), m3ae.py:77, and its derivative
__device__ __forceinline__ float gelu_fwd(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.39894228040143268f * expf(-0.5f * u * u);
}

static inline int pack_tap(int dy, int dx, int wt) { return (dy & 0xff) | ((dx & 0xff) << 8) | (wt << 16); }
__device__ __forceinline__ int tap_dy(int t) { return (int)(signed char)(t & 0xff); }
__device__ __forceinline__ int tap_dx(int t) { return (int)(signed char)((t >> 8) & 0xff); }
__device__ __forceinline__ int tap_wt(int t) { return t >> 16; }

// ---------------------------------------------------------------------------------------------
// Epilogue shared by the gather-GEMM kernels: residual add, bias, ReLU/GELU backward mask, GELU second
// output, store, fused BatchNorm column statistics (forward) or BatchNorm-backward reductions (input gradient).  acc is in
// the 32x32 MFMA C/D layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  `red` is >= 4*WM*BN floats of LDS
// that no wave reads any more (the caller's trailing __syncthreads()).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const int4* rowinfo, float* red,
                                               float* Y, const float* R, const float* MASK, float* __restrict__ part,
                                               const float* __restrict__ BIAS, float* __restrict__ Y2, const IGemmGeom& g,
                                               int tm, int tn) {
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gCO = g.CO;
  const int h = lane >> 5, j = lane & 31;
  // BatchNorm statistics of the tile are accumulated in fp64: v * v is exact there (24 x 24 significand bits) and the later
  // E[x^2] - E[x]^2 then cancels against ~1e-16 instead of the ~1e-7 of fp32 tile sums, so channels with |mean| >> std keep
  // their variance (ATen uses Welford; tests/test_ops_gpu.py::test_bn_large_mean).  Only instantiated work when part != null.
  double csum[NI], csq[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) csum[ni] = csq[ni] = 0.0;
  // fused BatchNorm-backward reductions (see IGemmGeom::bn_x): b0 = sum v (shared by both requests), b1[q] = sum v * xhat_q
  const bool bnq0 = g.bn_x[0] != nullptr, bnq1 = g.bn_x[1] != nullptr;
  float b0[NI], b1[2][NI], bmu[2][NI], bis[2][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    b0[ni] = b1[0][ni] = b1[1][ni] = 0.f;
    const int c = tn * BN + wn * (BN / WN) + ni * 32 + j;
    bmu[0][ni] = bnq0 ? g.bn_mean[0][c] : 0.f;
    bis[0][ni] = bnq0 ? g.bn_invstd[0][c] : 0.f;
    bmu[1][ni] = bnq1 ? g.bn_mean[1][c] : 0.f;
    bis[1][ni] = bnq1 ? g.bn_invstd[1][c] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    size_t off[16];
    bool ok[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int4 info = rowinfo[wm * (BM / WM) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
      ok[e] = info.x >= 0;
      off[e] = (size_t)(ok[e] ? info.w : 0) * gCO + tn * BN + wn * (BN / WN) + j;
    }
    if (R) {  // all residual loads of this 32-row slab issue together (one wait, not one per element)
      float rv[16][NI];
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) rv[e][ni] = R[off[e] + ni * 32];
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni][e] += rv[e][ni];
    }
    if (BIAS) {  // Linear bias (one value per output column)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float bv = BIAS[tn * BN + wn * (BN / WN) + ni * 32 + j];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] += bv;
      }
    }
    if (MASK) {
      float mv[16][NI];
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mv[e][ni] = MASK[off[e] + ni * 32];
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (g.epi == 0) acc[mi][ni][e] = mv[e][ni] > 0.f ? acc[mi][ni][e] : 0.f;   // ReLU backward
          else acc[mi][ni][e] *= gelu_grad(mv[e][ni]);                                // GELU backward (erf form)
        }
    }
    if (Y2) {  // second output: GELU of the (bias-added) pre-activation that goes to Y
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          if (ok[e]) Y2[off[e] + ni * 32] = gelu_fwd(acc[mi][ni][e]);
    }
    if (bnq0) {  // rows past M may hold residual garbage (their loads read row 0): they are masked out of the sums
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && !bnq1) break;
        const float* __restrict__ bx = g.bn_x[q];
        float xv[16][NI];
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) xv[e][ni] = bx[off[e] + ni * 32];
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const float v = ok[e] ? acc[mi][ni][e] : 0.f;
            if (q == 0) b0[ni] += v;
            b1[q][ni] += v * ((xv[e][ni] - bmu[q][ni]) * bis[q][ni]);
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float v = acc[mi][ni][e];
        if (part) {             // rows past M hold exact zeros (zero A rows), so no masking is needed;
          const double vd = (double)v;   // the statistics are only requested by the forward conv (no R / MASK)
          csum[ni] += vd;
          csq[ni] += vd * vd;
        }
        if (ok[e]) Y[off[e] + ni * 32] = v;
      }
  }
  if (bnq0) {  // per-tile column sums of the fused BatchNorm-backward reductions: [tile][2][CO] floats per request
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      b0[ni] += __shfl_xor(b0[ni], 32, 64);
      b1[0][ni] += __shfl_xor(b1[0][ni], 32, 64);
      b1[1][ni] += __shfl_xor(b1[1][ni], 32, 64);
      if (h == 0) {
        const int c = wn * (BN / WN) + ni * 32 + j;
        red[(wm * 3 + 0) * BN + c] = b0[ni];
        red[(wm * 3 + 1) * BN + c] = b1[0][ni];
        red[(wm * 3 + 2) * BN + c] = b1[1][ni];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += red[(w * 3 + 0) * BN + tid];
        q0 += red[(w * 3 + 1) * BN + tid];
        q1 += red[(w * 3 + 2) * BN + tid];
      }
      const size_t o = ((size_t)(g.bn_tile0 + tm) * 2) * gCO + tn * BN + tid;
      g.bn_part[0][o] = s;
      g.bn_part[0][o + gCO] = q0;
      if (bnq1) {
        g.bn_part[1][o] = s;
        g.bn_part[1][o + gCO] = q1;
      }
    }
  }
  if (part) {  // fused BatchNorm statistics: per-tile column sum / sum of squares, fp64 [tiles][2][C]
    double* redd = reinterpret_cast<double*>(red);           // WM * 2 * BN doubles <= 8 KB of the (free) operand LDS
    double* partd = reinterpret_cast<double*>(part);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      csum[ni] += __shfl_xor(csum[ni], 32, 64);
      csq[ni] += __shfl_xor(csq[ni], 32, 64);
      if (h == 0) {
        const int c = wn * (BN / WN) + ni * 32 + j;
        redd[(wm * 2 + 0) * BN + c] = csum[ni];
        redd[(wm * 2 + 1) * BN + c] = csq[ni];
      }
    }
    __syncthreads();
    if (tid < BN) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += redd[(w * 2 + 0) * BN + tid];
        q += redd[(w * 2 + 1) * BN + tid];
      }
      partd[((size_t)tm * 2 + 0) * gCO + tn * BN + tid] = s;
      partd[((size_t)tm * 2 + 1) * gCO + tn * BN + tid] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side helpers
// ---------------------------------------------------------------------------------------------
static inline int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

static inline int check_conv(const char* who, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  MLA_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "%s: non-positive dims", who);
  MLA_REQUIRE(KH * KW <= MAX_TAPS && KH > 0 && KW > 0, "%s: kernel %dx%d unsupported (max %d taps)", who, KH, KW, MAX_TAPS);
  MLA_REQUIRE(stride == 1 || stride == 2, "%s: stride %d unsupported", who, stride);
  MLA_REQUIRE(Cout % 64 == 0 && Cout <= 1024 && Cin <= 1024, "%s: Cout=%d must be a multiple of 64; channels <= 1024", who, Cout);
  MLA_REQUIRE(Cin % 64 == 0 || Cin <= 4, "%s: Cin=%d must be a multiple of 64 or <= 4 (stem)", who, Cin);
  MLA_REQUIRE(pad >= 0 && pad < 64 && H < 32768 && W < 32768, "%s: pad/size out of range", who);
  MLA_REQUIRE((long)N * H * W < (1L << 31) / 4, "%s: too many pixels for 32-bit pixel indices", who);
  {   // both tensors of the convolution are addressed with 32-bit byte offsets
    const long oh = (H + 2 * pad - KH) / stride + 1, ow = (W + 2 * pad - KW) / stride + 1;
    MLA_REQUIRE((long)N * H * W * Cin * 4 < 0xFFFFFFF0L && (long)N * (oh > 0 ? oh : 0) * (ow > 0 ? ow : 0) * Cout * 4 < 0xFFFFFFF0L,
                "%s: input and output tensors must each be < 4 GiB (32-bit buffer offsets)", who);
  }
  return MLA_OK;
}

// Tile choice: every CU runs ceil(blocks/256) rounds of MFMA-bound tiles, so minimise
// rounds * tile area / efficiency (the 64x64 tile pays more barriers per flop).
enum { CFG_128x128 = 0, CFG_256x64 = 1, CFG_64x64 = 2, CFG_128x64 = 3, CFG_COUNT = 4 };
static inline int cfg_bm(int cfg) { return (cfg == CFG_128x128 || cfg == CFG_128x64) ? 128 : (cfg == CFG_256x64 ? 256 : 64); }
static inline int cfg_bn(int cfg) { return cfg == CFG_128x128 ? 128 : 64; }

#ifndef F32_EFF
#define F32_EFF 0.93, 0.90, 1.0, 1.02
#endif
extern int g_f32_cfg;   // measurement hook (conv_igemm.hip): force one tile
static inline int pick_cfg(const long* Ms, const int* weights, int n, int CO, bool scalar) {
  if (scalar) return CFG_64x64;   // stem: 2-5 K steps per block, so many small resident blocks overlap best (measured -12 % vs 256x64)
  // per-flop rates at full rounds, measured on the ResNet-18 layer shapes (scripts/f32_tile_probe.py), relative to the
  // 64x64 tile: 128x128 0.93, 256x64 0.90, 128x64 1.02 (a.l1: 124.6 vs 112.9 TF on 256x64); any tile loses ~10 % when
  // fewer than two workgroups are resident per CU.
  const double eff[CFG_COUNT] = {F32_EFF};
  int best = -1;
  double best_cost = 0;
  if (g_f32_cfg >= 0 && CO % cfg_bn(g_f32_cfg) == 0) return g_f32_cfg;
  for (int cfg = 0; cfg < CFG_COUNT; ++cfg) {
    if (CO % cfg_bn(cfg) != 0 || eff[cfg] <= 0.0) continue;
    double blocks = 0, wsum = 0, wblocks = 0;
    for (int k = 0; k < n; ++k) {
      const double b = (double)cdiv(Ms[k], cfg_bm(cfg)) * (CO / cfg_bn(cfg));
      blocks += b;
      wblocks += b * weights[k];
      wsum += weights[k];
    }
    if (blocks == 0) continue;
    const double avg_w = wblocks / blocks;                        // average taps per block
    const double rounds = (double)((long)((blocks + 255) / 256));
    const double cost = rounds * avg_w * cfg_bm(cfg) * cfg_bn(cfg) / (eff[cfg] * (blocks < 512 ? 0.9 : 1.0));
    (void)wsum;
    if (best < 0 || cost < best_cost) { best = cfg; best_cost = cost; }
  }
  return best;
}

static inline void make_fwd_geom(IGemmGeom& g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  g = IGemmGeom{};
  g.N = N; g.H = H; g.W = W; g.C = Cin;
  g.OH = conv_out(H, KH, stride, pad); g.OW = conv_out(W, KW, stride, pad);
  g.CO = Cout; g.sy = g.sx = stride;
  g.OHF = g.OH; g.OWF = g.OW; g.osy = g.osx = 1; g.ooy = g.oox = 0;
  g.T = KH * KW; g.M = N * g.OH * g.OW; g.K = g.T * Cin;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) g.tap[kh * KW + kw] = pack_tap(kh - pad, kw - pad, kh * KW + kw);
  g.x_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
  g.w_bytes = (unsigned)((size_t)g.T * Cin * Cout * 4);
}

// Input-gradient geometry of output parity class (py, px): the "input" of the gather-GEMM is dy
// (N,OH,OW,Cout), the "output" is dx (N,H,W,Cin).  T == 0 (1x1 stride-2, odd parity): no tap reaches this
// class; its blocks still run so the epilogue writes dx = residual (or 0) and applies the relu mask there.
static inline void make_dgrad_geom(IGemmGeom& g, int py, int px, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                   int stride, int pad) {
  const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
  g = IGemmGeom{};
  g.N = N; g.H = OH; g.W = OW; g.C = Cout; g.CO = Cin;
  g.OH = (H - py + stride - 1) / stride; g.OW = (W - px + stride - 1) / stride;
  g.sy = g.sx = 1;
  g.OHF = H; g.OWF = W; g.osy = g.osx = stride; g.ooy = py; g.oox = px;
  g.M = N * g.OH * g.OW;
  int T = 0;
  for (int kh = 0; kh < KH; ++kh) {
    if ((py + pad - kh) % stride != 0) continue;
    for (int kw = 0; kw < KW; ++kw) {
      if ((px + pad - kw) % stride != 0) continue;
      g.tap[T++] = pack_tap((py + pad - kh) / stride, (px + pad - kw) / stride, kh * KW + kw);  // exact division
    }
  }
  g.T = T; g.K = T * Cout;
  g.x_bytes = (unsigned)((size_t)N * OH * OW * Cout * 4);
  g.w_bytes = (unsigned)((size_t)KH * KW * Cin * Cout * 4);
}

// BatchNorm-backward reductions fused into an input-gradient launch (IGemmGeom::bn_x): validate the requests, attach them
static inline int attach_bn_reqs(const char* who, IGemmGeom& g, const mla_bn_reduce_req* reqs, int nreq, int tile0) {
  MLA_REQUIRE(nreq >= 0 && nreq <= 2 && (nreq == 0 || reqs), "%s: 0..2 BatchNorm reduction requests", who);
  for (int q = 0; q < nreq; ++q) {
    MLA_REQUIRE(reqs[q].x && reqs[q].mean && reqs[q].invstd && reqs[q].partial, "%s: null pointer in BatchNorm request %d", who, q);
    g.bn_x[q] = reqs[q].x;
    g.bn_mean[q] = reqs[q].mean;
    g.bn_invstd[q] = reqs[q].invstd;
    g.bn_part[q] = reqs[q].partial;
  }
  g.bn_tile0 = tile0;
  return MLA_OK;
}

// Linear layers as 1-tap gather-GEMMs over token rows (see the Linear entry points in conv_igemm.hip)
static inline int linear_geom(const char* who, IGemmGeom& g, int groups, int rows, int in_group_rows, int in_off,
                       int out_group_rows, int out_off, int K, int N) {
  MLA_REQUIRE(groups > 0 && rows > 0 && K > 0 && N > 0, "%s: non-positive dims", who);
  MLA_REQUIRE(K % 64 == 0 && N % 64 == 0 && K <= 8192 && N <= 8192, "%s: K=%d, N=%d must be multiples of 64", who, K, N);
  MLA_REQUIRE(in_off >= 0 && in_off + rows <= in_group_rows && out_off >= 0 && out_off + rows <= out_group_rows && in_off < 128,
              "%s: row window outside its group", who);
  MLA_REQUIRE((long)groups * in_group_rows * K * 4 < 0xFFFFFFF0L && (long)groups * out_group_rows * N * 4 < 0xFFFFFFF0L,
              "%s: tensors must be < 4 GiB", who);
  g = IGemmGeom{};
  g.N = groups; g.H = in_group_rows; g.W = 1; g.C = K; g.CO = N;
  g.OH = rows; g.OW = 1; g.sy = g.sx = 1;
  g.OHF = out_group_rows; g.OWF = 1; g.osy = g.osx = 1; g.ooy = out_off; g.oox = 0;
  g.T = 1; g.M = groups * rows; g.K = K;
  g.tap[0] = pack_tap(in_off, 0, 0);
  g.x_bytes = (unsigned)((size_t)groups * in_group_rows * K * 4);
  g.w_bytes = (unsigned)((size_t)K * N * 4);
  return MLA_OK;
}

