// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950).
//
// Replaces the ATen/cuDNN convolution forward + input-gradient + weight-gradient that
// nn.Conv2d triggers in the reference (models/backbone.py:4-12, 28, 31, 79-83, 127 and autograd).
//
// One "gather-GEMM" kernel serves the forward conv and every input-gradient case:
//     Y[out(m)][co] = sum_t sum_ci X[in(m,t)][ci] * Wt[wt[t]][ci][co]  (+ R) (* relu mask)
// rows m = (n, oy, ox) of a logical OHxOW grid, in(m,t) = (n, oy*sy+dy[t], ox*sx+dx[t]) with
// zero fill outside the tensor, out(m) = (n, oy*osy+ooy, ox*osx+oox).  The forward conv uses
// dy=kh-pad; dgrad uses the per-tap transposed weights and, for stride 2, one launch per output
// parity class so no MFMA work is spent on structurally-zero taps.
//
// Tiling: 256 threads = 4 waves; block tile BMxBN, K step 32 channels of one tap; A (pixels x
// channels) and B (channels x cout) staged through LDS with register prefetch of the next K step;
// every wave owns a (BM/WM)x(BN/WN) sub-tile as 32x32 MFMA tiles.  fp32 MFMA moves 512 B of LDS
// per 64-cycle instruction, so the kernel is MFMA-issue bound; two workgroups per CU hide the
// barrier and the global-load latency of each other.
#include "common.h"

#define MAX_TAPS 49
#define BK 32
#define LDA 36  // A row stride (floats): conflict-free ds_read_b128 for 16-lane groups (9*i mod 16 distinct)

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
  signed char dy[MAX_TAPS], dx[MAX_TAPS];
  unsigned char wt[MAX_TAPS];
};

// ---------------------------------------------------------------------------------------------
// MFMA over one staged K step.  As: [rows][LDA] (k contiguous), Bs: [BK][LDBS] (cout contiguous).
// Lane l = (i = l&31, h = l>>5).  One ds_read_b128 of A gives k = kk*8+4h+{0..3}; the matching B
// values are 4 ds_read_b32.  MFMA step jj sums k in {kk*8+jj, kk*8+4+jj}: all 32 k covered once.
// ---------------------------------------------------------------------------------------------
template <int MI, int NI, int LDBS>
__device__ __forceinline__ void mma_kstep(const float* __restrict__ As_w, const float* __restrict__ Bs_w,
                                          f32x16 (&acc)[MI][NI], int lane) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) {
    f32x4 a[MI];
    float b[NI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      a[mi] = *reinterpret_cast<const f32x4*>(As_w + (mi * 32 + i) * LDA + kk * 8 + 4 * h);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) b[ni][jj] = Bs_w[(kk * 8 + 4 * h + jj) * LDBS + ni * 32 + i];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][jj], b[ni][jj], acc[mi][ni], 0, 0, 0);
  }
}

// SCALAR_A = false: C % 32 == 0, float4 gathers (all convs except the stem).
// SCALAR_A = true : tiny C (stem: 1 or 3), K = T*C flattened, element-wise gather, zero-padded K.
template <int BM, int BN, int WM, int WN, bool SCALAR_A>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                        float* Y, const float* R, const float* MASK,
                                                        float* __restrict__ part, const IGemmGeom g) {
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int LDBS = BN;
  constexpr int APASS = BM / 32;            // float4 passes for A (8 float4 per row, 32 rows per pass)
  constexpr int BROWS = 256 / (BN / 4);     // B rows per pass
  constexpr int BPASS = BK / BROWS;
  static_assert(WM * WN == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) float As[BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDBS];
  __shared__ int4 rowinfo[BM];  // {n*H*W or -1, oy*sy, ox*sx, output pixel index}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = g.CO / BN;
  const int nwg = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tm = wg / gridN, tn = wg % gridN;

  for (int r = tid; r < BM; r += 256) {
    const int m = tm * BM + r;
    int4 info = make_int4(-1, 0, 0, 0);
    if (m < g.M) {
      const int ohw = g.OH * g.OW;
      const int n = m / ohw, rem = m - n * ohw;
      const int oy = rem / g.OW, ox = rem - oy * g.OW;
      info.x = n * g.H * g.W;
      info.y = oy * g.sy;
      info.z = ox * g.sx;
      info.w = (n * g.OHF + oy * g.osy + g.ooy) * g.OWF + ox * g.osx + g.oox;
    }
    rowinfo[r] = info;
  }
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int cpt = SCALAR_A ? 1 : g.C / BK;                        // K steps per tap
  const int nIter = SCALAR_A ? (g.K + BK - 1) / BK : g.T * cpt;

  f32x4 areg[SCALAR_A ? 1 : APASS];
  float asc[SCALAR_A ? BM / 8 : 1];
  f32x4 breg[BPASS];

  auto load_tiles = [&](int it) {
    if constexpr (!SCALAR_A) {
      const int t = it / cpt, c0 = (it - t * cpt) * BK;
      const int dy = g.dy[t], dx = g.dx[t];
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int4 info = rowinfo[p * 32 + (tid >> 3)];
        const int iy = info.y + dy, ix = info.z + dx;
        const bool ok = info.x >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(X + (size_t)(info.x + iy * g.W + ix) * g.C + c0 + (tid & 7) * 4);
        areg[p] = v;
      }
      const float* wsrc = Wt + ((size_t)g.wt[t] * g.C + c0) * g.CO + tn * BN;
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int row = p * BROWS + tid / (BN / 4), c4 = tid % (BN / 4);
        breg[p] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)row * g.CO + c4 * 4);
      }
    } else {
      const int kg = it * BK + (tid & 31);
      const bool kok = kg < g.K;
      const int t = kok ? kg / g.C : 0, ci = kok ? kg - t * g.C : 0;
      const int dy = g.dy[t], dx = g.dx[t];
#pragma unroll
      for (int p = 0; p < BM / 8; ++p) {
        const int4 info = rowinfo[p * 8 + (tid >> 5)];
        const int iy = info.y + dy, ix = info.z + dx;
        const bool ok = kok && info.x >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        asc[p] = ok ? X[(size_t)(info.x + iy * g.W + ix) * g.C + ci] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int row = p * BROWS + tid / (BN / 4), c4 = tid % (BN / 4);
        const int kr = it * BK + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kr < g.K) v = *reinterpret_cast<const f32x4*>(Wt + (size_t)kr * g.CO + tn * BN + c4 * 4);
        breg[p] = v;
      }
    }
  };
  auto store_tiles = [&]() {
    if constexpr (!SCALAR_A) {
#pragma unroll
      for (int p = 0; p < APASS; ++p)
        *reinterpret_cast<f32x4*>(&As[(p * 32 + (tid >> 3)) * LDA + (tid & 7) * 4]) = areg[p];
    } else {
#pragma unroll
      for (int p = 0; p < BM / 8; ++p) As[(p * 8 + (tid >> 5)) * LDA + (tid & 31)] = asc[p];
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) {
      const int row = p * BROWS + tid / (BN / 4), c4 = tid % (BN / 4);
      *reinterpret_cast<f32x4*>(&Bs[row * LDBS + c4 * 4]) = breg[p];
    }
  };

  if (nIter > 0) {  // nIter == 0: a dgrad parity class no tap reaches (1x1 stride 2): epilogue only
    load_tiles(0);
    store_tiles();
  }
  __syncthreads();
  const float* As_w = As + wm * (BM / WM) * LDA;
  const float* Bs_w = Bs + wn * (BN / WN);
  for (int it = 0; it < nIter; ++it) {
    const bool more = it + 1 < nIter;
    if (more) load_tiles(it + 1);
    mma_kstep<MI, NI, LDBS>(As_w, Bs_w, acc, lane);
    __syncthreads();
    if (more) {
      store_tiles();
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int h = lane >> 5, j = lane & 31;
  float csum[NI], csq[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) csum[ni] = csq[ni] = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * (BM / WM) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int4 info = rowinfo[row];
      if (info.x < 0) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = tn * BN + wn * (BN / WN) + ni * 32 + j;
        const size_t idx = (size_t)info.w * g.CO + col;
        float v = acc[mi][ni][e];
        csum[ni] += v;
        csq[ni] += v * v;
        if (R) v += R[idx];
        if (MASK) v = MASK[idx] > 0.f ? v : 0.f;
        Y[idx] = v;
      }
    }
  }
  if (part) {  // fused BatchNorm statistics: per-tile column sum / sum of squares
    float* red = As;  // reuse (all waves are past their last LDS read: trailing __syncthreads above)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      csum[ni] += __shfl_xor(csum[ni], 32, 64);
      csq[ni] += __shfl_xor(csq[ni], 32, 64);
      if (h == 0) {
        const int c = wn * (BN / WN) + ni * 32 + j;
        red[(wm * 2 + 0) * BN + c] = csum[ni];
        red[(wm * 2 + 1) * BN + c] = csq[ni];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += red[(w * 2 + 0) * BN + tid];
        q += red[(w * 2 + 1) * BN + tid];
      }
      part[((size_t)tm * 2 + 0) * g.CO + tn * BN + tid] = s;
      part[((size_t)tm * 2 + 1) * g.CO + tn * BN + tid] = q;
    }
  }
}

// per-tap transpose of HWIO weights: out[t][co][ci] = in[t][ci][co]
__global__ void weight_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int CI, int CO) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < CI && co < CO) ? in[((size_t)t * CI + ci) * CO + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (ci < CI && co < CO) out[((size_t)t * CO + co) * CI + ci] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[t][ci][co] = sum_m X[in(m,t)][ci] * dY[m][co]  (TN GEMM, K = pixels).
// grid.x = (ci tiles) x (co tiles) x taps, grid.y = split-K chunks of `chunk` pixels.  Partial
// slabs part[kc][t][ci][co] are summed in order by wgrad_reduce_kernel (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
#define WG_MAXCHUNK 2048
template <int BI, int BJ, int WI, int WJ, bool SCALAR_A>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                        float* __restrict__ part, const IGemmGeom g, int chunk) {
  constexpr int MI = BI / WI / 32, NI = BJ / WJ / 32;
  constexpr int XROWS = 256 / (BI / 4), XPASS = BK / XROWS;
  constexpr int YROWS = 256 / (BJ / 4), YPASS = BK / YROWS;
  __shared__ __attribute__((aligned(16))) float Xs[BK * BI];
  __shared__ __attribute__((aligned(16))) float Ys[BK * BJ];
  __shared__ int2 rowinfo[WG_MAXCHUNK];  // {n*H*W or -1, (oy*sy)<<16 | (ox*sx)&0xffff}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  const int tilesI = SCALAR_A ? (g.K + BI - 1) / BI : g.C / BI, tilesJ = g.CO / BJ;
  int b = blockIdx.x;
  const int tj = b % tilesJ; b /= tilesJ;
  const int ti = b % tilesI; b /= tilesI;
  const int t = b;  // tap (0 in scalar mode)
  const int m_begin = blockIdx.y * chunk, m_end = min(g.M, m_begin + chunk);

  for (int r = tid; r < chunk; r += 256) {
    const int m = m_begin + r;
    int2 info = make_int2(-1, 0);
    if (m < m_end) {
      const int ohw = g.OH * g.OW;
      const int n = m / ohw, rem = m - n * ohw;
      const int oy = rem / g.OW, ox = rem - oy * g.OW;
      info.x = n * g.H * g.W;
      info.y = ((oy * g.sy) << 16) | ((ox * g.sx) & 0xffff);
    }
    rowinfo[r] = info;
  }
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x4 xreg[SCALAR_A ? 1 : XPASS];
  float xsc[SCALAR_A ? BK * BI / 256 : 1];
  f32x4 yreg[YPASS];
  int dy = 0, dx = 0, ci_s = 0;
  bool kok = true;
  if constexpr (SCALAR_A) {
    const int kg = ti * BI + (tid % BI);
    kok = kg < g.K;
    const int tt = kok ? kg / g.C : 0;
    ci_s = kok ? kg - tt * g.C : 0;
    dy = g.dy[tt];
    dx = g.dx[tt];
  } else {
    dy = g.dy[t];
    dx = g.dx[t];
  }

  auto load_tiles = [&](int p0) {  // p0: first pixel (chunk-relative) of this K step
    if constexpr (!SCALAR_A) {
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int r = p0 + p * XROWS + tid / (BI / 4);
        const int2 info = rowinfo[r];
        const int iy = (info.y >> 16) + dy, ix = (int)(short)(info.y & 0xffff) + dx;
        const bool ok = info.x >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(X + (size_t)(info.x + iy * g.W + ix) * g.C + ti * BI + (tid % (BI / 4)) * 4);
        xreg[p] = v;
      }
    } else {
#pragma unroll
      for (int p = 0; p < BK * BI / 256; ++p) {
        const int r = p0 + p * (256 / BI) + tid / BI;
        const int2 info = rowinfo[r];
        const int iy = (info.y >> 16) + dy, ix = (int)(short)(info.y & 0xffff) + dx;
        const bool ok = kok && info.x >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        xsc[p] = ok ? X[(size_t)(info.x + iy * g.W + ix) * g.C + ci_s] : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < YPASS; ++p) {
      const int r = p0 + p * YROWS + tid / (BJ / 4);
      const int m = m_begin + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) v = *reinterpret_cast<const f32x4*>(dY + (size_t)m * g.CO + tj * BJ + (tid % (BJ / 4)) * 4);
      yreg[p] = v;
    }
  };
  auto store_tiles = [&]() {
    if constexpr (!SCALAR_A) {
#pragma unroll
      for (int p = 0; p < XPASS; ++p)
        *reinterpret_cast<f32x4*>(&Xs[(p * XROWS + tid / (BI / 4)) * BI + (tid % (BI / 4)) * 4]) = xreg[p];
    } else {
#pragma unroll
      for (int p = 0; p < BK * BI / 256; ++p) Xs[(p * (256 / BI) + tid / BI) * BI + (tid % BI)] = xsc[p];
    }
#pragma unroll
    for (int p = 0; p < YPASS; ++p)
      *reinterpret_cast<f32x4*>(&Ys[(p * YROWS + tid / (BJ / 4)) * BJ + (tid % (BJ / 4)) * 4]) = yreg[p];
  };

  const int nIter = (m_end - m_begin + BK - 1) / BK;
  const int i = lane & 31, h = lane >> 5;
  if (nIter > 0) {
    load_tiles(0);
    store_tiles();
  }
  __syncthreads();
  for (int it = 0; it < nIter; ++it) {
    const bool more = it + 1 < nIter;
    if (more) load_tiles((it + 1) * BK);
    const float* xa = Xs + wi * (BI / WI) + i;
    const float* yb = Ys + wj * (BJ / WJ) + i;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float a[MI], bb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = xa[(2 * s + h) * BI + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bb[ni] = yb[(2 * s + h) * BJ + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_tiles();
      __syncthreads();
    }
  }

  const int KR = SCALAR_A ? g.K : g.C;  // rows of one tap slab
  float* slab = part + ((size_t)blockIdx.y * (SCALAR_A ? 1 : g.T) + t) * KR * g.CO;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ti * BI + wi * (BI / WI) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= KR) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = tj * BJ + wj * (BJ / WJ) + ni * 32 + i;
        slab[(size_t)row * g.CO + col] = acc[mi][ni][e];
      }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n4, int splits) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(part)[idx];
  for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(part)[(size_t)k * n4 + idx];
  reinterpret_cast<f32x4*>(dw)[idx] = s;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

static int check_conv(const char* who, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  MLA_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "%s: non-positive dims", who);
  MLA_REQUIRE(KH * KW <= MAX_TAPS && KH > 0 && KW > 0, "%s: kernel %dx%d unsupported (max %d taps)", who, KH, KW, MAX_TAPS);
  MLA_REQUIRE(stride == 1 || stride == 2, "%s: stride %d unsupported", who, stride);
  MLA_REQUIRE(Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", who, Cout);
  MLA_REQUIRE(Cin % 64 == 0 || Cin <= 4, "%s: Cin=%d must be a multiple of 64 or <= 4 (stem)", who, Cin);
  MLA_REQUIRE(pad >= 0 && pad < 64 && H < 32768 && W < 32768, "%s: pad/size out of range", who);
  MLA_REQUIRE((long)N * H * W < (1L << 31) / 4, "%s: too many pixels for 32-bit pixel indices", who);
  return MLA_OK;
}

static int launch_igemm(const float* X, const float* Wt, float* Y, const float* R, const float* MASK, float* part,
                        const IGemmGeom& g, bool scalar, hipStream_t st) {
  if (g.M <= 0) return MLA_OK;
  if (scalar) {
    const int gm = cdiv(g.M, 256);
    igemm_kernel<256, 64, 4, 1, true><<<gm * (g.CO / 64), 256, 0, st>>>(X, Wt, Y, R, MASK, part, g);
  } else if (g.CO % 128 == 0) {
    const int gm = cdiv(g.M, 128);
    igemm_kernel<128, 128, 2, 2, false><<<gm * (g.CO / 128), 256, 0, st>>>(X, Wt, Y, R, MASK, part, g);
  } else {
    const int gm = cdiv(g.M, 256);
    igemm_kernel<256, 64, 4, 1, false><<<gm * (g.CO / 64), 256, 0, st>>>(X, Wt, Y, R, MASK, part, g);
  }
  MLA_CHECK_LAUNCH("igemm_kernel");
  return MLA_OK;
}

static int fwd_tile_m(int Cin, int Cout) { return (Cin % 64 == 0 && Cout % 128 == 0) ? 128 : 256; }

extern "C" size_t mla_conv2d_fwd_partial_elems(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  const long M = (long)N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad);
  return (size_t)cdiv(M, fwd_tile_m(Cin, Cout)) * 2 * Cout;
}

extern "C" int mla_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                              int KW, int stride, int pad, float* bn_partial, int* bn_tiles, void* stream) {
  if (int rc = check_conv("mla_conv2d_fwd", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && w && y, "mla_conv2d_fwd: null pointer");
  IGemmGeom g{};
  g.N = N; g.H = H; g.W = W; g.C = Cin;
  g.OH = conv_out(H, KH, stride, pad); g.OW = conv_out(W, KW, stride, pad);
  MLA_REQUIRE(g.OH > 0 && g.OW > 0, "mla_conv2d_fwd: empty output");
  g.CO = Cout; g.sy = g.sx = stride;
  g.OHF = g.OH; g.OWF = g.OW; g.osy = g.osx = 1; g.ooy = g.oox = 0;
  g.T = KH * KW; g.M = N * g.OH * g.OW; g.K = g.T * Cin;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      const int t = kh * KW + kw;
      g.dy[t] = (signed char)(kh - pad); g.dx[t] = (signed char)(kw - pad); g.wt[t] = (unsigned char)t;
    }
  if (bn_tiles) *bn_tiles = cdiv(g.M, fwd_tile_m(Cin, Cout));
  return launch_igemm(x, w, y, nullptr, nullptr, bn_partial, g, Cin % 64 != 0, (hipStream_t)stream);
}

extern "C" int mla_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                float* wt_ws, void* stream) {
  if (int rc = check_conv("mla_conv2d_dgrad", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(Cin % 64 == 0, "mla_conv2d_dgrad: Cin=%d must be a multiple of 64 (the stem needs no dgrad)", Cin);
  MLA_REQUIRE(dy && w && dx && wt_ws, "mla_conv2d_dgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
  weight_transpose_kernel<<<dim3(cdiv(Cout, 32), cdiv(Cin, 32), KH * KW), 256, 0, st>>>(w, wt_ws, KH * KW, Cin, Cout);
  MLA_CHECK_LAUNCH("weight_transpose_kernel");
  // "input" of the gather-GEMM is dy (N,OH,OW,Cout); "output" is dx (N,H,W,Cin)
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      IGemmGeom g{};
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
          // exact division of a possibly negative even number
          g.dy[T] = (signed char)((py + pad - kh) / stride);
          g.dx[T] = (signed char)((px + pad - kw) / stride);
          g.wt[T] = (unsigned char)(kh * KW + kw);
          ++T;
        }
      }
      g.T = T; g.K = T * Cout;
      if (g.M <= 0) continue;
      // T == 0 (1x1 stride-2, odd parity): no tap reaches this class; the launch still runs so the
      // epilogue writes dx = residual (or 0) and applies the relu mask there.
      if (int rc = launch_igemm(dy, wt_ws, dx, residual, relu_src, nullptr, g, false, st)) return rc;
    }
  return MLA_OK;
}

static void wgrad_plan(long M, int Cin, int Cout, int T, int* chunk, int* splits) {
  // enough workgroups to fill 256 CUs x 2, chunk a multiple of 32 pixels, <= WG_MAXCHUNK
  const bool scalar = Cin % 64 != 0;
  const int BI = scalar ? 64 : (Cin % 128 == 0 && Cout % 128 == 0 ? 128 : 64);
  const int BJ = scalar ? 64 : BI;
  const long tiles = (long)(scalar ? cdiv((long)T * Cin, BI) : (Cin / BI) * T) * (Cout / BJ);
  long want = (1024 + tiles - 1) / tiles;  // target ~1024 workgroups
  long c = (M + want - 1) / want;
  c = ((c + 31) / 32) * 32;
  if (c > WG_MAXCHUNK) c = WG_MAXCHUNK;
  if (c < 256) c = 256;
  *chunk = (int)c;
  *splits = (int)((M + c - 1) / c);
}

extern "C" size_t mla_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  const long M = (long)N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad);
  int chunk, splits;
  wgrad_plan(M, Cin, Cout, KH * KW, &chunk, &splits);
  return (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
}

extern "C" int mla_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_conv("mla_conv2d_wgrad", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && dy && dw && ws, "mla_conv2d_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  IGemmGeom g{};
  g.N = N; g.H = H; g.W = W; g.C = Cin; g.CO = Cout;
  g.OH = conv_out(H, KH, stride, pad); g.OW = conv_out(W, KW, stride, pad);
  g.sy = g.sx = stride; g.T = KH * KW; g.M = N * g.OH * g.OW; g.K = g.T * Cin;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      const int t = kh * KW + kw;
      g.dy[t] = (signed char)(kh - pad); g.dx[t] = (signed char)(kw - pad); g.wt[t] = (unsigned char)t;
    }
  int chunk, splits;
  wgrad_plan(g.M, Cin, Cout, g.T, &chunk, &splits);
  const size_t need = (size_t)splits * g.T * Cin * Cout * sizeof(float);
  if (ws_bytes < need) {
    mla_set_error("mla_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  const bool scalar = Cin % 64 != 0;
  if (scalar) {
    dim3 grid(cdiv(g.K, 64) * (Cout / 64), splits);
    wgrad_kernel<64, 64, 2, 2, true><<<grid, 256, 0, st>>>(x, dy, part, g, chunk);
  } else if (Cin % 128 == 0 && Cout % 128 == 0) {
    dim3 grid((Cin / 128) * (Cout / 128) * g.T, splits);
    wgrad_kernel<128, 128, 2, 2, false><<<grid, 256, 0, st>>>(x, dy, part, g, chunk);
  } else {
    dim3 grid((Cin / 64) * (Cout / 64) * g.T, splits);
    wgrad_kernel<64, 64, 2, 2, false><<<grid, 256, 0, st>>>(x, dy, part, g, chunk);
  }
  MLA_CHECK_LAUNCH("wgrad_kernel");
  const size_t n4 = (size_t)g.T * Cin * Cout / 4;
  wgrad_reduce_kernel<<<cdiv(n4, 256), 256, 0, st>>>(part, dw, n4, splits);
  MLA_CHECK_LAUNCH("wgrad_reduce_kernel");
  return MLA_OK;
}
