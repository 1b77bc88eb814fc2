// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950).
//
// Replaces the ATen/cuDNN convolution forward + input-gradient + weight-gradient that
// nn.Conv2d triggers in the reference (models/backbone.py:4-12, 28, 31, 79-83, 127 and autograd).
//
// One "gather-GEMM" kernel serves the forward conv and every input-gradient case:
//     Y[out(m)][co] = sum_t sum_ci X[in(m,t)][ci] * Wt[wt[t]][ci][co]  (+ R) (* relu mask)
// rows m = (n, oy, ox) of a logical OHxOW grid, in(m,t) = (n, oy*sy+dy[t], ox*sx+dx[t]) with
// zero fill outside the tensor, out(m) = (n, oy*osy+ooy, ox*osx+oox).  The forward conv uses
// dy=kh-pad; dgrad uses the per-tap transposed weights and, for stride 2, one geometry per output
// parity class (no MFMA work on structurally-zero taps), one launch per class.
//
// Tiling: 256 threads = 4 waves; block tile BMxBN (128x128, 256x64, 128x64 or 64x64, picked per problem to
// minimise the tail on 256 CUs), K step 32 channels of one tap; A (pixels x channels) and B
// (channels x cout) staged through LDS with register prefetch of the next K step; every wave owns a
// (BM/WM)x(BN/WN) sub-tile as 32x32 MFMA tiles.  fp32 MFMA moves 512 B of LDS per 64-cycle
// instruction, so the kernel is MFMA-issue bound; >= 2 workgroups per CU hide each other's barriers.
// Gathers are raw buffer loads (hardware range check = zero padding), row offsets and tap-validity masks
// are per-thread constants and the tap table is read with scalar loads, so a K step spends ~35 VALU
// instructions on addressing and nothing in the loop waits on vmcnt except the LDS store of the prefetch.
#include "igemm_common.h"

#define LDA 36  // A row stride (floats): conflict-free ds_read_b128 for 16-lane groups (9*i mod 16 distinct)

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
                                                        float* __restrict__ part, const float* __restrict__ BIAS,
                                                        float* __restrict__ Y2, const IGemmGeom g) {
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
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int gridN = g.CO / BN;
  int tm, tn;
  tile_coords(wg, (int)gridDim.x / gridN, gridN, tm, tn);
  const int gH = g.H, gW = g.W, gC = g.C, gCO = g.CO;

  for (int r = tid; r < BM; r += 256) {
    const int m = tm * BM + r;
    int4 info = make_int4(-1, -100000, -100000, 0);
    if (m < g.M) {
      const int ohw = g.OH * g.OW;
      const int n = m / ohw, rem = m - n * ohw;
      const int oy = rem / g.OW, ox = rem - oy * g.OW;
      info.x = n * gH * gW;
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

  const int cpt = SCALAR_A ? 1 : gC / BK;                         // K steps per tap
  const int nIter = SCALAR_A ? (g.K + BK - 1) / BK : g.T * cpt;

  f32x4 areg[SCALAR_A ? 1 : APASS];
  float asc[SCALAR_A ? BM / 8 : 1];
  f32x4 breg[BPASS];

  const rsrc_t xr = make_rsrc(X, g.x_bytes), wr = make_rsrc(Wt, g.w_bytes);
  // Per-thread gather state, computed once: byte offset of each of this thread's rows at tap (0,0) and a
  // bitmask of the taps that fall inside the image.  Per K step a row then costs 3 VALU ops + 1 buffer load.
  unsigned rowoff[SCALAR_A ? 1 : APASS], tmask[SCALAR_A ? 1 : APASS], boff[BPASS];
  if constexpr (!SCALAR_A) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int4 info = rowinfo[p * 32 + (tid >> 3)];
      rowoff[p] = ((unsigned)(info.x + info.y * gW + info.z) * (unsigned)gC + (tid & 7) * 4) * 4u;
      unsigned m = 0;
      for (int t = 0; t < g.T; ++t) {
        const int tp = g.tap[t];
        const int iy = info.y + tap_dy(tp), ix = info.z + tap_dx(tp);
        m |= ((unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW) ? (1u << t) : 0u;   // invalid rows: -100000
      }
      tmask[p] = m;
    }
  }
#pragma unroll
  for (int p = 0; p < BPASS; ++p) boff[p] = ((p * BROWS + tid / (BN / 4)) * gCO + (tid % (BN / 4)) * 4) * 4u;

  auto load_tiles = [&](int it) {
    if constexpr (!SCALAR_A) {
      const int t = it / cpt, c0 = (it - t * cpt) * BK;
      const int tp = g.tap[t];
      const unsigned toff = (unsigned)(((tap_dy(tp) * gW + tap_dx(tp)) * gC + c0) * 4);   // wave-uniform (SGPR)
#pragma unroll
      for (int p = 0; p < APASS; ++p)
        areg[p] = buf_load4(xr, ((tmask[p] >> t) & 1u) ? rowoff[p] + toff : OOB_OFF, 0);
      const unsigned wsoff = (unsigned)(((tap_wt(tp) * gC + c0) * gCO + tn * BN) * 4);    // wave-uniform
#pragma unroll
      for (int p = 0; p < BPASS; ++p) breg[p] = buf_load4(wr, boff[p], wsoff);
    } else {
      const int kg = it * BK + (tid & 31);
      const bool kok = kg < g.K;
      const int t = kok ? kg / gC : 0, ci = kok ? kg - t * gC : 0;
      const int tp = g.tap[t];
      const int dy = tap_dy(tp), dx = tap_dx(tp);
#pragma unroll
      for (int p = 0; p < BM / 8; ++p) {
        const int4 info = rowinfo[p * 8 + (tid >> 5)];
        const int iy = info.y + dy, ix = info.z + dx;
        const bool ok = kok && (unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW;
        asc[p] = buf_load1(xr, ok ? ((unsigned)(info.x + iy * gW + ix) * (unsigned)gC + ci) * 4u : OOB_OFF, 0);
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int kr = it * BK + p * BROWS + tid / (BN / 4);
        breg[p] = buf_load4(wr, kr < g.K ? (unsigned)(kr * gCO + tn * BN + (tid % (BN / 4)) * 4) * 4u : OOB_OFF, 0);
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

  // Single LDS buffer + register prefetch, two barriers per K step.  Measured alternatives (a.l2/a.l3/a.l4
  // shapes, same run): double-buffered LDS with one barrier per step -2 % (70 KB LDS -> 2 instead of 3
  // resident workgroups), also in the branch-free peeled form that pays off for the split-bf16 kernels (-3...5 %: here
  // a K step is 4096 MFMA cycles per wave, so there is little staging to hide); staggering co-resident workgroups +-0 %.  Timing-only ablations put the ceiling of
  // this structure (LDS reads + MFMA only) at 147 TF, global loads cost ~5 %, LDS stores + 2nd barrier ~5 %;
  // what recovers them is more resident workgroups per CU, hence the 64x64 tile for small / ragged problems.
  const float* As_w = As + wm * (BM / WM) * LDA;
  const float* Bs_w = Bs + wn * (BN / WN);
  if (nIter > 0) {  // nIter == 0: a dgrad parity class no tap reaches (1x1 stride 2): epilogue only
    load_tiles(0);
    store_tiles();
  }
  __syncthreads();
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

  igemm_epilogue<BM, BN, WM, WN>(acc, rowinfo, As, Y, R, MASK, part, BIAS, Y2, g, tm, tn);
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
// grid.x = (ci tiles) x (co tiles) x taps, grid.y = split-K ranges of `span` pixels, walked in
// sub-chunks of WG_CHUNK pixels (row table in LDS).  Partial slabs part[split][t][ci][co] are summed
// in a fixed order by wgrad_reduce_kernel (bitwise reproducible, no atomics).
// ---------------------------------------------------------------------------------------------
#define WG_CHUNK 1024
template <int BI, int BJ, int WI, int WJ, bool SCALAR_A>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                        float* __restrict__ part, const IGemmGeom g, int span) {
  constexpr int MI = BI / WI / 32, NI = BJ / WJ / 32;
  constexpr int XROWS = 256 / (BI / 4), XPASS = BK / XROWS;
  constexpr int YROWS = 256 / (BJ / 4), YPASS = BK / YROWS;
  __shared__ __attribute__((aligned(16))) float Xs[BK * BI];
  __shared__ __attribute__((aligned(16))) float Ys[BK * BJ];
  __shared__ int2 rowinfo[WG_CHUNK];  // vector mode: {byte offset of the gathered X row for THIS block's tap or OOB_OFF, -};
                                      // scalar mode: {n*H*W or -1, (oy*sy)<<16 | (ox*sx)&0xffff}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  const int tilesI = SCALAR_A ? (g.K + BI - 1) / BI : g.C / BI, tilesJ = g.CO / BJ;
  // all workgroups of one pixel span (every tap and channel tile) on ONE XCD: they gather the same X / dY rows, which its
  // L2 then serves 9 x tiles times; in dispatch order they would be spread round-robin over the 8 XCDs (8 L2 fills per span).
  // Same-box A/B: 64x64 tiles (layer1) -4...5 %, 128x128 tiles +-0.
  const int lwg = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = lwg / (int)gridDim.x;
  int b = lwg - by * (int)gridDim.x;
  const int tj = b % tilesJ; b /= tilesJ;
  const int ti = b % tilesI; b /= tilesI;
  const int t = b;  // tap (0 in scalar mode)
  const int m_begin = by * span, m_end = min(g.M, m_begin + span);
  const int gH = g.H, gW = g.W, gC = g.C, gCO = g.CO;

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
    const int tt = kok ? kg / gC : 0;
    ci_s = kok ? kg - tt * gC : 0;
    dy = tap_dy(g.tap[tt]);
    dx = tap_dx(g.tap[tt]);
  } else {
    dy = tap_dy(g.tap[t]);
    dx = tap_dx(g.tap[t]);
  }
  const int i = lane & 31, h = lane >> 5;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, g.y_bytes);

  for (int c_begin = m_begin; c_begin < m_end; c_begin += WG_CHUNK) {
    const int c_end = min(m_end, c_begin + WG_CHUNK);
    __syncthreads();  // previous sub-chunk's readers are done with rowinfo / Xs / Ys
    for (int r = tid; r < WG_CHUNK; r += 256) {
      const int m = c_begin + r;
      int2 info = make_int2(SCALAR_A ? -1 : (int)OOB_OFF, (int)0x80008000);   // coordinates -32768: never in range
      if (m < c_end) {
        const int ohw = g.OH * g.OW;
        const int n = m / ohw, rem = m - n * ohw;
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        if constexpr (SCALAR_A) {
          info.x = n * gH * gW;
          info.y = ((oy * g.sy) << 16) | ((ox * g.sx) & 0xffff);
        } else {
          const int iy = oy * g.sy + dy, ix = ox * g.sx + dx;
          if ((unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW)
            info.x = (int)(((unsigned)(n * gH * gW + iy * gW + ix) * (unsigned)gC + ti * BI) * 4u);
        }
      }
      rowinfo[r] = info;
    }
    __syncthreads();

    auto load_tiles = [&](int p0) {  // p0: first pixel (sub-chunk-relative) of this K step
      if constexpr (!SCALAR_A) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
          const unsigned ro = (unsigned)rowinfo[p0 + p * XROWS + tid / (BI / 4)].x;
          xreg[p] = buf_load4(xr, ro == OOB_OFF ? OOB_OFF : ro + (tid % (BI / 4)) * 16u, 0);   // no wrap past OOB_OFF
        }
      } else {
#pragma unroll
        for (int p = 0; p < BK * BI / 256; ++p) {
          const int2 info = rowinfo[p0 + p * (256 / BI) + tid / BI];
          const int iy = (info.y >> 16) + dy, ix = (int)(short)(info.y & 0xffff) + dx;
          const bool ok = kok && (unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW;
          xsc[p] = buf_load1(xr, ok ? ((unsigned)(info.x + iy * gW + ix) * (unsigned)gC + ci_s) * 4u : OOB_OFF, 0);
        }
      }
#pragma unroll
      for (int p = 0; p < YPASS; ++p) {
        const int m = c_begin + p0 + p * YROWS + tid / (BJ / 4);
        yreg[p] = buf_load4(yr, m < c_end ? ((unsigned)m * (unsigned)gCO + tj * BJ + (tid % (BJ / 4)) * 4) * 4u : OOB_OFF, 0);
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

    const int nIter = (c_end - c_begin + BK - 1) / BK;
    load_tiles(0);
    store_tiles();
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
  }

  const int KR = SCALAR_A ? g.K : gC;  // rows of one tap slab
  float* slab = part + ((size_t)by * (SCALAR_A ? 1 : g.T) + t) * KR * gCO;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ti * BI + wi * (BI / WI) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= KR) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = tj * BJ + wj * (BJ / WJ) + ni * 32 + i;
        slab[(size_t)row * gCO + col] = acc[mi][ni][e];
      }
    }
}

// dw[idx] = sum over splits, in a fixed order: split-lane l adds splits l, l+L, ... then the lanes are
// added 0..L-1.  256 threads = (256/L) float4 columns x L split lanes.
template <int L>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            size_t n4, int splits) {
  __shared__ f32x4 red[256];
  constexpr int COLS = 256 / L;
  const int col = threadIdx.x % COLS, l = threadIdx.x / COLS;
  const size_t idx = (size_t)blockIdx.x * COLS + col;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (idx < n4)
    for (int k = l; k < splits; k += L) s += reinterpret_cast<const f32x4*>(part)[(size_t)k * n4 + idx];
  if (L > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (l == 0) {
#pragma unroll
      for (int k = 1; k < L; ++k) s += red[k * COLS + col];
    }
  }
  if (l == 0 && idx < n4) reinterpret_cast<f32x4*>(dw)[idx] = s;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern "C" size_t mla_bn_partial_scratch_elems(int C);   // bn.hip: scratch tail every BN partial buffer carries
int g_f32_cfg = -1;
extern "C" int mla_conv2d_f32_cfg(int cfg) { g_f32_cfg = (cfg >= 0 && cfg < CFG_COUNT) ? cfg : -1; return g_f32_cfg; }

static int launch_igemm(const float* X, const float* Wt, float* Y, const float* R, const float* MASK, float* part,
                        const IGemmGeom& mg, bool scalar, int cfg, hipStream_t st, const float* BIAS = nullptr,
                        float* Y2 = nullptr) {
  const int total = cdiv(mg.M, cfg_bm(cfg)) * (mg.CO / cfg_bn(cfg));
  if (total <= 0) return MLA_OK;
  if (scalar) {
    if (cfg == CFG_64x64) igemm_kernel<64, 64, 2, 2, true><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
    else igemm_kernel<256, 64, 4, 1, true><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
  } else if (cfg == CFG_128x128) {
    igemm_kernel<128, 128, 2, 2, false><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
  } else if (cfg == CFG_256x64) {
    igemm_kernel<256, 64, 4, 1, false><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
  } else if (cfg == CFG_128x64) {
    igemm_kernel<128, 64, 2, 2, false><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
  } else {
    igemm_kernel<64, 64, 2, 2, false><<<total, 256, 0, st>>>(X, Wt, Y, R, MASK, part, BIAS, Y2, mg);
  }
  MLA_CHECK_LAUNCH("igemm_kernel");
  return MLA_OK;
}

static int fwd_cfg(long M, int Cin, int Cout) {
  const int w = 1;
  return pick_cfg(&M, &w, 1, Cout, Cin % 64 != 0);
}

extern "C" size_t mla_conv2d_fwd_partial_elems(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  const long M = (long)N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad);
  return (size_t)cdiv(M, 64) * 2 * Cout * 2 + mla_bn_partial_scratch_elems(Cout);  // fp64 partials (2 floats each), upper bound over the tile choices, + reduce scratch
}

extern "C" int mla_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                              int KW, int stride, int pad, float* bn_partial, int* bn_tiles, void* stream) {
  if (int rc = check_conv("mla_conv2d_fwd", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && w && y, "mla_conv2d_fwd: null pointer");
  IGemmGeom g;
  make_fwd_geom(g, N, H, W, Cin, Cout, KH, KW, stride, pad);
  MLA_REQUIRE(g.OH > 0 && g.OW > 0, "mla_conv2d_fwd: empty output");
  const int cfg = fwd_cfg(g.M, Cin, Cout);
  if (bn_tiles) *bn_tiles = cdiv(g.M, cfg_bm(cfg));
  return launch_igemm(x, w, y, nullptr, nullptr, bn_partial, g, Cin % 64 != 0, cfg, (hipStream_t)stream);
}

extern "C" size_t mla_conv2d_dgrad_bn_partial_elems(int N, int H, int W, int Cin) {
  // [tiles][2][Cin] floats, tiles <= ceil(pixels / 64) + one ragged tile per stride-2 parity class, + the reduce scratch tail
  return ((size_t)cdiv((long)N * H * W, 64) + 4) * 2 * Cin + mla_bn_partial_scratch_elems(Cin);
}

extern "C" int mla_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                float* wt_ws, void* stream) {
  return mla_conv2d_dgrad_bn(dy, w, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, residual, relu_src, wt_ws, nullptr, 0, nullptr, stream);
}

extern "C" int mla_conv2d_dgrad_bn(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                   int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                   float* wt_ws, const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, void* stream) {
  return mla_conv2d_dgrad_classes(dy, w, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, residual, relu_src, wt_ws, reqs, nreq, bn_tiles,
                                  0xF, 0xF, stream);
}

extern "C" int mla_conv2d_dgrad_classes(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                        int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                        float* wt_ws, const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, int class_mask,
                                        int residual_mask, void* stream) {
  if (int rc = check_conv("mla_conv2d_dgrad", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(Cin % 64 == 0, "mla_conv2d_dgrad: Cin=%d must be a multiple of 64 (the stem needs no dgrad)", Cin);
  MLA_REQUIRE(dy && w && dx && wt_ws, "mla_conv2d_dgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  weight_transpose_kernel<<<dim3(cdiv(Cout, 32), cdiv(Cin, 32), KH * KW), 256, 0, st>>>(w, wt_ws, KH * KW, Cin, Cout);
  MLA_CHECK_LAUNCH("weight_transpose_kernel");
  // "input" of the gather-GEMM is dy (N,OH,OW,Cout); "output" is dx (N,H,W,Cin).  Stride 2: one launch per
  // output parity class, each with its own tile choice (measured: 25 % faster than all classes merged in one
  // launch with interleaved workgroups, 1.56 vs 2.09 ms over the six stride-2 convs).
  int tiles = 0;   // row tiles launched so far = first tile index of the next parity class in the BatchNorm partial buffers
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      const int cls = py * stride + px;               // output parity class (py, px): bit of class_mask / residual_mask
      if (!((class_mask >> cls) & 1)) continue;
      const float* res = ((residual_mask >> cls) & 1) ? residual : nullptr;
      IGemmGeom g;
      make_dgrad_geom(g, py, px, N, H, W, Cin, Cout, KH, KW, stride, pad);
      const int T = g.T;
      if (g.M <= 0) continue;
      const long Mc = g.M;
      const int wt = T > 0 ? T : 1;
      const int cfg = pick_cfg(&Mc, &wt, 1, Cin, false);
      if (int rc = attach_bn_reqs("mla_conv2d_dgrad_bn", g, reqs, nreq, tiles)) return rc;
      if (int rc = launch_igemm(dy, wt_ws, dx, res, relu_src, nullptr, g, false, cfg, st)) return rc;
      tiles += cdiv(g.M, cfg_bm(cfg));
    }
  if (bn_tiles) *bn_tiles = tiles;
  return MLA_OK;
}

// split-K plan: as close to (and not above) the workgroup budget as the tile count allows
static void wgrad_plan(long M, int Cin, int Cout, int T, int* span, int* splits) {
  const bool scalar = Cin % 64 != 0;
  const int BI = scalar ? 64 : (Cin % 128 == 0 && Cout % 128 == 0 ? 128 : 64);
  const int BJ = scalar ? 64 : BI;
  const long tiles = (long)(scalar ? cdiv((long)T * Cin, BI) : (Cin / BI) * T) * (Cout / BJ);
  // workgroup budget: 3 rounds of 256 for the 128x128 tile (134 VGPRs: 3 resident per CU), 8 rounds for the
  // 64x64 tile (60 VGPRs).  Measured on the layer shapes: 64x64 layers gain 4-8 % from 768 -> 2048.
  // stem (scalar gathers, 60 VGPRs: 6 workgroups per CU): one round of 1536 (-4 % against 2048 = 1.33 rounds, same-box sweep)
  long want = (scalar ? 1536 : (BI == 64 ? 2048 : 768)) / tiles;
  if (want < 1) want = 1;
  long s = (M + want - 1) / want;
  s = ((s + BK - 1) / BK) * BK;                  // whole K steps; spans longer than WG_CHUNK are walked in sub-chunks
  if (s < 256) s = 256;
  *span = (int)s;
  *splits = (int)((M + s - 1) / s);
}

// ordered reduce of split-K slabs (shared with conv_igemm_split.hip)
int mla_wgrad_reduce(const float* part, float* dw, size_t n4, int splits, hipStream_t st) {
  if (splits >= 64 || n4 < 16384) {
    wgrad_reduce_kernel<16><<<cdiv(n4, 16), 256, 0, st>>>(part, dw, n4, splits);
  } else if (splits >= 8) {
    wgrad_reduce_kernel<4><<<cdiv(n4, 64), 256, 0, st>>>(part, dw, n4, splits);
  } else {
    wgrad_reduce_kernel<1><<<cdiv(n4, 256), 256, 0, st>>>(part, dw, n4, splits);
  }
  MLA_CHECK_LAUNCH("wgrad_reduce_kernel");
  return MLA_OK;
}

extern "C" size_t mla_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  const long M = (long)N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad);
  int span, splits;
  wgrad_plan(M, Cin, Cout, KH * KW, &span, &splits);
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
    for (int kw = 0; kw < KW; ++kw) g.tap[kh * KW + kw] = pack_tap(kh - pad, kw - pad, kh * KW + kw);
  g.x_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
  g.y_bytes = (unsigned)((size_t)g.M * Cout * 4);
  int span, splits;
  wgrad_plan(g.M, Cin, Cout, g.T, &span, &splits);
  const size_t need = (size_t)splits * g.T * Cin * Cout * sizeof(float);
  if (ws_bytes < need) {
    mla_set_error("mla_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  const bool scalar = Cin % 64 != 0;
  if (scalar) {
    dim3 grid(cdiv(g.K, 64) * (Cout / 64), splits);
    wgrad_kernel<64, 64, 2, 2, true><<<grid, 256, 0, st>>>(x, dy, part, g, span);
  } else if (Cin % 128 == 0 && Cout % 128 == 0) {
    dim3 grid((Cin / 128) * (Cout / 128) * g.T, splits);
    wgrad_kernel<128, 128, 2, 2, false><<<grid, 256, 0, st>>>(x, dy, part, g, span);
  } else {
    dim3 grid((Cin / 64) * (Cout / 64) * g.T, splits);
    wgrad_kernel<64, 64, 2, 2, false><<<grid, 256, 0, st>>>(x, dy, part, g, span);
  }
  MLA_CHECK_LAUNCH("wgrad_kernel");
  return mla_wgrad_reduce(part, dw, (size_t)g.T * Cin * Cout / 4, splits, st);
}

// ---------------------------------------------------------------------------------------------
// Linear layers of the transformer encoders (nn.Linear in models/m3ae.py:70-71, 97-98, 308) on the same
// gather-GEMM: a Linear is a 1-tap "convolution" over token rows.  Rows are addressed as `groups` groups of
// `rows` consecutive tokens, taken at offset *_off inside groups of *_group_rows tokens, so the sub-range
// "tokens 1..256 of each 257-token sequence" needs no copy.  w is [K][N] (= nn.Linear.weight transposed).
// ---------------------------------------------------------------------------------------------
extern "C" int mla_linear_fwd(const float* x, const float* w_kn, const float* bias, const float* residual, float* y,
                              float* y_gelu, int groups, int rows, int x_group_rows, int x_off, int y_group_rows,
                              int y_off, int K, int N, void* stream) {
  MLA_REQUIRE(x && w_kn && y, "mla_linear_fwd: null pointer");
  IGemmGeom g;
  if (int rc = linear_geom("mla_linear_fwd", g, groups, rows, x_group_rows, x_off, y_group_rows, y_off, K, N)) return rc;
  const long M = g.M;
  const int one = 1;
  return launch_igemm(x, w_kn, y, residual, nullptr, nullptr, g, false, pick_cfg(&M, &one, 1, N, false),
                      (hipStream_t)stream, bias, y_gelu);
}

// dx = dy W^T (+ residual) (* gelu'(gelu_src)).  wt_ws: K*N floats.
extern "C" int mla_linear_dgrad(const float* dy, const float* w_kn, float* dx, const float* residual,
                                const float* gelu_src, float* wt_ws, int groups, int rows, int dy_group_rows, int dy_off,
                                int dx_group_rows, int dx_off, int K, int N, void* stream) {
  MLA_REQUIRE(dy && w_kn && dx && wt_ws, "mla_linear_dgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  weight_transpose_kernel<<<dim3(cdiv(N, 32), cdiv(K, 32), 1), 256, 0, st>>>(w_kn, wt_ws, 1, K, N);
  MLA_CHECK_LAUNCH("weight_transpose_kernel");
  IGemmGeom g;   // GEMM: [M][N] x [N][K] -> [M][K]
  if (int rc = linear_geom("mla_linear_dgrad", g, groups, rows, dy_group_rows, dy_off, dx_group_rows, dx_off, N, K)) return rc;
  g.epi = 1;
  const long M = g.M;
  const int one = 1;
  return launch_igemm(dy, wt_ws, dx, residual, gelu_src, nullptr, g, false, pick_cfg(&M, &one, 1, K, false), st);
}

extern "C" size_t mla_linear_wgrad_ws_bytes(int M, int K, int N) {
  int span, splits;
  wgrad_plan(M, K, N, 1, &span, &splits);
  return (size_t)splits * K * N * sizeof(float);
}

// dw[K][N] = sum_rows x^T dy; x rows at (x_group_rows, x_off), dy rows dense [groups*rows][N].
extern "C" int mla_linear_wgrad(const float* x, const float* dy, float* dw_kn, int groups, int rows, int x_group_rows,
                                int x_off, int K, int N, void* ws, size_t ws_bytes, void* stream) {
  MLA_REQUIRE(x && dy && dw_kn && ws, "mla_linear_wgrad: null pointer");
  IGemmGeom g;
  if (int rc = linear_geom("mla_linear_wgrad", g, groups, rows, x_group_rows, x_off, rows, 0, K, N)) return rc;
  g.y_bytes = (unsigned)((size_t)g.M * N * 4);
  hipStream_t st = (hipStream_t)stream;
  int span, splits;
  wgrad_plan(g.M, K, N, 1, &span, &splits);
  const size_t need = (size_t)splits * K * N * sizeof(float);
  if (ws_bytes < need) {
    mla_set_error("mla_linear_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  if (K % 128 == 0 && N % 128 == 0) {
    wgrad_kernel<128, 128, 2, 2, false><<<dim3((K / 128) * (N / 128), splits), 256, 0, st>>>(x, dy, part, g, span);
  } else {
    wgrad_kernel<64, 64, 2, 2, false><<<dim3((K / 64) * (N / 64), splits), 256, 0, st>>>(x, dy, part, g, span);
  }
  MLA_CHECK_LAUNCH("wgrad_kernel");
  return mla_wgrad_reduce(part, dw_kn, (size_t)K * N / 4, splits, st);
}

