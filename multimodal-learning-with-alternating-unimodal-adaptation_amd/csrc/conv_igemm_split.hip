// Gather-GEMM convolution with fp32 operands split into bf16 terms, on v_mfma_f32_32x32x16_bf16 (gfx950).
//
// Same contraction, geometry, gathers and epilogue as conv_igemm.hip (forward conv and input gradient of
// nn.Conv2d, models/backbone.py:4-12, 28, 31, 79-83, 127), different arithmetic: every fp32 operand is written
// as x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (round-to-nearest, exact:
// 3 x 8 significand bits + signs cover the 24-bit fp32 significand), and a*b is accumulated in fp32 from the
// bf16 products a_i*b_j with i + j <= 2 (six MFMAs).  The dropped terms a1*b2 + a2*b1 + a2*b2 are
// <= 2^-23 |a*b|, the size of one fp32 rounding of the product, and the 32x32x16 MFMA rounds into the
// accumulator once per 16 products instead of once per 2, so the result is as close to the exact sum as the
// fp32-MFMA path's (tests/test_ops_gpu.py compares both with an fp64 reference).  The bf16 MFMA issues 16x
// the fp32 MFMA's FLOPs per clock, so six of them cost 6/16 of the fp32 instruction they replace.
//
// Weights are split once per optimizer step into three bf16 planes [plane][tap][n][k] (k contiguous:
// the layout the MFMA B operand reads with one ds_read_b128) by mla_conv2d_wsplit; activations are split
// in the kernel on their way from registers to LDS (v_cvt_pk_bf16_f32, ~5.5 VALU ops per element).
#include "split_common.h"
#include <stdlib.h>
#include <type_traits>

// LDS image of one operand plane: [rows][32 bf16] = 16 dwords per row, no padding; the 16-byte chunk q of row r
// sits at chunk slot q ^ ((r >> 2) & 3), which makes the MFMA fragment reads (ds_read_b128, 16-lane groups
// {0-3,12-15,20-27}, ...) and the staging stores conflict-free.
#define LROW 16
#ifndef SPLIT_EFF
#define SPLIT_EFF 1.0, 0.95, 0.8, 0.7, 0.9, 0.93
#endif

// One workgroup of WM x WN waves per CU, two LDS buffers, one barrier per K step: while the MFMAs of K step `it`
// read buffer it&1, the same waves split the register-staged tile it+1 into the other buffer and issue the global
// loads of tile it+2.  (Two co-resident workgroups with a single buffer each fall into lock step -- both in the
// MFMA phase, then both in the staging phase -- and the phases add instead of overlapping; measured.)
// BKS = K per stage: 32 (64-B LDS rows, two MFMA k-chunks per barrier) or 16 (32-B rows, one k-chunk per barrier, half
// the LDS: two 8-wave workgroups per CU, whose prologues / epilogues / barriers then overlap each other).
// Chunk swizzle of row r: BKS 32: 16-B chunk q -> q ^ ((r >> 2) & 3);  BKS 16: 16-B half q -> q ^ ((r >> 3) & 1).
template <int BM, int BN, int WM, int WN, int BKS>
struct SplitTileCfg {
  static_assert(BKS == 32 || BKS == 16, "stage depth");
  static constexpr int NT = 64 * WM * WN;
  static constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  static constexpr int LR = BKS / 2;                                  // dwords per LDS row
  static constexpr int ASZ = 3 * BM * LR, BSZ = 3 * BN * LR;          // dwords per buffer
};

template <int BM>
__device__ __forceinline__ void fill_rowinfo(int4* rowinfo, const IGemmGeom& g, int tm, int tid, int nt) {
  for (int r = tid; r < BM; r += nt) {
    const int m = g.m0 + tm * BM + r;
    int4 info = make_int4(-1, -100000, -100000, 0);
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
}

// One output tile (tm, tn) of the geometry g: rowinfo, K loop over its T * (C / BKS) stages, epilogue.  A device function so that one launch
// can serve several geometries (igemm_split_classes_kernel: the parity classes of a stride-2 input gradient).
template <int BM, int BN, int WM, int WN, int TERMS, int BKS>
__device__ __forceinline__ void igemm_split_tile(const float* __restrict__ X, const void* __restrict__ Wsp, float* Y, const float* R,
                                                 const float* MASK, float* __restrict__ part, const float* __restrict__ BIAS,
                                                 float* __restrict__ Y2, const IGemmGeom& g, unsigned* As, unsigned* Bs, int4* rowinfo,
                                                 int tm, int tn) {
  constexpr int NT = 64 * WM * WN;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int LR = BKS / 2;                                  // dwords per LDS row
  constexpr int ACH = BKS / 4;                                 // float4 chunks per A row and stage
  constexpr int AROWS = NT / ACH, APASS = BM / AROWS;
  constexpr int BROWS = NT / 4;                                // B: 4 chunks of BKS/2 bytes per row, plane and stage
  constexpr int BPASS = (BN + BROWS - 1) / BROWS;
  constexpr bool BPART = BN < BROWS;                           // only the first BN*4 threads stage B
  constexpr int NKK = BKS / 16;                                // MFMA k-chunks per stage
  static_assert(APASS >= 1 && AROWS % 16 == 0 && (BN % BROWS == 0 || BPART), "tile / workgroup mismatch");
  constexpr int ASZ = 3 * BM * LR, BSZ = 3 * BN * LR;          // dwords per buffer
  typedef typename std::conditional<BKS == 32, u32x4, u32x2>::type bchunk_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gH = g.H, gW = g.W, gC = g.C, gCO = g.CO;

  fill_rowinfo<BM>(rowinfo, g, tm, tid, NT);
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nIter = g.T * (gC / BKS);   // K stages: taps x channel chunks
  const unsigned plane_bytes = g.w_bytes / 2;   // one bf16 plane of the whole weight tensor

  struct Stage { f32x4 a[APASS]; bchunk_t b[3][BPASS]; };   // one K stage of both operands in registers, on its way to LDS
  Stage rs0;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), wr = make_rsrc(Wsp, 3 * plane_bytes);
  const int arow = tid / ACH, ac = tid % ACH;       // A: row within a pass, float4 chunk
  const int brow = tid >> 2, bq = tid & 3;          // B: row within a pass, chunk
  const bool bact = !BPART || tid < BN * 4;         // wave-uniform (BN*4 is a multiple of 64)
  unsigned rowoff[APASS], tmask[APASS], boff[BPASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int4 info = rowinfo[p * AROWS + arow];
    rowoff[p] = ((unsigned)(info.x + info.y * gW + info.z) * (unsigned)gC + ac * 4) * 4u;
    unsigned m = 0;
    for (int t = 0; t < g.T; ++t) {
      const int tp = g.tap[t];
      const int iy = info.y + tap_dy(tp), ix = info.z + tap_dx(tp);
      m |= ((unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW) ? (1u << t) : 0u;   // invalid rows: -100000
    }
    tmask[p] = m;
  }
#pragma unroll
  for (int p = 0; p < BPASS; ++p) boff[p] = ((unsigned)(p * BROWS + brow) * (unsigned)gC + bq * (BKS / 4)) * 2u;

  // (tap, first channel) of the next tile to load, carried as counters instead of it / cpt (+1-2 %; going further and
  // reading per-tap byte offsets from an LDS table is 5-10 % SLOWER: the lgkmcnt(0) wait drains the fragment reads)
  int ld_t = 0, ld_c0 = 0;
  auto load_tiles = [&](Stage& rs) {       // tiles are loaded strictly in order 0, 1, 2, ...
    const int t = ld_t, c0 = ld_c0;
    ld_c0 += BKS;
    if (ld_c0 == gC) { ld_c0 = 0; ++ld_t; }
    const int tp = g.tap[t];
    const unsigned toff = (unsigned)(((tap_dy(tp) * gW + tap_dx(tp)) * gC + c0) * 4);       // wave-uniform (SGPR)
#pragma unroll
    for (int p = 0; p < APASS; ++p)
      rs.a[p] = buf_load4(xr, ((tmask[p] >> t) & 1u) ? rowoff[p] + toff : OOB_OFF, 0);
    const unsigned wsoff = (unsigned)(((tap_wt(tp) * gCO + tn * BN) * gC + c0) * 2);        // wave-uniform
    if (bact) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
          if constexpr (BKS == 32) rs.b[pl][p] = buf_load4u(wr, boff[p], wsoff + pl * plane_bytes);
          else rs.b[pl][p] = __builtin_amdgcn_raw_buffer_load_b64(wr, (int)boff[p], (int)(wsoff + pl * plane_bytes), 0);
        }
    }
  };
  // staging stores: A thread (row, c) owns k = 4c..4c+3 (8 B per plane); B thread (row, q) owns chunk q (BKS/2 bytes)
  const int a_st = BKS == 32 ? arow * LR + ((((ac >> 1) ^ ((arow >> 2) & 3)) << 2) + (ac & 1) * 2)
                             : arow * LR + ((((ac >> 1) ^ ((arow >> 3) & 1)) << 2) + (ac & 1) * 2);
  const int b_st = BKS == 32 ? brow * LR + ((bq ^ ((brow >> 2) & 3)) << 2)
                             : brow * LR + ((((bq >> 1) ^ ((brow >> 3) & 1)) << 2) + (bq & 1) * 2);
  auto store_tiles = [&](int buf, const Stage& rs) {
    unsigned* Ad = As + buf * ASZ + a_st;
    unsigned* Bd = Bs + buf * BSZ + b_st;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      unsigned h0, m0, l0, h1, m1, l1;
      split_pair<true>(rs.a[p][0], rs.a[p][1], h0, m0, l0);
      split_pair<true>(rs.a[p][2], rs.a[p][3], h1, m1, l1);
      unsigned* dst = Ad + p * AROWS * LR;
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + BM * LR) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * BM * LR) = u32x2{l0, l1};
    }
    if (bact) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int p = 0; p < BPASS; ++p)
          *reinterpret_cast<bchunk_t*>(Bd + (pl * BN + p * BROWS) * LR) = rs.b[pl][p];
    }
  };

  const int i = lane & 31, h = lane >> 5;
  const int swz = BKS == 32 ? (i >> 2) & 3 : (i >> 3) & 1;
  const int a_rd = (wm * (BM / WM) + i) * LR, b_rd = (wn * (BN / WN) + i) * LR;
  struct Frags { bf16x8_t a[3][MI], b[3][NI]; };
  auto load_frags = [&](int buf, int kk, Frags& f) {   // lane (i, h) holds k = kk*16 + 8h .. +7 of row / column i
    const unsigned* Ar = As + buf * ASZ + a_rd + (((kk * 2 + h) ^ swz) << 2);
    const unsigned* Br = Bs + buf * BSZ + b_rd + (((kk * 2 + h) ^ swz) << 2);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        f.a[pl][mi] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ar + (pl * BM + mi * 32) * LR));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        f.b[pl][ni] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Br + (pl * BN + ni * 32) * LR));
    }
  };
  auto mma_frags = [&](const Frags& f) {
#pragma unroll
    for (int term = 0; term < TERMS; ++term)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[TERM_A[term]][mi], f.b[TERM_B[term]][ni], acc[mi][ni], 0, 0, 0);
  };
  Frags f0, f1;

  if (nIter > 0) {  // nIter == 0: a dgrad parity class no tap reaches (1x1 stride 2): epilogue only
    load_tiles(rs0);
    store_tiles(0, rs0);
    if (nIter > 1) load_tiles(rs0);
  }
  __syncthreads();
  // Steady state is branch-free so the scheduler can interleave the staging work with the MFMAs; the last two
  // stages (nothing left to load / to stage) are peeled.  Measured alternatives (l2 shape, 128x128 tile, 177 TF):
  // a second register set with the loads of tile it+2 pinned at the top of step it: 3-7 % slower, unpinned with a
  // two-stage load distance: +2 % on the K >= 1152 layers, -4 % on the stride-2 parity classes; a ring of three
  // LDS buffers with the next step's first-half fragments fetched before the barrier: +-0; plane-wise fragment reads
  // issued one MFMA group ahead of their use, with and without sched_group_barrier pinning: +-1 %; a hand-scheduled
  // inline-asm stage (all 18 fragment reads in flight, counted lgkmcnt waits, fixed registers) with SIMD partners
  // staging in opposite halves of the step: -15 % -- what pays is VALU / LDS work inside the MFMA shadows of the SAME
  // wave, which the compiler's interleaved schedule already gives.  Timing-only ablations:
  // MFMAs alone reach the 6-product ceiling (333 TF at the sustained bf16 rate) once the tail of the last round is
  // taken out; adding the fragment reads costs ~20 %, the barrier nothing, staging another ~17 %; a tile's
  // prologue / epilogue are exposed with one workgroup per CU (~10 us per tile round on the 1x1 convs).
  int it = 0;
  for (; it + 2 < nIter; ++it) {
    __builtin_amdgcn_iglp_opt(1);   // the compiler's single-wave small-GEMM interleave of DS / VMEM / MFMA for this block: -1.7 % (same-box A/B;
                                    // strategy 0, the multi-wave one, +0.8 %)
    const int cur = it & 1;
    load_frags(cur, 0, f0);
    if constexpr (NKK == 2) load_frags(cur, 1, f1);   // both halves' fragments in flight before the first MFMA
    mma_frags(f0);
    store_tiles(cur ^ 1, rs0);   // readers of that buffer finished before the previous barrier
    load_tiles(rs0);
    if constexpr (NKK == 2) mma_frags(f1);
    __syncthreads();
  }
  if (it + 1 < nIter) {
    const int cur = it & 1;
    load_frags(cur, 0, f0);
    if constexpr (NKK == 2) load_frags(cur, 1, f1);
    mma_frags(f0);
    store_tiles(cur ^ 1, rs0);
    if constexpr (NKK == 2) mma_frags(f1);
    __syncthreads();
    ++it;
  }
  if (it < nIter) {
    load_frags(it & 1, 0, f0);
    if constexpr (NKK == 2) load_frags(it & 1, 1, f1);
    mma_frags(f0);
    if constexpr (NKK == 2) mma_frags(f1);
    __syncthreads();
  }
  igemm_epilogue<BM, BN, WM, WN>(acc, rowinfo, reinterpret_cast<float*>(As), Y, R, MASK, part, BIAS, Y2, g, tm, tn);
}

template <int BM, int BN, int WM, int WN, int TERMS, int BKS, int WPE>
__global__ __launch_bounds__(64 * WM * WN, WPE) void igemm_split_kernel(const float* __restrict__ X, const void* __restrict__ Wsp,
                                                                         float* Y, const float* R, const float* MASK,
                                                                         float* __restrict__ part, const float* __restrict__ BIAS,
                                                                         float* __restrict__ Y2, const IGemmGeom g) {
  typedef SplitTileCfg<BM, BN, WM, WN, BKS> C;
  __shared__ __attribute__((aligned(16))) unsigned As[2 * C::ASZ];
  __shared__ __attribute__((aligned(16))) unsigned Bs[2 * C::BSZ];
  __shared__ int4 rowinfo[BM];  // {n*H*W or -1, oy*sy, ox*sx, output pixel index}
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int gridN = g.CO / BN;
  int tm, tn;
  tile_coords(wg, (int)gridDim.x / gridN, gridN, tm, tn);
  igemm_split_tile<BM, BN, WM, WN, TERMS, BKS>(X, Wsp, Y, R, MASK, part, BIAS, Y2, g, As, Bs, rowinfo, tm, tn);
}

// The output parity classes of a stride-2 input gradient in ONE launch.  Each class is its own gather-GEMM (1, 2, 2 and 4 of the nine
// taps of a 3x3 kernel) over a quarter of the pixels; launched one after the other, layer3.0 / layer4.0 put 128-148 workgroups on the
// 256 CUs four times over (the kernel trace of the round-3 step: 27 + 39 + 39 + 65 us for 22 GFLOP).  Here the classes share the
// grid, longest first (4-tap tiles, then the 2-tap ones, the 1-tap class fills the tail), so the chip stays full until the end.
struct IGemmClasses {
  IGemmGeom g[4];
  const float* R[4];         // residual per class (null: none)
  int first[5];              // first workgroup of class slot c; first[n] = grid size
  int n;
};
static_assert(sizeof(IGemmClasses) + 32 <= 4096, "kernel arguments must fit the 4 KB kernarg segment");
template <int BM, int BN, int WM, int WN, int BKS, int WPE>
__global__ __launch_bounds__(64 * WM * WN, WPE) void igemm_split_classes_kernel(const float* __restrict__ X, const void* __restrict__ Wsp,
                                                                                 float* Y, const float* MASK, const IGemmClasses cs) {
  typedef SplitTileCfg<BM, BN, WM, WN, BKS> C;
  __shared__ __attribute__((aligned(16))) unsigned As[2 * C::ASZ];
  __shared__ __attribute__((aligned(16))) unsigned Bs[2 * C::BSZ];
  __shared__ int4 rowinfo[BM];
  const int wg = (int)blockIdx.x;            // dispatch order = class order (no XCD remap across classes of different length)
  int c = 0;
  while (c + 1 < cs.n && wg >= cs.first[c + 1]) ++c;
  const IGemmGeom& g = cs.g[c];
  const int nwg = cs.first[c + 1] - cs.first[c];
  const int gridN = g.CO / BN;
  int tm, tn;
  tile_coords(xcd_remap(wg - cs.first[c], nwg), nwg / gridN, gridN, tm, tn);
  igemm_split_tile<BM, BN, WM, WN, 6, BKS>(X, Wsp, Y, cs.R[c], MASK, nullptr, nullptr, nullptr, g, As, Bs, rowinfo, tm, tn);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient on the same arithmetic: dW[t][ci][co] = sum_m X[in(m,t)][ci] * dY[m][co], GEMM-K = pixels.
// Both operands arrive pixel-major (channels contiguous) but the MFMA wants k = pixels contiguous per channel
// row, so the staging pass transposes in registers: a thread gathers the same 4 channels of PX consecutive
// pixels, splits them and writes PX bf16 (8 B for PX = 4) per channel row and plane.  Lanes run over the pixel
// groups first, so a wave's loads are 128-B row pieces and its LDS stores are at most 2-way conflicted.
// grid.x = (ci tiles) x (co tiles) x taps, grid.y = split-K ranges of `span` pixels walked in sub-chunks of
// WGS_CHUNK pixels; partial slabs and the ordered reduce are those of the fp32 kernel (conv_igemm.hip).
// ---------------------------------------------------------------------------------------------
#define WGS_CHUNK 2048
template <int BI, int BJ, int WI, int WJ>
__global__ __launch_bounds__(64 * WI * WJ, 1) void wgrad_split_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                        float* __restrict__ part, const IGemmGeom g, int span,
                                                                        float* __restrict__ bias_part = nullptr) {
  static_assert(WI * WJ == 4 || WI * WJ == 8, "4 or 8 waves");
  constexpr int NT = 64 * WI * WJ;
  constexpr int MI = BI / WI / 32, NI = BJ / WJ / 32;
  constexpr int XG = NT / (BI / 4), XPX = BK / XG;     // pixel groups per K step, pixels per thread (X operand)
  constexpr int YG = NT / (BJ / 4), YPX = BK / YG;
  static_assert((XPX == 4 || XPX == 2) && (YPX == 4 || YPX == 2), "tile / workgroup mismatch");
  constexpr int ASZ = 3 * BI * LROW, BSZ = 3 * BJ * LROW;
  __shared__ __attribute__((aligned(16))) unsigned As[2 * ASZ];
  __shared__ __attribute__((aligned(16))) unsigned Bs[2 * BSZ];
  __shared__ unsigned rowoff[WGS_CHUNK];   // (64x64 tile with half the chunk = three workgroups per CU: 7-17 % slower, measured)
                                           // byte offset of the gathered X row (this block's tap, channel tile) or OOB_OFF

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  const int tilesI = g.C / BI, tilesJ = g.CO / BJ;
  // all workgroups of one pixel span (every tap and channel tile) on ONE XCD: they gather the same X / dY rows, which its
  // L2 then serves 9 x tiles times; in dispatch order they would be spread round-robin over the 8 XCDs (8 L2 fills per span).
  // Same-box A/B: 64x64 tiles (layer1) -4...5 %, 128x128 tiles +-0.
  const int lwg = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = lwg / (int)gridDim.x;
  int b = lwg - by * (int)gridDim.x;
  const int tj = b % tilesJ; b /= tilesJ;
  const int ti = b % tilesI; b /= tilesI;
  const int t = b;
  const int m_begin = by * span, m_end = min(g.M, m_begin + span);
  const int gH = g.H, gW = g.W, gC = g.C, gCO = g.CO;
  const int dy = tap_dy(g.tap[t]), dx = tap_dx(g.tap[t]);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, g.y_bytes);
  f32x4 xreg[XPX], yreg[YPX];
  const int xg = tid % XG, xc = tid / XG;   // pixel group, 4-channel chunk
  const int yg = tid % YG, yc = tid / YG;
  // LDS store address of channel row 4c+e, pixels g*PX .. +PX-1 (bytes 2*PX*g of the 64-B row, chunk-swizzled by row>>2 & 3 = c & 3)
  const int x_st = (xc * 4) * LROW + (XPX == 4 ? ((((xg >> 1) ^ (xc & 3)) << 2) + (xg & 1) * 2) : ((((xg >> 2) ^ (xc & 3)) << 2) + (xg & 3)));
  const int y_st = (yc * 4) * LROW + (YPX == 4 ? ((((yg >> 1) ^ (yc & 3)) << 2) + (yg & 1) * 2) : ((((yg >> 2) ^ (yc & 3)) << 2) + (yg & 3)));
  const int i = lane & 31, h = lane >> 5, swz = (i >> 2) & 3;
  const int a_rd = (wi * (BI / WI) + i) * LROW, b_rd = (wj * (BJ / WJ) + i) * LROW;

  struct Frags { bf16x8_t a[3][MI], b[3][NI]; };
  auto load_frags = [&](int buf, int kk, Frags& f) {
    const unsigned* Ar = As + buf * ASZ + a_rd + (((kk * 2 + h) ^ swz) << 2);
    const unsigned* Br = Bs + buf * BSZ + b_rd + (((kk * 2 + h) ^ swz) << 2);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        f.a[pl][mi] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ar + (pl * BI + mi * 32) * LROW));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        f.b[pl][ni] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Br + (pl * BJ + ni * 32) * LROW));
    }
  };
  auto mma_frags = [&](const Frags& f) {
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[TERM_A[term]][mi], f.b[TERM_B[term]][ni], acc[mi][ni], 0, 0, 0);
  };
  Frags f0, f1;

  const bool do_bias = bias_part != nullptr && ti == 0 && t == 0;   // one workgroup per (column tile, pixel range)
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
  for (int c_begin = m_begin; c_begin < m_end; c_begin += WGS_CHUNK) {
    const int c_end = min(m_end, c_begin + WGS_CHUNK);
    __syncthreads();  // previous sub-chunk's readers are done with rowoff / As / Bs
    for (int r = tid; r < WGS_CHUNK; r += NT) {
      const int m = c_begin + r;
      unsigned off = OOB_OFF;
      if (m < c_end) {
        const int ohw = g.OH * g.OW;
        const int n = m / ohw, rem = m - n * ohw;
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        const int iy = oy * g.sy + dy, ix = ox * g.sx + dx;
        if ((unsigned)iy < (unsigned)gH && (unsigned)ix < (unsigned)gW)
          off = ((unsigned)(n * gH * gW + iy * gW + ix) * (unsigned)gC + ti * BI) * 4u;
      }
      rowoff[r] = off;
    }
    __syncthreads();

    auto load_tiles = [&](int p0) {  // p0: first pixel (sub-chunk-relative) of this K step
#pragma unroll
      for (int q = 0; q < XPX; ++q) {
        const unsigned ro = rowoff[p0 + xg * XPX + q];
        xreg[q] = buf_load4(xr, ro == OOB_OFF ? OOB_OFF : ro + xc * 16u, 0);   // no wrap past OOB_OFF
      }
#pragma unroll
      for (int q = 0; q < YPX; ++q) {
        const int m = c_begin + p0 + yg * YPX + q;
        yreg[q] = buf_load4(yr, m < c_end ? ((unsigned)m * (unsigned)gCO + tj * BJ + yc * 4) * 4u : OOB_OFF, 0);
      }
    };
    auto store_op = [&](unsigned* base, int plane_rows, const f32x4* reg, auto px_tag) {
      constexpr int PX = decltype(px_tag)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // channel row 4c+e
        unsigned* dst = base + e * LROW;
        if constexpr (PX == 4) {
          unsigned h0, m0, l0, h1, m1, l1;
          split_pair<true>(reg[0][e], reg[1][e], h0, m0, l0);
          split_pair<true>(reg[2][e], reg[3][e], h1, m1, l1);
          *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
          *reinterpret_cast<u32x2*>(dst + plane_rows * LROW) = u32x2{m0, m1};
          *reinterpret_cast<u32x2*>(dst + 2 * plane_rows * LROW) = u32x2{l0, l1};
        } else {
          unsigned h0, m0, l0;
          split_pair<true>(reg[0][e], reg[1][e], h0, m0, l0);
          dst[0] = h0;
          dst[plane_rows * LROW] = m0;
          dst[2 * plane_rows * LROW] = l0;
        }
      }
    };
    auto store_tiles = [&](int buf) {
      store_op(As + buf * ASZ + x_st, BI, xreg, std::integral_constant<int, XPX>{});
      store_op(Bs + buf * BSZ + y_st, BJ, yreg, std::integral_constant<int, YPX>{});
      if (do_bias) {            // Linear bias gradient = column sums of dY: the tile is in registers here anyway
#pragma unroll
        for (int q = 0; q < YPX; ++q) bacc += yreg[q];
      }
    };

    const int nIter = (c_end - c_begin + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    if (nIter > 1) load_tiles(BK);
    __syncthreads();
    int it = 0;
    for (; it + 2 < nIter; ++it) {   // branch-free steady state (see igemm_split_kernel); unrolling by two with compile-time
                                     // LDS buffers (address arithmetic folded into offset fields) measured +0.5 % slower
      const int cur = it & 1;
      load_frags(cur, 0, f0);
      load_frags(cur, 1, f1);
      mma_frags(f0);
      store_tiles(cur ^ 1);
      load_tiles((it + 2) * BK);
      mma_frags(f1);
      __syncthreads();
    }
    if (it + 1 < nIter) {
      const int cur = it & 1;
      load_frags(cur, 0, f0);
      load_frags(cur, 1, f1);
      mma_frags(f0);
      store_tiles(cur ^ 1);
      mma_frags(f1);
      __syncthreads();
      ++it;
    }
    if (it < nIter) {
      load_frags(it & 1, 0, f0);
      load_frags(it & 1, 1, f1);
      mma_frags(f0);
      mma_frags(f1);
    }
  }

  if (do_bias) {     // reduce the per-thread sums over the YG pixel groups (fixed order), write this range's partial bias row
    __syncthreads();                                  // the last stage's fragment reads of As are done
    float* red = reinterpret_cast<float*>(As);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[yg * BJ + yc * 4 + e] = bacc[e];
    __syncthreads();
    if (tid < BJ) {
      float sb = 0.f;
      for (int k2 = 0; k2 < YG; ++k2) sb += red[k2 * BJ + tid];
      bias_part[(size_t)by * gCO + tj * BJ + tid] = sb;
    }
  }
  float* slab = part + ((size_t)by * g.T + t) * gC * gCO;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ti * BI + wi * (BI / WI) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = tj * BJ + wj * (BJ / WJ) + ni * 32 + i;
        slab[(size_t)row * gCO + col] = acc[mi][ni][e];
      }
    }
}

// out[plane][t][n][k] (bf16): transposed = 1: n = co, k = ci (forward conv); 0: n = ci, k = co (input gradient).
// in is HWIO fp32 [t][ci][co].  One 32x32 (ci x co) tile of one tap per workgroup.
__device__ __forceinline__ void weight_split_tile(const float* __restrict__ in, unsigned short* __restrict__ out, int T, int CI,
                                                  int CO, int transposed, int t, int ci0, int co0, float (*tile)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  const size_t plane = (size_t)T * CI * CO;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < CI && co < CO) ? in[((size_t)t * CI + ci) * CO + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int ci, co;
    float v;
    size_t o;
    if (transposed) { co = co0 + r; ci = ci0 + tx; v = tile[tx][r]; o = ((size_t)t * CO + co) * CI + ci; }
    else            { ci = ci0 + r; co = co0 + tx; v = tile[r][tx]; o = ((size_t)t * CI + ci) * CO + co; }
    if (ci < CI && co < CO) {
      unsigned hi, mid, lo;
      split_pair(v, 0.f, hi, mid, lo);
      out[o] = (unsigned short)hi;
      out[plane + o] = (unsigned short)mid;
      out[2 * plane + o] = (unsigned short)lo;
    }
  }
}

__global__ __launch_bounds__(256) void weight_split_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                            int T, int CI, int CO, int transposed) {
  __shared__ float tile[32][33];
  weight_split_tile(in, out, T, CI, CO, transposed, blockIdx.z, blockIdx.y * 32, blockIdx.x * 32, tile);
}

// Every conv of an encoder in one launch: desc[j] = {w_off, out_off, T, CI, CO, transposed, first_block, -} (device memory)
__global__ __launch_bounds__(256) void weight_split_batch_kernel(const float* __restrict__ params, unsigned short* __restrict__ out,
                                                                  const int* __restrict__ desc, int n) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int j = 0, hi = n - 1;                                  // last row whose first_block <= b (scalar binary search)
  while (j < hi) {
    const int mid = (j + hi + 1) >> 1;
    if (desc[mid * 8 + 6] <= b) j = mid; else hi = mid - 1;
  }
  const int* d = desc + j * 8;
  const int T = d[2], CI = d[3], CO = d[4];
  const int tc = (CO + 31) / 32, tr = (CI + 31) / 32;
  int lb = b - d[6];
  const int bx = lb % tc; lb /= tc;
  const int by = lb % tr; lb /= tr;
  if (lb >= T) return;
  weight_split_tile(params + d[0], out + d[1], T, CI, CO, d[5], lb, by * 32, bx * 32, tile);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int g_split_terms = 6;
extern "C" int mla_conv2d_split_terms(int terms) {   // measurement hook: 3, 6 (default, fp32-equivalent) or 8 products
  if (terms == 3 || terms == 6 || terms == 8) g_split_terms = terms;
  return g_split_terms;
}

// Tiles: 256x128, 192x128 (96x32 wave tiles: for row counts that leave 256-row tiles on ~half of the CUs, e.g. 9 408 = 49 x 192) and 128x128 (8 waves, K stage 32, one workgroup per CU), 256x64 (8 waves, K stage 16, two per CU:
// the Cout = 64 layers, +10 % over 128x64), 128x64 and 64x64 (4 waves, K stage 32, two / three per CU).  Measured the other
// way round too: 128x128 at K stage 16 with two workgroups per CU is 6-15 % slower than one at K stage 32, 256x64 at
// K stage 32 (one per CU) 5-15 % slower than two at K stage 16.
enum { SCFG_256x128 = 0, SCFG_128x128 = 1, SCFG_128x64 = 2, SCFG_64x64 = 3, SCFG_256x64 = 4, SCFG_192x128 = 5, SCFG_COUNT = 6 };
static int scfg_bm(int c) { return (c == SCFG_256x128 || c == SCFG_256x64) ? 256 : (c == SCFG_64x64 ? 64 : (c == SCFG_192x128 ? 192 : 128)); }
static int scfg_bn(int c) { return (c <= SCFG_128x128 || c == SCFG_192x128) ? 128 : 64; }
static int g_split_cfg = -1;   // measurement hook: force one tile
extern "C" int mla_conv2d_split_cfg(int cfg) { g_split_cfg = (cfg >= 0 && cfg < SCFG_COUNT) ? cfg : -1; return g_split_cfg; }

// Minimise rounds * resident workgroups * tile area / efficiency.  Small tiles stage more bytes per MFMA.  The
// efficiencies are the measured per-flop rates at the ResNet-18 layer shapes relative to the 256x128 tile
// (scripts/split_probe.py); the ranking they give matches the measured ranking on l1..l4 of both modalities.
static int pick_scfg(long M, int CO, int weight, int k_total = 1 << 30) {
  if (g_split_cfg >= 0 && CO % scfg_bn(g_split_cfg) == 0) return g_split_cfg;
  // one-tap convolutions with a short K (the 1x1 stride-2 downsample convs, K <= 512: 2-16 stages per tile): prologue and epilogue
  // dominate, the 64x64 tile with three workgroups per CU is fastest at every such shape (forced-tile probe, 10-45 % over the model)
  if (k_total <= 512 && M <= (1L << 18)) return SCFG_64x64;
  const double eff[SCFG_COUNT] = {SPLIT_EFF};
  const int per_cu_tab[SCFG_COUNT] = {1, 1, 2, 3, 2, 1};         // resident workgroups per CU (LDS / VGPRs)
  int best = -1;
  double best_cost = 0;
  for (int c = 0; c < SCFG_COUNT; ++c) {
    if (CO % scfg_bn(c) != 0) continue;
    const double blocks = (double)cdiv(M, scfg_bm(c)) * (CO / scfg_bn(c));
    const int per_cu = per_cu_tab[c];
    const double rounds = (double)((long)((blocks + 256 * per_cu - 1) / (256 * per_cu)));
    const double cost = rounds * per_cu * scfg_bm(c) * scfg_bn(c) / eff[c];
    if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  (void)weight;
  return best;
}

// conv_patch_split.hip: 3x3 / stride 1 / pad 1 forward and input gradient with the A operand served from an LDS-resident patch
bool mla_patch_supported(const IGemmGeom& g, bool force);
int mla_patch_launch(const float* X, const void* Wsp, float* Y, const float* R, const float* MASK, float* part, const IGemmGeom& g,
                     int* bn_tiles, hipStream_t st);
bool mla_patch64p_usable(const IGemmGeom& g);
static int g_wgrad_tr = 1;                            // mla_conv2d_wgrad_tr: the all-taps weight-gradient kernels (wgrad_tr_split.hip)
static int g_patch = -1;                              // -1: not yet read from $MLA_CONV_PATCH (default 1)
static int patch_mode() {
  if (g_patch < 0) {
    const char* e = getenv("MLA_CONV_PATCH");
    g_patch = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
  }
  return g_patch;
}
// measurement / test hook: 0 = per-tap gather-GEMM everywhere, 1 = default (patch kernel where its grid fills the chip), 2 = patch
// kernel wherever the geometry allows; other values: query
extern "C" int mla_conv2d_patch(int on) {
  if (on >= 0 && on <= 2) g_patch = on;
  return patch_mode();
}
static bool use_patch(const IGemmGeom& g) {
  return patch_mode() && g_split_terms == 6 && g_split_cfg < 0 && mla_patch_supported(g, patch_mode() == 2);
}

template <int TERMS>
static void launch_split_t(int cfg, int total, hipStream_t st, const float* X, const void* Wsp, float* Y, const float* R,
                           const float* MASK, float* part, const float* BIAS, float* Y2, const IGemmGeom& mg) {
  if (cfg == SCFG_256x128) igemm_split_kernel<256, 128, 4, 2, TERMS, 32, 1><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else if (cfg == SCFG_128x128) igemm_split_kernel<128, 128, 2, 4, TERMS, 32, 1><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else if (cfg == SCFG_256x64) igemm_split_kernel<256, 64, 4, 2, TERMS, 16, 4><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else if (cfg == SCFG_192x128) igemm_split_kernel<192, 128, 2, 4, TERMS, 32, 1><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else if (cfg == SCFG_128x64) igemm_split_kernel<128, 64, 2, 2, TERMS, 32, 1><<<total, 256, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else igemm_split_kernel<64, 64, 2, 2, TERMS, 32, 1><<<total, 256, 0, st>>>(X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
}

static double scfg_cost(int c, long M, int CO) {       // the cost of pick_scfg: rounds x resident workgroups x tile area / relative rate
  const double eff[SCFG_COUNT] = {SPLIT_EFF};
  const int per_cu_tab[SCFG_COUNT] = {1, 1, 2, 3, 2, 1};
  const double blocks = (double)cdiv(M, scfg_bm(c)) * (CO / scfg_bn(c));
  const int per_cu = per_cu_tab[c];
  const double rounds = (double)((long)((blocks + 256 * per_cu - 1) / (256 * per_cu)));
  return rounds * per_cu * scfg_bm(c) * scfg_bn(c) / eff[c];
}

// Two-phase schedule for row counts between whole rounds (visual layer3: 37 632 rows x 256 columns = 2.3 rounds of 128x128 tiles, 1.5 of
// 192x128 ones -- every single tile shape pays for a mostly empty last round): whole rounds of a big tile over the first rows, then ONE
// launch of the best tile for the rows that are left (24 576 rows as 256 tiles of 192x128 + 13 056 rows as 204 tiles of 128x128: 1.25 instead
// of 1.5 tile-round units).  Each output row belongs to exactly one launch; statistics rows of the second launch follow the first's.
static int g_two_phase = -1;                          // -1: $MLA_CONV_TWO_PHASE (default 1)
extern "C" int mla_conv2d_two_phase(int on) {         // measurement / test hook: 0 = always one launch, 1 = default; other: query
  if (on == 0 || on == 1) g_two_phase = on;
  if (g_two_phase < 0) { const char* e = getenv("MLA_CONV_TWO_PHASE"); g_two_phase = (e && e[0] == '0') ? 0 : 1; }
  return g_two_phase;
}
static bool plan_two_phase(const IGemmGeom& g, int cfg1, int k_stages, int* cfgA, int* rowsA, int* cfgB) {
  if (!mla_conv2d_two_phase(-1) || g_split_cfg >= 0 || g.CO % 128 != 0 || g.m0 != 0 || k_stages < 16) return false;
  const double eff[SCFG_COUNT] = {SPLIT_EFF};
  double best = 0.93 * scfg_cost(cfg1, g.M, g.CO);
  bool found = false;
  const int gridN = g.CO / 128;
  const int bigs[2] = {SCFG_256x128, SCFG_192x128};
  for (int k = 0; k < 2; ++k) {
    const int cA = bigs[k];
    const long tiles = (long)cdiv(g.M, scfg_bm(cA)) * gridN;
    const long full = tiles / 256;                                     // whole rounds of the 256 CUs
    // whole ROW tiles that fit those rounds (wide outputs -- the transformer Linears' 6 / 18 / 24 column tiles -- leave up to
    // gridN - 1 workgroup slots of the last round empty)
    const long row_tiles = full * 256 / gridN;
    const long rA = row_tiles * scfg_bm(cA), rem = g.M - rA;
    if (full < 1 || row_tiles < 1 || rem <= 0) continue;
    const double costA = (double)full * scfg_bm(cA) * 128 / eff[cA];
    int cB = -1;
    double costB = 0;
    for (int c = 0; c < SCFG_COUNT; ++c) {
      if (g.CO % scfg_bn(c) != 0) continue;
      const double cc = scfg_cost(c, rem, g.CO);
      if (cB < 0 || cc < costB) { cB = c; costB = cc; }
    }
    const double total = costA + costB + 0.03 * 256 * 128;
    if (total < best) { best = total; *cfgA = cA; *rowsA = (int)rA; *cfgB = cB; found = true; }
  }
  return found;
}

static int launch_split_one(const float* X, const void* Wsp, float* Y, const float* R, const float* MASK, float* part,
                            const IGemmGeom& mg, int cfg, hipStream_t st, const float* BIAS, float* Y2) {
  const int total = cdiv(mg.M - mg.m0, scfg_bm(cfg)) * (mg.CO / scfg_bn(cfg));
  if (total <= 0) return MLA_OK;
  if (g_split_terms == 6) launch_split_t<6>(cfg, total, st, X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else if (g_split_terms == 8) launch_split_t<8>(cfg, total, st, X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  else launch_split_t<3>(cfg, total, st, X, Wsp, Y, R, MASK, part, BIAS, Y2, mg);
  MLA_CHECK_LAUNCH("igemm_split_kernel");
  return MLA_OK;
}

// row_tiles (optional): number of statistics rows ([tile][2][CO]) the launch(es) wrote / advanced
static int launch_split(const float* X, const void* Wsp, float* Y, const float* R, const float* MASK, float* part,
                        const IGemmGeom& mg, int cfg, hipStream_t st, const float* BIAS = nullptr, float* Y2 = nullptr,
                        int* row_tiles = nullptr) {
  int cfgA, rowsA, cfgB;
  if (plan_two_phase(mg, cfg, mg.T * (mg.C / 32), &cfgA, &rowsA, &cfgB)) {
    IGemmGeom ga = mg;
    ga.M = rowsA;
    if (int rc = launch_split_one(X, Wsp, Y, R, MASK, part, ga, cfgA, st, BIAS, Y2)) return rc;
    const int tilesA = rowsA / scfg_bm(cfgA);
    IGemmGeom gb = mg;
    gb.m0 = rowsA;
    gb.bn_tile0 = mg.bn_tile0 + tilesA;
    float* partB = part ? part + (size_t)tilesA * 2 * mg.CO * 2 : nullptr;       // fp64 rows [tile][2][CO]
    if (row_tiles) *row_tiles = tilesA + cdiv(mg.M - rowsA, scfg_bm(cfgB));
    return launch_split_one(X, Wsp, Y, R, MASK, partB, gb, cfgB, st, BIAS, Y2);
  }
  if (row_tiles) *row_tiles = cdiv(mg.M, scfg_bm(cfg));
  return launch_split_one(X, Wsp, Y, R, MASK, part, mg, cfg, st, BIAS, Y2);
}

extern "C" size_t mla_conv2d_wsplit_bytes(int Cin, int Cout, int KH, int KW) {
  return (size_t)3 * KH * KW * Cin * Cout * sizeof(unsigned short);
}

extern "C" int mla_conv2d_wsplit(const float* w, void* wsplit, int Cin, int Cout, int KH, int KW, int transposed,
                                 void* stream) {
  MLA_REQUIRE(w && wsplit, "mla_conv2d_wsplit: null pointer");
  MLA_REQUIRE(Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && KH * KW <= MAX_TAPS, "mla_conv2d_wsplit: bad dims");
  weight_split_kernel<<<dim3(cdiv(Cout, 32), cdiv(Cin, 32), KH * KW), 256, 0, (hipStream_t)stream>>>(
      w, (unsigned short*)wsplit, KH * KW, Cin, Cout, transposed ? 1 : 0);
  MLA_CHECK_LAUNCH("weight_split_kernel");
  return MLA_OK;
}

extern "C" int mla_conv2d_wsplit_batch(const float* params, void* wsplit, const int* desc, int n, int total_blocks, void* stream) {
  MLA_REQUIRE(params && wsplit && desc, "mla_conv2d_wsplit_batch: null pointer");
  MLA_REQUIRE(n > 0 && n <= 4096 && total_blocks > 0, "mla_conv2d_wsplit_batch: n=%d (1..4096), total_blocks=%d", n, total_blocks);
  weight_split_batch_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(params, (unsigned short*)wsplit, desc, n);
  MLA_CHECK_LAUNCH("weight_split_batch_kernel");
  return MLA_OK;
}

static int g_dgrad_merge = -1;                        // -1: $MLA_DGRAD_MERGE (default 1)
static bool dgrad_merge_on() {
  if (g_dgrad_merge < 0) {
    const char* e = getenv("MLA_DGRAD_MERGE");
    g_dgrad_merge = (e && e[0] == '0') ? 0 : 1;
  }
  return g_dgrad_merge != 0;
}
extern "C" int mla_conv2d_dgrad_merge(int on) {       // measurement / test hook: 0 = one launch per parity class, 1 = default; other: query
  if (on == 0 || on == 1) g_dgrad_merge = on;
  return dgrad_merge_on() ? 1 : 0;
}

extern "C" int mla_conv2d_fwd_split(const float* x, const void* wsplit_t, float* y, int N, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad, float* bn_partial, int* bn_tiles, void* stream) {
  if (int rc = check_conv("mla_conv2d_fwd_split", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(Cin % 64 == 0, "mla_conv2d_fwd_split: Cin=%d must be a multiple of 64 (the stem runs on mla_conv2d_fwd)", Cin);
  MLA_REQUIRE(x && wsplit_t && y, "mla_conv2d_fwd_split: null pointer");
  IGemmGeom g;
  make_fwd_geom(g, N, H, W, Cin, Cout, KH, KW, stride, pad);
  MLA_REQUIRE(g.OH > 0 && g.OW > 0, "mla_conv2d_fwd_split: empty output");
  if (use_patch(g)) return mla_patch_launch(x, wsplit_t, y, nullptr, nullptr, bn_partial, g, bn_tiles, (hipStream_t)stream);
  const int cfg = pick_scfg(g.M, Cout, 1, KH * KW * Cin);
  return launch_split(x, wsplit_t, y, nullptr, nullptr, bn_partial, g, cfg, (hipStream_t)stream, nullptr, nullptr, bn_tiles);
}

// ---- BatchNorm folded into the operands of the 64 -> 64 channel 3x3 / 1 / 1 convolutions (conv2 of the layer1 BasicBlocks,
// models/backbone.py:38-46: conv1 -> bn1 -> relu -> conv2): relu(bn1(y1)) is formed by the consumers -- conv2's forward patch staging, conv2's
// weight-gradient staging, the ReLU mask of conv2's input-gradient epilogue -- from y1 with bn_apply_kernel's expression (bit-identical to
// the materialised path), so the activation tensor is never written (154 MB per block at batch 64) nor read.
extern "C" int mla_conv2d_bnfold_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  if (!(Cin == 64 && Cout == 64 && KH == 3 && KW == 3 && stride == 1 && pad == 1) || N <= 0 || H <= 0 || W <= 0) return 0;
  if (g_split_terms != 6 || g_split_cfg >= 0 || !patch_mode() || !g_wgrad_tr) return 0;
  IGemmGeom g;
  make_fwd_geom(g, N, H, W, Cin, Cout, KH, KW, stride, pad);
  // only where the persistent patch kernel is what the unfolded convolution would run on anyway (its grid fills the chip, or it is forced):
  // folding then never changes which kernel -- and so which summation order -- a layer gets
  return (use_patch(g) && mla_patch64p_usable(g)) ? 1 : 0;
}

extern "C" int mla_conv2d_fwd_split_bnin(const float* x, const void* wsplit_t, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                         int KW, int stride, int pad, const float* in_mean, const float* in_invstd, const float* in_gamma,
                                         const float* in_beta, float* bn_partial, int* bn_tiles, void* stream) {
  if (int rc = check_conv("mla_conv2d_fwd_split_bnin", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && wsplit_t && y && in_mean && in_invstd && in_gamma && in_beta, "mla_conv2d_fwd_split_bnin: null pointer");
  MLA_REQUIRE(mla_conv2d_bnfold_supported(N, H, W, Cin, Cout, KH, KW, stride, pad), "mla_conv2d_fwd_split_bnin: unsupported (mla_conv2d_bnfold_supported)");
  IGemmGeom g;
  make_fwd_geom(g, N, H, W, Cin, Cout, KH, KW, stride, pad);
  g.in_bn[0] = in_mean; g.in_bn[1] = in_invstd; g.in_bn[2] = in_gamma; g.in_bn[3] = in_beta;
  return mla_patch_launch(x, wsplit_t, y, nullptr, nullptr, bn_partial, g, bn_tiles, (hipStream_t)stream);
}

extern "C" int mla_conv2d_dgrad_split_bnmask(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout, int KH,
                                             int KW, int stride, int pad, const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles,
                                             const float* mask_gamma, const float* mask_beta, void* stream) {
  if (int rc = check_conv("mla_conv2d_dgrad_split_bnmask", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(dy && wsplit && dx && reqs && nreq == 1 && mask_gamma && mask_beta, "mla_conv2d_dgrad_split_bnmask: one reduction request and the mask's gamma / beta are required");
  MLA_REQUIRE(mla_conv2d_bnfold_supported(N, H, W, Cin, Cout, KH, KW, stride, pad), "mla_conv2d_dgrad_split_bnmask: unsupported (mla_conv2d_bnfold_supported)");
  IGemmGeom g;
  make_dgrad_geom(g, 0, 0, N, H, W, Cin, Cout, KH, KW, stride, pad);
  int tiles = 0;
  if (int rc = attach_bn_reqs("mla_conv2d_dgrad_split_bnmask", g, reqs, nreq, tiles)) return rc;
  g.mask_gb[0] = mask_gamma; g.mask_gb[1] = mask_beta;
  int ptiles = 0;
  if (int rc = mla_patch_launch(dy, wsplit, dx, nullptr, nullptr, nullptr, g, &ptiles, (hipStream_t)stream)) return rc;
  if (bn_tiles) *bn_tiles = ptiles;
  return MLA_OK;
}

extern "C" int mla_conv2d_dgrad_split(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                      void* stream) {
  return mla_conv2d_dgrad_split_bn(dy, wsplit, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, residual, relu_src, nullptr, 0, nullptr, stream);
}

extern "C" int mla_conv2d_dgrad_split_bn(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout,
                                         int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                         const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, void* stream) {
  return mla_conv2d_dgrad_split_classes(dy, wsplit, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, residual, relu_src, reqs, nreq, bn_tiles,
                                        0xF, 0xF, stream);
}

extern "C" int mla_conv2d_dgrad_split_classes(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout,
                                              int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                              const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, int class_mask,
                                              int residual_mask, void* stream) {
  if (int rc = check_conv("mla_conv2d_dgrad_split", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(Cin % 64 == 0, "mla_conv2d_dgrad_split: Cin=%d must be a multiple of 64 (the stem needs no dgrad)", Cin);
  MLA_REQUIRE(dy && wsplit && dx, "mla_conv2d_dgrad_split: null pointer");
  int tiles = 0;
  if (stride == 2 && dgrad_merge_on() && g_split_terms == 6 && g_split_cfg < 0) {
    // all requested parity classes in one launch, longest K first
    IGemmClasses cs;
    cs.n = 0;
    int order[4], nord = 0;
    IGemmGeom gc[4];
    for (int cls = 0; cls < 4; ++cls) {
      if (!((class_mask >> cls) & 1)) continue;
      make_dgrad_geom(gc[cls], cls / 2, cls % 2, N, H, W, Cin, Cout, KH, KW, stride, pad);
      if (gc[cls].M <= 0) continue;
      int pos = nord++;
      while (pos > 0 && gc[order[pos - 1]].T < gc[cls].T) { order[pos] = order[pos - 1]; --pos; }
      order[pos] = cls;
    }
    if (nord >= 2) {
      const IGemmGeom& big = gc[order[0]];
      const int cfg = pick_scfg(big.M, Cin, big.T > 0 ? big.T : 1, KH * KW == 1 ? Cout : 1 << 30);
      const int bm = scfg_bm(cfg), bn = scfg_bn(cfg);
      cs.first[0] = 0;
      for (int k = 0; k < nord; ++k) {
        const int cls = order[k];
        cs.g[k] = gc[cls];
        if (int rc = attach_bn_reqs("mla_conv2d_dgrad_split_bn", cs.g[k], reqs, nreq, tiles)) return rc;
        cs.R[k] = ((residual_mask >> cls) & 1) ? residual : nullptr;
        const int tm = cdiv(cs.g[k].M, bm);
        cs.first[k + 1] = cs.first[k] + tm * (Cin / bn);
        tiles += tm;
      }
      cs.n = nord;
      const int grid = cs.first[nord];
      hipStream_t st = (hipStream_t)stream;
      if (cfg == SCFG_256x128) igemm_split_classes_kernel<256, 128, 4, 2, 32, 1><<<grid, 512, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      else if (cfg == SCFG_128x128) igemm_split_classes_kernel<128, 128, 2, 4, 32, 1><<<grid, 512, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      else if (cfg == SCFG_256x64) igemm_split_classes_kernel<256, 64, 4, 2, 16, 4><<<grid, 512, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      else if (cfg == SCFG_192x128) igemm_split_classes_kernel<192, 128, 2, 4, 32, 1><<<grid, 512, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      else if (cfg == SCFG_128x64) igemm_split_classes_kernel<128, 64, 2, 2, 32, 1><<<grid, 256, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      else igemm_split_classes_kernel<64, 64, 2, 2, 32, 1><<<grid, 256, 0, st>>>(dy, wsplit, dx, relu_src, cs);
      MLA_CHECK_LAUNCH("igemm_split_classes_kernel");
      if (bn_tiles) *bn_tiles = tiles;
      return MLA_OK;
    }
  }
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      const int cls = py * stride + px;               // output parity class (py, px): bit of class_mask / residual_mask
      if (!((class_mask >> cls) & 1)) continue;
      const float* res = ((residual_mask >> cls) & 1) ? residual : nullptr;
      IGemmGeom g;
      make_dgrad_geom(g, py, px, N, H, W, Cin, Cout, KH, KW, stride, pad);
      if (g.M <= 0) continue;
      if (int rc = attach_bn_reqs("mla_conv2d_dgrad_split_bn", g, reqs, nreq, tiles)) return rc;
      if (use_patch(g)) {                               // stride 1: the one parity class is a 3x3 "same" convolution over dy
        int ptiles = 0;
        if (int rc = mla_patch_launch(dy, wsplit, dx, res, relu_src, nullptr, g, &ptiles, (hipStream_t)stream)) return rc;
        tiles += ptiles;
        continue;
      }
      const int cfg = pick_scfg(g.M, Cin, g.T > 0 ? g.T : 1, KH * KW == 1 ? Cout : 1 << 30);
      int rt = 0;
      if (int rc = launch_split(dy, wsplit, dx, res, relu_src, nullptr, g, cfg, (hipStream_t)stream, nullptr, nullptr, &rt)) return rc;
      tiles += rt;
    }
  if (bn_tiles) *bn_tiles = tiles;
  return MLA_OK;
}

// split-K plan of the split weight gradient: the number of pixel ranges that minimises  rounds x (pixels per range + c0),
// rounds = ceil(workgroups / resident slots) (one 128x128 workgroup per CU, two 64x64), c0 = the fixed cost of a workgroup
// (prologue, slab write, its share of the ordered reduce) in pixel-equivalents.  Calibrated on a same-box sweep of the split
// count at the ResNet-18 shapes: layer2 / layer3 (9 / 36 tiles) want ONE round of ~252 workgroups (-3 % against two), layer4
// (144 tiles) wants 5 ranges = 2.8 rounds (-5 % against 3 ranges = 1.7 rounds), the 64x64 layers two full rounds of 512.
static void wgrad_split_plan(long M, int Cin, int Cout, int T, int* span, int* splits) {
  const int BI = (Cin % 128 == 0 && Cout % 128 == 0) ? 128 : 64;
  const long tiles = (long)(Cin / BI) * T * (Cout / BI);
  const long slots = BI == 64 ? 512 : 256;
  const double c0 = 250.0;
  long best_s = 1;
  double best_cost = 0;
  const long smax = M / 256 > 1 ? (M / 256 < 512 ? M / 256 : 512) : 1;
  for (long sp = 1; sp <= smax; ++sp) {
    const long rounds = (tiles * sp + slots - 1) / slots;
    if (rounds > 4) break;                                   // more rounds only add slabs to reduce
    const double cost = (double)rounds * ((double)((M + sp - 1) / sp) + c0);
    if (sp == 1 || cost < best_cost * 0.995) { best_s = sp; best_cost = cost; }
  }
  long s = (M + best_s - 1) / best_s;
  s = ((s + BK - 1) / BK) * BK;
  if (s < 256) s = 256;
  *span = (int)s;
  *splits = (int)((M + s - 1) / s);
}

// wgrad_tr_split.hip: persistent all-taps kernel for the 64 -> 64 channel 3x3 / 1 / 1 convolutions (layer1)
bool mla_wgrad_tr_supported(int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
size_t mla_wgrad_tr_ws_bytes(int N, int H, int W, int Cin, int Cout);
int mla_wgrad_tr_launch(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                        hipStream_t st, const float* const* in_bn);
extern "C" int mla_conv2d_wgrad_tr(int on) {          // measurement hook: 0 = the per-tap kernel everywhere, 1 = default; other: query
  if (on == 0 || on == 1) g_wgrad_tr = on;
  return g_wgrad_tr;
}

extern "C" size_t mla_conv2d_wgrad_split_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  const long M = (long)N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad);
  int span, splits;
  wgrad_split_plan(M, Cin, Cout, KH * KW, &span, &splits);
  size_t b = (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
  if (mla_wgrad_tr_supported(W, Cin, Cout, KH, KW, stride, pad) && mla_wgrad_tr_ws_bytes(N, H, W, Cin, Cout) > b)
    b = mla_wgrad_tr_ws_bytes(N, H, W, Cin, Cout);
  return b;
}

int mla_wgrad_reduce(const float* part, float* dw, size_t n4, int splits, hipStream_t st);   // conv_igemm.hip

extern "C" int mla_conv2d_wgrad_split(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_conv("mla_conv2d_wgrad_split", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(Cin % 64 == 0, "mla_conv2d_wgrad_split: Cin=%d must be a multiple of 64 (the stem runs on mla_conv2d_wgrad)", Cin);
  MLA_REQUIRE(x && dy && dw && ws, "mla_conv2d_wgrad_split: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (g_wgrad_tr && mla_wgrad_tr_supported(W, Cin, Cout, KH, KW, stride, pad))
    return mla_wgrad_tr_launch(x, dy, dw, N, H, W, Cin, Cout, ws, ws_bytes, st, nullptr);
  IGemmGeom g;
  make_fwd_geom(g, N, H, W, Cin, Cout, KH, KW, stride, pad);
  g.y_bytes = (unsigned)((size_t)g.M * Cout * 4);
  int span, splits;
  wgrad_split_plan(g.M, Cin, Cout, g.T, &span, &splits);
  const size_t need = (size_t)splits * g.T * Cin * Cout * sizeof(float);
  if (ws_bytes < need) {
    mla_set_error("mla_conv2d_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  if (Cin % 128 == 0 && Cout % 128 == 0) {
    // 8 waves (64x32 wave tiles): two waves per SIMD with one workgroup per CU; same-box A/B against the 4-wave form (64x64 wave
    // tiles, a third fewer fragment reads): -9 % time on the 128x128-tile layers (2x4 and 4x2 wave grids measure the same)
    wgrad_split_kernel<128, 128, 2, 4><<<dim3((Cin / 128) * (Cout / 128) * g.T, splits), 512, 0, st>>>(x, dy, part, g, span);
  } else {
    wgrad_split_kernel<64, 64, 2, 2><<<dim3((Cin / 64) * (Cout / 64) * g.T, splits), 256, 0, st>>>(x, dy, part, g, span);
  }
  MLA_CHECK_LAUNCH("wgrad_split_kernel");
  return mla_wgrad_reduce(part, dw, (size_t)g.T * Cin * Cout / 4, splits, st);
}

extern "C" int mla_conv2d_wgrad_split_bnin(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                           int stride, int pad, const float* in_mean, const float* in_invstd, const float* in_gamma,
                                           const float* in_beta, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_conv("mla_conv2d_wgrad_split_bnin", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && dy && dw && ws && in_mean && in_invstd && in_gamma && in_beta, "mla_conv2d_wgrad_split_bnin: null pointer");
  MLA_REQUIRE(mla_conv2d_bnfold_supported(N, H, W, Cin, Cout, KH, KW, stride, pad), "mla_conv2d_wgrad_split_bnin: unsupported (mla_conv2d_bnfold_supported)");
  const float* in_bn[4] = {in_mean, in_invstd, in_gamma, in_beta};
  return mla_wgrad_tr_launch(x, dy, dw, N, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream, in_bn);
}

// ---------------------------------------------------------------------------------------------
// Linear layers of the transformer encoders on the split arithmetic (same row-window geometry as mla_linear_*).
// wsplit_t / wsplit = mla_conv2d_wsplit(w_kn, ., Cin = K, Cout = N, 1, 1, transposed = 1 / 0, .).
// ---------------------------------------------------------------------------------------------
extern "C" int mla_linear_fwd_split(const float* x, const void* wsplit_t, const float* bias, const float* residual, float* y,
                                    float* y_gelu, int groups, int rows, int x_group_rows, int x_off, int y_group_rows,
                                    int y_off, int K, int N, void* stream) {
  MLA_REQUIRE(x && wsplit_t && y, "mla_linear_fwd_split: null pointer");
  IGemmGeom g;
  if (int rc = linear_geom("mla_linear_fwd_split", g, groups, rows, x_group_rows, x_off, y_group_rows, y_off, K, N)) return rc;
  return launch_split(x, wsplit_t, y, residual, nullptr, nullptr, g, pick_scfg(g.M, N, 1), (hipStream_t)stream, bias, y_gelu);
}

extern "C" int mla_linear_dgrad_split(const float* dy, const void* wsplit, float* dx, const float* residual,
                                      const float* gelu_src, int groups, int rows, int dy_group_rows, int dy_off,
                                      int dx_group_rows, int dx_off, int K, int N, void* stream) {
  MLA_REQUIRE(dy && wsplit && dx, "mla_linear_dgrad_split: null pointer");
  IGemmGeom g;   // GEMM: [M][N] x [N][K] -> [M][K]
  if (int rc = linear_geom("mla_linear_dgrad_split", g, groups, rows, dy_group_rows, dy_off, dx_group_rows, dx_off, N, K)) return rc;
  g.epi = 1;
  return launch_split(dy, wsplit, dx, residual, gelu_src, nullptr, g, pick_scfg(g.M, K, 1), (hipStream_t)stream);
}

// wgrad_tr_split.hip: Linear weight gradient on transposing LDS reads (192 x 192 tiles; dense rows, K and N multiples of 192)
bool mla_linear_wgrad_tr_supported(long M, int K, int N);
size_t mla_linear_wgrad_tr_ws_bytes(long M, int K, int N);
int mla_linear_wgrad_tr_launch(const float* x, const float* dy, float* dw_kn, float* dbias, int M, int K, int N, void* ws, size_t ws_bytes,
                               hipStream_t st);

extern "C" size_t mla_linear_wgrad_split_ws_bytes(int M, int K, int N) {
  int span, splits;
  wgrad_split_plan(M, K, N, 1, &span, &splits);
  size_t b = (size_t)splits * K * N * sizeof(float) + (size_t)splits * N * sizeof(float);     // weight slabs + bias rows
  if (mla_linear_wgrad_tr_supported(M, K, N) && mla_linear_wgrad_tr_ws_bytes(M, K, N) > b) b = mla_linear_wgrad_tr_ws_bytes(M, K, N);
  return b;
}

extern "C" int mla_linear_wgrad_split(const float* x, const float* dy, float* dw_kn, int groups, int rows, int x_group_rows,
                                      int x_off, int K, int N, void* ws, size_t ws_bytes, void* stream) {
  return mla_linear_wgrad_split_bias(x, dy, dw_kn, nullptr, groups, rows, x_group_rows, x_off, K, N, ws, ws_bytes, stream);
}

// ... and, with dbias != null, the bias gradient dbias[N] = column sums of dy out of the same pass (the dy tiles are staged
// for the MFMA anyway; replaces a separate column-reduction kernel pair per Linear layer)
extern "C" int mla_linear_wgrad_split_bias(const float* x, const float* dy, float* dw_kn, float* dbias, int groups, int rows,
                                           int x_group_rows, int x_off, int K, int N, void* ws, size_t ws_bytes, void* stream) {
  MLA_REQUIRE(x && dy && dw_kn && ws, "mla_linear_wgrad_split: null pointer");
  IGemmGeom g;
  if (int rc = linear_geom("mla_linear_wgrad_split", g, groups, rows, x_group_rows, x_off, rows, 0, K, N)) return rc;
  g.y_bytes = (unsigned)((size_t)g.M * N * 4);
  hipStream_t st = (hipStream_t)stream;
  if (g_wgrad_tr && (groups == 1 || (x_group_rows == rows && x_off == 0)) && x_off == 0 && mla_linear_wgrad_tr_supported(g.M, K, N))
    return mla_linear_wgrad_tr_launch(x, dy, dw_kn, dbias, g.M, K, N, ws, ws_bytes, st);     // dense token rows
  int span, splits;
  wgrad_split_plan(g.M, K, N, 1, &span, &splits);
  const size_t need = (size_t)splits * K * N * sizeof(float) + (dbias ? (size_t)splits * N * sizeof(float) : 0);
  if (ws_bytes < need) {
    mla_set_error("mla_linear_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  float* bias_part = dbias ? part + (size_t)splits * K * N : nullptr;
  if (K % 128 == 0 && N % 128 == 0) {
    wgrad_split_kernel<128, 128, 2, 4><<<dim3((K / 128) * (N / 128), splits), 512, 0, st>>>(x, dy, part, g, span, bias_part);
  } else {
    wgrad_split_kernel<64, 64, 2, 2><<<dim3((K / 64) * (N / 64), splits), 256, 0, st>>>(x, dy, part, g, span, bias_part);
  }
  MLA_CHECK_LAUNCH("wgrad_split_kernel");
  if (int rc = mla_wgrad_reduce(part, dw_kn, (size_t)K * N / 4, splits, st)) return rc;
  return dbias ? mla_wgrad_reduce(bias_part, dbias, (size_t)N / 4, splits, st) : MLA_OK;
}
