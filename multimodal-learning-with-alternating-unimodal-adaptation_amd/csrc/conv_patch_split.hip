// 3x3 / stride 1 / pad 1 convolution forward and input gradient (models/backbone.py:28, 31 and autograd: 26 of the 40 conv
// layers of the two ResNet-18 encoders, forward and backward) on the split-bf16 arithmetic, with the A operand served from an
// LDS-resident input PATCH instead of per-tap gathers.
//
// igemm_split_kernel (conv_igemm_split.hip) treats every (tap, 32-channel chunk) as an independent K stage: each stage
// gathers its 256 x 32 A tile from global memory again, splits it into bf16 planes again (88 VALU per wave and stage) and
// stores it to LDS again (12 ds_write), although the nine taps of a chunk read the SAME input pixels shifted by one row or
// column -- the round-2 ablations attribute ~17 % of the kernel to that staging.  Here, per 32-channel chunk,
//   * the input rows the tile's 256 output pixels touch (its image rows plus one halo row above / below, one zero column left /
//     right) are loaded ONCE, split ONCE and kept in LDS as three bf16 planes [pixel][32 channels] (64-B rows, the XOR
//     chunk swizzle of igemm_split_kernel): a tap is a row offset into that patch, and all nine taps read their A fragments
//     from it with the same conflict-free ds_read_b128 -- 9x fewer gathers and operand splits;
//   * the pre-split weight planes of a (tap, chunk) stage are register-staged one stage ahead into a two-slot ring (three
//     16-B loads and three ds_write_b128 per thread: the B side was never the cost);
//   * a K stage is then fragment reads + 48 MFMAs per wave, one barrier per stage as before; the next chunk's patch is in
//     flight in registers during the nine stages of this chunk.
// Geometry, tap tables (forward: (kh - 1, kw - 1); input gradient: (1 - kh, 1 - kw) on the transposed weights) and the whole
// epilogue (BatchNorm statistics, residual, ReLU mask, fused BatchNorm-backward reductions) are those of igemm_split_kernel.
#include "split_common.h"
#include <stdlib.h>

namespace {

constexpr int PT_BM = 256;
constexpr int PT_MAXPX = 480;                 // patch pixels incl. the zero pixel (index PT_MAXPX - 1)
constexpr int PT_PPL = PT_MAXPX * 16;         // dwords per patch plane
constexpr int PT_NPRE = (PT_MAXPX * 8 + 511) / 512;   // float4 patch slots per thread (32 channels = 8 float4 per pixel): 8
constexpr int PT_ZP = PT_MAXPX - 1;

template <int AUX>
__device__ __forceinline__ float buf_load1i(rsrc_t r, unsigned voff, int imm) {       // imm: folded into the instruction's 12-bit offset
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(voff + (unsigned)imm), 0, AUX));
}
__device__ __forceinline__ void buf_store1i(rsrc_t r, float v, unsigned voff, int imm) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)(voff + (unsigned)imm), 0, 0);
}

template <int BN>
__global__ __launch_bounds__(512, 2) void patch_split_kernel(const float* __restrict__ X, const void* __restrict__ Wsp, float* Y,
                                                              const float* R, const float* MASK, float* __restrict__ part,
                                                              const IGemmGeom g) {
  constexpr int WM = 4, WN = 2, MI = 2, NI = BN / WN / 32;
  constexpr int TPS = 128 / BN;                      // taps per K stage: the B slot always holds 128 rows x 32 k x 3 planes
  constexpr int BSLOT = 3 * 128 * 16;                // dwords per B slot
  __shared__ __attribute__((aligned(16))) unsigned P[3 * PT_PPL];
  __shared__ __attribute__((aligned(16))) unsigned Bs[2 * BSLOT];
  __shared__ int4 rowinfo[PT_BM];  // {n*H*W or -1, patch pixel of the row at tap (0,0), row-validity bits (1: y-1 ok, 2: y+1 ok), output pixel}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int gridN = g.CO / BN;
  int tm, tn;
  tile_coords(wg, (int)gridDim.x / gridN, gridN, tm, tn);
  const int gH = g.H, gW = g.W, gC = g.C, gCO = g.CO;
  const int prow = gW + 2;
  const int m0 = tm * PT_BM;
  const int r0 = m0 / gW;                                       // first global image row (n * H + y) of the tile
  const int mlast = min(g.M, m0 + PT_BM) - 1;
  const int nrows = mlast / gW - r0 + 3;                        // + one halo row above and below
  const int npx = nrows * prow;

  for (int r = tid; r < PT_BM; r += 512) {
    const int m = m0 + r;
    int4 info = make_int4(-1, PT_ZP, 0, 0);
    if (m < g.M) {
      const int gr = m / gW, ox = m - gr * gW;
      const int n = gr / gH, oy = gr - n * gH;
      info.x = n * gH * gW;
      info.y = (gr - r0 + 1) * prow + ox + 1;
      info.z = (oy >= 1 ? 1 : 0) | (oy + 1 < gH ? 2 : 0);
      info.w = m;                                               // stride-1 "same" convolution: output pixel = m
    }
    rowinfo[r] = info;
  }
  if (tid < 48) P[(tid >> 4) * PT_PPL + PT_ZP * 16 + (tid & 15)] = 0u;         // the zero pixel of the three planes
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int i = lane & 31, h = lane >> 5;
  int pb[MI], pv[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int4 info = rowinfo[wm * 64 + mi * 32 + i];
    pb[mi] = info.y;
    pv[mi] = info.x >= 0 ? (info.z | 4) : 0;                    // bit 2: the row exists
  }
  const int swz = (i >> 2) & 3;
  const int b_rd = (wn * (BN / WN) + i) * 16;
  const unsigned plane_bytes = g.w_bytes / 2;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), wr = make_rsrc(Wsp, 3 * plane_bytes);

  // ---- patch staging: slot s = tid + 512 u -> (patch pixel s >> 3, float4 s & 7 of the chunk's 32 channels)
  f32x4 pre[PT_NPRE];
  auto patch_load = [&](int c0) {
#pragma unroll
    for (int u = 0; u < PT_NPRE; ++u) {
      const int s = tid + 512 * u, pp = s >> 3, c4 = s & 7;
      const int pr = pp / prow, pc = pp - pr * prow;
      const int gr = r0 - 1 + pr, x = pc - 1;
      const int ok = (int)(pp < npx) & (int)((unsigned)gr < (unsigned)(g.N * gH)) & (int)((unsigned)x < (unsigned)gW);
      const unsigned off = ((unsigned)(gr * gW + x) * (unsigned)gC + (unsigned)(c0 + 4 * c4)) * 4u;
      pre[u] = buf_load4(xr, off | ((unsigned)ok - 1u), 0);
    }
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int u = 0; u < PT_NPRE; ++u) {
      const int s = tid + 512 * u, pp = s >> 3, c4 = s & 7;
      if (pp >= PT_ZP) continue;                               // (never the zero pixel; pixels in [npx, ZP) receive zeros: harmless)
      unsigned h0, m0_, l0, h1, m1, l1;
      split_pair<true>(pre[u][0], pre[u][1], h0, m0_, l0);
      split_pair<true>(pre[u][2], pre[u][3], h1, m1, l1);
      unsigned* dst = P + pp * 16 + ((((c4 >> 1) ^ ((pp >> 2) & 3)) << 2) + (c4 & 1) * 2);
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + PT_PPL) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * PT_PPL) = u32x2{l0, l1};
    }
  };
  // ---- weight planes of K stage (first tap t0, chunk c0) -> registers -> B slot: thread (row = tid >> 2, LDS chunk slot
  // tid & 3) of each plane, 16 B each (the swizzle is applied on the source side); 128 rows = TPS taps x BN output columns.
  // (LDS-DMA -- buffer_load ... lds -- was tried first: hipcc serialises the DMA instructions with vmcnt(0) and waits for all
  // of them before the first fragment read of every stage, which exposes the whole load latency: register staging it is.)
  u32x4 breg[3];
  auto b_load = [&](int t0, int c0) {
    const int row = tid >> 2;
    const int tt = row / BN, n = row - tt * BN;
    const int t = t0 + tt;
    const int q = (tid & 3) ^ ((row >> 2) & 3);
    const int tp = g.tap[t < g.T ? t : g.T - 1];
    const unsigned off = (unsigned)(((tap_wt(tp) * gCO + tn * BN + n) * gC + c0) * 2 + q * 16) | ((unsigned)(t < g.T) - 1u);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) breg[pl] = buf_load4u(wr, off, pl * plane_bytes);
  };
  auto b_store = [&](int slot) {
    unsigned* dst = Bs + slot * BSLOT + tid * 4;            // row (tid >> 2) * 16 dwords + slot (tid & 3) * 4 = tid * 4
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 128 * 16) = breg[pl];
  };

  struct Frags { bf16x8_t a[3][MI], b[3][NI]; };
  auto load_frags = [&](int slot, int tt, int ppv[MI], int kk, Frags& f) {
    const unsigned* Br = Bs + slot * BSLOT + tt * BN * 16 + b_rd + (((kk * 2 + h) ^ swz) << 2);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned* Ar = P + pl * PT_PPL + ppv[mi] * 16 + (((kk * 2 + h) ^ ((ppv[mi] >> 2) & 3)) << 2);
        f.a[pl][mi] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ar));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        f.b[pl][ni] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Br + (pl * 128 + ni * 32) * 16));
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

  const int NC = gC / 32;                        // channel chunks
  constexpr int SPC = (9 + TPS - 1) / TPS;       // K stages per chunk
  const int NS = NC * SPC;
  patch_load(0);
  b_load(0, 0);
  patch_store();
  b_store(0);
  __syncthreads();
  Frags f0, f1;
  int c = 0, st = 0;                             // chunk, stage within the chunk
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1;
    {                                            // next stage's weights -> registers (stored into the other slot mid-stage)
      int st1 = st + 1, c1 = c;
      if (st1 == SPC) { st1 = 0; ++c1; }
      if (s + 1 < NS) b_load(st1 * TPS, c1 * 32);
    }
    if (st == 0 && c + 1 < NC) patch_load((c + 1) * 32);       // next chunk's patch: in flight for the nine taps of this one
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int t = st * TPS + tt;
      if (t < 9) {
        const int tp = g.tap[t];
        const int dy = tap_dy(tp), dx = tap_dx(tp);
        const int toff = dy * prow + dx;                        // wave-uniform
        const int need = 4 | (dy < 0 ? 1 : 0) | (dy > 0 ? 2 : 0);
        int ppv[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ppv[mi] = (pv[mi] & need) == need ? pb[mi] + toff : PT_ZP;
        load_frags(cur, tt, ppv, 0, f0);
        load_frags(cur, tt, ppv, 1, f1);
        mma_frags(f0);
        if (tt == 0 && s + 1 < NS) b_store(cur ^ 1);           // that slot's readers passed the previous barrier
        mma_frags(f1);
      }
    }
    ++st;
    if (st == SPC) {
      st = 0;
      ++c;
      if (c < NC) {
        __syncthreads();                         // every wave is done with this chunk's patch
        patch_store();
      }
    }
    __syncthreads();
  }
  igemm_epilogue<PT_BM, BN, WM, WN>(acc, rowinfo, reinterpret_cast<float*>(P), Y, R, MASK, part, nullptr, nullptr, g, tm, tn);
}


// ---------------------------------------------------------------------------------------------------------------------
// Persistent variant for the 64 -> 64 channel layers (layer1: 16 of the 76 forward / input-gradient calls of a step, and the
// most memory-heavy ones: 38.7 GFLOP against 270 MB (forward) ... 670 MB (input gradient with residual, ReLU mask and the
// BatchNorm-backward reduction reading its input) per call).  With one 145-KB workgroup per CU nothing overlaps a tile's
// epilogue: the round-3 kernel trace shows patch_split_kernel<64> at 266 us per call against 212 us with the epilogue
// compiled out.  Here a workgroup walks tiles; the whole weight / patch stage stream runs on across tile boundaries (the
// next tile's first patch is prefetched during this tile's last chunk), and the EPILOGUE of tile t -- its loads of residual /
// mask / BatchNorm input, the arithmetic, the stores and the column statistics -- is cut into eight 4-row pieces that ride in
// the MFMA stages of tile t + 1: a piece's operands are loaded in one stage and consumed in the next.  Statistics are kept per
// lane for the whole launch and reduced once at the end: one partial row per workgroup (no barrier inside the epilogue).
// Cin = Cout = 64 only: 2 chunks x 5 two-tap stages = 10 unrolled stages per tile.
// ---------------------------------------------------------------------------------------------------------------------
// BNIN (forward): the gathered tensor is a convolution output y and the operand is relu(bn(y)) -- BatchNorm + ReLU applied to every loaded
// value on its way into the patch planes (padding stays zero), so the activation tensor between conv1 and conv2 of a BasicBlock is never
// written or read.  BNMASK (input gradient): the ReLU mask of that activation is re-formed from the BatchNorm input the epilogue reads
// anyway for the fused reduction (bn_x[0]) instead of loading the activation.  Both use bn_val1, the expression of bn_apply_kernel.
template <bool HAS_R, bool HAS_MASK, int NREQ, bool STATS, bool BNIN = false, bool BNMASK = false>
__global__ __launch_bounds__(512) void patch64p_kernel(const float* __restrict__ X, const void* __restrict__ Wsp, float* Y,
                                                           const float* R, const float* MASK, double* __restrict__ part,
                                                           const IGemmGeom g, int ntiles) {
  constexpr int BN = 64, WN = 2, MI = 2, TPS = 2, SPC = 5, NC = 2, NSTG = NC * SPC;
  constexpr int BSLOT = 3 * 128 * 16;
  static_assert(!BNMASK || (NREQ >= 1 && !HAS_MASK), "the mask is re-formed from the first reduction request's BatchNorm input");
  __shared__ __attribute__((aligned(16))) unsigned P[3 * PT_PPL];
  __shared__ __attribute__((aligned(16))) unsigned Bs[2 * BSLOT];
  __shared__ __attribute__((aligned(16))) float bnt[BNIN ? 4 : 1][64];       // BNIN: {mean, invstd, gamma, beta} of the 64 input channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i = lane & 31, h = lane >> 5;
  const int gH = g.H, gW = g.W;
  const int prow = gW + 2;
  const int col = wn * 32 + i;                                  // this lane's output column
  if constexpr (BNIN) {
    if (tid < 256) bnt[tid >> 6][tid & 63] = g.in_bn[tid >> 6][tid & 63];        // (read after the first barrier)
  }
  const unsigned plane_bytes = g.w_bytes / 2;
  const unsigned t_bytes = (unsigned)g.M * 64u * 4u;           // every (M, 64) fp32 tensor of this launch
  const rsrc_t xr = make_rsrc(X, g.x_bytes), wr = make_rsrc(Wsp, 3 * plane_bytes), yr = make_rsrc(Y, t_bytes);
  const rsrc_t rr = make_rsrc(HAS_R ? R : Y, t_bytes), mr = make_rsrc(HAS_MASK ? MASK : Y, t_bytes);
  const rsrc_t b0r = make_rsrc(NREQ > 0 ? g.bn_x[0] : Y, t_bytes), b1r = make_rsrc(NREQ > 1 ? g.bn_x[1] : Y, t_bytes);

  if (tid < 48) P[(tid >> 4) * PT_PPL + PT_ZP * 16 + (tid & 15)] = 0u;         // the zero pixel of the three planes

  // ---- patch staging of (tile, chunk)
  f32x4 pre[PT_NPRE];
  unsigned pre_ok = 0;                           // BNIN: which of the eight staged slots hold real pixels (padding must stay 0 after BatchNorm)
  const unsigned prow_inv = (65536u + (unsigned)prow - 1u) / (unsigned)prow;      // pp / prow == (pp * prow_inv) >> 16 for pp < 512, prow < 64
  auto patch_load = [&](int tile, int c0) {
    const int m0 = tile * PT_BM;
    const int r0 = m0 / gW;
    const int mlast = min(g.M, m0 + PT_BM) - 1;
    const int npx = tile < ntiles ? (mlast / gW - r0 + 3) * prow : 0;
    int zero = 0;
    asm volatile("" : "+v"(zero));               // (keeps the eight slot addresses from being hoisted out of the tile loop: 30 VGPRs)
    unsigned okbits = 0;
#pragma unroll
    for (int u = 0; u < PT_NPRE; ++u) {
      const int s = tid + 512 * u + zero, pp = s >> 3, c4 = s & 7;
      const int pr = (int)(((unsigned)pp * prow_inv) >> 16), pc = pp - pr * prow;
      const int gr = r0 - 1 + pr, x = pc - 1;
      const int ok = (int)(pp < npx) & (int)((unsigned)gr < (unsigned)(g.N * gH)) & (int)((unsigned)x < (unsigned)gW);
      const unsigned off = ((unsigned)(gr * gW + x) * 64u + (unsigned)(c0 + 4 * c4)) * 4u;
      pre[u] = buf_load4(xr, off | ((unsigned)ok - 1u), 0);
      okbits |= (unsigned)ok << u;
    }
    if constexpr (BNIN) pre_ok = okbits;
  };
  auto patch_store = [&](int c0) {               // c0: the channel chunk the staged registers hold (BNIN: selects the BatchNorm parameters)
    f32x4 bmu_, bis_, bga_, bbe_;
    if constexpr (BNIN) {
      const int c = c0 + 4 * (tid & 7);          // slot s = tid + 512 u: the thread's four channels are the same in every pass
      bmu_ = *reinterpret_cast<const f32x4*>(&bnt[0][c]);
      bis_ = *reinterpret_cast<const f32x4*>(&bnt[1][c]);
      bga_ = *reinterpret_cast<const f32x4*>(&bnt[2][c]);
      bbe_ = *reinterpret_cast<const f32x4*>(&bnt[3][c]);
    }
#pragma unroll
    for (int u = 0; u < PT_NPRE; ++u) {
      const int s = tid + 512 * u, pp = s >> 3, c4 = s & 7;
      if constexpr (BNIN) {
        const unsigned okm = 0u - ((pre_ok >> u) & 1u);          // all ones / zero: branch-free
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pre[u][e] = __uint_as_float(__float_as_uint(fmaxf(bn_val1(pre[u][e], bmu_[e], bis_[e], bga_[e], bbe_[e]), 0.f)) & okm);
      }
      unsigned h0, m0_, l0, h1, m1, l1;
      split_pair<true>(pre[u][0], pre[u][1], h0, m0_, l0);
      split_pair<true>(pre[u][2], pre[u][3], h1, m1, l1);
      // (the last slots would fall on the zero pixel: redirected onto pixel ZP - 1, which no tile reaches and which receives zeros)
      const int ppc = pp < PT_ZP ? pp : PT_ZP - 1;
      unsigned* dst = P + ppc * 16 + ((((c4 >> 1) ^ ((ppc >> 2) & 3)) << 2) + (c4 & 1) * 2);
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + PT_PPL) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * PT_PPL) = u32x2{l0, l1};
    }
  };
  u32x4 breg[3];
  const int b_row = tid >> 2, b_tt = b_row / BN, b_n = b_row - b_tt * BN;
  const unsigned b_base = (unsigned)(b_n * 64 * 2 + ((tid & 3) ^ ((b_row >> 2) & 3)) * 16);
  auto b_load = [&](int t0, int c0) {
    const int t = t0 + b_tt;
    const int tp = g.tap[t < 9 ? t : 8];
    unsigned off = b_base;
    asm volatile("" : "+v"(off));
    off = (off + (unsigned)(tap_wt(tp) * 64 * 64 * 2 + c0 * 2)) | ((unsigned)(t < 9) - 1u);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) breg[pl] = buf_load4u(wr, off, pl * plane_bytes);
  };
  auto b_store = [&](int slot) {
    unsigned* dst = Bs + slot * BSLOT + tid * 4;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 128 * 16) = breg[pl];
  };
  const int swz = (i >> 2) & 3;
  const int b_rd = (wn * 32 + i) * 16;
  struct Frags { bf16x8_t a[3][MI], b[3]; };
  auto load_frags = [&](int slot, int tt, const int (&ppv)[MI], int kk, Frags& f) {
    const unsigned* Br = Bs + slot * BSLOT + tt * BN * 16 + b_rd + (((kk * 2 + h) ^ swz) << 2);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned* Ar = P + pl * PT_PPL + ppv[mi] * 16 + (((kk * 2 + h) ^ ((ppv[mi] >> 2) & 3)) << 2);
        f.a[pl][mi] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ar));
      }
      f.b[pl] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Br + pl * 128 * 16));
    }
  };

  // ---- epilogue state: the finished accumulators of the previous tile and the launch-long column statistics
  f32x16 prev[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) prev[mi][e] = 0.f;
  double csum = 0.0, csq = 0.0;                                 // STATS: forward BatchNorm statistics of this lane's column
  float rb0 = 0.f, rb1[2] = {0.f, 0.f};                         // NREQ: sum v, sum v * xhat_q over every tile of this workgroup
  float bmu[2] = {0.f, 0.f}, bis[2] = {0.f, 0.f};
  float mga = 0.f, mbe = 0.f;
  if constexpr (BNMASK) { mga = g.mask_gb[0][col]; mbe = g.mask_gb[1][col]; }
  if constexpr (NREQ > 0) { bmu[0] = g.bn_mean[0][col]; bis[0] = g.bn_invstd[0][col]; }
  if constexpr (NREQ > 1) { bmu[1] = g.bn_mean[1][col]; bis[1] = g.bn_invstd[1][col]; }
  struct Piece { float r[4], mk[4], bx[2][4]; };
  // row (e4, piece p) of the previous tile, this lane's column: tile base + a constant; rows past M fall outside the buffer ranges
  // (loads give 0, stores are dropped) and their accumulators are exact zeros (the zero pixel): no row masks anywhere.
  unsigned prev_off = (unsigned)g.M * 256u;                     // no previous tile yet: every row is out of range
  auto piece_load = [&](int p, Piece& pc) {
    const unsigned off = prev_off + (unsigned)(((p >> 2) * 32 + 8 * (p & 3)) * 256);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      if constexpr (HAS_R) pc.r[e4] = buf_load1i<0>(rr, off, e4 * 256);
      if constexpr (HAS_MASK) pc.mk[e4] = buf_load1i<0>(mr, off, e4 * 256);
      if constexpr (NREQ > 0) pc.bx[0][e4] = buf_load1i<0>(b0r, off, e4 * 256);
      if constexpr (NREQ > 1) pc.bx[1][e4] = buf_load1i<0>(b1r, off, e4 * 256);
    }
  };
  auto piece_finish = [&](int p, const Piece& pc) {
    const int mi = p >> 2, eq = p & 3;
    const unsigned off = prev_off + (unsigned)((mi * 32 + 8 * eq) * 256);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      float v = prev[mi][4 * eq + e4];
      if constexpr (HAS_R) v += pc.r[e4];
      if constexpr (HAS_MASK) v = pc.mk[e4] > 0.f ? v : 0.f;
      if constexpr (BNMASK) v = fmaxf(bn_val1(pc.bx[0][e4], bmu[0], bis[0], mga, mbe), 0.f) > 0.f ? v : 0.f;     // relu(bn(y)) > 0, as bn_apply formed it
      buf_store1i(yr, v, off, e4 * 256);
      if constexpr (STATS) {                                      // fp64 as in igemm_epilogue (rows past M are exact zeros); rides in the MFMA shadow
        const double vd = (double)v;
        csum += vd;
        csq = fma(vd, vd, csq);
      }
      if constexpr (NREQ > 0) {
        rb0 += v;
        rb1[0] = fmaf(v, (pc.bx[0][e4] - bmu[0]) * bis[0], rb1[0]);
      }
      if constexpr (NREQ > 1) rb1[1] = fmaf(v, (pc.bx[1][e4] - bmu[1]) * bis[1], rb1[1]);
    }
  };
  // ---- stage stream
  int tile = blockIdx.x;
  patch_load(tile, 0);
  b_load(0, 0);
  __syncthreads();                               // the zero pixel (and the BatchNorm table) is written
  patch_store(0);
  b_store(0);
  __syncthreads();
  Frags f0;
  Piece pcA;
  for (; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * PT_BM;
    const int r0 = m0 / gW;
    int pb[MI], pv[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {            // this lane's two output rows: patch pixel at tap (0, 0) and validity bits
      const int m = m0 + wm * 64 + mi * 32 + i;
      const int gr = m / gW, ox = m - gr * gW;
      const int n = gr / gH, oy = gr - n * gH;
      pb[mi] = (gr - r0 + 1) * prow + ox + 1;
      pv[mi] = m < g.M ? (4 | (oy >= 1 ? 1 : 0) | (oy + 1 < gH ? 2 : 0)) : 0;
    }
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      const int c = k / SPC, st = k - c * SPC, cur = k & 1;
      {                                          // next stage's weights (the stage sequence repeats every tile)
        const int k1 = (k + 1) % NSTG;
        b_load((k1 % SPC) * TPS, (k1 / SPC) * 32);
      }
      if (k == 0) patch_load(tile, 32);                          // this tile's second chunk
      if (k == SPC) patch_load(tile + (int)gridDim.x, 0);        // the next tile's first chunk (masked off past the last tile)
      // previous tile's epilogue: piece k's operands are requested now, piece k - 1 (requested a stage ago) is finished
      if (k >= 1 && k <= 8) piece_finish(k - 1, pcA);
      if (k < 8) piece_load(k, pcA);
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int t = st * TPS + tt;
        if (t < 9) {
          const int tp = g.tap[t];
          const int dy = tap_dy(tp), dx = tap_dx(tp);
          const int toff = dy * prow + dx;
          const int need = 4 | (dy < 0 ? 1 : 0) | (dy > 0 ? 2 : 0);
          int ppv[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ppv[mi] = (pv[mi] & need) == need ? pb[mi] + toff : PT_ZP;
          load_frags(cur, tt, ppv, 0, f0);
#pragma unroll
          for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.a[TERM_A[term]][mi], f0.b[TERM_B[term]], acc[mi], 0, 0, 0);
          if (tt == 0) b_store(cur ^ 1);                          // that slot's readers passed the previous barrier
          load_frags(cur, tt, ppv, 1, f0);
#pragma unroll
          for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.a[TERM_A[term]][mi], f0.b[TERM_B[term]], acc[mi], 0, 0, 0);
        }
      }
      if (st == SPC - 1) {                       // chunk boundary: every wave is done with this patch
        __syncthreads();
        patch_store(c == 0 ? 32 : 0);            // this tile's second chunk / the next tile's first
      }
      __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) prev[mi] = acc[mi];
    prev_off = ((unsigned)(m0 + wm * 64 + 4 * h) * 64u + (unsigned)col) * 4u;
  }
  // ---- the last tile's epilogue, then the launch-long statistics: reduce over the row halves (h) and the four row waves
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    piece_load(p, pcA);
    piece_finish(p, pcA);
  }
  if constexpr (STATS || NREQ > 0) {
    __syncthreads();
    double* redd = reinterpret_cast<double*>(P);                // [wm 4][2][64] doubles / [wm 4][3][64] floats
    float* redf = reinterpret_cast<float*>(P) + 4096;
    if constexpr (STATS) {
      csum += __shfl_xor(csum, 32, 64);
      csq += __shfl_xor(csq, 32, 64);
      if (h == 0) { redd[(wm * 2 + 0) * 64 + col] = csum; redd[(wm * 2 + 1) * 64 + col] = csq; }
    }
    if constexpr (NREQ > 0) {
      rb0 += __shfl_xor(rb0, 32, 64);
      rb1[0] += __shfl_xor(rb1[0], 32, 64);
      rb1[1] += __shfl_xor(rb1[1], 32, 64);
      if (h == 0) { redf[(wm * 3 + 0) * 64 + col] = rb0; redf[(wm * 3 + 1) * 64 + col] = rb1[0]; redf[(wm * 3 + 2) * 64 + col] = rb1[1]; }
    }
    __syncthreads();
    if (tid < 64) {
      if constexpr (STATS) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += redd[(w * 2 + 0) * 64 + tid]; q += redd[(w * 2 + 1) * 64 + tid]; }
        part[((size_t)blockIdx.x * 2 + 0) * 64 + tid] = s;
        part[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = q;
      }
      if constexpr (NREQ > 0) {
        float s = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += redf[(w * 3 + 0) * 64 + tid]; q0 += redf[(w * 3 + 1) * 64 + tid]; q1 += redf[(w * 3 + 2) * 64 + tid]; }
        const size_t o = ((size_t)(g.bn_tile0 + blockIdx.x) * 2) * 64 + tid;
        g.bn_part[0][o] = s;
        g.bn_part[0][o + 64] = q0;
        if constexpr (NREQ > 1) { g.bn_part[1][o] = s; g.bn_part[1][o + 64] = q1; }
      }
    }
  }
}

}  // namespace

// (not part of the C ABI: called by launch_split in conv_igemm_split.hip)
bool mla_patch_supported(const IGemmGeom& g, bool force) {
  if (g.T != 9 || g.sy != 1 || g.sx != 1 || g.osy != 1 || g.osx != 1 || g.ooy != 0 || g.oox != 0) return false;
  if (g.OH != g.H || g.OW != g.W || g.OHF != g.H || g.OWF != g.W || g.C % 32 != 0 || g.CO % 64 != 0) return false;
  int seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = (int)(signed char)(g.tap[t] & 0xff), dx = (int)(signed char)((g.tap[t] >> 8) & 0xff);
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    seen |= 1 << ((dy + 1) * 3 + dx + 1);
  }
  if (seen != 0x1ff) return false;
  if ((255 / g.W + 4) * (g.W + 2) > PT_MAXPX - 1) return false;             // the patch of any 256-pixel tile must fit
  if (force) return true;
  // One 8-wave workgroup per CU and 256-row tiles only: the kernel wins (same-box layer sweep: +3 ... +12 % over the per-tap
  // gather-GEMM at layer1, layer2 and audio layer3) where its grid fills the chip in whole rounds, and loses where the last
  // round is mostly empty (visual layer3: 294 workgroups, layer4: 128 / 148) -- there the per-tap kernel's 128-row tiles win.
  const int BN = g.CO % 128 == 0 ? 128 : 64;
  const long wgs = (long)cdiv(g.M, PT_BM) * (g.CO / BN);
  const long rounds = (wgs + 255) / 256;
  static int min_fill = -1;                                                  // percent of the slots of its rounds that must be used
  if (min_fill < 0) {
    const char* e = getenv("MLA_PATCH_MIN_FILL");
    min_fill = e ? atoi(e) : 75;
  }
  return wgs * 100 >= rounds * 256 * min_fill;
}

static int g_patch_persistent = -1;       // -1: $MLA_PATCH_PERSISTENT (default 1)
static bool patch_persistent_on() {
  if (g_patch_persistent < 0) {
    const char* e = getenv("MLA_PATCH_PERSISTENT");
    g_patch_persistent = (e && e[0] == '0') ? 0 : 1;
  }
  return g_patch_persistent != 0;
}

// does the persistent 64 -> 64 kernel (and with it the folded-BatchNorm variants) serve this geometry?
bool mla_patch64p_usable(const IGemmGeom& g) {
  return g.C == 64 && g.CO == 64 && patch_persistent_on() && (long)g.M * 64 * 4 < 0xFFFFFFF0L && cdiv(g.M, PT_BM) >= 2 &&
         mla_patch_supported(g, true);
}

int mla_patch_launch(const float* X, const void* Wsp, float* Y, const float* R, const float* MASK, float* part, const IGemmGeom& g,
                     int* bn_tiles, hipStream_t st) {
  const int BN = g.CO % 128 == 0 ? 128 : 64;
  const int tiles = cdiv(g.M, PT_BM);
  const int total = tiles * (g.CO / BN);
  if (total <= 0) { if (bn_tiles) *bn_tiles = 0; return MLA_OK; }
  const int nreq = g.bn_x[0] ? (g.bn_x[1] ? 2 : 1) : 0;
  // layer1 (64 -> 64): the persistent kernel with the epilogue riding in the next tile's MFMA stages, for the operand
  // combinations the training step uses (everything else: the one-tile-per-workgroup kernel below)
  if (g.C == 64 && g.CO == 64 && patch_persistent_on() && (long)g.M * 64 * 4 < 0xFFFFFFF0L && tiles >= 2) {
    const int grid = tiles < 256 ? tiles : 256;
    double* pd = reinterpret_cast<double*>(part);
    bool done = true;
#define P64(R_, M_, N_, S_) patch64p_kernel<R_, M_, N_, S_><<<grid, 512, 0, st>>>(X, Wsp, Y, R, MASK, pd, g, tiles)
    if (g.in_bn[0]) {                                                            // forward over relu(bn(x))
      if (!R && !MASK && nreq == 0 && part) patch64p_kernel<false, false, 0, true, true, false><<<grid, 512, 0, st>>>(X, Wsp, Y, R, MASK, pd, g, tiles);
      else if (!R && !MASK && nreq == 0) patch64p_kernel<false, false, 0, false, true, false><<<grid, 512, 0, st>>>(X, Wsp, Y, R, MASK, pd, g, tiles);
      else done = false;
    } else if (g.mask_gb[0]) {                                                   // input gradient, ReLU mask from the BatchNorm input of request 0
      if (!R && !MASK && nreq == 1 && !part) patch64p_kernel<false, false, 1, false, false, true><<<grid, 512, 0, st>>>(X, Wsp, Y, R, MASK, pd, g, tiles);
      else done = false;
    }
    else if (!R && !MASK && nreq == 0 && part) P64(false, false, 0, true);       // forward, training
    else if (!R && !MASK && nreq == 0) P64(false, false, 0, false);              // forward, evaluation
    else if (!R && MASK && nreq == 1 && !part) P64(false, true, 1, false);       // conv2 input gradient
    else if (R && MASK && nreq == 1 && !part) P64(true, true, 1, false);         // conv1 input gradient, block below without downsample
    else if (R && !MASK && nreq == 0 && !part) P64(true, false, 0, false);       // first block (after the max-pool)
    else if (R && MASK && nreq == 0 && !part) P64(true, true, 0, false);         // (reductions not fused)
    else if (!R && MASK && nreq == 0 && !part) P64(false, true, 0, false);
    else done = false;
#undef P64
    if (done) {
      MLA_CHECK_LAUNCH("patch64p_kernel");
      if (bn_tiles) *bn_tiles = grid;
      return MLA_OK;
    }
  }
  if (g.in_bn[0] || g.mask_gb[0]) {           // folded BatchNorm exists in the persistent 64 -> 64 kernel only: never drop it silently
    mla_set_error("mla_conv2d_*_bn{in,mask}: unsupported shape / operand combination (see mla_conv2d_bnfold_supported)");
    return MLA_ERR_INVALID_ARG;
  }
  if (bn_tiles) *bn_tiles = tiles;
  if (BN == 128) patch_split_kernel<128><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, g);
  else patch_split_kernel<64><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, g);
  MLA_CHECK_LAUNCH("patch_split_kernel");
  return MLA_OK;
}
