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

namespace {

constexpr int PT_BM = 256;
constexpr int PT_MAXPX = 480;                 // patch pixels incl. the zero pixel (index PT_MAXPX - 1)
constexpr int PT_PPL = PT_MAXPX * 16;         // dwords per patch plane
constexpr int PT_NPRE = (PT_MAXPX * 8 + 511) / 512;   // float4 patch slots per thread (32 channels = 8 float4 per pixel): 8
constexpr int PT_ZP = PT_MAXPX - 1;

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
  return wgs * 4 >= rounds * 256 * 3;                                        // >= 75 % of the slots of its rounds are used
}

int mla_patch_launch(const float* X, const void* Wsp, float* Y, const float* R, const float* MASK, float* part, const IGemmGeom& g,
                     int* bn_tiles, hipStream_t st) {
  const int BN = g.CO % 128 == 0 ? 128 : 64;
  const int total = cdiv(g.M, PT_BM) * (g.CO / BN);
  if (bn_tiles) *bn_tiles = cdiv(g.M, PT_BM);
  if (total <= 0) return MLA_OK;
  if (BN == 128) patch_split_kernel<128><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, g);
  else patch_split_kernel<64><<<total, 512, 0, st>>>(X, Wsp, Y, R, MASK, part, g);
  MLA_CHECK_LAUNCH("patch_split_kernel");
  return MLA_OK;
}
