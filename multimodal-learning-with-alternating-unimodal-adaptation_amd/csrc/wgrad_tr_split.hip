// Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolutions (ResNet-18 layer1, models/backbone.py:28, 31:
// eight launches per step, the largest below-par block of round 2 at 113-128 TFLOP/s) on the split-bf16 arithmetic, as a
// persistent all-taps kernel:  dW[t][ci][co] = sum over pixels  x[pixel + tap t][ci] * dy[pixel][co].
//
// wgrad_split_kernel (conv_igemm_split.hip) gives every (tap, pixel span) its own workgroup, so x and dy are gathered and split
// nine times over and a 64 x 64 tile sees 16 FLOP per gathered byte.  Here one workgroup owns ALL NINE taps:
//   * it walks 8 x 8-pixel output tiles; the 10 x 10 x 64 input patch and the 8 x 8 x 64 dy tile are loaded once, split once
//     into three bf16 planes and stored in LDS in their NATURAL pixel-major layout ([channel half][pixel][32 channels], 64-B
//     rows), double-buffered (the next tile is in flight in registers while this one computes);
//   * the contraction runs over pixels, so both MFMA operands are the TRANSPOSE of those images: they are read with
//     ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads 4 pixel rows x 16 channels and every lane
//     receives 4 pixels of ITS channel).  A tap is then nothing but a different row address -- no im2col, no shifted copies,
//     no alignment problem -- and the same dy fragments serve all nine taps of a step;
//   * a wave keeps its 9 x (32 x 32) accumulators for the whole launch: one slab per workgroup at the end, then the ordered
//     reduce of conv_igemm.hip (no atomics, bitwise reproducible).
// 8 waves = 2 step groups x (2 ci blocks x 2 co blocks); a step = 16 pixels (two tile rows), 54 MFMAs per wave.
#include "split_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int WT_TH = 8, WT_TW = 8;                      // output tile
constexpr int WT_SW = WT_TW + 2, WT_SH = WT_TH + 2;      // input patch (halo 1)
constexpr int WT_SPX = WT_SW * WT_SH;                    // 100 patch pixels
constexpr int WT_TPX = WT_TH * WT_TW;                    // 64 tile pixels
constexpr int WT_XPL = 2 * WT_SPX * 16;                  // dwords per x plane: [2 halves][100 px][16 dwords]
constexpr int WT_YPL = 2 * WT_TPX * 16;                  // dwords per dy plane
constexpr int WT_BUF = 3 * WT_XPL + 3 * WT_YPL;          // dwords per buffer (15744 = 61.5 KB)
constexpr int WT_XU = (WT_SPX * 16 + 511) / 512;         // staging passes (512 float4 slots each) over the input patch: 4 (1600 slots)
constexpr int WT_YU = WT_TPX * 16 / 512;                 // ... and over the dy tile: 2 (1024 slots)
constexpr int WT_NLD = WT_XU + WT_YU;                    // float4 held per thread: 6
constexpr int WT_RACC = 9 * 64 * 64;                     // floats of one slab

struct WtGeom {
  int N, H, W, tilesY, tilesX, ntiles;
  unsigned x_bytes;
  const float* in_bn[4];     // BNIN: x is a convolution output y, the operand is relu(bn(y)): {mean, invstd, gamma, beta} per channel
};

__device__ __forceinline__ unsigned wt_off_or_oob(int ok, unsigned off) { return off | ((unsigned)ok - 1u); }

__device__ __forceinline__ s16x4 lds_tr(const unsigned* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

template <bool BNIN>
__global__ __launch_bounds__(512, 2) void wgrad_tr_split_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                 float* __restrict__ slabs, const WtGeom g) {
  __shared__ __attribute__((aligned(16))) unsigned S[(2 * WT_BUF + 64 > WT_RACC ? 2 * WT_BUF + 64 : WT_RACC)];
  __shared__ __attribute__((aligned(16))) float bnt[BNIN ? 4 : 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;     // step group, ci block, co block
  const int i = lane & 31, h = lane >> 5;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, g.x_bytes);    // dy has the shape of x (64 -> 64, stride 1)

  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    const int per = g.tilesY * g.tilesX;
    n = t / per;
    const int rem = t - n * per;
    const int ty = rem / g.tilesX;
    oy0 = ty * WT_TH;
    ox0 = (rem - ty * g.tilesX) * WT_TW;
  };
  // ---- staging: pass u < WT_XU covers float4 slot tid + 512 u of the input patch (pixel = slot >> 4, 4 channels = slot & 15;
  // slots past 1600 are masked), pass u >= WT_XU slot tid + 512 (u - WT_XU) of the dy tile
  f32x4 pre[WT_NLD];
  unsigned pre_ok = 0;                             // BNIN: which input-patch slots hold real pixels (image borders must stay 0 after BatchNorm)
  if constexpr (BNIN) {
    if (tid < 256) bnt[tid >> 6][tid & 63] = g.in_bn[tid >> 6][tid & 63];
  }
  auto stage_load = [&](int t) {
    int n, oy0, ox0;
    decode(t, n, oy0, ox0);
    unsigned okbits = 0;
#pragma unroll
    for (int u = 0; u < WT_NLD; ++u) {
      const bool isx = u < WT_XU;                                                // compile-time
      const int sl = tid + 512 * (isx ? u : u - WT_XU);
      const int px = sl >> 4, c4 = sl & 15;
      const int wdt = isx ? WT_SW : WT_TW;
      const int r = px / wdt, c = px - r * wdt;
      const int iy = oy0 + r - (isx ? 1 : 0), ix = ox0 + c - (isx ? 1 : 0);     // the patch starts one pixel up / left of the tile
      const int ok = (int)(px < (isx ? WT_SPX : WT_TPX)) & (int)((unsigned)iy < (unsigned)g.H) & (int)((unsigned)ix < (unsigned)g.W);
      const unsigned off = ((unsigned)((n * g.H + iy) * g.W + ix) * 64u + (unsigned)c4 * 4u) * 4u;
      pre[u] = buf_load4(isx ? xr : yr, wt_off_or_oob(ok, off), 0);
      okbits |= (unsigned)ok << u;
    }
    if constexpr (BNIN) pre_ok = okbits;
  };
  int pz = 0;                                      // an opaque 0, renewed per tile in the BNIN variant (see wgrad_flat_tr_kernel): the LDS store
                                                   // addresses of the six passes are then recomputed instead of living in ~12 hoisted registers
  auto stage_store = [&](int buf, int u) {
    const bool isx = u < WT_XU;
    const int sl = tid + pz + 512 * (isx ? u : u - WT_XU);
    const int c4 = sl & 15;
    int px = sl >> 4;
    if (isx && px >= WT_SPX) px = WT_SPX - 1 + 0 * px;                          // masked slots (zeros) land on ... see below
    if constexpr (BNIN) {
      if (isx) {                                   // relu(bn(y)) of the thread's four channels (c4 = tid & 15 in every pass), bn_apply's expression
        const int c = 4 * c4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(&bnt[0][c]), is = *reinterpret_cast<const f32x4*>(&bnt[1][c]);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(&bnt[2][c]), be = *reinterpret_cast<const f32x4*>(&bnt[3][c]);
        const unsigned okm = 0u - ((pre_ok >> u) & 1u);          // all ones / zero: branch-free (a select makes the compiler skip the block by exec mask)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pre[u][e] = __uint_as_float(__float_as_uint(fmaxf(bn_val1(pre[u][e], mu[e], is[e], ga[e], be[e]), 0.f)) & okm);
      }
    }
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair<true>(pre[u][0], pre[u][1], h0, m0, l0);
    split_pair<true>(pre[u][2], pre[u][3], h1, m1, l1);
    const int npx = isx ? WT_SPX : WT_TPX, pl = isx ? WT_XPL : WT_YPL;
    // image [half = c4 >> 3][pixel][(c4 & 7) * 2 dwords].  Patch slots past pixel 99 (pass 3, lanes of pixels 100..127) must not
    // store: their pixel index is redirected to the scratch pixel row behind the dy planes (branch-free)
    const bool live = !isx || (sl >> 4) < WT_SPX;
    unsigned* dst = S + buf * WT_BUF + (isx ? 0 : 3 * WT_XPL) + ((c4 >> 3) * npx + px) * 16 + (c4 & 7) * 2;
    if (!live) dst = S + 2 * WT_BUF + (tid & 31) * 2;                           // 64 scratch dwords after the two buffers
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(live ? dst + pl : dst) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(live ? dst + 2 * pl : dst) = u32x2{l0, l1};
  };

  // ---- transposing-read addressing (ds_read_b64_tr_b16): within a 16-lane group lane 4q + p supplies the address of pixel
  // row q, channels 4p .. 4p+3 of the group's 16 channels; the group's lane c receives channel c of the 4 pixels.
  // MFMA lanes 0-31 = 32 channels (two groups of 16), h = pixel half.  dwords: pixel row = 16 dwords, 16 channels = 8 dwords.
  const int q = (lane & 15) >> 2, p = lane & 3, grp = (lane >> 4) & 1;
  // A (x patch, channel half wi): step s of group sg covers tile rows 2 (2 sg + s) + h, pixels x = 0..7: lane pixel (row, q | q + 4)
  // B (dy tile, channel half wj): tile pixel (2 (2 sg + s) + h) * 8 + q | q + 4
  const int a_lane = (wi * WT_SPX + (2 * (2 * sg) + h + 1) * WT_SW + q + 1) * 16 + grp * 8 + p * 2;      // tap (0, 0), step 0, read 0
  const int b_lane = 3 * WT_XPL + (wj * WT_TPX + (2 * (2 * sg) + h) * WT_TW + q) * 16 + grp * 8 + p * 2;

  f32x16 acc[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t9][e] = 0.f;

  struct Fr { s16x4 lo[3], hi[3]; };                  // one operand fragment: pixels 0-3 | 4-7 of this lane's half, three planes
  auto read_a = [&](const unsigned* Sc, int s, int t9, Fr& f) {      // compile-time s, t9
    const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
    const unsigned* ap = Sc + a_lane + ((2 * s + dy) * WT_SW + dx) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      f.lo[pl] = lds_tr(ap + pl * WT_XPL);
      f.hi[pl] = lds_tr(ap + pl * WT_XPL + 4 * 16);
    }
  };
  auto read_b = [&](const unsigned* Sc, int s, Fr& f) {
    const unsigned* bp = Sc + b_lane + (2 * s * WT_TW) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      f.lo[pl] = lds_tr(bp + pl * WT_YPL);
      f.hi[pl] = lds_tr(bp + pl * WT_YPL + 4 * 16);
    }
  };
  auto frag = [&](const Fr& f, int pl) {              // dword-granular: lets the register coalescer place lo | hi without moves
    const u32x2 lo = __builtin_bit_cast(u32x2, f.lo[pl]), hi = __builtin_bit_cast(u32x2, f.hi[pl]);
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  for (int idx = tid; idx < 2 * WT_BUF; idx += 512) S[idx] = 0u;
  int cur = 0;
  int t = blockIdx.x;
  if (t < g.ntiles) stage_load(t);
  __syncthreads();
  if (t < g.ntiles) {
#pragma unroll
    for (int u = 0; u < WT_NLD; ++u) stage_store(0, u);
  }
  __syncthreads();
  constexpr int U = 2 * 9;                             // units of a tile for one wave: (step s, tap)
  for (; t < g.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if constexpr (BNIN) asm volatile("" : "+v"(pz));
    if (tn < g.ntiles) stage_load(tn);                 // in flight while this tile computes
    const unsigned* Sc = S + cur * WT_BUF;
    Fr a0, a1, b0, b1;
    read_b(Sc, 0, b0);
    read_a(Sc, 0, 0, a0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = u / 9, t9 = u - s * 9;
      Fr& ac = (u & 1) ? a1 : a0;
      Fr& an = (u & 1) ? a0 : a1;
      Fr& bc = (s && !BNIN) ? b1 : b0;
      if (BNIN && u == 9) read_b(Sc, 1, b0);           // (BNIN: one dy fragment set -- the BatchNorm temporaries take its 12 registers --, one exposed LDS latency per tile)
      if (u + 1 < U) read_a(Sc, (u + 1) / 9, (u + 1) % 9, an);
      if (!BNIN && u == 4) read_b(Sc, 1, b1);          // the second step's dy fragments, well ahead
      if (u >= U - WT_NLD) stage_store(cur ^ 1, u - (U - WT_NLD));   // unconditional (stale registers without a next tile: never read)
#pragma unroll
      for (int term = 0; term < 6; ++term)
        acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ac, TERM_A[term]), frag(bc, TERM_B[term]), acc[t9], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);          // the LDS reads of the next unit first
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- the two step groups' accumulators are summed in order through LDS; slab [9][64 ci][64 co] of this workgroup
  float* R = reinterpret_cast<float*>(S);
  for (int w = 0; w < 2; ++w) {
    if (sg == w) {
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int o = (t9 * 64 + row) * 64 + wj * 32 + i;
          if (w == 0) R[o] = acc[t9][e];
          else slabs[(size_t)blockIdx.x * WT_RACC + o] = R[o] + acc[t9][e];
        }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same all-taps scheme for the 128 / 256 / 512-channel 3x3 / stride 1 / pad 1 convolutions (layer2..layer4: 18 launches
// per step on wgrad_split_kernel<128, 128> at 165-180 TFLOP/s, where every (tap, channel tile) re-gathers and re-splits both
// operands).  Their maps are 28 / 14 / 7 pixels wide (audio: 16 / 8 / 4): 8 x 8 spatial tiles would waste up to half of the MFMA
// work on them, so a tile here is 64 consecutive FLAT pixels of the (N, H, W) tensor and the input patch is the flat range
// [p0 - W - 1, p0 + 64 + W + 1): a tap is still one row offset (dy W + dx), and a tile pixel whose neighbour falls off its image
// (mask bit per pixel and tap, written to LDS by the staging pass) supplies the address of a zero row to the transposing read
// instead.  A workgroup owns one (64 input channels) x (64 output channels) block pair and every `splits`-th tile; its slab is a
// 64 x 64 window of a full-size [9][Cin][Cout] slab per split, so the ordered reduce of conv_igemm.hip applies unchanged.
// All block pairs of one split are neighbours in the XCD-remapped order: the tiles they share come out of one L2.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WF_NPMAX = 122;                            // patch pixels 64 + 2 W + 2 for W <= 28
constexpr int WF_ZR = WF_NPMAX;                          // the zero row behind every half plane of x
constexpr int WF_XPL = 2 * (WF_NPMAX + 1) * 16;          // dwords per x plane: [2 halves][123 rows][16 dwords]
constexpr int WF_YPL = 2 * 64 * 16;
constexpr int WF_VT = 3 * WF_XPL + 3 * WF_YPL;           // validity table of the buffer's tile: 64 dwords
constexpr int WF_BUF = WF_VT + 64;                       // 18016 dwords = 70.4 KB
constexpr int WF_XU = 4, WF_YU = 2, WF_NLD = WF_XU + WF_YU;

struct WfGeom {
  int M, H, W, Cin, Cout, ntiles, splits, pairs, pairs_j;
  unsigned x_bytes, y_bytes;
};

__global__ __launch_bounds__(512, 1) void wgrad_flat_tr_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                float* __restrict__ slabs, const WfGeom g) {
  __shared__ __attribute__((aligned(16))) unsigned S[(2 * WF_BUF + 64 > WT_RACC ? 2 * WF_BUF + 64 : WT_RACC)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, g.y_bytes);
  const int lw = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = lw / g.pairs, pair = lw - split * g.pairs;
  const int ci0 = (pair / g.pairs_j) * 64, co0 = (pair % g.pairs_j) * 64;
  const int gW = g.W, NP = 64 + 2 * gW + 2;

  f32x4 pre[WF_NLD];
  auto stage_load = [&](int t, int vbuf) {         // vbuf: the LDS buffer this tile will be staged into (its validity table is written here)
    const int p0 = t * 64;
#pragma unroll
    for (int u = 0; u < WF_NLD; ++u) {
      const bool isx = u < WF_XU;
      const int sl = tid + 512 * (isx ? u : u - WF_XU);
      const int px = sl >> 4, c4 = sl & 15;
      const int pix = isx ? p0 - gW - 1 + px : p0 + px;
      const int ok = (int)(px < (isx ? NP : 64)) & (int)((unsigned)pix < (unsigned)g.M);
      const unsigned off = isx ? ((unsigned)pix * (unsigned)g.Cin + (unsigned)(ci0 + c4 * 4)) * 4u
                               : ((unsigned)pix * (unsigned)g.Cout + (unsigned)(co0 + c4 * 4)) * 4u;
      pre[u] = buf_load4(isx ? xr : yr, wt_off_or_oob(ok, off), 0);
    }
    if (wave == 0) {                                  // tap-validity bits of tile pixel `lane`
      const int pix = p0 + lane;
      const int hw = g.H * gW;
      const int rem = pix - (pix / hw) * hw;
      const int y = rem / gW, x = rem - y * gW;
      unsigned m = 0;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int yy = y + t9 / 3 - 1, xx = x + t9 % 3 - 1;
        m |= (unsigned)((int)((unsigned)yy < (unsigned)g.H) & (int)((unsigned)xx < (unsigned)gW) & (int)(pix < g.M)) << t9;
      }
      S[vbuf * WF_BUF + WF_VT + lane] = m;            // no reader: that buffer's tile was finished before the last barrier
    }
  };
  int pz = 0;                                       // an opaque 0, renewed per tile: keeps tile-invariant address arithmetic (LDS store
                                                    // addresses of the six staging passes, the 36 patch rows of the fragment reads) from
                                                    // being hoisted out of the tile loop into ~45 registers that then spill
  auto stage_store = [&](int buf, int u) {
    const bool isx = u < WF_XU;
    const int sl = tid + pz + 512 * (isx ? u : u - WF_XU);
    const int c4 = sl & 15, px = sl >> 4;
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair<true>(pre[u][0], pre[u][1], h0, m0, l0);
    split_pair<true>(pre[u][2], pre[u][3], h1, m1, l1);
    const int rows = isx ? WF_NPMAX + 1 : 64, pl = isx ? WF_XPL : WF_YPL;
    const bool live = !isx || px < NP;                // patch slots past the last row (and the zero row itself) are not written
    unsigned* dst = S + buf * WF_BUF + (isx ? 0 : 3 * WF_XPL) + ((c4 >> 3) * rows + px) * 16 + (c4 & 7) * 2;
    if (!live) dst = S + 2 * WF_BUF + (tid & 31) * 2;
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(live ? dst + pl : dst) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(live ? dst + 2 * pl : dst) = u32x2{l0, l1};
  };

  const int q = (lane & 15) >> 2, p = lane & 3, grp = (lane >> 4) & 1;
  const int a_lane = wi * (WF_NPMAX + 1) * 16 + grp * 8 + p * 2;               // + row * 16
  const int b_lane = 3 * WF_XPL + wj * 64 * 16 + grp * 8 + p * 2;             // + pixel * 16
  int pixl[2];                                                                 // this lane's tile pixel (first of lo | hi = +4) in step s
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) pixl[s2] = (2 * (2 * sg + s2) + h) * 8 + q;

  f32x16 acc[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t9][e] = 0.f;

  struct Fr { s16x4 lo[3], hi[3]; };
  unsigned vm[2];
  auto read_a = [&](const unsigned* Sc, int s2, int t9, Fr& f) {               // compile-time s2, t9
    const int toff = (t9 / 3 - 1) * gW + (t9 % 3 - 1) + gW + 1;                // wave-uniform
    const int rl = ((vm[s2] >> t9) & 1u) ? pixl[s2] + pz + toff : WF_ZR;
    const int rh = ((vm[s2] >> (t9 + 9)) & 1u) ? pixl[s2] + pz + 4 + toff : WF_ZR;
    const unsigned* al = Sc + a_lane + rl * 16;
    const unsigned* ah = Sc + a_lane + rh * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      f.lo[pl] = lds_tr(al + pl * WF_XPL);
      f.hi[pl] = lds_tr(ah + pl * WF_XPL);
    }
  };
  auto read_b = [&](const unsigned* Sc, int s2, Fr& f) {
    const unsigned* bp = Sc + b_lane + pixl[s2] * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      f.lo[pl] = lds_tr(bp + pl * WF_YPL);
      f.hi[pl] = lds_tr(bp + pl * WF_YPL + 4 * 16);
    }
  };
  auto frag = [&](const Fr& f, int pl) {
    const u32x2 lo = __builtin_bit_cast(u32x2, f.lo[pl]), hi = __builtin_bit_cast(u32x2, f.hi[pl]);
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  for (int idx = tid; idx < 2 * WF_BUF + 64; idx += 512) S[idx] = 0u;
  __syncthreads();
  int cur = 0;
  int t = split;
  if (t < g.ntiles) stage_load(t, 0);
  __syncthreads();
  if (t < g.ntiles) {
#pragma unroll
    for (int u = 0; u < WF_NLD; ++u) stage_store(0, u);
  }
  __syncthreads();
  constexpr int U = 2 * 9;
  for (; t < g.ntiles; t += g.splits) {
    const int tn = t + g.splits;
    if (tn < g.ntiles) stage_load(tn, cur ^ 1);
    const unsigned* Sc = S + cur * WF_BUF;
    asm volatile("" : "+v"(pz));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) vm[s2] = Sc[WF_VT + pixl[s2]] | (Sc[WF_VT + pixl[s2] + 4] << 9);   // tap bits of the lo | hi pixel
    Fr a0, a1, b0;
    read_b(Sc, 0, b0);
    read_a(Sc, 0, 0, a0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s2 = u / 9, t9 = u - s2 * 9;
      Fr& ac = (u & 1) ? a1 : a0;
      Fr& an = (u & 1) ? a0 : a1;
      Fr& bc = b0;
      if (u == 9) read_b(Sc, 1, b0);                     // (one exposed LDS latency per tile: a second dy fragment set does not fit the 256 VGPRs)
      if (u + 1 < U) read_a(Sc, (u + 1) / 9, (u + 1) % 9, an);
      if (u >= U - WF_NLD) stage_store(cur ^ 1, u - (U - WF_NLD));
#pragma unroll
      for (int term = 0; term < 6; ++term)
        acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ac, TERM_A[term]), frag(bc, TERM_B[term]), acc[t9], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- the two step groups' accumulators are summed in order through LDS; window (ci0, co0) of slab `split`
  float* R = reinterpret_cast<float*>(S);
  float* out = slabs + (size_t)split * 9 * g.Cin * g.Cout + (size_t)(ci0 + wi * 32 + 4 * h) * g.Cout + co0 + wj * 32 + i;
  const int rl = ((wi * 32 + 4 * h) * 64) + wj * 32 + i;
  if (sg == 0) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e = 0; e < 16; ++e) R[rl + (t9 * 64 + (e & 3) + 8 * (e >> 2)) * 64] = acc[t9][e];
  }
  __syncthreads();
  if (sg == 1) {
    const unsigned tap_stride = (unsigned)g.Cin * (unsigned)g.Cout;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        out[t9 * tap_stride + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)g.Cout] = R[rl + (t9 * 64 + (e & 3) + 8 * (e >> 2)) * 64] + acc[t9][e];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear weight gradient dW[k][n] = sum_m X[m][k] dY[m][n] (the transformer rows: 97 launches per M3AE step on
// wgrad_split_kernel<128, 128>, 30 % of that step) with the same transposing-read scheme: no taps, so a wave spends its nine
// 32 x 32 accumulators on a 96 x 96 block of dW instead -- 3 x-fragments and 3 dy-fragments feed 54 MFMAs per 16-row step (0.67
// LDS reads per MFMA against 2 for one block).  Workgroup = 192 x 192 outputs: 2 step groups x (2 x 2) waves; a tile is 32 token
// rows of both operands ([6 column groups][32 rows][32 columns] per bf16 plane, natural layout, 72 KB per buffer, two buffers);
// grid = (K / 192) x (N / 192) x splits over the rows, one 192 x 192 window of a full-size slab per split, ordered reduce.
// The bias gradient (column sums of dY) is accumulated from the staging registers of the workgroups with k-tile 0.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LW_MS = 32;                                // token rows per tile
constexpr int LW_PL = 6 * LW_MS * 16;                    // dwords per plane of one operand: 3072
constexpr int LW_BUF = 6 * LW_PL;                        // x planes 0-2, dy planes 3-5: 18432 dwords = 72 KB
constexpr int LW_NLD = 6;                                // float4 per thread and tile: 3 passes per operand (1536 slots each)

struct LwGeom {
  int M, K, N, tilesN, pairs, splits, mtiles;
  unsigned x_bytes, y_bytes;
};

template <bool BIAS>
__global__ __launch_bounds__(512, 1) void linear_wgrad_tr_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                  float* __restrict__ slabs, float* __restrict__ bias_part,
                                                                  const LwGeom g) {
  __shared__ __attribute__((aligned(16))) unsigned S[(2 * LW_BUF + 64 > 192 * 192 ? 2 * LW_BUF + 64 : 192 * 192)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, g.y_bytes);
  const int lw = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = lw / g.pairs, pair = lw - split * g.pairs;
  const int kt = pair / g.tilesN, nt = pair - kt * g.tilesN;
  const int k0 = kt * 192, n0 = nt * 192;
  const bool do_bias = BIAS && kt == 0;

  f32x4 pre[LW_NLD];
  f32x4 bacc[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) bacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int pz = 0;                                            // opaque 0 per tile (see wgrad_flat_tr_kernel)
  auto stage_load = [&](int t) {
    const int m0 = t * LW_MS;
#pragma unroll
    for (int u = 0; u < LW_NLD; ++u) {
      const bool isx = u < 3;
      const int sl = tid + pz + 512 * (isx ? u : u - 3);
      const int row = sl / 48, c4 = sl - row * 48;
      const int m = m0 + row;
      const unsigned off = isx ? ((unsigned)m * (unsigned)g.K + (unsigned)(k0 + c4 * 4)) * 4u
                               : ((unsigned)m * (unsigned)g.N + (unsigned)(n0 + c4 * 4)) * 4u;
      pre[u] = buf_load4(isx ? xr : yr, wt_off_or_oob((int)(m < g.M), off), 0);
    }
  };
  auto stage_store = [&](int buf, int u) {
    const bool isx = u < 3;
    const int sl = tid + pz + 512 * (isx ? u : u - 3);
    const int row = sl / 48, c4 = sl - row * 48;
    if (BIAS && !isx) bacc[u - 3] += pre[u];             // rows past M were loaded as zeros
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair<true>(pre[u][0], pre[u][1], h0, m0, l0);
    split_pair<true>(pre[u][2], pre[u][3], h1, m1, l1);
    unsigned* dst = S + buf * LW_BUF + (isx ? 0 : 3 * LW_PL) + ((c4 >> 3) * LW_MS + row) * 16 + (c4 & 7) * 2;
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + LW_PL) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(dst + 2 * LW_PL) = u32x2{l0, l1};
  };

  const int q = (lane & 15) >> 2, p = lane & 3, grp = (lane >> 4) & 1;
  const int rd_lane = (16 * sg + 8 * h + q) * 16 + grp * 8 + p * 2;            // row of the lo read; hi = + 4 rows
  const int a_lane = (3 * wi) * LW_MS * 16 + rd_lane;
  const int b_lane = 3 * LW_PL + (3 * wj) * LW_MS * 16 + rd_lane;

  f32x16 acc[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t9][e] = 0.f;

  struct Fr { s16x4 lo[3], hi[3]; };
  auto read_fr = [&](const unsigned* base, int cgi, Fr& f) {                   // column group cgi of this wave's three
    const unsigned* ap = base + cgi * LW_MS * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      f.lo[pl] = lds_tr(ap + pl * LW_PL);
      f.hi[pl] = lds_tr(ap + pl * LW_PL + 4 * 16);
    }
  };
  auto frag = [&](const Fr& f, int pl) {
    const u32x2 lo = __builtin_bit_cast(u32x2, f.lo[pl]), hi = __builtin_bit_cast(u32x2, f.hi[pl]);
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  int cur = 0;
  int t = split;
  if (t < g.mtiles) stage_load(t);
  if (t < g.mtiles) {
#pragma unroll
    for (int u = 0; u < LW_NLD; ++u) stage_store(0, u);
  }
  __syncthreads();
  for (; t < g.mtiles; t += g.splits) {
    const int tn = t + g.splits;
    asm volatile("" : "+v"(pz));
    if (tn < g.mtiles) {
      stage_load(tn);                                    // in flight while this tile computes
    } else if (BIAS) {                                   // no next tile: the staging passes below still run (their output is never read),
#pragma unroll
      for (int u = 3; u < LW_NLD; ++u) pre[u] = f32x4{0.f, 0.f, 0.f, 0.f};     // but must not count the last tile's dy twice
    }
    const unsigned* Sc = S + cur * LW_BUF;
    Fr a0, a1, b[3];
    read_fr(Sc + b_lane, 0, b[0]);
    read_fr(Sc + a_lane, 0, a0);
    read_fr(Sc + b_lane, 1, b[1]);
    read_fr(Sc + b_lane, 2, b[2]);
#pragma unroll
    for (int ai = 0; ai < 3; ++ai) {
      // x fragments: two sets in turn (the next one is read while this one multiplies); the bias variant has 12 registers less (its
      // column sums) and keeps one set, re-read after the MFMAs that use it were issued
      Fr& ac = BIAS ? a0 : ((ai & 1) ? a1 : a0);
      Fr& an = BIAS ? a0 : ((ai & 1) ? a0 : a1);
      if (!BIAS && ai + 1 < 3) read_fr(Sc + a_lane, ai + 1, an);
      stage_store(cur ^ 1, 2 * ai);                      // unconditional (stale registers without a next tile: never read)
      stage_store(cur ^ 1, 2 * ai + 1);
#pragma unroll
      for (int bj = 0; bj < 3; ++bj)
#pragma unroll
        for (int term = 0; term < 6; ++term)
          acc[ai * 3 + bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ac, TERM_A[term]), frag(b[bj], TERM_B[term]), acc[ai * 3 + bj], 0, 0, 0);
      if (BIAS && ai + 1 < 3) read_fr(Sc + a_lane, ai + 1, an);
      if (!BIAS) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);           // the next x fragment's reads first
#pragma unroll
      for (int m = 0; m < 18; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        if (m % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- the two step groups' accumulators are summed in order through LDS; window (k0, n0) of slab `split`
  float* R = reinterpret_cast<float*>(S);
  const int rl = ((wi * 96 + 4 * h) * 192) + wj * 96 + i;
  if (sg == 0) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e = 0; e < 16; ++e) R[rl + ((t9 / 3) * 32 + (e & 3) + 8 * (e >> 2)) * 192 + (t9 % 3) * 32] = acc[t9][e];
  }
  __syncthreads();
  if (sg == 1) {
    float* out = slabs + (size_t)split * g.K * g.N + (size_t)(k0 + wi * 96 + 4 * h) * g.N + n0 + wj * 96 + i;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (t9 / 3) * 32 + (e & 3) + 8 * (e >> 2), c = (t9 % 3) * 32;
        out[(unsigned)r * (unsigned)g.N + c] = R[rl + r * 192 + c] + acc[t9][e];
      }
  }
  if (BIAS) {
    __syncthreads();
    if (do_bias) {                                       // (uniform per workgroup) per-thread sums -> [32 row slots][192] -> column sums in row order
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int sl = tid + 512 * u, row = sl / 48, c4 = sl - row * 48;
        *reinterpret_cast<f32x4*>(R + row * 192 + c4 * 4) = bacc[u];
      }
    }
    __syncthreads();
    if (do_bias && tid < 192) {
      float sb = 0.f;
      for (int r = 0; r < LW_MS; ++r) sb += R[r * 192 + tid];
      bias_part[(size_t)split * g.N + n0 + tid] = sb;
    }
  }
}

int wt_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

}  // namespace

int mla_wgrad_reduce(const float* part, float* dw, size_t n4, int splits, hipStream_t st);   // conv_igemm.hip

// (not part of the C ABI: called by mla_conv2d_wgrad_split in conv_igemm_split.hip)
static bool wf_supported(int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % 64 == 0 && Cout % 64 == 0 && 64 + 2 * W + 2 <= WF_NPMAX;
}
static void wf_plan(long M, int Cin, int Cout, int* pairs, int* splits) {
  const long ntiles = (M + 63) / 64;
  *pairs = (Cin / 64) * (Cout / 64);
  long s = wt_cus() / *pairs;
  if (s < 1) s = 1;
  if (s > ntiles) s = ntiles;
  *splits = (int)s;
}
bool mla_wgrad_tr_supported(int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  if (Cin == 64 && Cout == 64 && KH == 3 && KW == 3 && stride == 1 && pad == 1) return true;      // 8 x 8 spatial tiles, any map
  return wf_supported(W, Cin, Cout, KH, KW, stride, pad);
}
size_t mla_wgrad_tr_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  if (Cin == 64 && Cout == 64) return (size_t)wt_cus() * WT_RACC * sizeof(float);
  int pairs, splits;
  wf_plan((long)N * H * W, Cin, Cout, &pairs, &splits);
  return (size_t)splits * 9 * Cin * Cout * sizeof(float);
}

int mla_wgrad_tr_launch(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                        hipStream_t st, const float* const* in_bn) {
  const size_t need = mla_wgrad_tr_ws_bytes(N, H, W, Cin, Cout);
  if (ws_bytes < need) {
    mla_set_error("mla_conv2d_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  MLA_REQUIRE((size_t)N * H * W * Cin * 4 < 0xFFFFFFF0UL && (size_t)N * H * W * Cout * 4 < 0xFFFFFFF0UL,
              "mla_conv2d_wgrad_split: tensors must be < 4 GiB");
  if (Cin == 64 && Cout == 64) {
    WtGeom g;
    g.N = N; g.H = H; g.W = W;
    g.tilesY = cdiv(H, WT_TH); g.tilesX = cdiv(W, WT_TW);
    g.ntiles = N * g.tilesY * g.tilesX;
    g.x_bytes = (unsigned)((size_t)N * H * W * 64 * 4);
    for (int k = 0; k < 4; ++k) g.in_bn[k] = in_bn ? in_bn[k] : nullptr;
    const int grid = g.ntiles < wt_cus() ? g.ntiles : wt_cus();
    if (in_bn) wgrad_tr_split_kernel<true><<<grid, 512, 0, st>>>(x, dy, (float*)ws, g);
    else wgrad_tr_split_kernel<false><<<grid, 512, 0, st>>>(x, dy, (float*)ws, g);
    MLA_CHECK_LAUNCH("wgrad_tr_split_kernel");
    return mla_wgrad_reduce((const float*)ws, dw, (size_t)WT_RACC / 4, grid, st);
  }
  MLA_REQUIRE(!in_bn, "mla_conv2d_wgrad_split_bnin: folded BatchNorm exists for the 64 -> 64 channel convolutions only");
  WfGeom g;
  g.M = N * H * W; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
  g.ntiles = cdiv(g.M, 64);
  wf_plan(g.M, Cin, Cout, &g.pairs, &g.splits);
  g.pairs_j = Cout / 64;
  g.x_bytes = (unsigned)((size_t)g.M * Cin * 4);
  g.y_bytes = (unsigned)((size_t)g.M * Cout * 4);
  wgrad_flat_tr_kernel<<<g.pairs * g.splits, 512, 0, st>>>(x, dy, (float*)ws, g);
  MLA_CHECK_LAUNCH("wgrad_flat_tr_kernel");
  return mla_wgrad_reduce((const float*)ws, dw, (size_t)9 * Cin * Cout / 4, g.splits, st);
}

// Linear weight gradient (not part of the C ABI: called by mla_linear_wgrad_split_bias in conv_igemm_split.hip).  Dense rows only.
static void lw_plan(long M, int K, int N, int* pairs, int* splits) {
  *pairs = (K / 192) * (N / 192);
  const long mtiles = (M + LW_MS - 1) / LW_MS;
  long s = wt_cus() / *pairs;
  if (s < 1) s = 1;
  if (s > mtiles) s = mtiles;
  *splits = (int)s;
}
bool mla_linear_wgrad_tr_supported(long M, int K, int N) { return K % 192 == 0 && N % 192 == 0 && M >= 4 * LW_MS; }
size_t mla_linear_wgrad_tr_ws_bytes(long M, int K, int N) {
  int pairs, splits;
  lw_plan(M, K, N, &pairs, &splits);
  return (size_t)splits * K * N * sizeof(float) + (size_t)splits * N * sizeof(float);
}
int mla_linear_wgrad_tr_launch(const float* x, const float* dy, float* dw_kn, float* dbias, int M, int K, int N, void* ws, size_t ws_bytes,
                               hipStream_t st) {
  const size_t need = mla_linear_wgrad_tr_ws_bytes(M, K, N);
  if (ws_bytes < need) {
    mla_set_error("mla_linear_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  MLA_REQUIRE((size_t)M * K * 4 < 0xFFFFFFF0UL && (size_t)M * N * 4 < 0xFFFFFFF0UL, "mla_linear_wgrad_split: tensors must be < 4 GiB");
  LwGeom g;
  g.M = M; g.K = K; g.N = N;
  g.tilesN = N / 192;
  lw_plan(M, K, N, &g.pairs, &g.splits);
  g.mtiles = cdiv(M, LW_MS);
  g.x_bytes = (unsigned)((size_t)M * K * 4);
  g.y_bytes = (unsigned)((size_t)M * N * 4);
  float* part = (float*)ws;
  float* bias_part = part + (size_t)g.splits * K * N;
  if (dbias) linear_wgrad_tr_kernel<true><<<g.pairs * g.splits, 512, 0, st>>>(x, dy, part, bias_part, g);
  else linear_wgrad_tr_kernel<false><<<g.pairs * g.splits, 512, 0, st>>>(x, dy, part, nullptr, g);
  MLA_CHECK_LAUNCH("linear_wgrad_tr_kernel");
  if (int rc = mla_wgrad_reduce(part, dw_kn, (size_t)K * N / 4, g.splits, st)) return rc;
  return dbias ? mla_wgrad_reduce(bias_part, dbias, (size_t)N / 4, g.splits, st) : MLA_OK;
}

