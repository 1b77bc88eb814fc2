// The ResNet stem convolution (7x7, stride 2, pad 3, 1 or 3 -> 64 channels; models/backbone.py:79-83, 149) on the split-bf16
// arithmetic of conv_igemm_split.hip, as a PERSISTENT patch-loader kernel pair (forward here, weight gradient below).
//
// Why its own kernel: with Cin = 1 / 3 the gather-GEMM of conv_igemm.hip degenerates -- K = 49 / 147 is 2-5 K steps, every
// step waits a full load latency behind a one-step prefetch (357 us with MFMAs, loads and stores compiled
// out), the scalar element gathers fetch every input pixel ~12 times, and the exact-fp32 MFMA bounds the visual stem at
// 288 us.  Here
//   * a workgroup stays resident and walks 16 x 16 output tiles; the 7 x 7 x Cin x 64 weights are split once per workgroup
//     into three bf16 planes that live in LDS for the kernel's lifetime ([plane][co][k], k contiguous = the MFMA B operand);
//   * the input patch of a tile (37 rows x 37 pixels x Cin) is loaded ONCE from global memory, split into three bf16 planes
//     and kept in LDS as a strip; all 49 taps are served from it.  The strip of tile t+1 is in flight (registers) while tile t
//     computes: one load latency per tile, not per K step;
//   * K is laid out as kernel rows padded to whole groups of 8 elements (Cin = 3: 21 -> 24, Cin = 1: 7 -> 8): an MFMA lane's
//     8 consecutive k of one kernel row are 8 consecutive bf16 of one strip row, i.e. four aligned ds_read_b32 per plane --
//     no im2col image, no per-K-step staging pass.  The padding elements are forced to zero on the A side by a per-lane
//     AND mask (so a non-finite neighbour pixel outside the window cannot leak in through a zero weight);
//   * products: the same six bf16 x bf16 terms (i + j <= 2), fp32 accumulate; BatchNorm column statistics in
//     fp64 as in igemm_epilogue, accumulated over all tiles of the workgroup: one partial row per workgroup.
// Bound: HBM for the audio stem (537 MB of output at 13 GFLOP), MFMA / LDS for the visual one (K = 176 padded, 617 MB).
#include "split_common.h"
#ifndef WG_VALU
#define WG_VALU 6
#endif

namespace {

template <int CIN>
struct StemCfg {
  static_assert(CIN == 1 || CIN == 3, "stem: 1 (audio) or 3 (visual) input channels");
  static constexpr int ROWE = 7 * CIN;                 // real elements of one kernel row (kw, ci)
  static constexpr int G = (ROWE + 7) / 8;             // 8-element groups per kernel row
  static constexpr int NG = 7 * G;                     // groups that carry weights
  static constexpr int NCH = (NG + 1) / 2;             // MFMA k-chunks of 16 (two groups: lane half h picks one)
  static constexpr int KPAD = NCH * 16;
  static constexpr int KROW = KPAD + ((KPAD / 8) % 2 == 0 ? 8 : 0);   // weight row stride (bf16): KROW / 8 odd -> conflict-free b128
  static constexpr int SR = 37;                        // strip rows: 2 * 16 + 5
  static constexpr int SE = 37 * CIN;                  // strip elements per row
  static constexpr int NPQ = (SE + 1) / 2;             // bf16 pairs (dwords) per strip row that are filled
  static constexpr int PD = CIN == 1 ? 24 : 56;        // strip row pitch in dwords: 2 * PD = 16 (mod 32) -> conflict-free ds_read_b32
  static constexpr int SBUF = SR * PD + 8;             // dwords per plane and buffer (+ slack for the masked over-read)
  static constexpr int NLD = (SR * NPQ + 255) / 256;   // strip dword slots per thread
  static constexpr int WPC = CIN == 1 ? 2 : 1;         // resident workgroups per CU (LDS: 50 KB / 118 KB)
};

struct StemGeom {
  int N, H, W, OH, OW, tilesY, tilesX, ntiles;
  unsigned x_bytes, y_bytes;
};

// Branch-free "offset or out of range": the compiler turns `ok ? off : OOB_OFF` with short-circuit conditions into control flow
// (exec-masked branches with vmcnt(0) waits inside the MFMA loop: measured, it serialised the dy stream).  ok is 0 / 1.
__device__ __forceinline__ unsigned off_or_oob(int ok, unsigned off) { return off | ((unsigned)ok - 1u); }
__device__ __forceinline__ int in_range(int v, int n) { return (int)((unsigned)v < (unsigned)n); }

__device__ __forceinline__ void buf_store1(rsrc_t r, float v, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}

// WAVES = 4: every wave owns two 32-pixel M blocks (MI = 2); WAVES = 8: one (MI = 1), two waves per SIMD.
// RAGGED = false: OH and OW are multiples of 16 (every CREMA-D shape): no per-pixel range checks at all.
template <int CIN, bool STATS, bool RAGGED, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (CIN == 3 && WAVES == 4) ? 1 : 2) void stem_fwd_split_kernel(
    const float* __restrict__ X, const float* __restrict__ Wf, float* __restrict__ Y, double* __restrict__ part, const StemGeom g) {
  using C = StemCfg<CIN>;
  constexpr int G = C::G, NG = C::NG, NCH = C::NCH, KROW = C::KROW, PD = C::PD, SBUF = C::SBUF, NPQ = C::NPQ;
  constexpr int NT = 64 * WAVES, MI = 8 / WAVES;
  constexpr int NLD = (C::SR * NPQ + NT - 1) / NT;            // strip dword slots per thread
  static_assert(WAVES == 4 || WAVES == 8, "4 or 8 waves");
  __shared__ __attribute__((aligned(16))) unsigned short Wl[3 * 64 * KROW];
  __shared__ __attribute__((aligned(16))) unsigned S[2 * 3 * SBUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;

  // ---- weights: fp32 HWIO [kh][kw][ci][co] -> three bf16 planes [co][k], k = (kh * G + part) * 8 + e, zero padded
  for (int idx = tid; idx < C::KPAD * 64; idx += NT) {
    const int co = idx & 63, k = idx >> 6;
    const int gq = k >> 3, e = k & 7;
    const int kh = gq / G, pt = gq - kh * G, re = pt * 8 + e;
    const float v = (gq < NG && re < C::ROWE) ? Wf[(kh * C::ROWE + re) * 64 + co] : 0.f;
    unsigned hi, mid, lo;
    split_pair(v, 0.f, hi, mid, lo);
    Wl[co * KROW + k] = (unsigned short)hi;
    Wl[64 * KROW + co * KROW + k] = (unsigned short)mid;
    Wl[2 * 64 * KROW + co * KROW + k] = (unsigned short)lo;
  }
  for (int idx = tid; idx < 2 * 3 * SBUF; idx += NT) S[idx] = 0u;      // slack / unfilled pitch dwords: defined (they are masked anyway)

  // ---- strip staging: slot s = tid + NT u -> (row r, pair q); elements 2q, 2q + 1 of strip row r
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(Y, g.y_bytes);
  // (slot -> (row, pair) is recomputed per use: index registers held across the MFMA loop cost more than the few VALU)
  auto slot_row = [&](int u) { const int s = tid + NT * u; return s < C::SR * NPQ ? s / NPQ : -1; };
  auto slot_pair = [&](int u) { const int s = tid + NT * u; return s - (s / NPQ) * NPQ; };
  float pre0[NLD], pre1[NLD];
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    const int per = g.tilesY * g.tilesX;
    n = t / per;
    const int rem = t - n * per;
    const int ty = rem / g.tilesX;
    oy0 = ty * 16;
    ox0 = (rem - ty * g.tilesX) * 16;
  };
  auto stage_load = [&](int t) {
    int n, oy0, ox0;
    decode(t, n, oy0, ox0);
    const int iy0 = 2 * oy0 - 3, ex0 = (2 * ox0 - 3) * CIN, rowE = g.W * CIN;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int sr = slot_row(u), sp = slot_pair(u);
      const int iy = iy0 + sr, e0 = ex0 + 2 * sp;
      const int rok = (int)(sr >= 0) & in_range(iy, g.H);
      const unsigned base = ((unsigned)(n * g.H + iy) * (unsigned)rowE + (unsigned)e0) * 4u;
      const int ok0 = rok & in_range(e0, rowE) & (int)(2 * sp < C::SE);
      const int ok1 = rok & in_range(e0 + 1, rowE) & (int)(2 * sp + 1 < C::SE);
      pre0[u] = buf_load1(xr, off_or_oob(ok0, base), 0);
      pre1[u] = buf_load1(xr, off_or_oob(ok1, base + 4u), 0);
    }
  };
  constexpr int NPART = NLD < 3 ? NLD : 3, PER = (NLD + NPART - 1) / NPART;   // the strip stores of a tile are spread over MFMA blocks
  auto stage_store = [&](int buf, int part_) {                // part_ < 0: everything
    unsigned* Sb = S + buf * 3 * SBUF;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (part_ >= 0 && u / PER != part_) continue;
      unsigned hi, mid, lo;
      split_pair<true>(pre0[u], pre1[u], hi, mid, lo);
      const int sr = slot_row(u);
      const int d = sr >= 0 ? sr * PD + slot_pair(u) : C::SR * PD + 4;           // slots past the strip: the slack dwords (branch-free)
      Sb[d] = hi;
      Sb[SBUF + d] = mid;
      Sb[2 * SBUF + d] = lo;
    }
  };

  // ---- per-lane fragment addressing: wave w owns M-blocks w * MI ..: block b = output rows 2b, 2b+1 of the tile, 16 columns each
  int a_base[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int oyl = 2 * (MI * wave + mi) + (i >> 4), oxl = i & 15;
    a_base[mi] = 2 * oyl * PD + oxl * CIN;
  }
  const int b_base = i * KROW + 8 * h;       // bf16 index of this lane's B fragment at chunk 0, column block 0

  // BatchNorm column statistics: per tile and lane the 16 MI values of a column are summed in fp32 as deviations from the
  // lane's first value v0 (sum d, sum d^2 with d = v - v0: exact-ish whatever |mean| / std is, Sterbenz), then folded into
  // the fp64 running sums as  sum v = n v0 + sum d,  sum v^2 = n v0^2 + 2 v0 sum d + sum d^2  -- 3 fp32 VALU per value instead
  // of a conversion and two fp64 operations (which cost a quarter of the visual stem's issue slots).
  double csum[2] = {0.0, 0.0}, csq[2] = {0.0, 0.0};
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, v0[2] = {0.f, 0.f}, cnt[2] = {0.f, 0.f};
  constexpr unsigned OOB_ST = 0xFFFFFF00u;                 // out-of-range store offset that stays out of range after + 128
  // Epilogue of a tile (stores + statistics), in 4 MI pieces of 4 accumulator rows x 2 column blocks.  Branch-free: pixels
  // outside the output (ragged tiles) are dropped by the buffer range check and skipped in the statistics by a select.
  auto epi_piece = [&](int piece, const f32x16 (&pa)[MI][2], int pn, int poy0, int pox0) {
    const int mi = piece >> 2, eq = piece & 3;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const int e = 4 * eq + e4;
      const int row = e4 + 8 * eq + 4 * h;                  // = (e & 3) + 8 * (e >> 2) + 4 h
      const int oy = poy0 + 2 * (MI * wave + mi) + (row >> 4), ox = pox0 + (row & 15);
      int ok;
      if constexpr (RAGGED) ok = (int)(oy < g.OH) & (int)(ox < g.OW);
      else ok = (int)(poy0 < (1 << 27));                     // only the dummy first drain is "outside"
      const unsigned off = (((unsigned)((pn * g.OH + oy) * g.OW + ox) * 64u + (unsigned)i) * 4u) | (((unsigned)ok - 1u) & OOB_ST);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const float v = pa[mi][ni][e];
        buf_store1(yr, v, off + ni * 128u);
        if constexpr (STATS) {
          if (piece == 0 && e4 == 0) {                       // first value of this lane's column in the tile: the shift
            v0[ni] = RAGGED ? (ok ? v : 0.f) : v;             // (value selects compile to v_cndmask)
            if constexpr (RAGGED) cnt[ni] = 0.f;
          }
          const float d = v - v0[ni];
          if constexpr (RAGGED) {
            const float dz = ok ? d : 0.f;
            s1[ni] += dz;
            s2[ni] = fmaf(dz, dz, s2[ni]);
            cnt[ni] += ok ? 1.f : 0.f;
          } else {
            s1[ni] += d;
            s2[ni] = fmaf(d, d, s2[ni]);
          }
        }
      }
    }
  };
  auto fold_stats = [&](bool valid) {                        // fp32 tile sums -> fp64 running sums (once per tile)
    if constexpr (STATS) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const double n_ = RAGGED ? (double)cnt[ni] : (valid ? 16.0 * MI : 0.0);
        const double a = (double)v0[ni], b = (double)s1[ni], c = (double)s2[ni];
        // RAGGED: v0 may belong to an out-of-range pixel (then it was replaced by 0 and the deviations are the values)
        csum[ni] += n_ * a + b;
        csq[ni] += n_ * a * a + 2.0 * a * b + c;
        s1[ni] = 0.f;
        s2[ni] = 0.f;
      }
    }
  };

  int cur = 0;
  int t = blockIdx.x;
  if (t < g.ntiles) {
    stage_load(t);
    __syncthreads();          // the zero fill above is complete before any strip store
    stage_store(0, -1);
  }
  __syncthreads();
  // The epilogue of tile t runs INSIDE the MFMA blocks of tile t + 1 (a separate epilogue phase would leave the matrix pipe
  // idle for a third of the tile time), so does the LDS store of the prefetched strip; `prev` holds the finished accumulators
  // meanwhile.  The first pass drains an all-zero `prev` whose pixels are all out of range.
  f32x16 prev[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) prev[mi][ni][e] = 0.f;
  int pn = 0, poy0 = 1 << 28, pox0 = 0;
  constexpr int NPC = 4 * MI;                                                   // epilogue pieces per tile
  constexpr int NEB = NCH >= NPC + NPART ? NCH - NPART : (NCH < NPC ? NCH : NPC);   // MFMA blocks that carry epilogue pieces
  constexpr int EPB = (NPC + NEB - 1) / NEB;                                   // pieces per block
  for (; t < g.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool has_next = tn < g.ntiles;
    if (has_next) stage_load(tn);                      // in flight while this tile computes
    f32x16 acc[MI][2];
    const unsigned* Sc = S + cur * 3 * SBUF;
    // Fragments of chunk kk + 1 are read (into the other register set) before the MFMAs of chunk kk issue.  The blocks are
    // fenced so that the reads stay a whole MFMA block ahead.
    struct Frags { u32x4 a[3][MI]; bf16x8_t b[3][2]; };
    auto load_frags = [&](int kk, Frags& f) {
      const int g0 = 2 * kk, g1 = 2 * kk + 1;                                       // groups of lane halves h = 0 / 1
      const int off0 = g0 < NG ? (g0 / G) * PD + (g0 % G) * 4 : 0, off1 = g1 < NG ? (g1 / G) * PD + (g1 % G) * 4 : 0;
      const int loff = h ? off1 : off0;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const unsigned* p = Sc + pl * SBUF + a_base[mi] + loff;
          f.a[pl][mi] = u32x4{p[0], p[1], p[2], p[3]};
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          f.b[pl][ni] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Wl + pl * 64 * KROW + ni * 32 * KROW + b_base + kk * 16));
      }
    };
    auto mask_frags = [&](int kk, Frags& f) {                                         // zero the padding elements of the group (see header)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned m[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int gq = 2 * kk + hh;
          const int pt = gq % G;
          const int nreal = gq < NG ? (C::ROWE - pt * 8 < 8 ? C::ROWE - pt * 8 : 8) : 0;
          m[hh] = nreal >= 2 * d + 2 ? 0xFFFFFFFFu : (nreal == 2 * d + 1 ? 0x0000FFFFu : 0u);
        }
        if (m[0] == 0xFFFFFFFFu && m[1] == 0xFFFFFFFFu) continue;
        const unsigned lm = h ? m[1] : m[0];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) f.a[pl][mi][d] &= lm;
      }
    };
    auto mma_frags = [&](const Frags& f, bool first) {
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            f32x16 c = acc[mi][ni];
            if (first && term == 0) {                  // the tile's first product starts from zero: no accumulator clear
#pragma unroll
              for (int e = 0; e < 16; ++e) c[e] = 0.f;
            }
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.a[TERM_A[term]][mi]), f.b[TERM_B[term]][ni], c, 0, 0, 0);
          }
    };
    // (audio with two 4-wave workgroups per CU: one fragment set -- the co-resident workgroup covers the LDS latency and the
    // second set would spill at 256 registers)
    constexpr bool DBUF = !(CIN == 1 && WAVES == 4);
    Frags fr0, fr1;
    if constexpr (DBUF) load_frags(0, fr0);
#pragma unroll
    for (int kk = 0; kk < NCH; ++kk) {
      Frags& fc = (DBUF && (kk & 1)) ? fr1 : fr0;
      Frags& fn = (kk & 1) ? fr0 : fr1;
      if constexpr (DBUF) {
        if (kk + 1 < NCH) load_frags(kk + 1, fn);
      } else {
        load_frags(kk, fr0);
      }
      mask_frags(kk, fc);
      mma_frags(fc, kk == 0);
      // previous tile's epilogue pieces and this tile's strip stores, spread over the blocks
      if (kk < NEB) {
#pragma unroll
        for (int q = 0; q < EPB; ++q)
          if (kk * EPB + q < NPC) epi_piece(kk * EPB + q, prev, pn, poy0, pox0);
      }
      // strip stores: unconditional (without a next tile the registers hold stale values and the buffer is never read again):
      // a branch here would end the scheduling region and push this work behind the MFMAs
      if (kk >= NCH - NPART) stage_store(cur ^ 1, kk - (NCH - NPART));   // that buffer's readers finished before the previous barrier
      __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);      // this block: the LDS reads of chunk kk + 1 first ...
      __builtin_amdgcn_sched_group_barrier(0x002, 12 * MI, 0); // ... the masks of chunk kk ...
#pragma unroll
      for (int m = 0; m < 12 * MI; ++m) {                      // ... then its MFMAs, the other work in their shadows
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    fold_stats(poy0 < (1 << 27));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) prev[mi][ni] = acc[mi][ni];
    decode(t, pn, poy0, pox0);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int piece = 0; piece < NPC; ++piece) epi_piece(piece, prev, pn, poy0, pox0);     // the last tile's epilogue
  fold_stats(poy0 < (1 << 27));
  if constexpr (STATS) {                               // one partial row per workgroup: [gridDim.x][2][64] doubles
    double* red = reinterpret_cast<double*>(S);        // WAVES x 2 x 64 doubles <= 8 KB of the (now idle) strip buffers
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      csum[ni] += __shfl_xor(csum[ni], 32, 64);
      csq[ni] += __shfl_xor(csq[ni], 32, 64);
      if (h == 0) {
        red[(wave * 2 + 0) * 64 + ni * 32 + i] = csum[ni];
        red[(wave * 2 + 1) * 64 + ni * 32 + i] = csq[ni];
      }
    }
    __syncthreads();
    if (tid < 64) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        s += red[(w * 2 + 0) * 64 + tid];
        q += red[(w * 2 + 1) * 64 + tid];
      }
      part[((size_t)blockIdx.x * 2 + 0) * 64 + tid] = s;
      part[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem weight gradient dW[k][co] = sum over pixels  x[pixel, tap k] * dy[pixel][co],  k = (kh * 7 + kw) * Cin + ci (HWIO order),
// on the same arithmetic and the same persistent patch-loader structure, transposed: the contraction runs over PIXELS, so
//   * MFMA rows = k (KB = ceil(49 Cin / 32) blocks of 32), columns = co (2 blocks), 16 pixels per instruction; a wave keeps
//     the whole KB x 2 x (32 x 32) accumulator in registers for ALL tiles of its workgroup and never writes a partial
//     until the kernel ends: one slab per workgroup, then the ordered reduce of conv_igemm.hip (no atomics, reproducible);
//   * an MFMA lane needs 8 consecutive PIXELS of one tap: output pixels ox .. ox+7 of a row read input columns
//     2 ox - 3 + kw (stride 2 Cin floats).  The strip is therefore split ONCE while it is staged (each element would
//     otherwise be split by ~12 tap lanes: measured, the kernel was VALU-issue-bound) and kept DE-INTERLEAVED in LDS as bf16
//     planes [copy][plane][column parity][ci][row][x / 2]: the 8 pixels are 8 consecutive bf16 = 4 dwords.  Their start
//     x/2 = 8h + (kw >> 1) is odd for half of the taps, so two copies are kept, packing element pairs (2d, 2d+1) and
//     (2d+1, 2d+2): every lane reads whole aligned dwords (ds_read_b32 x 4 per plane) and needs no VALU on the A side;
//   * dy is streamed straight from global memory in MFMA B layout (lane = channel: 128-B row pieces per instruction), one
//     step ahead of its use, and split in registers: it is read exactly once (617 MB / 537 MB per launch: the HBM floor).
// A "step" = 16 pixels of one output row; a 16 x 16 tile = 16 steps, 4 per wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int CIN>
struct StemWCfg {
  static constexpr int K = 49 * CIN;
  static constexpr int KB = (K + 31) / 32;
  static constexpr int SR = 37;
  static constexpr int XQ = 10;                          // dwords (element pairs) per strip row of one (parity, ci) array
  static constexpr int XPD = CIN == 1 ? 13 : 10;         // row pitch in dwords (bank spread of the tap lanes, scripts in DESIGN)
  static constexpr int ARR = SR * XPD;                   // one (parity, ci) array
  static constexpr int HALF = CIN * ARR + (CIN == 1 ? 9 : 0);      // parity stride
  static constexpr int ZROW = 2 * HALF;                  // 16 zero dwords per plane for the lanes whose k >= K
  static constexpr int PLANE = ZROW + 16;
  static constexpr int COPY = 3 * PLANE + (CIN == 1 ? 20 : 0);
  static constexpr int SBUF = 2 * COPY;                  // dwords per strip buffer
  static constexpr int NSL = 2 * CIN * SR * XQ;          // staging slots: (array, row, pair)
  static constexpr int RACC = KB * 32 * 64;
  static constexpr int LDSF = 2 * SBUF > RACC ? 2 * SBUF : RACC;
};

// 8 waves: wave w = (column half ni = w >> 2, row group wq = w & 3).  Two waves per SIMD hide each other's latencies, the
// 80-register accumulator (KB x 1 x 16) leaves room for a dy prefetch two steps deep (one step deep with one wave per SIMD
// the kernel was latency-bound on the dy stream: 16 KB in flight per CU), and both column halves share one x strip.
template <int CIN>
__global__ __launch_bounds__(512, 2) void stem_wgrad_split_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                   float* __restrict__ slabs, const StemGeom g, unsigned dy_bytes) {
  using C = StemWCfg<CIN>;
  constexpr int KB = C::KB, XPD = C::XPD, ARR = C::ARR, HALF = C::HALF, PLANE = C::PLANE, COPY = C::COPY, SBUF = C::SBUF;
  constexpr int NT = 512, NLD = (C::NSL + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) unsigned S[C::LDSF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int ni = wave >> 2, wq = wave & 3;
  for (int idx = tid; idx < C::LDSF; idx += NT) S[idx] = 0u;

  // ---- strip staging: slot -> (array a = parity * CIN + ci, row r, pair q); copy 0 packs strip columns (2q, 2q+1) of that
  // array, copy 1 packs (2q+1, 2q+2); array column c = strip column 2c + parity.  (Indices are recomputed per use: holding
  // them in registers across the MFMA loop costs more than the ~15 VALU per slot and tile.)
  const rsrc_t xr = make_rsrc(X, g.x_bytes), yr = make_rsrc(dY, dy_bytes);
  float pre0[NLD], pre1[NLD], pre2[NLD];              // (three plain arrays: a [NLD][3] array was left in scratch memory by SROA)
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    const int per = g.tilesY * g.tilesX;
    n = t / per;
    const int rem = t - n * per;
    const int ty = rem / g.tilesX;
    oy0 = ty * 16;
    ox0 = (rem - ty * g.tilesX) * 16;
  };
  auto stage_load = [&](int t) {
    int n, oy0, ox0;
    decode(t, n, oy0, ox0);
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int sl = tid + NT * u;
      const int a = sl / (C::SR * C::XQ), rq = sl - a * (C::SR * C::XQ);
      const int r = rq / C::XQ, q = rq - r * C::XQ;
      const int par = a / CIN, ci = a - par * CIN;
      const int iy = iy0 + r;
      const int rok = (int)(sl < C::NSL) & in_range(iy, g.H);
      auto ld = [&](int j) {
        const int sc = 2 * (2 * q + j) + par;                    // strip column 0..36 (beyond: zero)
        const int ix = ix0 + sc;
        const int ok = rok & (int)(sc <= 36) & in_range(ix, g.W);
        return buf_load1(xr, off_or_oob(ok, (((unsigned)(n * g.H + iy) * (unsigned)g.W + (unsigned)ix) * CIN + ci) * 4u), 0);
      };
      pre0[u] = ld(0);
      pre1[u] = ld(1);
      pre2[u] = ld(2);
    }
  };
  auto stage_store = [&](int buf, int u) {            // slot u of this thread
    unsigned* Sb = S + buf * SBUF;
    const int sl = tid + NT * u;
    const int a = sl / (C::SR * C::XQ), rq = sl - a * (C::SR * C::XQ);
    const int r = rq / C::XQ, q = rq - r * C::XQ;
    const int par = a / CIN, ci = a - par * CIN;
    const int d = sl < C::NSL ? par * HALF + ci * ARR + r * XPD + q : C::ZROW + 8;     // surplus slots: a scratch dword (branch-free)
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair<true>(pre0[u], pre1[u], h0, m0, l0);
    split_pair<true>(pre1[u], pre2[u], h1, m1, l1);
    Sb[d] = h0; Sb[PLANE + d] = m0; Sb[2 * PLANE + d] = l0;
    Sb[COPY + d] = h1; Sb[COPY + PLANE + d] = m1; Sb[COPY + 2 * PLANE + d] = l1;
  };

  // per-lane A addressing: k = kb * 32 + i -> tap (kh, kw, ci); lanes past K read the zero row (dwords, plane 0 of copy p)
  int a_base[KB], a_rstep[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int k = kb * 32 + i;
    if (k < C::K) {
      const int kh = k / (7 * CIN), rem = k - kh * 7 * CIN;
      const int kw = rem / CIN, ci = rem - kw * CIN;
      const int sft = kw >> 1;                                   // first array column of output pixel 0: x / 2 = sft
      a_base[kb] = (sft & 1) * COPY + (kw & 1) * HALF + ci * ARR + (kh + 8 * wq) * XPD + 4 * h + (sft >> 1);
      a_rstep[kb] = 2 * XPD;
    } else {
      a_base[kb] = C::ZROW + 4 * h;
      a_rstep[kb] = 0;
    }
  }

  f32x16 acc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[kb][e] = 0.f;

  // ---- software pipeline.  The work of a tile is a flat sequence of U = 4 * KB units (step s, k block kb), 6 MFMAs each.
  // In the block of unit u the wave issues the LDS reads of unit u + 1's A fragments (other register set), splits -- in the
  // first two blocks of a step -- half of the NEXT step's dy into bf16 planes and, in the last blocks of the tile, a slot of
  // the next tile's prefetched strip, interleaved with the MFMAs of unit u, and after the second block of step s issues the
  // global loads of dy for step s + 3 (two steps in flight).  Steps run on across tile boundaries; B-fragment and dy register
  // sets alternate with the step parity (4 steps per tile keeps the parity fixed).
  constexpr int U = 4 * KB;
  float dyv[2][8];                                    // dy in flight: slot (s & 1) holds step s; [pixel e] of this lane's channel
  auto load_dy = [&](int t, int s, float (&dst)[8]) { // step s of tile t: output row oy0 + 4 * wq + s, pixels ox0 + 8h + e
    int n, oy0, ox0;
    decode(t, n, oy0, ox0);
    const int oy = oy0 + 4 * wq + s;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ox = ox0 + 8 * h + e;
      const int ok = (int)(t < g.ntiles) & (int)(oy < g.OH) & (int)(ox < g.OW);
      dst[e] = buf_load1(yr, off_or_oob(ok, ((unsigned)((n * g.OH + oy) * g.OW + ox) * 64u + (unsigned)(ni * 32 + i)) * 4u), 0);
    }
  };
  auto load_dy_step = [&](int t, int tn, int s, float (&dst)[8]) {   // step index s may run on into the next tile(s)
    if (s < 4) load_dy(t, s, dst);
    else load_dy(tn, s - 4, dst);                     // (lanes masked off when there is no next tile)
  };
  struct BFr { u32x4 p[3]; };                         // dy planes of one step
  auto split_dy_half = [&](int j, const float (&src)[8], BFr& b) {   // half j = 0, 1: pixel pairs 2j, 2j + 1
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = 2 * j + qq;
      unsigned hi, mid, lo;
      split_pair<true>(src[2 * q], src[2 * q + 1], hi, mid, lo);
      b.p[0][q] = hi; b.p[1][q] = mid; b.p[2][q] = lo;
    }
  };
  struct AFr { u32x4 p[3]; };
  auto read_a = [&](const unsigned* Sc, int u, AFr& a) {
    const int s_ = u / KB, kb = u - s_ * KB;          // compile-time after unrolling
    const unsigned* ap = Sc + a_base[kb] + s_ * a_rstep[kb];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) a.p[pl] = u32x4{ap[pl * PLANE], ap[pl * PLANE + 1], ap[pl * PLANE + 2], ap[pl * PLANE + 3]};
  };

  int cur = 0;
  int t = blockIdx.x;
  BFr bf0, bf1;
  if (t < g.ntiles) stage_load(t);
  __syncthreads();                                    // zero fill complete
  if (t < g.ntiles) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) stage_store(0, u);
    load_dy(t, 0, dyv[0]);
    split_dy_half(0, dyv[0], bf0);                    // step 0's planes (latency exposed once per kernel)
    split_dy_half(1, dyv[0], bf0);
    load_dy(t, 1, dyv[1]);
    load_dy(t, 2, dyv[0]);
  }
  __syncthreads();
  constexpr int ST0 = U - NLD < 0 ? 0 : U - NLD;      // the last NLD blocks of a tile each store one strip slot
  for (; t < g.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool has_next = tn < g.ntiles;
    if (has_next) stage_load(tn);
    const unsigned* Sc = S + cur * SBUF;
    AFr a0, a1;
    read_a(Sc, 0, a0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s_ = u / KB, kb = u - s_ * KB;
      AFr& acur = (u & 1) ? a1 : a0;
      AFr& anext = (u & 1) ? a0 : a1;
      BFr& bcur = (s_ & 1) ? bf1 : bf0;
      BFr& bnext = (s_ & 1) ? bf0 : bf1;
      if (u + 1 < U) read_a(Sc, u + 1, anext);
      // the next step's dy (slot (s + 1) & 1, landed two steps ago) -> planes, half per block; then that slot is free for
      // step s + 3
      if (kb < 2) split_dy_half(kb, dyv[(s_ + 1) & 1], bnext);
      // the next tile's strip (loads issued at the top of this tile): one slot per block; unconditional -- without a next tile
      // the registers are stale and the buffer is never read (a branch would end the scheduling region)
      if (u >= ST0) stage_store(cur ^ 1, u - ST0);
#pragma unroll
      for (int term = 0; term < 6; ++term)
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, acur.p[TERM_A[term]]),
                                                          __builtin_bit_cast(bf16x8_t, bcur.p[TERM_B[term]]), acc[kb], 0, 0, 0);
      if (kb == 1) load_dy_step(t, tn, s_ + 3, dyv[(s_ + 1) & 1]);
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);          // LDS reads of unit u + 1 first
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, WG_VALU, 0);   // ... then a few VALU of the splits
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- sum the four row-group waves of each column half in order through LDS, write this workgroup's slab [K][64]
  float* R = reinterpret_cast<float*>(S);
  for (int w = 0; w < 4; ++w) {
    if (wq == w) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int o = row * 64 + ni * 32 + i;
          const float v = w == 0 ? acc[kb][e] : R[o] + acc[kb][e];
          if (w == 3) {
            if (row < C::K) slabs[(size_t)blockIdx.x * C::K * 64 + o] = v;
          } else {
            R[o] = v;
          }
        }
      }
    }
    __syncthreads();
  }
}

int stem_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

int stem_check(const char* who, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  MLA_REQUIRE(KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64 && (Cin == 1 || Cin == 3),
              "%s: the stem kernels are 7x7 / stride 2 / pad 3, 1 or 3 -> 64 channels (got %dx%d s%d p%d, %d -> %d)", who, KH, KW, stride, pad, Cin, Cout);
  MLA_REQUIRE(N > 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "%s: bad dims", who);
  const long oh = (H + 6 - 7) / 2 + 1, ow = (W + 6 - 7) / 2 + 1;
  MLA_REQUIRE(oh > 0 && ow > 0 && (long)N * H * W * Cin * 4 < 0xFFFFFFF0L && (long)N * oh * ow * 64 * 4 < 0xFFFFFFF0L,
              "%s: input and output tensors must each be < 4 GiB (32-bit buffer offsets)", who);
  return MLA_OK;
}

}  // namespace

static int g_stem_waves = 0;                           // 0 = automatic: 4 (two workgroups per CU) for Cin = 1, 8 for Cin = 3 (measured)
extern "C" int mla_conv2d_stem_waves(int waves) {       // measurement hook: force 4 or 8 waves per stem-forward workgroup, 0 = automatic
  if (waves == 4 || waves == 8 || waves == 0) g_stem_waves = waves;
  return g_stem_waves;
}

extern "C" int mla_conv2d_stem_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
  return KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64 && (Cin == 1 || Cin == 3);
}

extern "C" size_t mla_conv2d_stem_fwd_partial_elems(void) { return (size_t)stem_cus() * 2 * 2 * 64 * 2; }   // floats: [workgroups][2][64] doubles

extern "C" int mla_conv2d_stem_fwd_split(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                         int KW, int stride, int pad, float* bn_partial, int* bn_tiles, void* stream) {
  if (int rc = stem_check("mla_conv2d_stem_fwd_split", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && w && y, "mla_conv2d_stem_fwd_split: null pointer");
  MLA_REQUIRE(!bn_partial || ((uintptr_t)bn_partial % 8) == 0, "mla_conv2d_stem_fwd_split: bn_partial must be 8-byte aligned");
  StemGeom g;
  g.N = N; g.H = H; g.W = W;
  g.OH = conv_out(H, 7, 2, 3); g.OW = conv_out(W, 7, 2, 3);
  g.tilesY = cdiv(g.OH, 16); g.tilesX = cdiv(g.OW, 16);
  g.ntiles = N * g.tilesY * g.tilesX;
  g.x_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
  g.y_bytes = (unsigned)((size_t)N * g.OH * g.OW * 64 * 4);
  hipStream_t st = (hipStream_t)stream;
  const int waves = g_stem_waves ? g_stem_waves : (Cin == 1 ? 4 : 8);
  const int slots = stem_cus() * ((Cin == 1 && waves == 4) ? 2 : 1);      // resident workgroups per CU (registers / LDS)
  const int grid = g.ntiles < slots ? g.ntiles : slots;
  double* pd = reinterpret_cast<double*>(bn_partial);
  const bool ragged = (g.OH % 16) != 0 || (g.OW % 16) != 0;
#define STEM_LAUNCH(CIN_, ST_, RG_, WV_) stem_fwd_split_kernel<CIN_, ST_, RG_, WV_><<<grid, 64 * WV_, 0, st>>>(x, w, y, pd, g)
#define STEM_PICK(CIN_, WV_)                                                            \
  do {                                                                                  \
    if (pd && ragged) STEM_LAUNCH(CIN_, true, true, WV_);                               \
    else if (pd) STEM_LAUNCH(CIN_, true, false, WV_);                                   \
    else if (ragged) STEM_LAUNCH(CIN_, false, true, WV_);                               \
    else STEM_LAUNCH(CIN_, false, false, WV_);                                          \
  } while (0)
  if (Cin == 1 && waves == 8) STEM_PICK(1, 8);
  else if (Cin == 1) STEM_PICK(1, 4);
  else if (waves == 8) STEM_PICK(3, 8);
  else STEM_PICK(3, 4);
#undef STEM_PICK
#undef STEM_LAUNCH
  MLA_CHECK_LAUNCH("stem_fwd_split_kernel");
  if (bn_tiles) *bn_tiles = grid;
  return MLA_OK;
}

int mla_wgrad_reduce(const float* part, float* dw, size_t n4, int splits, hipStream_t st);   // conv_igemm.hip

extern "C" size_t mla_conv2d_stem_wgrad_split_ws_bytes(int Cin) { return (size_t)stem_cus() * 49 * Cin * 64 * sizeof(float); }

extern "C" int mla_conv2d_stem_wgrad_split(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                                           int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = stem_check("mla_conv2d_stem_wgrad_split", N, H, W, Cin, Cout, KH, KW, stride, pad)) return rc;
  MLA_REQUIRE(x && dy && dw && ws, "mla_conv2d_stem_wgrad_split: null pointer");
  StemGeom g;
  g.N = N; g.H = H; g.W = W;
  g.OH = conv_out(H, 7, 2, 3); g.OW = conv_out(W, 7, 2, 3);
  g.tilesY = cdiv(g.OH, 16); g.tilesX = cdiv(g.OW, 16);
  g.ntiles = N * g.tilesY * g.tilesX;
  g.x_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
  const size_t dyb = (size_t)N * g.OH * g.OW * 64 * 4;
  MLA_REQUIRE(dyb < 0xFFFFFFF0UL, "mla_conv2d_stem_wgrad_split: dy must be < 4 GiB (32-bit buffer offsets)");
  const int grid = g.ntiles < stem_cus() ? g.ntiles : stem_cus();
  const size_t need = (size_t)grid * 49 * Cin * 64 * sizeof(float);
  if (ws_bytes < need) {
    mla_set_error("mla_conv2d_stem_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
    return MLA_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 1) stem_wgrad_split_kernel<1><<<grid, 512, 0, st>>>(x, dy, (float*)ws, g, (unsigned)dyb);
  else stem_wgrad_split_kernel<3><<<grid, 512, 0, st>>>(x, dy, (float*)ws, g, (unsigned)dyb);
  MLA_CHECK_LAUNCH("stem_wgrad_split_kernel");
  return mla_wgrad_reduce((const float*)ws, dw, (size_t)49 * Cin * 64 / 4, grid, st);
}
