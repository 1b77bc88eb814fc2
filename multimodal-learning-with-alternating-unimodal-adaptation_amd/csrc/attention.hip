// Fused multi-head attention for the transformer rows, forward and backward, exact fp32 on v_mfma_f32_32x32x2_f32.
//
//   reference: Attention.forward, models/m3ae.py:102-125 (and timm's Attention inside cav_mae.py:93):
//     attention = (q @ k^T) * scale;  attention = where(padding_mask > 0, -1e7, attention)  (m3ae.py:109-117)
//     attention = softmax(attention, -1);  x = attention @ v  -> (B, n, H*hd)                  (m3ae.py:118-122)
//
// q, k, v are read in place from the (B, n, 3, H, 64) buffer the fused qkv Linear writes; the output lands in (B, n, H*64).
// The n x n score / probability matrices never reach HBM (the materialised form moved 2 x 203 MB per layer at B = 64 and
// ran its batched GEMMs at 43 TFLOP/s): forward keeps an online softmax, backward recomputes the probabilities from the
// saved log-sum-exp.  Everything is deterministic (no atomics): the backward is two kernels, one that owns query rows
// (dQ) and one that owns key rows (dK, dV).
//
// Wave-level formulation (64-wide wavefronts, 32x32x2 MFMA): every wave owns 32 rows of its output operand and walks the
// other sequence dimension in tiles of 32 through LDS.  Scores are produced TRANSPOSED to the operand the wave owns
// (forward / dQ: S^T = K Q^T, so a lane holds 16 keys of ONE query), which makes the softmax statistics per-lane scalars
// (one cross-half shuffle per tile instead of 32-lane reductions) and -- because the k index of an MFMA contraction may
// be enumerated in any order as long as A and B agree -- lets the probability accumulator registers be fed straight back
// as the B operand of the next product (P^T as [key][query]) without a round trip through LDS.
//
// Ragged lengths: a sequence of n = 32 m + r tokens with r <= 3 (M3AE: 257 = [cls] + 256) is split into m full tiles for the
// MFMA kernels and r remainder rows that are handled by vector code -- as extra keys / queries at the end of each MFMA wave's
// loop (a dot product and a rank-1 update per row) and, as OWNED rows, by wave 0 of workgroup 0 of every (b, h), which runs plain
// vector code on the tiles it has in LDS anyway, next to its own MFMA rows; a ninth 32-wide tile and a ninth wave block for one row made
// n = 257 cost 1.6x of n = 256.
//
// Masking: keys beyond n do not exist (-inf, probability exactly 0); padded keys (mask > 0) have their score REPLACED by
// -1e7 like the reference, which underflows to probability exactly 0 in fp32 unless a whole row is padded (never: the
// [cls] key is always present, m3ae.py:347).
#include "common.h"

#define ATT_HD 64          // head dim (ViT-B: 768 / 12)
#define ATT_LD 68          // padded LDS row (floats): 17 x 16 B -> conflict-free ds_read_b128 across 32 rows
#define ATT_TAIL_MAX 3     // n % 32 <= this: the remainder rows take the vector path
#define ATT_TAIL_N 4096    // longest sequence the one-wave tail kernels hold in LDS

namespace {

// exp via v_exp_f32 (2^x): the accurate expf expands to ~20 VALU instructions, 16 of them per lane and key tile were a
// quarter of the forward's issue slots.  Relative error <= |x| 2^-23 (the rounding of x * log2 e), i.e. < 3e-6 for the
// score range a softmax row can hold before the result underflows anyway; arguments of -inf / -1e7 give exactly 0.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ int acc_row(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

struct AttGeom {
  const float* qkv;        // (B, n, 3, H, 64)
  const float* pm;         // (B, n) or null
  int B, H, n;
  int nm;                  // rows [0, nm) are handled by the MFMA kernels (nm % 32 == 0 or nm == n); rows [nm, n) -- at most
                           // ATT_TAIL_MAX of them -- by vector code: the [cls] token makes n = 257 = 8 tiles + 1, and a ninth
                           // 32-wide tile / a ninth wave block for that one row cost 60 % more time than n = 256
  float scale;
};

// byte-free helpers: element offset of (b, t, which, h, 0)
__device__ __forceinline__ size_t qkv_off(const AttGeom& g, int b, int t, int which, int h) {
  return (((size_t)b * g.n + t) * 3 + which) * (size_t)(g.H * ATT_HD) + (size_t)h * ATT_HD;
}

// One 32 x 64 tile (rows t0.., zero-filled beyond n) from global into registers / from registers into padded LDS.
// NT threads: float4 slot idx = tid + NT p of the 512 (row = idx / 16, float4 column idx % 16).
template <int NT>
struct TileRegs {
  static constexpr int P = (512 + NT - 1) / NT;
  f32x4 r[P];
  __device__ __forceinline__ void load(const float* base, size_t row_stride, int t0, int n, int tid) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int idx = tid + NT * p, row = idx >> 4;
      r[p] = (idx < 512 && t0 + row < n) ? *reinterpret_cast<const f32x4*>(base + (size_t)(t0 + row) * row_stride + (idx & 15) * 4)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int idx = tid + NT * p;
      if (idx < 512) *reinterpret_cast<f32x4*>(&lds[(idx >> 4) * ATT_LD + (idx & 15) * 4]) = r[p];
    }
  }
};

// 32 registers of the B operand of a wave-owned row block: lane (row = lane % 32, half) holds X[row][half * 32 + kk]
__device__ __forceinline__ void own_rows_load(float (&x)[32], const float* base, size_t row_stride, int row, int n, int half) {
  if (row < n) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base + (size_t)row * row_stride + half * 32);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 v = p[c];
      x[4 * c + 0] = v[0]; x[4 * c + 1] = v[1]; x[4 * c + 2] = v[2]; x[4 * c + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = 0.f;
  }
}

// acc(32 x 32) = T(32 rows from LDS, b128 fragments) x own^T : acc[e] <-> (tile row acc_row(e, half), own row lane % 32)
__device__ __forceinline__ void mma_tile_own(f32x16& acc, const float* tile, const float (&own)[32], int lane) {
  const float* rowp = tile + (lane & 31) * ATT_LD + (lane >> 5) * 32;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], own[4 * c + j], acc, 0, 0, 0);
  }
}

// out^T(64 x 32, two 32-row blocks) += tile^T (64 x 32 tile rows) x w(32 tile rows x 32 own rows, accumulator layout)
__device__ __forceinline__ void mma_tileT_acc(f32x16& o0, f32x16& o1, const float* tile, const float (&w)[16], int lane) {
  const int half = lane >> 5, c = lane & 31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float* rowp = tile + acc_row(e, half) * ATT_LD;
    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[c], w[e], o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[32 + c], w[e], o1, 0, 0, 0);
  }
}

// Write a wave's transposed 64 x 32 result (o0 / o1: row = feature, column = own row) to global rows of 64 floats.
// stage: this wave's 32 x ATT_LD floats of LDS.
__device__ __forceinline__ void store_ownT(float* stage, const f32x16& o0, const f32x16& o1, float* dst, size_t row_stride,
                                            int row0, int n, int lane, float mul) {
  const int half = lane >> 5, c = lane & 31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    stage[c * ATT_LD + acc_row(e, half)] = o0[e] * mul;
    stage[c * ATT_LD + 32 + acc_row(e, half)] = o1[e] * mul;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = p * 64 + lane, r = idx >> 4, c4 = idx & 15;
    if (row0 + r < n)
      *reinterpret_cast<f32x4*>(dst + (size_t)(row0 + r) * row_stride + c4 * 4) = *reinterpret_cast<const f32x4*>(&stage[r * ATT_LD + c4 * 4]);
  }
}

// key state of a tile: 0 = attend, 1 = padded (score := -1e7), 2 = beyond n
__device__ __forceinline__ float key_state(const AttGeom& g, int b, int key, int limit) {
  if (key >= limit) return 2.f;
  return (g.pm && g.pm[(size_t)b * g.n + key] > 0.f) ? 1.f : 0.f;
}

// dot product of a wave-owned row (lane: row lane % 32, dims half * 32 ..) with one broadcast row of 64 floats
__device__ __forceinline__ float own_dot(const float (&own)[32], const float* __restrict__ row, int half) {
  const f32x4* p = reinterpret_cast<const f32x4*>(row + half * 32);
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = p[c];
    d += own[4 * c] * v[0] + own[4 * c + 1] * v[1] + own[4 * c + 2] * v[2] + own[4 * c + 3] * v[3];
  }
  return d + __shfl_xor(d, 32, 64);
}
// out^T (+)= row (x) w : rank-1 update of a transposed 64 x 32 accumulator pair with one broadcast row and a per-lane weight
__device__ __forceinline__ void own_axpy(f32x16& o0, f32x16& o1, const float* __restrict__ row, float w, int half) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    o0[e] += row[acc_row(e, half)] * w;
    o1[e] += row[32 + acc_row(e, half)] * w;
  }
}

// ---- vector code for OWNED remainder rows, run by one wave per (b, h) on the tiles in LDS: a dot product per tile row (lane = row,
// half = dim half), softmax statistics by wave reductions, weighted row sums with the weights broadcast by v_readlane.
__device__ __forceinline__ float vec_row_dot(const float* __restrict__ tile, const float* __restrict__ vec, int lane) {
  const f32x4* rowp = reinterpret_cast<const f32x4*>(tile + (lane & 31) * ATT_LD + (lane >> 5) * 32);
  const f32x4* vp = reinterpret_cast<const f32x4*>(vec + (lane >> 5) * 32);
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 a = rowp[c], v = vp[c];
    d += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
  }
  return d + __shfl_xor(d, 32, 64);
}
// sum_{r < 32} w_r * tile[r][lane], w_r = the value lane r holds
__device__ __forceinline__ float vec_wsum(const float* __restrict__ tile, float w, int lane) {
  float a = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r)
    a += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), r)) * tile[r * ATT_LD + lane];
  return a;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one (b, h); loop over key tiles of 32.  o (B, n, H*64), lse (B, H, n).
// ---------------------------------------------------------------------------------------------------------------------
template <int W, int TAIL>
// (with remainder rows the kernel needs ~190 VGPRs: at three 4-wave workgroups per CU it spilled 11 registers whose reloads wait `vmcnt(0)`
//  inside the key loop; two workgroups per CU and no scratch is faster -- n = 257: 170 -> see DESIGN 8)
__global__ __launch_bounds__(64 * W, (W >= 3 && !TAIL) ? 3 : 2) void attn_fwd_kernel(const AttGeom g, float* __restrict__ O, float* __restrict__ LSE) {
  constexpr int TMAX = TAIL > 0 ? TAIL : 1;      // remainder rows this instantiation can own (TAIL = 1: the [cls] + 256 case, 3: any)
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 32];
  float* Ks = smem;                       // [2][32][ATT_LD]
  float* Vs = smem + 2 * 32 * ATT_LD;     // [2][32][ATT_LD]
  float* Kst = smem + 4 * 32 * ATT_LD;    // [2][32] key states
  __shared__ __attribute__((aligned(16))) float tailK[ATT_TAIL_MAX][ATT_HD], tailV[ATT_TAIL_MAX][ATT_HD];   // the remainder keys [nm, n)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q = blockIdx.x * (32 * W) + wave * 32 + (lane & 31);
  // With remainder rows (TAIL) workgroup 0 of every (b, h) owns them IN ADDITION to its MFMA rows (see vec_row_dot): tile jt is
  // handled by wave jt % W, each wave keeps a partial (max, sum, weighted row) state, wave 0 merges them at the end -- ~7 % more
  // work per wave of that workgroup.  Measured alternatives at n = 257 (forward, us; n = 256 takes 124): everything on wave 0
  // (155), a fifth wave per workgroup (192), one more workgroup per (b, h) that only stages tiles and runs the vector code (173),
  // stand-alone vector kernels reading global memory (164, and 2 x 40 in the backward), a ninth padded tile / wave block (202).
  const bool vec = TAIL && blockIdx.x == 0;
  constexpr bool vecwg = false;
  constexpr bool stager = true;
  const bool wave_active = blockIdx.x * (32 * W) + wave * 32 < g.nm;
  __shared__ __attribute__((aligned(16))) float vecQ[ATT_TAIL_MAX][ATT_HD];
  __shared__ float vred[W][ATT_TAIL_MAX][ATT_HD + 2], tailSt[ATT_TAIL_MAX];
  const int nt = g.n - g.nm;
  float vm[TMAX], vl[TMAX], vo[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { vm[t] = -INFINITY; vl[t] = 0.f; vo[t] = 0.f; }
  const size_t rs = (size_t)3 * g.H * ATT_HD;                  // token stride inside qkv
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);

  float qreg[32];
  own_rows_load(qreg, Qb, rs, vecwg ? g.nm : q, g.nm, half);   // (the vector workgroup loads nothing here: row nm is "beyond")
  if (vec && wave == 0)
    for (int t = 0; t < nt; ++t) vecQ[t][lane] = Qb[(size_t)(g.nm + t) * rs + lane];
  if (tid < nt) tailSt[tid] = key_state(g, b, g.nm + tid, g.n);
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
  float m_i = -INFINITY, l_i = 0.f;

  const int ntiles = (g.nm + 31) / 32;
  TileRegs<64 * W> kr, vr;
  if (stager) {
    kr.load(Kb, rs, 0, g.nm, tid);
    vr.load(Vb, rs, 0, g.nm, tid);
    kr.store(Ks, tid);
    vr.store(Vs, tid);
  }
  if (tid < 32) Kst[tid] = key_state(g, b, tid, g.nm);
  for (int i = tid; stager && i < (g.n - g.nm) * ATT_HD; i += 64 * W) {
    tailK[i >> 6][i & 63] = Kb[(size_t)(g.nm + (i >> 6)) * rs + (i & 63)];
    tailV[i >> 6][i & 63] = Vb[(size_t)(g.nm + (i >> 6)) * rs + (i & 63)];
  }
  __syncthreads();
  for (int jt = 0; jt < ntiles; ++jt) {
    const int cur = jt & 1, nxt = cur ^ 1;
    const bool more = jt + 1 < ntiles;
    float kst_next = 0.f;
    if (more && stager) {
      kr.load(Kb, rs, (jt + 1) * 32, g.nm, tid);
      vr.load(Vb, rs, (jt + 1) * 32, g.nm, tid);
      if (tid < 32) kst_next = key_state(g, b, (jt + 1) * 32 + tid, g.nm);    // pad-mask load issued here, consumed after the tile's work
    }
    if (vec && jt % W == wave) {
      const float* kt = Ks + cur * 32 * ATT_LD;
      const float* vt = Vs + cur * 32 * ATT_LD;
      const float st = Kst[cur * 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          float sv = vec_row_dot(kt, vecQ[t], lane);
          sv = st == 0.f ? sv * g.scale : (st == 1.f ? -1e7f : -INFINITY);
          const float m_new = fmaxf(vm[t], wave_max(sv));
          const float alpha = fast_exp(vm[t] - m_new), pv = fast_exp(sv - m_new);
          vl[t] = vl[t] * alpha + wave_sum(half == 0 ? pv : 0.f);
          vm[t] = m_new;
          vo[t] = vo[t] * alpha + vec_wsum(vt, pv, lane);
        }
    }
    if (wave_active) {
      const float* kt = Ks + cur * 32 * ATT_LD;
      const float* vt = Vs + cur * 32 * ATT_LD;
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      mma_tile_own(s, kt, qreg, lane);                          // S^T = K Q^T: s[e] <-> (key acc_row(e, half), query lane)
      float p[16];
      float mt = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float st = Kst[cur * 32 + acc_row(e, half)];
        p[e] = st == 0.f ? s[e] * g.scale : (st == 1.f ? -1e7f : -INFINITY);
        mt = fmaxf(mt, p[e]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_i, mt);
      const float alpha = fast_exp(m_i - m_new);                     // first tile: exp(-inf) = 0
      float lt = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        p[e] = fast_exp(p[e] - m_new);
        lt += p[e];
      }
      lt += __shfl_xor(lt, 32, 64);
      l_i = l_i * alpha + lt;
      m_i = m_new;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        o0[e] *= alpha;
        o1[e] *= alpha;
      }
      mma_tileT_acc(o0, o1, vt, p, lane);                       // O^T += V^T P^T
    }
    if (more && stager) {
      kr.store(Ks + nxt * 32 * ATT_LD, tid);
      vr.store(Vs + nxt * 32 * ATT_LD, tid);
      if (tid < 32) Kst[nxt * 32 + tid] = kst_next;
    }
    __syncthreads();
  }
  if (vec) {        // remainder queries: wave 0 adds the remainder keys to its partial, all partials are merged, wave 0 writes the rows
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < nt) {
        if (wave == 0)
          for (int j = g.nm; j < g.n; ++j) {
            const float dot = wave_sum(vecQ[t][lane] * tailK[j - g.nm][lane]);
            const float sj = tailSt[j - g.nm] == 0.f ? dot * g.scale : -1e7f;
            const float m_new = fmaxf(vm[t], sj);
            const float alpha = fast_exp(vm[t] - m_new), pj = fast_exp(sj - m_new);
            vl[t] = vl[t] * alpha + pj;
            vm[t] = m_new;
            vo[t] = vo[t] * alpha + pj * tailV[j - g.nm][lane];
          }
        vred[wave][t][lane] = vo[t];
        if (lane == 0) {
          vred[wave][t][ATT_HD] = vm[t];
          vred[wave][t][ATT_HD + 1] = vl[t];
        }
      }
    __syncthreads();
    if (wave == 0)
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          float m = -INFINITY;
#pragma unroll
          for (int w = 0; w < W; ++w) m = fmaxf(m, vred[w][t][ATT_HD]);
          float l = 0.f, o = 0.f;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const float sc = fast_exp(vred[w][t][ATT_HD] - m);          // a wave that saw no tile: exp(-inf) = 0
            l += vred[w][t][ATT_HD + 1] * sc;
            o += vred[w][t][lane] * sc;
          }
          O[((size_t)b * g.n + g.nm + t) * (g.H * ATT_HD) + (size_t)h * ATT_HD + lane] = o / l;
          if (lane == 0) LSE[(size_t)bh * g.n + g.nm + t] = m + logf(l);
        }
    __syncthreads();                                               // vred shares nothing with the staging below, but keep the waves together
  }
  if (!wave_active) return;
  for (int j = g.nm; j < g.n; ++j) {                             // remainder keys (e.g. the 257th token): vector code
    const float sj = tailSt[j - g.nm] == 0.f ? own_dot(qreg, tailK[j - g.nm], half) * g.scale : -1e7f;
    const float m_new = fmaxf(m_i, sj);
    const float alpha = fast_exp(m_i - m_new), pj = fast_exp(sj - m_new);
    l_i = l_i * alpha + pj;
    m_i = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      o0[e] *= alpha;
      o1[e] *= alpha;
    }
    own_axpy(o0, o1, tailV[j - g.nm], pj, half);
  }
  // all waves have passed the last barrier: the K/V buffers are free, reuse them as per-wave staging
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, o0, o1, O + (size_t)b * g.n * (g.H * ATT_HD) + (size_t)h * ATT_HD, (size_t)g.H * ATT_HD,
             blockIdx.x * (32 * W) + wave * 32, g.nm, lane, 1.0f / l_i);
  if (half == 0 && q < g.nm) LSE[(size_t)bh * g.n + q] = m_i + logf(l_i);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, query side: workgroup = 128 queries; loop over key tiles.  Recomputes P^T from LSE, writes dQ and
// Dvec[b, h, q] = sum_d dO * O (consumed by the key-side kernel).
// ---------------------------------------------------------------------------------------------------------------------
template <int W, int TAIL>
__global__ __launch_bounds__(64 * W, 2) void attn_bwd_dq_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ O,
                                                           const float* __restrict__ LSE, float* __restrict__ Dvec,
                                                           float* __restrict__ dqkv) {
  constexpr int TMAX = TAIL > 0 ? TAIL : 1;
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 32];
  float* Ks = smem;
  float* Vs = smem + 2 * 32 * ATT_LD;
  float* Kst = smem + 4 * 32 * ATT_LD;
  __shared__ __attribute__((aligned(16))) float tailK[ATT_TAIL_MAX][ATT_HD], tailV[ATT_TAIL_MAX][ATT_HD];   // the remainder keys [nm, n)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q0 = blockIdx.x * (32 * W) + wave * 32, q = q0 + (lane & 31);
  const bool vec = TAIL && blockIdx.x == 0;                   // this workgroup also owns the remainder queries [nm, n) (see attn_fwd_kernel)
  constexpr bool vecwg = false;
  constexpr bool stager = true;
  const bool wave_active = q0 < g.nm;
  __shared__ __attribute__((aligned(16))) float vecQ[ATT_TAIL_MAX][ATT_HD], vecD[ATT_TAIL_MAX][ATT_HD];
  __shared__ float vred[W][ATT_TAIL_MAX][ATT_HD], tailSt[ATT_TAIL_MAX];
  const int nt = g.n - g.nm;
  float vlse[TMAX], vD[TMAX], vdq[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { vlse[t] = INFINITY; vD[t] = 0.f; vdq[t] = 0.f; }
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  const float* dOb = dO + (size_t)b * g.n * os + (size_t)h * ATT_HD;
  const float* Ob = O + (size_t)b * g.n * os + (size_t)h * ATT_HD;

  float qreg[32], doreg[32];
  own_rows_load(qreg, Qb, rs, vecwg ? g.nm : q, g.nm, half);
  own_rows_load(doreg, dOb, os, vecwg ? g.nm : q, g.nm, half);
  if (tid < nt) tailSt[tid] = key_state(g, b, g.nm + tid, g.n);
  if (vec)
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < nt) {
        const float dv = dOb[(size_t)(g.nm + t) * os + lane];
        if (wave == 0) {
          vecQ[t][lane] = Qb[(size_t)(g.nm + t) * rs + lane];
          vecD[t][lane] = dv;
        }
        vD[t] = wave_sum(dv * Ob[(size_t)(g.nm + t) * os + lane]);
        vlse[t] = LSE[(size_t)bh * g.n + g.nm + t];
      }
  float dsum = 0.f;
  {
    float oreg[32];
    own_rows_load(oreg, Ob, os, vecwg ? g.nm : q, g.nm, half);
#pragma unroll
    for (int c = 0; c < 32; ++c) dsum += doreg[c] * oreg[c];
  }
  dsum += __shfl_xor(dsum, 32, 64);
  const float lse = (!vecwg && q < g.nm) ? LSE[(size_t)bh * g.n + q] : INFINITY;   // rows beyond nm: p = exp(s - inf) = 0
  if (!vecwg && half == 0 && q < g.nm) Dvec[(size_t)bh * g.n + q] = dsum;
  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) dq0[e] = dq1[e] = 0.f;

  const int ntiles = (g.nm + 31) / 32;
  TileRegs<64 * W> kr, vr;
  if (stager) {
    kr.load(Kb, rs, 0, g.nm, tid);
    vr.load(Vb, rs, 0, g.nm, tid);
    kr.store(Ks, tid);
    vr.store(Vs, tid);
  }
  if (tid < 32) Kst[tid] = key_state(g, b, tid, g.nm);
  for (int i = tid; stager && i < (g.n - g.nm) * ATT_HD; i += 64 * W) {
    tailK[i >> 6][i & 63] = Kb[(size_t)(g.nm + (i >> 6)) * rs + (i & 63)];
    tailV[i >> 6][i & 63] = Vb[(size_t)(g.nm + (i >> 6)) * rs + (i & 63)];
  }
  __syncthreads();
  for (int jt = 0; jt < ntiles; ++jt) {
    const int cur = jt & 1, nxt = cur ^ 1;
    const bool more = jt + 1 < ntiles;
    float kst_next = 0.f;
    if (more && stager) {
      kr.load(Kb, rs, (jt + 1) * 32, g.nm, tid);
      vr.load(Vb, rs, (jt + 1) * 32, g.nm, tid);
      if (tid < 32) kst_next = key_state(g, b, (jt + 1) * 32 + tid, g.nm);    // pad-mask load issued here, consumed after the tile's work
    }
    if (vec && jt % W == wave) {
      const float* kt = Ks + cur * 32 * ATT_LD;
      const float* vt = Vs + cur * 32 * ATT_LD;
      const bool att = Kst[cur * 32 + (lane & 31)] == 0.f;
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          const float sv = vec_row_dot(kt, vecQ[t], lane), dpv = vec_row_dot(vt, vecD[t], lane);
          const float pv = att ? fast_exp(sv * g.scale - vlse[t]) : 0.f;
          vdq[t] += vec_wsum(kt, pv * (dpv - vD[t]) * g.scale, lane);
        }
    }
    if (wave_active) {
      const float* kt = Ks + cur * 32 * ATT_LD;
      const float* vt = Vs + cur * 32 * ATT_LD;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
      mma_tile_own(s, kt, qreg, lane);                          // S^T  = K Q^T
      mma_tile_own(dp, vt, doreg, lane);                        // dP^T = V dO^T
      float ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float st = Kst[cur * 32 + acc_row(e, half)];
        const float pe = st == 0.f ? fast_exp(s[e] * g.scale - lse) : 0.f;   // padded keys: exp(-1e7 - lse) == 0 exactly
        ds[e] = pe * (dp[e] - dsum) * g.scale;
      }
      mma_tileT_acc(dq0, dq1, kt, ds, lane);                    // dQ^T += K^T dS^T
    }
    if (more && stager) {
      kr.store(Ks + nxt * 32 * ATT_LD, tid);
      vr.store(Vs + nxt * 32 * ATT_LD, tid);
      if (tid < 32) Kst[nxt * 32 + tid] = kst_next;
    }
    __syncthreads();
  }
  if (vec) {        // remainder queries: wave 0 adds the remainder keys, the waves' partial dQ rows are summed, wave 0 writes
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < nt) {
        if (wave == 0)
          for (int j = g.nm; j < g.n; ++j) {
            const float sj = wave_sum(vecQ[t][lane] * tailK[j - g.nm][lane]), dpj = wave_sum(vecD[t][lane] * tailV[j - g.nm][lane]);
            const float pj = tailSt[j - g.nm] == 0.f ? fast_exp(sj * g.scale - vlse[t]) : 0.f;
            vdq[t] += pj * (dpj - vD[t]) * g.scale * tailK[j - g.nm][lane];
          }
        vred[wave][t][lane] = vdq[t];
      }
    __syncthreads();
    if (wave == 0)
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < W; ++w) a += vred[w][t][lane];
          dqkv[qkv_off(g, b, g.nm + t, 0, h) + lane] = a;
          if (lane == 0) Dvec[(size_t)bh * g.n + g.nm + t] = vD[t];
        }
    __syncthreads();
  }
  if (!wave_active) return;
  for (int j = g.nm; j < g.n; ++j) {                             // remainder keys: vector code
    const float sj = own_dot(qreg, tailK[j - g.nm], half), dpj = own_dot(doreg, tailV[j - g.nm], half);
    const float pj = tailSt[j - g.nm] == 0.f ? fast_exp(sj * g.scale - lse) : 0.f;
    own_axpy(dq0, dq1, tailK[j - g.nm], pj * (dpj - dsum) * g.scale, half);
  }
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, dq0, dq1, dqkv + qkv_off(g, b, 0, 0, h), rs, q0, g.nm, lane, 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, key side: workgroup = 128 keys; loop over query tiles of 32 (Q, dO, LSE, Dvec through LDS).  Writes dK, dV.
// ---------------------------------------------------------------------------------------------------------------------
template <int W, int TAIL>
__global__ __launch_bounds__(64 * W, 2) void attn_bwd_dkv_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ LSE,
                                                            const float* __restrict__ Dvec, float* __restrict__ dqkv) {
  constexpr int TMAX = TAIL > 0 ? TAIL : 1;
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 64];
  float* Qs = smem;
  float* dOs = smem + 2 * 32 * ATT_LD;
  float* Rs = smem + 4 * 32 * ATT_LD;      // [2][64]: lse (32) | Dvec (32) of the query tile
  __shared__ __attribute__((aligned(16))) float tailQ[ATT_TAIL_MAX][ATT_HD], tailD[ATT_TAIL_MAX][ATT_HD];   // remainder queries: Q, dO rows
  __shared__ float tailS[ATT_TAIL_MAX][2];                                                                  // their lse, rowsum(dO * O)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int k0 = blockIdx.x * (32 * W) + wave * 32, key = k0 + (lane & 31);
  const bool vec = TAIL && blockIdx.x == 0;                   // this workgroup also owns the remainder keys [nm, n) (see attn_fwd_kernel)
  constexpr bool vecwg = false;
  constexpr bool stager = true;
  const bool wave_active = k0 < g.nm;
  __shared__ __attribute__((aligned(16))) float vecK[ATT_TAIL_MAX][ATT_HD], vecV[ATT_TAIL_MAX][ATT_HD];
  __shared__ float vred[W][ATT_TAIL_MAX][2][ATT_HD];
  const int nt = g.n - g.nm;
  float vdk[TMAX], vdv[TMAX];
  bool vatt[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { vdk[t] = 0.f; vdv[t] = 0.f; vatt[t] = false; }
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  const float* dOb = dO + (size_t)b * g.n * os + (size_t)h * ATT_HD;

  float kreg[32], vreg[32];
  own_rows_load(kreg, Kb, rs, vecwg ? g.nm : key, g.nm, half);
  own_rows_load(vreg, Vb, rs, vecwg ? g.nm : key, g.nm, half);
  if (vec)
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < nt) {
        if (wave == 0) {
          vecK[t][lane] = Kb[(size_t)(g.nm + t) * rs + lane];
          vecV[t][lane] = Vb[(size_t)(g.nm + t) * rs + lane];
        }
        vatt[t] = key_state(g, b, g.nm + t, g.n) == 0.f;
      }
  const bool attend = key_state(g, b, key, g.nm) == 0.f;               // padded / missing keys: P == 0, so dK = dV = 0
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int e = 0; e < 16; ++e) dk0[e] = dk1[e] = dv0[e] = dv1[e] = 0.f;

  const int ntiles = (g.nm + 31) / 32;
  auto row_stat = [&](int t0) -> float {                        // lse (tid < 32) / rowsum(dO * O) (tid 32..63) of query t0 + tid % 32
    float v = tid < 32 ? INFINITY : 0.f;                         // queries beyond nm: lse = +inf -> p = 0
    if (tid < 64) {
      const int qq = t0 + (tid & 31);
      if (qq < g.nm) v = tid < 32 ? LSE[(size_t)bh * g.n + qq] : Dvec[(size_t)bh * g.n + qq];
    }
    return v;
  };
  TileRegs<64 * W> qr, dr;
  if (stager) {
    qr.load(Qb, rs, 0, g.nm, tid);
    dr.load(dOb, os, 0, g.nm, tid);
    qr.store(Qs, tid);
    dr.store(dOs, tid);
  }
  if (tid < 64) Rs[tid] = row_stat(0);
  for (int i = tid; stager && i < (g.n - g.nm) * ATT_HD; i += 64 * W) {
    tailQ[i >> 6][i & 63] = Qb[(size_t)(g.nm + (i >> 6)) * rs + (i & 63)];
    tailD[i >> 6][i & 63] = dOb[(size_t)(g.nm + (i >> 6)) * os + (i & 63)];
  }
  if (tid < 2 * (g.n - g.nm)) tailS[tid >> 1][tid & 1] = (tid & 1) ? Dvec[(size_t)bh * g.n + g.nm + (tid >> 1)] : LSE[(size_t)bh * g.n + g.nm + (tid >> 1)];
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    const bool more = it + 1 < ntiles;
    float stat_next = 0.f;
    if (more && stager) {
      qr.load(Qb, rs, (it + 1) * 32, g.nm, tid);
      dr.load(dOb, os, (it + 1) * 32, g.nm, tid);
      stat_next = row_stat((it + 1) * 32);                      // issued with the tile loads, consumed after the tile's work
    }
    if (vec && it % W == wave) {
      const float* qt = Qs + cur * 32 * ATT_LD;
      const float* dt = dOs + cur * 32 * ATT_LD;
      const float lse_q = Rs[cur * 64 + (lane & 31)], d_q = Rs[cur * 64 + 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          const float sv = vec_row_dot(qt, vecK[t], lane), dpv = vec_row_dot(dt, vecV[t], lane);
          const float pv = vatt[t] ? fast_exp(sv * g.scale - lse_q) : 0.f;
          vdv[t] += vec_wsum(dt, pv, lane);
          vdk[t] += vec_wsum(qt, pv * (dpv - d_q) * g.scale, lane);
        }
    }
    if (wave_active) {
      const float* qt = Qs + cur * 32 * ATT_LD;
      const float* dt = dOs + cur * 32 * ATT_LD;
      const float* st = Rs + cur * 64;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
      mma_tile_own(s, qt, kreg, lane);                          // S  = Q K^T : s[e] <-> (query acc_row(e, half), key lane)
      mma_tile_own(dp, dt, vreg, lane);                         // dP = dO V^T
      float p[16], ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = acc_row(e, half);
        p[e] = attend ? fast_exp(s[e] * g.scale - st[r]) : 0.f;
        ds[e] = p[e] * (dp[e] - st[32 + r]) * g.scale;
      }
      mma_tileT_acc(dv0, dv1, dt, p, lane);                     // dV^T += dO^T P
      mma_tileT_acc(dk0, dk1, qt, ds, lane);                    // dK^T += Q^T dS
    }
    if (more && stager) {
      qr.store(Qs + nxt * 32 * ATT_LD, tid);
      dr.store(dOs + nxt * 32 * ATT_LD, tid);
      if (tid < 64) Rs[nxt * 64 + tid] = stat_next;
    }
    __syncthreads();
  }
  if (vec) {        // remainder keys: wave 0 adds the remainder queries, the waves' partial dK / dV rows are summed, wave 0 writes
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < nt) {
        if (wave == 0)
          for (int qq = g.nm; qq < g.n; ++qq) {
            const float sq = wave_sum(vecK[t][lane] * tailQ[qq - g.nm][lane]), dpq = wave_sum(vecV[t][lane] * tailD[qq - g.nm][lane]);
            const float pq = vatt[t] ? fast_exp(sq * g.scale - tailS[qq - g.nm][0]) : 0.f;
            vdv[t] += pq * tailD[qq - g.nm][lane];
            vdk[t] += pq * (dpq - tailS[qq - g.nm][1]) * g.scale * tailQ[qq - g.nm][lane];
          }
        vred[wave][t][0][lane] = vdk[t];
        vred[wave][t][1][lane] = vdv[t];
      }
    __syncthreads();
    if (wave == 0)
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < nt) {
          float a = 0.f, c = 0.f;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            a += vred[w][t][0][lane];
            c += vred[w][t][1][lane];
          }
          dqkv[qkv_off(g, b, g.nm + t, 1, h) + lane] = a;
          dqkv[qkv_off(g, b, g.nm + t, 2, h) + lane] = c;
        }
    __syncthreads();
  }
  if (!wave_active) return;
  for (int qq = g.nm; qq < g.n; ++qq) {                          // remainder queries: vector code
    const float* qrow = tailQ[qq - g.nm];
    const float* drow = tailD[qq - g.nm];
    const float sq = own_dot(kreg, qrow, half), dpq = own_dot(vreg, drow, half);
    const float pq = attend ? fast_exp(sq * g.scale - tailS[qq - g.nm][0]) : 0.f;
    own_axpy(dv0, dv1, drow, pq, half);
    own_axpy(dk0, dk1, qrow, pq * (dpq - tailS[qq - g.nm][1]) * g.scale, half);
  }
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, dk0, dk1, dqkv + qkv_off(g, b, 0, 1, h), rs, k0, g.nm, lane, 1.0f);
  __builtin_amdgcn_wave_barrier();
  store_ownT(stage, dv0, dv1, dqkv + qkv_off(g, b, 0, 2, h), rs, k0, g.nm, lane, 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone vector kernels for sequences shorter than one tile (n <= ATT_TAIL_MAX, nm == 0; otherwise the vector wave of the
// tiled kernels owns the remainder rows): one 4-wave workgroup per (b, h, row), plain vector code over the whole sequence (scores of one
// row against n keys / queries in LDS, lanes over the other axis for the dot products, lanes over the 64 features for the
// weighted sums).  ~n x 128 FMAs per wave: microseconds for the whole grid.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_dot(const float* __restrict__ row, const float* __restrict__ vec_lds) {
  const f32x4* p = reinterpret_cast<const f32x4*>(row);
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const f32x4 v = p[c];
    d += v[0] * vec_lds[4 * c] + v[1] * vec_lds[4 * c + 1] + v[2] * vec_lds[4 * c + 2] + v[3] * vec_lds[4 * c + 3];
  }
  return d;
}

// weighted sum over the sequence: out[d] = sum_j w[j] * rows[j][d]; ATT_TW waves = ATT_TW row lanes x 64 features, 16 independent
// loads per lane in flight per round (a tail kernel is a few dependent memory round trips, nothing else), lanes combined through
// LDS in a fixed order.  Result in threads 0..63 (feature = tid).
#define ATT_TW 4
__device__ __forceinline__ float tail_wsum(const float* __restrict__ rows, size_t row_stride, const float* __restrict__ w, int n,
                                            float (*red)[ATT_HD]) {
  const int d = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float acc = 0.f;
  for (int j0 = rl; j0 < n; j0 += 16 * ATT_TW) {
    float v[16], ww[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int j = j0 + k * ATT_TW;
      v[k] = j < n ? rows[(size_t)j * row_stride + d] : 0.f;
      ww[k] = j < n ? w[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += ww[k] * v[k];
  }
  red[rl][d] = acc;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x < 64)
#pragma unroll
    for (int k = 0; k < ATT_TW; ++k) t += red[k][d];
  return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* redw) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) redw[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = redw[0];
#pragma unroll
  for (int k = 1; k < ATT_TW; ++k) t = fmaxf(t, redw[k]);
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* redw) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) redw[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < ATT_TW; ++k) t += redw[k];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(64 * ATT_TW) void attn_fwd_tail_kernel(const AttGeom g, float* __restrict__ O, float* __restrict__ LSE) {
  __shared__ float qs[ATT_HD], ps[ATT_TAIL_N], red[ATT_TW][ATT_HD], red4[ATT_TW];
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / g.H, h = bh - b * g.H, q = g.nm + blockIdx.y;
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  if (tid < ATT_HD) qs[tid] = g.qkv[qkv_off(g, b, q, 0, h) + tid];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < g.n; j += 64 * ATT_TW) {
    const float sj = key_state(g, b, j, g.n) == 0.f ? row_dot(Kb + (size_t)j * rs, qs) * g.scale : -1e7f;
    ps[j] = sj;
    mx = fmaxf(mx, sj);
  }
  mx = block_reduce_max(mx, red4);
  float sum = 0.f;
  for (int j = tid; j < g.n; j += 64 * ATT_TW) {
    const float pj = fast_exp(ps[j] - mx);
    ps[j] = pj;
    sum += pj;
  }
  sum = block_reduce_sum(sum, red4);
  const float acc = tail_wsum(Vb, rs, ps, g.n, red);
  if (tid < ATT_HD) O[((size_t)b * g.n + q) * os + (size_t)h * ATT_HD + tid] = acc / sum;
  if (tid == 0) LSE[(size_t)bh * g.n + q] = mx + logf(sum);
}

__global__ __launch_bounds__(64 * ATT_TW) void attn_bwd_dq_tail_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ O,
                                                                const float* __restrict__ LSE, float* __restrict__ Dvec,
                                                                float* __restrict__ dqkv) {
  __shared__ float qs[ATT_HD], dos[ATT_HD], dss[ATT_TAIL_N], red[ATT_TW][ATT_HD], red4[ATT_TW];
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / g.H, h = bh - b * g.H, q = g.nm + blockIdx.y;
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  const size_t orow = ((size_t)b * g.n + q) * os + (size_t)h * ATT_HD;
  float prod = 0.f;
  if (tid < ATT_HD) {
    const float dov = dO[orow + tid];
    qs[tid] = g.qkv[qkv_off(g, b, q, 0, h) + tid];
    dos[tid] = dov;
    prod = dov * O[orow + tid];
  }
  const float dsum = block_reduce_sum(prod, red4);              // also orders the LDS writes above before the reads below
  const float lse = LSE[(size_t)bh * g.n + q];
  for (int j = tid; j < g.n; j += 64 * ATT_TW) {
    const float sj = row_dot(Kb + (size_t)j * rs, qs), dpj = row_dot(Vb + (size_t)j * rs, dos);
    const float pj = key_state(g, b, j, g.n) == 0.f ? fast_exp(sj * g.scale - lse) : 0.f;
    dss[j] = pj * (dpj - dsum) * g.scale;
  }
  __syncthreads();
  const float acc = tail_wsum(Kb, rs, dss, g.n, red);
  if (tid < ATT_HD) dqkv[qkv_off(g, b, q, 0, h) + tid] = acc;
  if (tid == 0) Dvec[(size_t)bh * g.n + q] = dsum;
}

__global__ __launch_bounds__(64 * ATT_TW) void attn_bwd_dkv_tail_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ LSE,
                                                                 const float* __restrict__ Dvec, float* __restrict__ dqkv) {
  __shared__ float ks[ATT_HD], vs[ATT_HD], ps[ATT_TAIL_N], dss[ATT_TAIL_N], red[ATT_TW][ATT_HD];
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / g.H, h = bh - b * g.H, key = g.nm + blockIdx.y;
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* dOb = dO + (size_t)b * g.n * os + (size_t)h * ATT_HD;
  if (tid < ATT_HD) {
    ks[tid] = g.qkv[qkv_off(g, b, key, 1, h) + tid];
    vs[tid] = g.qkv[qkv_off(g, b, key, 2, h) + tid];
  }
  const bool attend = key_state(g, b, key, g.n) == 0.f;
  __syncthreads();
  for (int qq = tid; qq < g.n; qq += 64 * ATT_TW) {
    const float sq = row_dot(Qb + (size_t)qq * rs, ks), dpq = row_dot(dOb + (size_t)qq * os, vs);
    const float pq = attend ? fast_exp(sq * g.scale - LSE[(size_t)bh * g.n + qq]) : 0.f;
    ps[qq] = pq;
    dss[qq] = pq * (dpq - Dvec[(size_t)bh * g.n + qq]) * g.scale;
  }
  __syncthreads();
  const float dv = tail_wsum(dOb, os, ps, g.n, red);
  __syncthreads();
  const float dk = tail_wsum(Qb, rs, dss, g.n, red);
  if (tid < ATT_HD) {
    dqkv[qkv_off(g, b, key, 1, h) + tid] = dk;
    dqkv[qkv_off(g, b, key, 2, h) + tid] = dv;
  }
}

// rows the MFMA kernels own: everything, unless the last partial tile holds <= ATT_TAIL_MAX rows
int att_main_rows(int n) {
  const int r = n % 32;
  return (r > 0 && r <= ATT_TAIL_MAX && n <= ATT_TAIL_N) ? n - r : n;
}

// Waves (= blocks of 32 owned rows) per workgroup: the sequence is cut into ceil(n / 32) wave blocks and a workgroup whose
// last waves have no rows keeps its CU slot for the whole key loop, so pick the width that leaves the fewest idle waves
// (n = 257 = cls + 256: nine blocks -> three workgroups of three waves instead of 4 + 4 + 1; n = 512: four).
int att_waves(int n) {
  const int blocks = (n + 31) / 32;
  int best = 4, best_idle = 1 << 30;
  for (int w = 4; w >= 2; --w) {
    const int idle = ((blocks + w - 1) / w) * w - blocks;
    if (idle < best_idle) { best = w; best_idle = idle; }
  }
  return best;
}

int check_att(const char* who, int B, int H, int n, int hd) {
  MLA_REQUIRE(B > 0 && H > 0 && n > 0, "%s: non-positive dims", who);
  MLA_REQUIRE(hd == ATT_HD, "%s: head dim %d unsupported (64)", who, hd);
  MLA_REQUIRE((long)B * H < 65536 && (long)B * n * 3 * H * hd < (1L << 31), "%s: problem too large", who);
  return MLA_OK;
}

}  // namespace

extern "C" int mla_attention_fwd(const float* qkv, const float* pad_mask, float* o, float* lse, int B, int H, int n, int hd,
                                 void* stream) {
  MLA_REQUIRE(qkv && o && lse, "mla_attention_fwd: null pointer");
  if (int rc = check_att("mla_attention_fwd", B, H, n, hd)) return rc;
  const int nm = att_main_rows(n);
  AttGeom g{qkv, pad_mask, B, H, n, nm, 1.0f / sqrtf((float)hd)};
  hipStream_t st = (hipStream_t)stream;
  if (nm > 0) {
    const int W = att_waves(nm);
    const dim3 grid(cdiv(nm, 32 * W), B * H);
    if (nm + 1 == n) {    // wave 0 of workgroup 0 also owns the remainder rows; one row ([cls] + 256 patches): 3 instead of 9 state registers
      if (W == 4) attn_fwd_kernel<4, 1><<<grid, 256, 0, st>>>(g, o, lse);
      else if (W == 3) attn_fwd_kernel<3, 1><<<grid, 192, 0, st>>>(g, o, lse);
      else attn_fwd_kernel<2, 1><<<grid, 128, 0, st>>>(g, o, lse);
    } else if (nm < n) {
      if (W == 4) attn_fwd_kernel<4, ATT_TAIL_MAX><<<grid, 256, 0, st>>>(g, o, lse);
      else if (W == 3) attn_fwd_kernel<3, ATT_TAIL_MAX><<<grid, 192, 0, st>>>(g, o, lse);
      else attn_fwd_kernel<2, ATT_TAIL_MAX><<<grid, 128, 0, st>>>(g, o, lse);
    } else {
      if (W == 4) attn_fwd_kernel<4, 0><<<grid, 256, 0, st>>>(g, o, lse);
      else if (W == 3) attn_fwd_kernel<3, 0><<<grid, 192, 0, st>>>(g, o, lse);
      else attn_fwd_kernel<2, 0><<<grid, 128, 0, st>>>(g, o, lse);
    }
  } else {                // n <= ATT_TAIL_MAX: nothing for the MFMA kernels, the stand-alone vector kernel does all rows
    attn_fwd_tail_kernel<<<dim3(B * H, n - nm), 64 * ATT_TW, 0, st>>>(g, o, lse);
  }
  MLA_CHECK_LAUNCH("attn_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_attention_bwd(const float* d_o, const float* qkv, const float* o, const float* lse, const float* pad_mask,
                                 float* dqkv, float* dvec, int B, int H, int n, int hd, void* stream) {
  MLA_REQUIRE(d_o && qkv && o && lse && dqkv && dvec, "mla_attention_bwd: null pointer");
  if (int rc = check_att("mla_attention_bwd", B, H, n, hd)) return rc;
  const int nm = att_main_rows(n);
  AttGeom g{qkv, pad_mask, B, H, n, nm, 1.0f / sqrtf((float)hd)};
  hipStream_t st = (hipStream_t)stream;
  const int W = att_waves(nm > 0 ? nm : 32);
  const dim3 grid(cdiv(nm > 0 ? nm : 1, 32 * W), B * H), tail(B * H, n - nm);
  // order matters: the query-side kernel writes dvec (rowsum(d_o * o)) for every row before the key-side kernel reads it
  if (nm > 0) {
    if (nm + 1 == n) {
      if (W == 4) attn_bwd_dq_kernel<4, 1><<<grid, 256, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dq_kernel<3, 1><<<grid, 192, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else attn_bwd_dq_kernel<2, 1><<<grid, 128, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      MLA_CHECK_LAUNCH("attn_bwd_dq_kernel");
      if (W == 4) attn_bwd_dkv_kernel<4, 1><<<grid, 256, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dkv_kernel<3, 1><<<grid, 192, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else attn_bwd_dkv_kernel<2, 1><<<grid, 128, 0, st>>>(g, d_o, lse, dvec, dqkv);
    } else if (nm < n) {
      if (W == 4) attn_bwd_dq_kernel<4, ATT_TAIL_MAX><<<grid, 256, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dq_kernel<3, ATT_TAIL_MAX><<<grid, 192, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else attn_bwd_dq_kernel<2, ATT_TAIL_MAX><<<grid, 128, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      MLA_CHECK_LAUNCH("attn_bwd_dq_kernel");
      if (W == 4) attn_bwd_dkv_kernel<4, ATT_TAIL_MAX><<<grid, 256, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dkv_kernel<3, ATT_TAIL_MAX><<<grid, 192, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else attn_bwd_dkv_kernel<2, ATT_TAIL_MAX><<<grid, 128, 0, st>>>(g, d_o, lse, dvec, dqkv);
    } else {
      if (W == 4) attn_bwd_dq_kernel<4, 0><<<grid, 256, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dq_kernel<3, 0><<<grid, 192, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      else attn_bwd_dq_kernel<2, 0><<<grid, 128, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
      MLA_CHECK_LAUNCH("attn_bwd_dq_kernel");
      if (W == 4) attn_bwd_dkv_kernel<4, 0><<<grid, 256, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else if (W == 3) attn_bwd_dkv_kernel<3, 0><<<grid, 192, 0, st>>>(g, d_o, lse, dvec, dqkv);
      else attn_bwd_dkv_kernel<2, 0><<<grid, 128, 0, st>>>(g, d_o, lse, dvec, dqkv);
    }
  } else {                // n <= ATT_TAIL_MAX: the stand-alone vector kernels do all rows
    attn_bwd_dq_tail_kernel<<<tail, 64 * ATT_TW, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
    MLA_CHECK_LAUNCH("attn_bwd_dq_tail_kernel");
    attn_bwd_dkv_tail_kernel<<<tail, 64 * ATT_TW, 0, st>>>(g, d_o, lse, dvec, dqkv);
  }
  MLA_CHECK_LAUNCH("attn_bwd_dkv_kernel");
  return MLA_OK;
}
