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
// Masking: keys beyond n do not exist (-inf, probability exactly 0); padded keys (mask > 0) have their score REPLACED by
// -1e7 like the reference, which underflows to probability exactly 0 in fp32 unless a whole row is padded (never: the
// [cls] key is always present, m3ae.py:347).
#include "common.h"

#define ATT_HD 64          // head dim (ViT-B: 768 / 12)
#define ATT_LD 68          // padded LDS row (floats): 17 x 16 B -> conflict-free ds_read_b128 across 32 rows

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
__device__ __forceinline__ float key_state(const AttGeom& g, int b, int key) {
  if (key >= g.n) return 2.f;
  return (g.pm && g.pm[(size_t)b * g.n + key] > 0.f) ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one (b, h); loop over key tiles of 32.  o (B, n, H*64), lse (B, H, n).
// ---------------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64 * W, 2) void attn_fwd_kernel(const AttGeom g, float* __restrict__ O, float* __restrict__ LSE) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 32];
  float* Ks = smem;                       // [2][32][ATT_LD]
  float* Vs = smem + 2 * 32 * ATT_LD;     // [2][32][ATT_LD]
  float* Kst = smem + 4 * 32 * ATT_LD;    // [2][32] key states
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q = blockIdx.x * (32 * W) + wave * 32 + (lane & 31);
  const bool wave_active = blockIdx.x * (32 * W) + wave * 32 < g.n;
  const size_t rs = (size_t)3 * g.H * ATT_HD;                  // token stride inside qkv
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);

  float qreg[32];
  own_rows_load(qreg, Qb, rs, q, g.n, half);
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
  float m_i = -INFINITY, l_i = 0.f;

  const int ntiles = (g.n + 31) / 32;
  TileRegs<64 * W> kr, vr;
  kr.load(Kb, rs, 0, g.n, tid);
  vr.load(Vb, rs, 0, g.n, tid);
  kr.store(Ks, tid);
  vr.store(Vs, tid);
  if (tid < 32) Kst[tid] = key_state(g, b, tid);
  __syncthreads();
  for (int jt = 0; jt < ntiles; ++jt) {
    const int cur = jt & 1, nxt = cur ^ 1;
    const bool more = jt + 1 < ntiles;
    if (more) {
      kr.load(Kb, rs, (jt + 1) * 32, g.n, tid);
      vr.load(Vb, rs, (jt + 1) * 32, g.n, tid);
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
    if (more) {
      kr.store(Ks + nxt * 32 * ATT_LD, tid);
      vr.store(Vs + nxt * 32 * ATT_LD, tid);
      if (tid < 32) Kst[nxt * 32 + tid] = key_state(g, b, (jt + 1) * 32 + tid);
    }
    __syncthreads();
  }
  if (!wave_active) return;
  // all waves have passed the last barrier: the K/V buffers are free, reuse them as per-wave staging
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, o0, o1, O + (size_t)b * g.n * (g.H * ATT_HD) + (size_t)h * ATT_HD, (size_t)g.H * ATT_HD,
             blockIdx.x * (32 * W) + wave * 32, g.n, lane, 1.0f / l_i);
  if (half == 0 && q < g.n) LSE[(size_t)bh * g.n + q] = m_i + logf(l_i);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, query side: workgroup = 128 queries; loop over key tiles.  Recomputes P^T from LSE, writes dQ and
// Dvec[b, h, q] = sum_d dO * O (consumed by the key-side kernel).
// ---------------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64 * W, 2) void attn_bwd_dq_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ O,
                                                           const float* __restrict__ LSE, float* __restrict__ Dvec,
                                                           float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 32];
  float* Ks = smem;
  float* Vs = smem + 2 * 32 * ATT_LD;
  float* Kst = smem + 4 * 32 * ATT_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q0 = blockIdx.x * (32 * W) + wave * 32, q = q0 + (lane & 31);
  const bool wave_active = q0 < g.n;
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  const float* dOb = dO + (size_t)b * g.n * os + (size_t)h * ATT_HD;
  const float* Ob = O + (size_t)b * g.n * os + (size_t)h * ATT_HD;

  float qreg[32], doreg[32];
  own_rows_load(qreg, Qb, rs, q, g.n, half);
  own_rows_load(doreg, dOb, os, q, g.n, half);
  float dsum = 0.f;
  {
    float oreg[32];
    own_rows_load(oreg, Ob, os, q, g.n, half);
#pragma unroll
    for (int c = 0; c < 32; ++c) dsum += doreg[c] * oreg[c];
  }
  dsum += __shfl_xor(dsum, 32, 64);
  const float lse = q < g.n ? LSE[(size_t)bh * g.n + q] : INFINITY;    // rows beyond n: p = exp(s - inf) = 0
  if (half == 0 && q < g.n) Dvec[(size_t)bh * g.n + q] = dsum;
  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) dq0[e] = dq1[e] = 0.f;

  const int ntiles = (g.n + 31) / 32;
  TileRegs<64 * W> kr, vr;
  kr.load(Kb, rs, 0, g.n, tid);
  vr.load(Vb, rs, 0, g.n, tid);
  kr.store(Ks, tid);
  vr.store(Vs, tid);
  if (tid < 32) Kst[tid] = key_state(g, b, tid);
  __syncthreads();
  for (int jt = 0; jt < ntiles; ++jt) {
    const int cur = jt & 1, nxt = cur ^ 1;
    const bool more = jt + 1 < ntiles;
    if (more) {
      kr.load(Kb, rs, (jt + 1) * 32, g.n, tid);
      vr.load(Vb, rs, (jt + 1) * 32, g.n, tid);
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
    if (more) {
      kr.store(Ks + nxt * 32 * ATT_LD, tid);
      vr.store(Vs + nxt * 32 * ATT_LD, tid);
      if (tid < 32) Kst[nxt * 32 + tid] = key_state(g, b, (jt + 1) * 32 + tid);
    }
    __syncthreads();
  }
  if (!wave_active) return;
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, dq0, dq1, dqkv + qkv_off(g, b, 0, 0, h), rs, q0, g.n, lane, 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, key side: workgroup = 128 keys; loop over query tiles of 32 (Q, dO, LSE, Dvec through LDS).  Writes dK, dV.
// ---------------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64 * W, 2) void attn_bwd_dkv_kernel(const AttGeom g, const float* __restrict__ dO, const float* __restrict__ LSE,
                                                            const float* __restrict__ Dvec, float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 32 * ATT_LD + 2 * 64];
  float* Qs = smem;
  float* dOs = smem + 2 * 32 * ATT_LD;
  float* Rs = smem + 4 * 32 * ATT_LD;      // [2][64]: lse (32) | Dvec (32) of the query tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int k0 = blockIdx.x * (32 * W) + wave * 32, key = k0 + (lane & 31);
  const bool wave_active = k0 < g.n;
  const size_t rs = (size_t)3 * g.H * ATT_HD, os = (size_t)g.H * ATT_HD;
  const float* Qb = g.qkv + qkv_off(g, b, 0, 0, h);
  const float* Kb = g.qkv + qkv_off(g, b, 0, 1, h);
  const float* Vb = g.qkv + qkv_off(g, b, 0, 2, h);
  const float* dOb = dO + (size_t)b * g.n * os + (size_t)h * ATT_HD;

  float kreg[32], vreg[32];
  own_rows_load(kreg, Kb, rs, key, g.n, half);
  own_rows_load(vreg, Vb, rs, key, g.n, half);
  const bool attend = key_state(g, b, key) == 0.f;               // padded / missing keys: P == 0, so dK = dV = 0
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int e = 0; e < 16; ++e) dk0[e] = dk1[e] = dv0[e] = dv1[e] = 0.f;

  const int ntiles = (g.n + 31) / 32;
  auto row_stats = [&](int t0, int buf) {
    if (tid < 64) {
      const int qq = t0 + (tid & 31);
      float v = tid < 32 ? INFINITY : 0.f;                       // queries beyond n: lse = +inf -> p = 0
      if (qq < g.n) v = tid < 32 ? LSE[(size_t)bh * g.n + qq] : Dvec[(size_t)bh * g.n + qq];
      Rs[buf * 64 + tid] = v;
    }
  };
  TileRegs<64 * W> qr, dr;
  qr.load(Qb, rs, 0, g.n, tid);
  dr.load(dOb, os, 0, g.n, tid);
  qr.store(Qs, tid);
  dr.store(dOs, tid);
  row_stats(0, 0);
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    const bool more = it + 1 < ntiles;
    if (more) {
      qr.load(Qb, rs, (it + 1) * 32, g.n, tid);
      dr.load(dOb, os, (it + 1) * 32, g.n, tid);
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
    if (more) {
      qr.store(Qs + nxt * 32 * ATT_LD, tid);
      dr.store(dOs + nxt * 32 * ATT_LD, tid);
      row_stats((it + 1) * 32, nxt);
    }
    __syncthreads();
  }
  if (!wave_active) return;
  float* stage = smem + wave * 32 * ATT_LD;
  store_ownT(stage, dk0, dk1, dqkv + qkv_off(g, b, 0, 1, h), rs, k0, g.n, lane, 1.0f);
  __builtin_amdgcn_wave_barrier();
  store_ownT(stage, dv0, dv1, dqkv + qkv_off(g, b, 0, 2, h), rs, k0, g.n, lane, 1.0f);
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
  AttGeom g{qkv, pad_mask, B, H, n, 1.0f / sqrtf((float)hd)};
  const int W = att_waves(n);
  const dim3 grid(cdiv(n, 32 * W), B * H);
  hipStream_t st = (hipStream_t)stream;
  if (W == 4) attn_fwd_kernel<4><<<grid, 256, 0, st>>>(g, o, lse);
  else if (W == 3) attn_fwd_kernel<3><<<grid, 192, 0, st>>>(g, o, lse);
  else attn_fwd_kernel<2><<<grid, 128, 0, st>>>(g, o, lse);
  MLA_CHECK_LAUNCH("attn_fwd_kernel");
  return MLA_OK;
}

extern "C" int mla_attention_bwd(const float* d_o, const float* qkv, const float* o, const float* lse, const float* pad_mask,
                                 float* dqkv, float* dvec, int B, int H, int n, int hd, void* stream) {
  MLA_REQUIRE(d_o && qkv && o && lse && dqkv && dvec, "mla_attention_bwd: null pointer");
  if (int rc = check_att("mla_attention_bwd", B, H, n, hd)) return rc;
  AttGeom g{qkv, pad_mask, B, H, n, 1.0f / sqrtf((float)hd)};
  const int W = att_waves(n);
  const dim3 grid(cdiv(n, 32 * W), B * H);
  hipStream_t st = (hipStream_t)stream;
  if (W == 4) attn_bwd_dq_kernel<4><<<grid, 256, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
  else if (W == 3) attn_bwd_dq_kernel<3><<<grid, 192, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
  else attn_bwd_dq_kernel<2><<<grid, 128, 0, st>>>(g, d_o, o, lse, dvec, dqkv);
  MLA_CHECK_LAUNCH("attn_bwd_dq_kernel");
  if (W == 4) attn_bwd_dkv_kernel<4><<<grid, 256, 0, st>>>(g, d_o, lse, dvec, dqkv);
  else if (W == 3) attn_bwd_dkv_kernel<3><<<grid, 192, 0, st>>>(g, d_o, lse, dvec, dqkv);
  else attn_bwd_dkv_kernel<2><<<grid, 128, 0, st>>>(g, d_o, lse, dvec, dqkv);
  MLA_CHECK_LAUNCH("attn_bwd_dkv_kernel");
  return MLA_OK;
}
