// Latency-bound small kernels on the critical path between the two modalities:
//   shared head + cross entropy   ConcatFusion.fc_out = nn.Linear(D,C) (models/fusion_modules.py:19) applied per
//                                 modality (main.py:432, 444) + nn.CrossEntropyLoss (main.py:130) + autograd
//   head-gradient projection      GSPlugin.before_update (utils/utils.py:24-41), literal
//   SGD momentum + weight decay   torch.optim.SGD.step (main.py:749, 439, 451), one flat launch per group
// Wave-per-row kernels with 64-lane shuffle reductions; no atomics (bitwise reproducible).
#include "common.h"

#define HEAD_MAXC 128

// ---- head: one workgroup per sample ------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const int64_t* __restrict__ labels,
                                                        float* __restrict__ logits, float* __restrict__ rowloss,
                                                        float* __restrict__ dlogits, float* __restrict__ dX, int B, int D,
                                                        int C, float inv_batch) {
  // One workgroup (4 waves) per sample.  Wave w forms the logits of classes c = w, w+4, ... (each class by ONE wave, lanes
  // striding the features: the summation order of the one-wave-per-sample form, bit-identical results), wave 0 does the
  // softmax / loss, then wave w forms dX for the feature slots d = 64 (w + 4 m) + lane.  The one-wave form was a chain of
  // C dependent reductions and C dependent row reads of W: 206 us at C = 101, D = 768 on the step's critical path.
  __shared__ float lg[HEAD_MAXC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const float* x = X + (size_t)row * D;
  for (int c = wave; c < C; c += 4) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[d] * W[(size_t)c * D + d];
    s = wave_sum(s);
    if (lane == 0) lg[c] = s + bias[c];
  }
  __syncthreads();
  if (wave == 0) {
    float l0 = lane < C ? lg[lane] : -INFINITY;
    float l1 = lane + 64 < C ? lg[lane + 64] : -INFINITY;
    const float m = wave_max(fmaxf(l0, l1));
    const float e0 = lane < C ? expf(l0 - m) : 0.f, e1 = lane + 64 < C ? expf(l1 - m) : 0.f;
    const float s = wave_sum(e0 + e1);
    const long lab_raw = (long)labels[row];
    const bool lab_ok = lab_raw >= 0 && lab_raw < C;   // out of range: NaN loss (the reference's CE kernel asserts), no OOB LDS read
    const int lab = lab_ok ? (int)lab_raw : 0;
    const float lse = m + logf(s);
    if (lane < C) logits[(size_t)row * C + lane] = l0;
    if (lane + 64 < C) logits[(size_t)row * C + lane + 64] = l1;
    if (lane == 0) rowloss[row] = lab_ok ? (lse - lg[lab]) * inv_batch : NAN;
    __builtin_amdgcn_wave_barrier();
    const float d0 = (e0 / s - (lane == lab ? 1.f : 0.f)) * inv_batch;
    const float d1 = (e1 / s - (lane + 64 == lab ? 1.f : 0.f)) * inv_batch;
    if (lane < C) {
      lg[lane] = d0;
      dlogits[(size_t)row * C + lane] = d0;
    }
    if (lane + 64 < C) {
      lg[lane + 64] = d1;
      dlogits[(size_t)row * C + lane + 64] = d1;
    }
  }
  __syncthreads();
  for (int d = wave * 64 + lane; d < D; d += 256) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += lg[c] * W[(size_t)c * D + d];
    dX[(size_t)row * D + d] = a;
  }
}

// grid (ceil(D/256), C): dW[c][d] = sum_rows dl[row][c] X[row][d]; block (0,c) also db[c]; block (0,0) the loss.
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ X, const float* __restrict__ dlogits,
                                                         const float* __restrict__ rowloss, float* __restrict__ dW,
                                                         float* __restrict__ db, float* __restrict__ loss, int B, int D, int C) {
  const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d < D) {
    float a = 0.f;
    for (int r = 0; r < B; ++r) a += dlogits[(size_t)r * C + c] * X[(size_t)r * D + d];
    dW[(size_t)c * D + d] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float a = 0.f;
    for (int r = threadIdx.x; r < B; r += 64) a += dlogits[(size_t)r * C + c];
    a = wave_sum(a);
    if (threadIdx.x == 0) db[c] = a;
    if (c == 0) {
      float l = 0.f;
      for (int r = threadIdx.x; r < B; r += 64) l += rowloss[r];
      l = wave_sum(l);
      if (threadIdx.x == 0) *loss = l;
    }
  }
}

extern "C" size_t mla_head_ws_elems(int B, int C) { return (size_t)B * C + B; }

extern "C" int mla_head_ce_fwd_bwd(const float* X, const float* W, const float* b, const int64_t* labels, float* logits,
                                   float* loss, float* dW, float* db, float* dX, float* ws, int B, int D, int C,
                                   float inv_batch, void* stream) {
  MLA_REQUIRE(X && W && b && labels && logits && loss && dW && db && dX && ws, "mla_head_ce_fwd_bwd: null pointer");
  MLA_REQUIRE(B > 0 && D > 0 && C > 0 && C <= HEAD_MAXC, "mla_head_ce_fwd_bwd: need 0 < C <= %d (got %d)", HEAD_MAXC, C);
  hipStream_t st = (hipStream_t)stream;
  float* dlogits = ws;
  float* rowloss = ws + (size_t)B * C;
  head_fwd_kernel<<<B, 256, 0, st>>>(X, W, b, labels, logits, rowloss, dlogits, dX, B, D, C, inv_batch);
  MLA_CHECK_LAUNCH("head_fwd_kernel");
  head_grad_kernel<<<dim3(cdiv(D, 256), C), 256, 0, st>>>(X, dlogits, rowloss, dW, db, loss, B, D, C);
  MLA_CHECK_LAUNCH("head_grad_kernel");
  return MLA_OK;
}

// ---- autograd-protocol entry points: the same head, split where autograd splits it (main.py:432-435) -------------
// nn.CrossEntropyLoss (mean) forward + d logits in one launch: one wave per sample.  A label outside [0, C) poisons
// the loss with NaN (the reference's CUDA kernel asserts); its d logits row is zero.
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          float* __restrict__ rowloss, float* __restrict__ dlogits, int B, int C,
                                                          float inv_batch) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= B) return;
  const float* l = logits + (size_t)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, l[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(l[c] - m);
  s = wave_sum(s);
  const long lab = (long)labels[row];
  const bool ok = lab >= 0 && lab < C;
  if (lane == 0) rowloss[row] = ok ? (m + logf(s) - l[lab]) * inv_batch : NAN;
  for (int c = lane; c < C; c += 64)
    dlogits[(size_t)row * C + c] = ok ? (expf(l[c] - m) / s - (c == lab ? 1.f : 0.f)) * inv_batch : 0.f;
}
__global__ __launch_bounds__(64) void sum_to_scalar_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) a += v[i];
  a = wave_sum(a);
  if (threadIdx.x == 0) *out = a;
}
extern "C" int mla_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* ws, int B, int C,
                              float inv_batch, void* stream) {
  MLA_REQUIRE(logits && labels && loss && dlogits && ws && B > 0 && C > 0, "mla_ce_fwd_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ce_fwd_bwd_kernel<<<cdiv(B, 4), 256, 0, st>>>(logits, labels, ws, dlogits, B, C, inv_batch);
  MLA_CHECK_LAUNCH("ce_fwd_bwd_kernel");
  sum_to_scalar_kernel<<<1, 64, 0, st>>>(ws, loss, B);
  MLA_CHECK_LAUNCH("sum_to_scalar_kernel");
  return MLA_OK;
}

// nn.Linear backward from an arbitrary d logits (autograd hands it over): dX = scale * dl W (wave per sample),
// dW = scale * dl^T X, db = scale * colsum(dl).  scale = 1/world under data parallel.
__global__ __launch_bounds__(256) void head_dx_kernel(const float* __restrict__ dlogits, const float* __restrict__ W,
                                                       float* __restrict__ dX, int B, int D, int C, float scale) {
  const int d = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
  if (d >= D) return;
  float a = 0.f;
  for (int c = 0; c < C; ++c) a += dlogits[(size_t)row * C + c] * W[(size_t)c * D + d];
  dX[(size_t)row * D + d] = a * scale;
}
__global__ __launch_bounds__(256) void head_dw_kernel(const float* __restrict__ X, const float* __restrict__ dlogits,
                                                       float* __restrict__ dW, float* __restrict__ db, int B, int D, int C,
                                                       float scale) {
  const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d < D) {
    float a = 0.f;
    for (int r = 0; r < B; ++r) a += dlogits[(size_t)r * C + c] * X[(size_t)r * D + d];
    dW[(size_t)c * D + d] = a * scale;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float a = 0.f;
    for (int r = threadIdx.x; r < B; r += 64) a += dlogits[(size_t)r * C + c];
    a = wave_sum(a);
    if (threadIdx.x == 0) db[c] = a * scale;
  }
}
extern "C" int mla_head_bwd(const float* X, const float* W, const float* dlogits, float* dW, float* db, float* dX, int B, int D,
                            int C, float scale, void* stream) {
  MLA_REQUIRE(X && W && dlogits && dW && db && dX && B > 0 && D > 0 && C > 0, "mla_head_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  head_dx_kernel<<<dim3(cdiv(D, 256), B), 256, 0, st>>>(dlogits, W, dX, B, D, C, scale);
  MLA_CHECK_LAUNCH("head_dx_kernel");
  head_dw_kernel<<<dim3(cdiv(D, 256), C), 256, 0, st>>>(X, dlogits, dW, db, B, D, C, scale);
  MLA_CHECK_LAUNCH("head_dw_kernel");
  return MLA_OK;
}

// x[i] *= *scalar (the scalar lives on the device: the gradient autograd passes into a loss node)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, const float* __restrict__ s, size_t n) {
  const float f = *s;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= f;
}
extern "C" int mla_scale_by_device_scalar(float* x, const float* scalar, size_t n, void* stream) {
  MLA_REQUIRE(x && scalar, "mla_scale_by_device_scalar: null pointer");
  if (n == 0) return MLA_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  scale_dev_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, scalar, n);
  MLA_CHECK_LAUNCH("scale_dev_kernel");
  return MLA_OK;
}

// ---- column sum ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, float* __restrict__ r, int B, int D, float scale) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float a = 0.f;
  for (int i = 0; i < B; ++i) a += X[(size_t)i * D + d];
  r[d] = a * scale;
}

extern "C" int mla_colsum(const float* X, float* r, int B, int D, float scale, void* stream) {
  MLA_REQUIRE(X && r && B > 0 && D > 0, "mla_colsum: bad argument");
  colsum_kernel<<<cdiv(D, 256), 256, 0, (hipStream_t)stream>>>(X, r, B, D, scale);
  MLA_CHECK_LAUNCH("colsum_kernel");
  return MLA_OK;
}

// ---- GS projection: three row-parallel phases (wave per row of Pl) ---------------------------------
// A: k = Pl r^T
__global__ __launch_bounds__(256) void gs_k_kernel(const float* __restrict__ Pl, const float* __restrict__ r,
                                                    float* __restrict__ k, int D) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= D) return;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += Pl[(size_t)i * D + j] * r[j];
  s = wave_sum(s);
  if (lane == 0) k[i] = s;
}
// B: Pl[i][j] -= k_i k_j / (alpha + k_i r_j)   (element-wise DxD denominator, utils/utils.py:36); row sums of squares
__global__ __launch_bounds__(256) void gs_update_kernel(float* __restrict__ Pl, const float* __restrict__ r,
                                                         const float* __restrict__ k, float* __restrict__ rowsq, int D,
                                                         float alpha) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= D) return;
  const float ki = k[i];
  float q = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float v = Pl[(size_t)i * D + j] - (ki * k[j]) / (alpha + ki * r[j]);
    Pl[(size_t)i * D + j] = v;
    q += v * v;
  }
  q = wave_sum(q);
  if (lane == 0) rowsq[i] = q;
}
// C: Pl /= ||Pl||_F ; Gout[c][i] = sum_j G[c][j] Pl[i][j]
// One workgroup (4 waves) per row of Pl: every wave forms the norm (same order: bit-identical), wave w normalises the columns
// j = 64 (w + 4 m) + lane and, after a barrier, projects the classes c = w, w+4, ... (each class by one wave in the lane / stride
// order of the one-wave-per-row form).  At C = 101 that form was a chain of 101 dependent reductions per row: 300 us.
__global__ __launch_bounds__(256) void gs_finish_kernel(float* __restrict__ Pl, const float* __restrict__ rowsq,
                                                         const float* __restrict__ G, float* __restrict__ Gout, int D, int C) {
  const int i = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double t = 0.0;
  for (int j = lane; j < D; j += 64) t += (double)rowsq[j];
  t = wave_sum_d(t);
  const float nrm = (float)sqrt(t);
  for (int j = wave * 64 + lane; j < D; j += 256) Pl[(size_t)i * D + j] = Pl[(size_t)i * D + j] / nrm;
  __syncthreads();                       // the row is complete (and visible to the workgroup) before it is read back
  for (int c = wave; c < C; c += 4) {
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += G[(size_t)c * D + j] * Pl[(size_t)i * D + j];
    s = wave_sum(s);
    if (lane == 0) Gout[(size_t)c * D + i] = s;
  }
}

extern "C" size_t mla_gs_ws_elems(int D, int C) { return (size_t)2 * D + (size_t)C * D; }

extern "C" int mla_gs_project(float* Pl, const float* r, float* G, int D, int C, float alpha, float* ws, void* stream) {
  MLA_REQUIRE(Pl && r && G && ws && D > 0 && C > 0, "mla_gs_project: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float *k = ws, *rowsq = ws + D, *Gtmp = ws + 2 * (size_t)D;
  gs_k_kernel<<<cdiv(D, 4), 256, 0, st>>>(Pl, r, k, D);
  MLA_CHECK_LAUNCH("gs_k_kernel");
  gs_update_kernel<<<cdiv(D, 4), 256, 0, st>>>(Pl, r, k, rowsq, D, alpha);
  MLA_CHECK_LAUNCH("gs_update_kernel");
  gs_finish_kernel<<<D, 256, 0, st>>>(Pl, rowsq, G, Gtmp, D, C);
  MLA_CHECK_LAUNCH("gs_finish_kernel");
  if (hipMemcpyAsync(G, Gtmp, (size_t)C * D * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    mla_set_error("mla_gs_project: hipMemcpyAsync failed");
    return MLA_ERR_LAUNCH;
  }
  return MLA_OK;
}

// ---- SGD -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                   size_t n, float lr, float momentum, float wd, int first) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 d = pv * wd;
    if (g) d += reinterpret_cast<const f32x4*>(g)[i];
    f32x4 b = first ? d : reinterpret_cast<f32x4*>(buf)[i] * momentum + d;
    reinterpret_cast<f32x4*>(buf)[i] = b;
    reinterpret_cast<f32x4*>(p)[i] = pv - b * lr;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    const float pv = p[i];
    const float d = (g ? g[i] : 0.f) + wd * pv;
    const float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}

extern "C" int mla_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int first,
                            void* stream) {
  MLA_REQUIRE(p && buf, "mla_sgd_step: null pointer");
  if (n == 0) return MLA_OK;
  MLA_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)buf % 16 == 0) && (!g || (uintptr_t)g % 16 == 0),
              "mla_sgd_step: buffers must be 16-byte aligned");
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  sgd_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, buf, n, lr, momentum, wd, first);
  MLA_CHECK_LAUNCH("sgd_kernel");
  return MLA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Evaluation path (main.py:486-679 `valid`, gs_flag branch 622-651): per-modality logits from the shared head,
// fixed or entropy-gated fusion (main.py:65-106; SURVEY Q9: the "entropy" is taken over softmax(dim=0), i.e. over
// the BATCH axis, and summed over the whole tensor -> one scalar weight per modality per batch), arg-max and the
// per-class counters of main.py:659-676 -- on the device, one workgroup, instead of a per-sample .cpu() loop.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ logits, int B,
                                                           int D, int C) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= B) return;
  const float* x = X + (size_t)row * D;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[d] * W[(size_t)c * D + d];
    s = wave_sum(s);
    if (lane == 0) logits[(size_t)row * C + c] = s + bias[c];
  }
}

extern "C" int mla_head_logits(const float* X, const float* W, const float* b, float* logits, int B, int D, int C, void* stream) {
  MLA_REQUIRE(X && W && b && logits && B > 0 && D > 0 && C > 0, "mla_head_logits: bad argument");
  head_logits_kernel<<<cdiv(B, 4), 256, 0, (hipStream_t)stream>>>(X, W, b, logits, B, D, C);
  MLA_CHECK_LAUNCH("head_logits_kernel");
  return MLA_OK;
}

#define EVAL_MAXM 3
struct EvalFuseArgs {
  const float* out[EVAL_MAXM];   // logits per modality (B, C)
  float alpha[EVAL_MAXM];        // fixed fusion weights (used when dynamic == 0)
  int M, B, C, dynamic;
};

// counts layout (int32, accumulated across calls): [0..C) num per class, then for k = 0 (fused), 1..M (modalities):
// [C*(1+k) .. C*(2+k)) correct per class.  weights_out[M]: the fusion weights used for this batch.
__global__ __launch_bounds__(256) void eval_fuse_kernel(const EvalFuseArgs a, const int64_t* __restrict__ labels,
                                                         int* __restrict__ counts, float* __restrict__ weights_out) {
  __shared__ float ent[EVAL_MAXM];
  __shared__ float wgt[EVAL_MAXM];
  __shared__ float colpart[256];
  const int tid = threadIdx.x;
  if (a.dynamic) {
    // entropy_m = - sum_{rows, cols} p log p, p = softmax over ROWS (dim=0) of out_m   (main.py:65-70)
    for (int m = 0; m < a.M; ++m) {
      float acc = 0.f;
      for (int c = tid; c < a.C; c += 256) {           // one column per thread
        float mx = -INFINITY;
        for (int r = 0; r < a.B; ++r) mx = fmaxf(mx, a.out[m][(size_t)r * a.C + c]);
        float s = 0.f;
        for (int r = 0; r < a.B; ++r) s += expf(a.out[m][(size_t)r * a.C + c] - mx);
        const float ls = logf(s);
        for (int r = 0; r < a.B; ++r) {
          const float lp = a.out[m][(size_t)r * a.C + c] - mx - ls;   // log softmax
          acc -= expf(lp) * lp;
        }
      }
      colpart[tid] = acc;
      __syncthreads();
      if (tid == 0) {
        float e = 0.f;
        for (int k = 0; k < 256; ++k) e += colpart[k];
        ent[m] = e;
      }
      __syncthreads();
    }
    if (tid == 0) {                                    // main.py:72-106
      float mx = ent[0];
      for (int m = 1; m < a.M; ++m) mx = fmaxf(mx, ent[m]);
      float sum = 0.f;
      for (int m = 0; m < a.M; ++m) {
        wgt[m] = expf(mx - ent[m]);
        sum += wgt[m];
      }
      for (int m = 0; m < a.M; ++m) wgt[m] /= sum;
    }
  } else if (tid < a.M) {
    wgt[tid] = a.alpha[tid];                           // main.py:647-651
  }
  __syncthreads();
  if (tid < a.M && weights_out) weights_out[tid] = wgt[tid];
  for (int r = tid; r < a.B; r += 256) {               // one sample per thread: arg-max (first maximum, like np.argmax)
    const long lab_raw = (long)labels[r];
    if (lab_raw < 0 || lab_raw >= a.C) {               // never index the counters with a bad label; poison the weights instead
      if (weights_out) weights_out[0] = NAN;
      continue;
    }
    const int lab = (int)lab_raw;
    atomicAdd(&counts[lab], 1);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < a.C; ++c) {
      float v = 0.f;
      for (int m = 0; m < a.M; ++m) v += wgt[m] * a.out[m][(size_t)r * a.C + c];
      if (v > best) { best = v; arg = c; }
    }
    if (arg == lab) atomicAdd(&counts[a.C * 1 + lab], 1);
    for (int m = 0; m < a.M; ++m) {
      best = -INFINITY;
      arg = 0;
      for (int c = 0; c < a.C; ++c) {
        const float v = a.out[m][(size_t)r * a.C + c];
        if (v > best) { best = v; arg = c; }
      }
      if (arg == lab) atomicAdd(&counts[a.C * (2 + m) + lab], 1);
    }
  }
}

extern "C" int mla_eval_fuse(const float* out0, const float* out1, const float* out2, const int64_t* labels, int* counts,
                             float* weights_out, int M, int B, int C, int dynamic, float alpha0, float alpha1, float alpha2,
                             void* stream) {
  MLA_REQUIRE(out0 && out1 && labels && counts && (M == 2 || (M == 3 && out2)) && B > 0 && C > 0, "mla_eval_fuse: bad argument");
  EvalFuseArgs a;
  a.out[0] = out0; a.out[1] = out1; a.out[2] = out2;
  a.alpha[0] = alpha0; a.alpha[1] = alpha1; a.alpha[2] = alpha2;
  a.M = M; a.B = B; a.C = C; a.dynamic = dynamic;
  eval_fuse_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, labels, counts, weights_out);
  MLA_CHECK_LAUNCH("eval_fuse_kernel");
  return MLA_OK;
}

// invstd[i] = 1 / sqrt(var[i] + eps): eval-mode BatchNorm uses the running statistics (one launch per encoder over
// its flat running_var buffer)
__global__ __launch_bounds__(256) void invstd_kernel(const float* __restrict__ var, float* __restrict__ invstd, int n, float eps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) invstd[i] = 1.0f / sqrtf(var[i] + eps);
}
extern "C" int mla_bn_invstd(const float* var, float* invstd, int n, float eps, void* stream) {
  MLA_REQUIRE(var && invstd && n > 0, "mla_bn_invstd: bad argument");
  invstd_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(var, invstd, n, eps);
  MLA_CHECK_LAUNCH("invstd_kernel");
  return MLA_OK;
}
