/*
 * mla_hip.h -- C ABI of libmla_hip.so: the MI355X (gfx950) kernels of the MLA
 * alternating-unimodal training step (reference: main.py:419-476 and the modules it calls).
 *
 * The reference has no FFI: its "operator interface" for this path is the set of implicit ATen ops
 * that nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d / F.adaptive_avg_pool / nn.Linear /
 * nn.CrossEntropyLoss / GSPlugin.before_update / torch.optim.SGD trigger.  Each entry point below
 * replaces one of those op groups and cites the reference line that triggers it.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; nothing is allocated, freed or synchronised;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); kernels are enqueued on it;
 *   - activations are NHWC fp32 ("pixel-major": [N][H][W][C]); conv weights are HWIO fp32
 *     ([KH][KW][Cin][Cout]); the Python boundary converts from/to the reference's NCHW / OIHW;
 *   - return 0 on success, a negative mla_status on error (never throws, never aborts).
 */
#ifndef MLA_HIP_H
#define MLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum mla_status {
  MLA_OK = 0,
  MLA_ERR_INVALID_ARG = -1,   /* shape/pointer the kernels do not support */
  MLA_ERR_WORKSPACE = -2,     /* workspace too small */
  MLA_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after launch */
};

/* Library/ABI version and a human-readable description of the last error on this thread. */
int mla_abi_version(void);
const char* mla_last_error(void);

/* ---- layout (boundary) ---------------------------------------------------------------------- */
/* (B,C,T,H,W) video -> (B*T,H,W,C) frames.  Replaces backbone.py:144-147 permute+contiguous+view. */
int mla_video_to_nhwc(const float* src, float* dst, int B, int C, int T, int H, int W, void* stream);
/* generic NCHW <-> NHWC (tests / state import) */
int mla_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, void* stream);
int mla_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream);

/* ---- convolution: nn.Conv2d (backbone.py:4-12, 79-83, 28, 31, 127) --------------------------- */
/* Implicit-GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
 * bn_partial (nullable): if given, the epilogue also writes per-M-tile column sums and sums of
 * squares of y, layout [tiles][2][Cout]; *bn_tiles receives `tiles`.  Feed to mla_bn_finalize. */
/* measurement hook: force the fp32 kernels' tile 0..3 (128x128, 256x64, 64x64, 128x64) where Cout allows; -1 = automatic */
int mla_conv2d_f32_cfg(int cfg);
size_t mla_conv2d_fwd_partial_elems(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mla_conv2d_fwd(const float* x, const float* w_hwio, float* y,
                   int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   float* bn_partial, int* bn_tiles, void* stream);

/* Input gradient (autograd of the above).  dx = dgrad(dy) [+ residual] [* (relu_src > 0)].
 * `residual` may alias `dx` (accumulate).  wt_ws: Cin*Cout*KH*KW floats of scratch (per-tap
 * transposed weights).  (H,W) are the INPUT dims of the forward conv. */
int mla_conv2d_dgrad(const float* dy, const float* w_hwio, float* dx,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     const float* residual, const float* relu_src, float* wt_ws, void* stream);

/* Weight gradient: dw_hwio[kh][kw][ci][co] = sum_pixels x * dy.  Deterministic split-K:
 * partial slabs in `ws` (mla_conv2d_wgrad_ws_bytes) followed by an ordered reduce. */
size_t mla_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mla_conv2d_wgrad(const float* x, const float* dy, float* dw_hwio,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- convolution, split-bf16 arithmetic (same nn.Conv2d sites as above) -------------------------
 * The same gather-GEMM with every fp32 operand split exactly into three bf16 terms and the six products
 * a_i*b_j (i+j <= 2) accumulated in fp32 on v_mfma_f32_32x32x16_bf16: fp32 in, fp32 out, error against
 * the exact sum of the same order as the fp32-MFMA entry points (dropped terms <= 2^-23 |a*b|), at 6/16 of
 * their MFMA cycles.  Weights are pre-split by mla_conv2d_wsplit into mla_conv2d_wsplit_bytes of scratch:
 * transposed = 1 for mla_conv2d_fwd_split, 0 for mla_conv2d_dgrad_split.  Cin, Cout multiples of 64
 * (the stem stays on mla_conv2d_fwd).  Other arguments as in mla_conv2d_fwd / mla_conv2d_dgrad.
 * mla_conv2d_split_terms(t) (t = 3, 6, 8; anything else only queries) selects the product set for
 * measurements; 6 is the default and the only set the parity tests bless. */
size_t mla_conv2d_wsplit_bytes(int Cin, int Cout, int KH, int KW);
int mla_conv2d_wsplit(const float* w_hwio, void* wsplit, int Cin, int Cout, int KH, int KW, int transposed, void* stream);
/* Batched form (one launch for all convs of a flat parameter buffer).  desc: n <= 4096 rows of 8 ints in DEVICE memory,
 * {w_off (floats from params), out_off (16-bit elements from wsplit), taps, Cin, Cout, transposed, first_block, 0} with
 * first_block the running sum of taps*ceil(Cin/32)*ceil(Cout/32); total_blocks = that sum over all rows. */
int mla_conv2d_wsplit_batch(const float* params, void* wsplit, const int* desc, int n, int total_blocks, void* stream);
int mla_conv2d_fwd_split(const float* x, const void* wsplit_t, float* y,
                         int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                         float* bn_partial, int* bn_tiles, void* stream);
int mla_conv2d_dgrad_split(const float* dy, const void* wsplit, float* dx,
                           int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           const float* residual, const float* relu_src, void* stream);

/* Input gradient that also forms the reduction pass of the BatchNorm backward(s) consuming dx: for each request q (at most 2:
 * the BatchNorm before this convolution in the forward pass, and a downsample BatchNorm beside it) the epilogue writes the
 * per-tile column sums  sum dx  and  sum dx * (x_q - mean_q) * invstd_q  to partial_q[tile][2][Cin] (floats; size
 * mla_conv2d_dgrad_bn_partial_elems; *bn_tiles = number of tiles written), which mla_bn_bwd_from_partial then finalizes and
 * applies.  Saves re-reading dx and x in a separate reduction pass.  nreq = 0: identical to mla_conv2d_dgrad[_split]. */
typedef struct mla_bn_reduce_req {
  const float* x;        /* the BatchNorm's input (conv output), (N,H,W,Cin) like dx */
  const float* mean;     /* its saved batch mean / inverse standard deviation, (Cin) */
  const float* invstd;
  float* partial;        /* out */
} mla_bn_reduce_req;
size_t mla_conv2d_dgrad_bn_partial_elems(int N, int H, int W, int Cin);
int mla_conv2d_dgrad_bn(const float* dy, const float* w_hwio, float* dx, int N, int H, int W, int Cin, int Cout,
                        int KH, int KW, int stride, int pad, const float* residual, const float* relu_src, float* wt_ws,
                        const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, void* stream);
int mla_conv2d_dgrad_split_bn(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout,
                              int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                              const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, void* stream);
/* ... restricted to some output parity classes of a strided convolution (bit py * stride + px of class_mask; stride 1: bit 0),
 * with `residual` added only in the classes of residual_mask.  Lets the 1x1 / stride-2 downsample input gradient
 * (backbone.py:126-129) write just the one class it reaches, and the 3x3 / stride-2 conv1 beside it (backbone.py:28) accumulate
 * onto it there, instead of a full read-modify-write pass over dx in four launches.  (0xF, 0xF) = the plain entry points. */
int mla_conv2d_dgrad_classes(const float* dy, const float* w_hwio, float* dx, int N, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, const float* residual, const float* relu_src, float* wt_ws,
                             const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, int class_mask, int residual_mask,
                             void* stream);
int mla_conv2d_dgrad_split_classes(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout,
                                   int KH, int KW, int stride, int pad, const float* residual, const float* relu_src,
                                   const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, int class_mask, int residual_mask,
                                   void* stream);
/* weight gradient on the same arithmetic (both operands split in the kernel); workspace and reduce as mla_conv2d_wgrad */
size_t mla_conv2d_wgrad_split_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mla_conv2d_wgrad_split(const float* x, const float* dy, float* dw_hwio,
                           int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           void* ws, size_t ws_bytes, void* stream);
/* measurement hook: 0 = per-tap weight-gradient kernel for every layer, 1 (default) = the persistent all-taps kernel
 * (transposing LDS reads) for the 64 -> 64 channel 3x3 / stride 1 convolutions; other values: query.  Returns the setting. */
int mla_conv2d_wgrad_tr(int on);
/* measurement / test hook: 0 = per-tap gather-GEMM for every layer, 1 (default; $MLA_CONV_PATCH overrides) = LDS-resident
 * input patch (conv_patch_split.hip) for the 3x3 / stride 1 / pad 1 forward and input-gradient launches whose grid fills the chip,
 * 2 = for all of them; other values: query.  Returns the setting. */
int mla_conv2d_patch(int on);
/* BatchNorm folded into the operands of the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolutions (split arithmetic; conv2 of the layer1
 * BasicBlocks, models/backbone.py:38-46: conv1 -> bn1 -> relu -> conv2).  The consumers of a = relu(bn1(y1)) -- conv2's forward, conv2's weight
 * gradient and the ReLU mask of conv2's input gradient -- form it from y1 with the expression of mla_bn_apply, bit for bit, so `a` is
 * never written or read.  in_* / mask_*: per-channel BatchNorm parameters (training: batch statistics; evaluation: running statistics).
 * mla_conv2d_bnfold_supported: 1 where these entry points apply (they fail with MLA_ERR_INVALID_ARG elsewhere).
 *   mla_conv2d_fwd_split_bnin     = mla_conv2d_fwd_split   over relu(bn(x))
 *   mla_conv2d_wgrad_split_bnin   = mla_conv2d_wgrad_split over relu(bn(x))
 *   mla_conv2d_dgrad_split_bnmask = mla_conv2d_dgrad_split_bn with relu_src := relu(bn(reqs[0].x)) (gamma / beta given; mean / invstd: the request's) */
int mla_conv2d_bnfold_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mla_conv2d_fwd_split_bnin(const float* x, const void* wsplit_t, float* y, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                              int stride, int pad, const float* in_mean, const float* in_invstd, const float* in_gamma, const float* in_beta,
                              float* bn_partial, int* bn_tiles, void* stream);
int mla_conv2d_wgrad_split_bnin(const float* x, const float* dy, float* dw_hwio, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                int stride, int pad, const float* in_mean, const float* in_invstd, const float* in_gamma,
                                const float* in_beta, void* ws, size_t ws_bytes, void* stream);
int mla_conv2d_dgrad_split_bnmask(const float* dy, const void* wsplit, float* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                  int stride, int pad, const mla_bn_reduce_req* reqs, int nreq, int* bn_tiles, const float* mask_gamma,
                                  const float* mask_beta, void* stream);
/* measurement / test hook: 0 = one launch per output parity class of a stride-2 input gradient (split arithmetic), 1 (default;
 * $MLA_DGRAD_MERGE overrides) = all classes in one launch, longest K first; other values: query.  Returns the setting. */
int mla_conv2d_dgrad_merge(int on);
/* measurement / test hook: 0 = every split forward / input-gradient gather-GEMM is one launch, 1 (default; $MLA_CONV_TWO_PHASE overrides) =
 * row counts between whole rounds of the 256 CUs run whole rounds of a big tile and one launch of the best tile for the remaining rows;
 * other values: query.  Results do not depend on the setting (same K order per output element). */
int mla_conv2d_two_phase(int on);
int mla_conv2d_split_terms(int terms);
/* The ResNet stem (backbone.py:79-83, 149: 7x7, stride 2, pad 3, 1 or 3 -> 64 channels) on the split arithmetic, as persistent
 * patch-loader kernels: a workgroup keeps the weights (forward: three bf16 planes) resident in LDS, loads the 37 x 37 x Cin
 * input patch of a 16 x 16 output tile once and serves all 49 taps from it.  `w` / `dw` are the plain fp32 HWIO weights (the
 * forward splits them in its prologue).  mla_conv2d_stem_supported() != 0 for the shapes these entries accept; everything else
 * (and conv_math = f32) runs on mla_conv2d_fwd / mla_conv2d_wgrad.  bn_partial as in mla_conv2d_fwd, fp64 partials, one row per
 * workgroup (>= mla_conv2d_stem_fwd_partial_elems() floats); *bn_tiles receives the row count for mla_bn_finalize. */
int mla_conv2d_stem_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
/* measurement hook: force 4 or 8 waves per stem-forward workgroup, 0 = automatic (4 for Cin = 1, 8 for Cin = 3); other values: query */
int mla_conv2d_stem_waves(int waves);
size_t mla_conv2d_stem_fwd_partial_elems(void);
int mla_conv2d_stem_fwd_split(const float* x, const float* w_hwio, float* y,
                              int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                              float* bn_partial, int* bn_tiles, void* stream);
size_t mla_conv2d_stem_wgrad_split_ws_bytes(int Cin);
int mla_conv2d_stem_wgrad_split(const float* x, const float* dy, float* dw_hwio,
                                int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                void* ws, size_t ws_bytes, void* stream);
/* measurement hook: force tile 0..4 (256x128, 128x128, 128x64, 64x64, 256x64) where Cout allows; -1 = automatic */
int mla_conv2d_split_cfg(int cfg);

/* ---- BatchNorm2d, training mode (backbone.py:29, 32, 86, 128) -------------------------------- */
/* x is [M][C] (M = N*H*W).  Statistics: either mla_bn_stats (reads x) or conv-fused partials. */
size_t mla_bn_partial_scratch_elems(int C);   /* reduction scratch tail included in every *_partial_elems / *_ws_elems below */
size_t mla_bn_stats_partial_elems(int M, int C);
int mla_bn_stats_partial(const float* x, int M, int C, float* partial, int* tiles, void* stream);
/* Reduce partials in fp64 -> mean, invstd (biased var, eps), running stats (unbiased var, momentum). */
int mla_bn_finalize(const float* partial, int tiles, int M, int C, float eps, float momentum,
                    float* mean, float* invstd, float* running_mean, float* running_var, void* stream);
/* out = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ) */
int mla_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, const float* residual, float* out, int M, int C, int relu, void* stream);
/* Backward.  g = dout * (relu_out > 0) if relu_out else dout;
 * dgamma = sum g*xhat, dbeta = sum g, dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).
 * g_out (nullable) receives g (the gradient of the residual branch).  ws: mla_bn_bwd_ws_elems floats. */
size_t mla_bn_bwd_ws_elems(int M, int C);
int mla_bn_bwd(const float* dout, const float* relu_out, const float* x, const float* mean,
               const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
               float* g_out, float* ws, int M, int C, void* stream);

/* BatchNorm backward whose reduction pass was done by the producer of dout (mla_conv2d_dgrad[_split]_bn): finalize the
 * `tiles` per-tile sums of `partial` into dgamma / dbeta, then dx = gamma * invstd * (dout - dbeta/M - xhat * dgamma/M). */
int mla_bn_bwd_from_partial(const float* dout, const float* x, const float* mean, const float* invstd, const float* gamma,
                            float* dx, float* dgamma, float* dbeta, float* partial, int tiles, int M, int C, void* stream);
/* Stem conv1 -> bn1 -> relu -> maxpool (backbone.py:149-152) without materialising the ReLU output: the max-pool applies
 * BN + ReLU to y (NHWC conv output) on the fly; out (N,OH,OW,C), idx = window position 0..8 of the first maximum. */
int mla_bn_relu_maxpool_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float* out, uint8_t* idx, int N, int H, int W, int C, void* stream);
/* ... and its backward: BatchNorm backward whose upstream gradient is gathered from the POOLED gradient dpool (N,OH,OW,C)
 * through idx and masked by bn(y) > 0; dy (N,H,W,C) = gradient w.r.t. y, dgamma / dbeta written; ws >= mla_bn_bwd_ws_elems. */
int mla_bn_bwd_pooled(const float* dpool, const uint8_t* idx, const float* y, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta, float* ws,
                      int N, int H, int W, int C, void* stream);

/* ---- pooling ---------------------------------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1) (backbone.py:88,152); idx = window position 0..8 of the first maximum. */
int mla_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
/* dx = scatter(dy) [* (relu_src > 0)] */
int mla_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, const float* relu_src, float* dx,
                         int N, int H, int W, int C, void* stream);
/* adaptive_avg_pool2d/3d(.,1)+flatten (basic_model.py:56-65): x [NB][P][C] -> y [NB][C] */
int mla_avgpool_fwd(const float* x, float* y, int NB, int P, int C, void* stream);
/* dx[nb][p][c] = dy[nb][c]/P [* (relu_src > 0)]  (relu_src: the pooled tensor, output of the last ReLU) */
int mla_avgpool_bwd(const float* dy, const float* relu_src, float* dx, int NB, int P, int C, void* stream);

/* ---- shared head + cross entropy (fusion_modules.py:19; main.py:130, 432-435, 444-447) ------- */
/* logits = X W^T + b; loss = -sum_i log softmax(logits_i)[label_i] * inv_batch (rank-local part);
 * dlogits = (softmax - onehot) * inv_batch; dW = dlogits^T X; db = sum dlogits; dX = dlogits W.
 * Two launches (wave per sample, then the BxD contraction).  X [B][D], W [C][D], labels int64 [B];
 * C <= 128.  ws: mla_head_ws_elems(B, C) floats. */
size_t mla_head_ws_elems(int B, int C);
int mla_head_ce_fwd_bwd(const float* X, const float* W, const float* b, const int64_t* labels,
                        float* logits, float* loss, float* dW, float* db, float* dX, float* ws,
                        int B, int D, int C, float inv_batch, void* stream);

/* The same head where autograd splits it, for the nn.Module / autograd.Function protocol (main.py:432-435 executed
 * literally: `out = fc_out(a)`; `loss = criterion(out, label)`; `loss.backward()`):
 *   mla_ce_fwd_bwd   nn.CrossEntropyLoss() (main.py:130), mean reduction: loss and dlogits = (softmax - onehot) * inv_batch
 *                    in one pass; ws: B floats.  A label outside [0, C) yields a NaN loss (the reference asserts).
 *   mla_head_bwd     autograd of nn.Linear (fusion_modules.py:19) for a given dlogits: dW = s dlogits^T X, db = s sum dlogits,
 *                    dX = s dlogits W  (s = 1/world under data parallel, else 1).
 *   mla_scale_by_device_scalar   x *= *scalar, the scalar read on the device (gradient flowing into a loss node). */
int mla_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* ws,
                   int B, int C, float inv_batch, void* stream);
int mla_head_bwd(const float* X, const float* W, const float* dlogits, float* dW, float* db, float* dX,
                 int B, int D, int C, float scale, void* stream);
int mla_scale_by_device_scalar(float* x, const float* scalar, size_t n, void* stream);

/* ---- GSPlugin.before_update (utils/utils.py:24-41) ------------------------------------------- */
/* r[j] = scale * sum_i X[i][j]   (column mean with scale = 1/B; rank-local column sum otherwise) */
int mla_colsum(const float* X, float* r, int B, int D, float scale, void* stream);
/* k = Pl r^T; Pl <- Pl - (k k^T) ./ (alpha + k r) ; Pl <- Pl/||Pl||_F ; G <- G Pl^T.
 * Pl [D][D] and G [C][D] are updated in place.  ws: mla_gs_ws_elems(D, C) floats. */
size_t mla_gs_ws_elems(int D, int C);
int mla_gs_project(float* Pl, const float* r, float* G, int D, int C, float alpha, float* ws, void* stream);

/* ---- OGM / OGM-GE gradient modulation (main.py:312-410, --modulation OGM | OGM_GE) ---------------------------------
 * mla_ogm_coeff: score_m = sum_i softmax(out_m)[i][label_i] (row order), ratios and the coefficient of every modality
 *   (main.py:373-384 for M = 2: {audio, visual}; 314-337 for M = 3: {audio, visual, text}) -> coeff[M] on the device;
 *   info (nullable, 6 floats): scores at [0..M), ratios at [3..3+M).
 * mla_ogm_modulate: for each segment (= one 4-D conv gradient; seg_desc int64 [n_seg][2] = {offset, numel} into `grad`,
 *   device memory): grad = grad * *coeff, and with ge != 0 additionally + N(0, std + 1e-8), std = unbiased standard
 *   deviation of the UNSCALED segment (main.py:397-400).  first_chunk (int [n_seg + 1], device): prefix sum of
 *   ceil(numel / mla_ogm_chunk_elems()) per segment; total_chunks = first_chunk[n_seg].  Noise: Philox4x32-10 keyed by
 *   `seed`, stream (step, segment), counter = element / 4 -> reproducible, independent of the launch geometry. */
int mla_ogm_coeff(const float* out0, const float* out1, const float* out2, const int64_t* labels, int M, int B, int C,
                  float alpha, float* coeff, float* info, void* stream);
int mla_ogm_chunk_elems(void);
size_t mla_ogm_ws_bytes(int total_chunks, int n_seg);
int mla_ogm_modulate(float* grad, const int64_t* seg_desc, const int* first_chunk, int n_seg, int total_chunks,
                     const float* coeff, int ge, uint64_t seed, uint64_t step, void* ws, size_t ws_bytes, void* stream);

/* ---- transformer encoders (M3AE, models/m3ae.py; CAV-MAE, models/cav_mae.py) ------------------- */
/* nn.Linear on the gather-GEMM: y[g][y_off+r][:] = x[g][x_off+r][:] . w_kn (+ bias) (+ residual);
 * rows are `groups` x `rows` tokens taken at an offset inside groups of *_group_rows tokens (no copies for
 * "tokens 1..256 of each 257-token sequence").  w_kn is [K][N] = nn.Linear.weight^T.  y_gelu (nullable)
 * additionally receives gelu(y) (erf form, F.gelu, m3ae.py:77).  K, N multiples of 64. */
int mla_linear_fwd(const float* x, const float* w_kn, const float* bias, const float* residual, float* y,
                   float* y_gelu, int groups, int rows, int x_group_rows, int x_off, int y_group_rows, int y_off,
                   int K, int N, void* stream);
/* dx = dy . w_kn^T (+ residual) (* gelu'(gelu_src)).  wt_ws: K*N floats. */
int mla_linear_dgrad(const float* dy, const float* w_kn, float* dx, const float* residual, const float* gelu_src,
                     float* wt_ws, int groups, int rows, int dy_group_rows, int dy_off, int dx_group_rows, int dx_off,
                     int K, int N, void* stream);
/* dw_kn[K][N] = sum_rows x^T dy (dy dense [groups*rows][N]); deterministic split-K like mla_conv2d_wgrad. */
size_t mla_linear_wgrad_ws_bytes(int M, int K, int N);
int mla_linear_wgrad(const float* x, const float* dy, float* dw_kn, int groups, int rows, int x_group_rows, int x_off,
                     int K, int N, void* ws, size_t ws_bytes, void* stream);
/* The same three on the split-bf16 arithmetic (see mla_conv2d_*_split): weights pre-split with
 * mla_conv2d_wsplit(w_kn, out, Cin = K, Cout = N, 1, 1, transposed, stream), transposed = 1 for the forward image
 * (wsplit_t), 0 for the input-gradient image (wsplit). */
int mla_linear_fwd_split(const float* x, const void* wsplit_t, const float* bias, const float* residual, float* y,
                         float* y_gelu, int groups, int rows, int x_group_rows, int x_off, int y_group_rows, int y_off,
                         int K, int N, void* stream);
int mla_linear_dgrad_split(const float* dy, const void* wsplit, float* dx, const float* residual, const float* gelu_src,
                           int groups, int rows, int dy_group_rows, int dy_off, int dx_group_rows, int dx_off,
                           int K, int N, void* stream);
size_t mla_linear_wgrad_split_ws_bytes(int M, int K, int N);
int mla_linear_wgrad_split(const float* x, const float* dy, float* dw_kn, int groups, int rows, int x_group_rows, int x_off,
                           int K, int N, void* ws, size_t ws_bytes, void* stream);
/* ... and the bias gradient dbias[N] = column sums of dy out of the same pass (dbias may be NULL); workspace as reported by
 * mla_linear_wgrad_split_ws_bytes (weight slabs + one bias row per split-K range). */
int mla_linear_wgrad_split_bias(const float* x, const float* dy, float* dw_kn, float* dbias, int groups, int rows,
                                int x_group_rows, int x_off, int K, int N, void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum_rows x[r][c]  (bias gradients).  ws: mla_colreduce_ws_elems(M, C) floats; C % 64 == 0. */
size_t mla_colreduce_ws_elems(int M, int C);
int mla_colsum_rows(const float* x, float* out, float* ws, int M, int C, void* stream);
/* nn.LayerNorm (m3ae.py:138,142,176) over rows of D (512/768/1024); saves mean and rstd per row. */
int mla_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                      int M, int D, float eps, void* stream);
/* dx = LN'(dy) (+ add), dw = sum dy*xhat, db = sum dy.  dx may alias dy or add.  ws: mla_colreduce_ws_elems. */
int mla_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                      const float* add, float* dx, float* dw, float* db, float* ws, int M, int D, void* stream);
/* Strided batched GEMM for attention (m3ae.py:109, 121): C[z] = alpha * A[z] . B[z], z = (batch, head);
 * strides in elements (>= 0): a = {batch, head, row i, k}, b = {batch, head, k, col j}, c = {batch, head, i, j}.
 * *_extent: elements addressable from each pointer; a descriptor that reaches beyond them returns MLA_ERR_INVALID_ARG
 * (the materialised attention path; the encoders use mla_attention_* by default). */
int mla_bgemm(const float* A, const float* B, float* C, int batches, int heads, int M, int N, int K,
              const long* a_strides, const long* b_strides, const long* c_strides,
              size_t a_extent, size_t b_extent, size_t c_extent, float alpha, void* stream);
/* In-place softmax over the last axis of S (B,H,n,n); pad_mask (B,n) float, > 0 -> score := -1e7 (m3ae.py:111-118). */
int mla_softmax_fwd(float* S, const float* pad_mask, int B, int H, int n, void* stream);
/* dS = P * (dP - rowsum(dP*P)), in place in dP. */
int mla_softmax_bwd(const float* P, float* dP, int B, int H, int n, void* stream);

/* Fused multi-head attention (models/m3ae.py:102-125; timm Attention behind cav_mae.py:93), exact fp32 MFMA, head dim 64.
 * qkv (B, n, 3, H, 64) as written by the fused qkv Linear; pad_mask (B, n) float or null (mask > 0: score := -1e7,
 * m3ae.py:111-117); o (B, n, H*64); lse (B, H, n) = log-sum-exp of every score row (saved for the backward).  The n x n
 * scores never reach HBM: online softmax forward, recomputation backward (two kernels: dQ owns query rows and also writes
 * dvec (B, H, n) = rowsum(d_o * o); dK / dV own key rows), no atomics -> bitwise reproducible.  dqkv has qkv's layout. */
int mla_attention_fwd(const float* qkv, const float* pad_mask, float* o, float* lse, int B, int H, int n, int hd, void* stream);
int mla_attention_bwd(const float* d_o, const float* qkv, const float* o, const float* lse, const float* pad_mask,
                      float* dqkv, float* dvec, int B, int H, int n, int hd, void* stream);
/* forward_representation token assembly (m3ae.py:342-366): x0[b][0] = cls; x0[b][1+i] = (table[ids[b][i]] if
 * table else x0[b][1+i]) + pos[i] + type.  x0 is (B, L+1, D).  cls == NULL (CAV-MAE, cav_mae.py:341-343): no
 * [cls] row, x0 is (B, L, D) and x0[b][i] += pos[i] + type.  V = rows of `table`; a token id outside [0, V) makes its row
 * NaN (nn.Embedding asserts there) and receives no gradient. */
int mla_tokens_assemble(float* x0, const float* table, const int64_t* ids, const float* pos, const float* type,
                        const float* cls, int B, int L, int D, int V, void* stream);
/* its gradients: dcls, dtype (needs colsum_all = column sum of dx0 over all B*(L+1) rows) and, for text,
 * dtable[ids] += dx0 rows (nn.Embedding backward, m3ae.py:306, 360; dtable pre-zeroed by the caller).  Deterministic since
 * ABI 3: ids are sorted on the device and the rows of one id are added in ascending token order (no float atomics), so
 * repeated runs agree bit for bit.  ws >= mla_tokens_assemble_bwd_ws_bytes (only read when dtable != NULL). */
size_t mla_tokens_assemble_bwd_ws_bytes(int B, int L, int D);
int mla_tokens_assemble_bwd(const float* dx0, const float* colsum_all, const int64_t* ids, float* dcls, float* dtype,
                            float* dtable, int B, int L, int D, int V, void* ws, size_t ws_bytes, void* stream);
/* einops 'b c (h p1) (w p2) -> b (h w) (c p1 p2)' (basic_model.py:184-186); also the im2col of CAV-MAE's
 * conv16x16/16 PatchEmbed (cav_mae.py:69-84).  transposed != 0: img is stored (B,C,W,H) (spectrogram (B,time,freq)
 * viewed as (B,1,freq,time), cav_mae.py:339-340). */
int mla_patchify(const float* img, float* out, int B, int C, int H, int W, int P, int transposed, void* stream);

/* ---- evaluation path (main.py:486-679 `valid`, gs_flag branch) --------------------------------- */
/* logits = X W^T + b only (main.py:636-639) */
int mla_head_logits(const float* X, const float* W, const float* b, float* logits, int B, int D, int C, void* stream);
/* Fusion + accuracy of one batch (main.py:640-676, 65-106): weights = fixed alphas or entropy gating
 * (entropy over softmax(dim=0) = the batch axis, summed over the tensor: one scalar per modality per batch, Q9);
 * fused = sum_m w_m out_m; arg-max (first maximum); int32 counters accumulated in `counts`:
 * [0,C) samples per class, [C,2C) fused-correct per class, [C(2+m), C(3+m)) modality-m-correct per class.
 * weights_out (nullable, M floats) receives the weights used.  M = 2 or 3. */
int mla_eval_fuse(const float* out0, const float* out1, const float* out2, const int64_t* labels, int* counts,
                  float* weights_out, int M, int B, int C, int dynamic, float alpha0, float alpha1, float alpha2,
                  void* stream);
/* invstd = 1/sqrt(var + eps) (eval-mode BatchNorm on running statistics) */
int mla_bn_invstd(const float* var, float* invstd, int n, float eps, void* stream);

/* ---- torch.optim.SGD(momentum, weight_decay) (main.py:749, 439, 451) ------------------------- */
/* d = g + wd*p; buf = first ? d : momentum*buf + d; p -= lr*buf.  g == NULL means zero gradient
 * (torch-1.8.1 zero_grad semantics).  One flat launch over n contiguous elements. */
int mla_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd,
                 int first, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLA_HIP_H */
