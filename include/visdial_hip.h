/* visdial_hip.h -- C ABI of libvisdial_hip.so (MI355X / gfx950).
 *
 * The reference (batra-mlp-lab/visdial) is Lua/Torch7; its only native boundary is the
 * LuaJIT-FFI call into THNN in model_utils/MaskSoftMax.lua:16-19,35-40.  Every arithmetic
 * op on its training hot path lives in un-vendored Lua rocks (nn, rnn, cunn).  This header is
 * the operator-level boundary that replaces those calls: each entry point names the reference
 * module call it stands in for (file:line under /root/reference).  A host (LuaJIT ffi.cdef,
 * Python ctypes, ...) composes them exactly as encoders/<name>.lua / decoders/<name>.lua compose nn
 * modules; see INTEGRATION.md for the Lua-side binding.
 *
 * Conventions
 *  - plain C types only; all pointers are DEVICE pointers unless the name says host.
 *  - every function returns 0 on success, <0 on error (never throws / longjmps);
 *    vd_last_error() returns a thread-local message.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue.
 *  - caller owns every buffer; float buffers must be 16-byte aligned, row strides multiples of 4.
 *  - fp32 throughout (IEEE, exact-fp32 MFMA); token ids are int32, 0 = padding; byte masks are uint8.
 *  - LSTM weights follow nn.SeqLSTM: weight [(D+H) x 4H] = [Wx ; Wh], gate column order i,f,o,g;
 *    nn.Linear weights are [out x in]; the embedding table is [(V+1) x E] with row 0 = pad (zero).
 */
#ifndef VISDIAL_HIP_H
#define VISDIAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD_ACT_NONE 0
#define VD_ACT_TANH 1

/* ---- library ------------------------------------------------------------------------- */
const char* vd_last_error(void);
/* Bumped whenever an exported symbol is removed or the meaning of an argument changes; hosts compare it with the VD_ABI_VERSION they
 * were written against right after loading (visdial_amd/_lib.py, lua/visdial_ffi.lua).  2 = round 4's surface: vd_tune_set /
 * vd_tune_clear / vd_lstm_seq_status removed, vd_model_params.lstmBf16 also takes 3 / 6 / 9 (exact-operand split). */
#define VD_ABI_VERSION 2
int vd_abi_version(void);
int vd_device_count(int* count);
int vd_set_device(int device);                 /* replaces cutorch.setDevice, train.lua:19 */
int vd_device_info(int device, char* name256, char* arch256, int* num_cus, int64_t* hbm_bytes);
int vd_malloc(void** ptr, int64_t bytes);      /* for hosts without a tensor library (Lua) */
int vd_free(void* ptr);
int vd_memset(void* ptr, int value, int64_t bytes, void* stream);
int vd_memcpy_h2d(void* dst, const void* src_host, int64_t bytes, void* stream); /* :cuda() dataloader.lua:410-416 */
int vd_memcpy_d2h(void* dst_host, const void* src, int64_t bytes, void* stream);
int vd_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream);
int vd_stream_synchronize(void* stream);
/* strided 2-D device copy of rows x cols floats (nn.JoinTable / nn.Narrow on column blocks,
 * encoders/lf-ques-im-hist.lua:49-55) */
int vd_copy_2d(float* dst, int64_t dst_ld, const float* src, int64_t src_ld, int64_t rows, int64_t cols,
               void* stream);

/* `flags` of the contraction / recurrence entry points: VD_FLAG_BF16 rounds the GEMM operands to bf16 (RNE) and
 * multiplies them on the bf16 MFMA with fp32 accumulation -- the opt-in "bf16 LSTM step" of BASELINE.json
 * configs[4].  0 = exact fp32 (the reference's arithmetic, the headline configuration). */
#define VD_FLAG_BF16 1
/* VD_FLAG_SPLIT9 (vd_lstm_forward / vd_lstm_backward, throughput shapes): the recurrent product h*Wh / da*Wh^T as the EXACT
 * three-way bf16 split of both fp32 operands -- nine bf16 MFMAs with fp32 accumulation per fp32 one, every product exact
 * (csrc/split_core.h): fp32-grade results at 9/16 of the matrix-pipe time.  Opt-in; VD_FLAG_SPLIT6 / VD_FLAG_SPLIT3 drop the
 * smallest products (NOT fp32-grade: they exist for the error table of tests/test_ops_gpu.py). */
#define VD_FLAG_SPLIT9 2
#define VD_FLAG_SPLIT6 4
#define VD_FLAG_SPLIT3 8

/* ---- dense contractions (nn.Linear / hoisted SeqLSTM input projection / weight grads) -- */
/* C[MxN] (+)= act(A[MxK] * W[NxK]^T + bias)   -- nn.Linear:updateOutput (+nn.Tanh),
 * e.g. encoders/mn-att-ques-im-hist.lua:64-65,77,88,106; also dX = dA * Wh^T style products.
 * accumulate: 0 = overwrite, 1 = C += (plain read-modify-write), 2 = C += with float atomics (C has other
 * concurrent atomic writers: the shared embedding gradient) */
int vd_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int M, int N, int K, int act, int accumulate, void* stream);
/* C[MxN] (+)= A[MxK] * B[KxN] + bias           -- x*Wx+b of nn.SeqLSTM (mn-att:27-41), nn.Linear:updateGradInput */
int vd_gemm_nn(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
               int64_t ldc, int M, int N, int K, int accumulate, void* stream);
/* C[MxN] += A[KxM]^T * B[KxN]                  -- accGradParameters of nn.Linear / nn.SeqLSTM */
int vd_gemm_tn_acc(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                   int N, int K, int flags, void* stream);
/* C[MxN] += sum_k A[a_rows[k], :M]^T * B[b_rows[k], :N], k < K  -- the same accGradParameters contraction over an
 * explicit list of (row of A, row of B) pairs: nn.SeqLSTM:maskZero() zeroes the gradient of padded (timestep, row)
 * pairs (encoders/mn-att-ques-im-hist.lua:27-41), so only the non-pad pairs are contracted (~55 % of T*N) */
int vd_gemm_tn_rows_acc(const float* A, int64_t lda, const int32_t* a_rows, const float* B, int64_t ldb,
                        const int32_t* b_rows, float* C, int64_t ldc, int M, int N, int K, void* stream);
/* out[N] += column sums of X[MxN]              -- gradBias */
int vd_colsum_acc(const float* X, int64_t ld, int M, int N, float* out, void* stream);

/* ---- nn.SeqLSTM (Element-Research rnn; call sites mn-att:27-45, disc.lua:4-15, gen.lua:17-22,
 *      lf-ques.lua:18-24, hre-ques-im-hist.lua:34,74,92) ---------------------------------------- */
/* Recurrence over T steps, time-major.  The input projection (x_t*Wx + b) is supplied already
 * computed: dense mode xproj[t] = xproj + t*x_tstride, row n at +n*x_ld ([N x 4H]); table mode
 * (tok_gather != NULL, [T x N]): row = xproj + tok_gather[t,n]*x_ld (xproj = Emb*Wx+b, [V+1 x 4H]).
 * tok_mask ([T x N] or NULL) implements :maskZero(): rows with token 0 get h = c = gates = 0.
 * h0/c0 ([N x H] or both NULL = zeros) are userPrevOutput/userPrevCell (gen.lua:32-38).
 * Outputs: gates [T x N x 4H] post-activation (i,f,o,g), h and c [T x N x H].
 * Limits: one step's slice of every tensor below 4 GB (N * 4H * 4 bytes; a projection table: its rows * x_ld * 4 bytes --
 * 524 288 table rows at H = 512): the epilogue uses 32-bit byte offsets.  Larger batches: call per row range.
 * Throughput shapes (N >= 2048) run the whole recurrence as ONE persistent launch (tile queues + per-row-tile
 * arrival counters; csrc/lstm.hip) on the LDS-DMA pipeline; its work buffers (4H x H gate-interleaved transpose
 * of Wh, queue heads, counters) are library-owned per (device, stream), so calls on different streams may overlap. */
int vd_lstm_forward(const float* xproj, int64_t x_tstride, int64_t x_ld, const int32_t* tok_gather,
                    const int32_t* tok_mask, const float* Wh, const float* h0, const float* c0, float* gates,
                    float* h, float* c, int T, int N, int H, int flags, void* stream);
/* Backward through time.  gates is overwritten IN PLACE by da (gradient w.r.t. the pre-activation
 * gates, = gradient of xproj).  dh_seq [T x N x H] or NULL: gradient arriving at every h_t;
 * dh_last [N x H] or NULL: extra gradient at h_{T-1} (nn.Select(1,-1)); dc_last or NULL:
 * userNextGradCell (gen.lua:49).  dc_work [N x H] scratch, on return = dL/dc0 (userGradPrevCell);
 * dh0 [N x H] or NULL receives dL/dh0 (userGradPrevOutput, gen.lua:50-58).
 * h_seq [T x N x H] + dWh_acc [H x 4H] (both or neither): also accumulate the recurrent weight gradient
 * dWh += sum_{t>=1} h_{t-1}^T da_t (accGradParameters). */
int vd_lstm_backward(const float* Wh, float* gates, const float* c, const float* c0, const float* dh_seq,
                     const float* dh_last, const float* dc_last, float* dc_work, float* dh0, const float* h_seq,
                     float* dWh_acc, int T, int N, int H, int flags, void* stream);


/* Two stacked nn.SeqLSTM layers (the pattern of every encoder branch: mn-att:27-45, lf-ques.lua:17-24)
 * advanced as a skewed wavefront, up to 2 independent stacks per call (history + question branches):
 * tick tau launches ONE grouped kernel doing L1 step tau, the layer-2 input projection of step tau-1
 * and L2 step tau-2 of every stack, so a T-step stack costs T+2 launches instead of 2T.
 * Forward: gates1 must hold x*Wx1+b1 on entry (vd_gemm_nn); outputs as vd_lstm_forward for both layers.
 * Backward: gates1/gates2 are overwritten by da1/da2; dh_last2 [N x H] is the gradient at the top
 * layer's last step; dh1_seq [T x N x H], dc1, dc2 [N x H] are scratch.  Weight gradients are then
 * formed by the caller with vd_gemm_tn_acc / vd_colsum_acc exactly as for vd_lstm_backward. */
typedef struct {
  int T, N;
  const int32_t* tok_mask;               /* [T x N] or NULL (maskZero) */
  const float *Wh1, *Wx2, *b2, *Wh2;
  float *gates1, *h1, *c1, *gates2, *h2, *c2;
  const int32_t* nact;                   /* HOST int32[T] or NULL.  Rows sorted by sequence length: only rows
                                            [0, nact[t]) are non-pad at step t and are computed; the caller
                                            zero-fills the rest (vd_zero_inactive_rows / memset) */
} vd_lstm2_fwd_t;
typedef struct {
  int T, N;
  const float *Wh1, *Wx2, *Wh2;
  float* gates1;
  const float* c1;
  float* gates2;
  const float* c2;
  const float* dh_last2;
  float* dh1_seq;
  float *dc1, *dc2;
  const int32_t* nact;                   /* HOST int32[T] or NULL, as in vd_lstm2_fwd_t */
} vd_lstm2_bwd_t;
int vd_lstm2_forward(const vd_lstm2_fwd_t* stacks, int nstacks, int H, void* stream);
int vd_lstm2_backward(const vd_lstm2_bwd_t* stacks, int nstacks, int H, void* stream);
/* the same wavefront with a pass's arithmetic: flags = 0 (fp32 MFMA, = the two calls above) or VD_FLAG_BF16 (both operands of every
 * recurrent product rounded to bf16 while staged, fp32 accumulate and state: the encoder ticks of a `lstmPrecision = bf16` pass) */
int vd_lstm2_forward_flags(const vd_lstm2_fwd_t* stacks, int nstacks, int H, int flags, void* stream);
int vd_lstm2_backward_flags(const vd_lstm2_bwd_t* stacks, int nstacks, int H, int flags, void* stream);

/* zero rows [nact[t], N) of every time slice of a [T x N x ld] buffer (nact_dev: DEVICE int32[T]) */
int vd_zero_inactive_rows(float* buf, int64_t tstride, int64_t ld, int ncols, const int32_t* nact_dev, int T,
                          int N, void* stream);

/* ---- nn.LookupTableMaskZero / nn.Dropout / small glue ------------------------------------- */
/* out[r,:] = emb[tok[r],:] * (mask ? mask*scale : 1)        mn-att:21,24-25; disc.lua:12 */
int vd_embed_gather(const float* emb, const int32_t* tok, const uint8_t* mask, float* out, int64_t rows,
                    int E, float scale, void* stream);
/* demb[tok[r],:] += dx[r,:] * mask*scale                    LookupTable:accGradParameters */
int vd_embed_scatter_acc(float* demb, const int32_t* tok, const uint8_t* mask, const float* dx, int64_t rows,
                         int E, float scale, void* stream);
/* nn.MaskTime (model_utils/MaskTime.lua:12-40): out[t,n,:] = tok[t,n] != 0 ? feat[n,:] : 0, and its
 * backward dfeat[n,:] = sum_t (tok[t,n] != 0) * dout[t,n,:]   (encoders/hre-ques-im-hist.lua:50-53) */
int vd_mask_time_forward(const float* feat, const int32_t* tok, float* out, int T, int N, int D, void* stream);
int vd_mask_time_backward(const float* dout, const int32_t* tok, float* dfeat, int T, int N, int D,
                          void* stream);
/* counting sort of token ids (prepares the option-table gradient); offset int32[V+1],
 * work int32[2V], perm int32[n] */
int vd_token_sort(const int32_t* tok, int64_t n, int V, int32_t* offset, int32_t* work, int32_t* perm,
                  void* stream);
/* out[tok[r], 0:ncol] += X[r, 0:ncol] for all n rows */
int vd_segment_rowsum_acc(const float* X, int64_t ldx, const int32_t* tok, const int32_t* perm, int64_t n,
                          int ncol, float* out, int64_t ldo, void* stream);
/* Bernoulli(1-p) keep-mask bytes from a counter-based generator (nn.Dropout noise) */
int vd_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, float p, void* stream);
/* y = x*mask*scale (nn.Dropout forward and backward) */
int vd_dropout_apply(const float* x, const uint8_t* mask, float* y, int64_t n, float scale, void* stream);
/* dx = dy*(1-y^2) (nn.Tanh:updateGradInput) */
int vd_tanh_backward(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* c = alpha*a + beta*b (b may be NULL)  (nn.CAddTable and gradient fan-in) */
int vd_axpby(const float* a, const float* b, float* c, int64_t n, float alpha, float beta, void* stream);

/* ---- memory-network attention: nn.MM(false,true) -> nn.MaskSoftMax -> nn.MM
 *      (mn-att:48-62; model_utils/MaskSoftMax.lua:5-46; mask from model.lua:280-294, 1 = hidden) */
int vd_mn_attention_forward(const float* Q, const float* Hm, const uint8_t* mask, float* P, float* hAtt, int B,
                            int R, int H, void* stream);
int vd_mn_attention_backward(const float* Q, const float* Hm, const float* P, const float* dhAtt, float* dQ,
                             float* dHm, int B, int R, int H, void* stream);

/* history attention of encoders/hrea-ques-im-hist.lua:83-131 (Linear(H,1) scores, Replicate+CAddTable,
 * model_utils/MaskFuture.lua, model_utils/ReplaceZero.lua(-inf), SoftMax, CMulTable+Sum):
 * sq, sh [B*R] scores; Hm [B x R x H]; P [B x R x R]; att [B x R x H] */
int vd_hrea_attention_forward(const float* sq, const float* sh, const float* Hm, float* P, float* att, int B, int R,
                              int H, void* stream);
int vd_hrea_attention_backward(const float* Hm, const float* P, const float* datt, float* dsq, float* dsh,
                               float* dHm, int B, int R, int H, void* stream);
/* nn.Linear(H, 1) (hrea:83-85): out[n] = <x[n,:], w> + b; backward accumulates dw, db and writes dx */
int vd_rowdot_forward(const float* x, const float* w, const float* bias, float* out, int N, int H, void* stream);
int vd_rowdot_backward(const float* x, const float* w, const float* dout, float* dw, float* db, float* dx, int N,
                       int H, void* stream);

/* ---- SAN image attention (mn-att:68-104).  pre = tanh(Linear(img)) per IMAGE [B*S2 x H];
 *      mask1/mask2 = dropout keep-masks of img_tr / img_ques_common per ROUND (NULL in evaluate()) */
int vd_img_common_forward(const float* pre, const uint8_t* mask1, const float* Wc, const float* bc,
                          const float* qc, const uint8_t* mask2, float* iqc, int N, int R, int S2, int H, int Kc,
                          float scale, void* stream);
int vd_img_att_forward(const float* iqc, const float* wa, const float* ba, const float* pre,
                       const uint8_t* mask1, const float* u0, float* p, float* u1, int N, int R, int S2, int H,
                       int Kc, float scale, void* stream);
int vd_img_att_backward(float* iqc_dz, const float* wa, const float* pre, const uint8_t* mask1,
                        const uint8_t* mask2, const float* p, const float* datt, float* dwa, float* dba,
                        float* dqc, float* work /* [N x S2] scratch */, int N, int R, int S2, int H, int Kc,
                        float scale, void* stream);
int vd_img_tr_backward(const float* dz, const float* Wc, const float* p, const float* datt,
                       const uint8_t* mask1, float* dpre, int N, int R, int S2, int H, int Kc, float scale,
                       void* stream);
int vd_img_common_wgrad(const float* dz, const float* pre, const uint8_t* mask1, float* dWc, int N, int R,
                        int S2, int H, int Kc, float scale, void* stream);

/* ---- discriminative head: nn.MM + nn.Squeeze (disc.lua:22-29) + nn.CrossEntropyCriterion
 *      (model.lua:37-38,330,334).  gt is 0-based here (the reference's answer_ind is 1-based). */
int vd_score_ce(const float* optH, const float* enc, const int32_t* gt, float* scores, float* loss_rows,
                float* dOptH, float* dEnc, int N, int O, int H, float gscale, void* stream);
/* generative head: nn.Sequencer(nn.MaskZero(nn.LogSoftMax(),1)) (decoders/gen.lua:24) +
 * SequencerCriterion(MaskZeroCriterion(ClassNLLCriterion, sizeAverage=false)) (model.lua:33-36).
 * logits [rows x ld] (V valid columns, ld = V rounded up to 4), tok_in/target int32[rows] (1-based
 * vocabulary ids, 0 = pad).  loss_rows[r] = -log p(target); with write_grad the row is overwritten
 * by d loss / d logits (softmax - onehot, zero rows at pads). */
int vd_logsoftmax_nll(float* logits, int64_t ld, int64_t rows, int V, const int32_t* tok_in,
                      const int32_t* target, float* loss_rows, int write_grad, void* stream);
/* in-place nn.LogSoftMax over `rows` rows of V logits (sampling / beam search, model.lua:432-613) */
int vd_log_softmax_rows(float* x, int64_t ld, int64_t rows, int V, void* stream);
/* utils.computeRanks (utils.lua:106-128): 1-based descending-sort position of every option */
int vd_ranks(const float* scores, int32_t* ranks, int N, int O, void* stream);

/* ---- wrapperdW:clamp(-5,5) + adam (model.lua:96-99; model_utils/optim_updates.lua:62-91) ---- */
int vd_clamp_adam(float* w, float* g, float* m, float* v, int64_t n, float gscale, float clip, float beta1,
                  float beta2, float eps, float step, void* stream);

/* ======================================================================================================
 * Model-level entry points (csrc/runtime.hip): the whole training / retrieval step behind the ABI, for every plug-in
 * pair of the reference: encoders lf-ques, lf-ques-im, lf-ques-hist, lf-ques-im-hist, lf-att-ques-im-hist,
 * hre-ques-hist, hre-ques-im-hist, hrea-ques-im-hist, mn-ques-hist, mn-ques-im-hist, mn-att-ques-im-hist
 * (encoders/<name>.lua) x decoders disc, gen (decoders/<name>.lua).  A LuaJIT host's model.lua needs only these
 * (INTEGRATION.md): stream fork/join, the skewed two-layer wavefront, the length sort, workspaces, forwardConnect /
 * backwardConnect and launch order live in the library.  One host thread per model; calls enqueue on library-owned
 * streams and block only where a host value is returned (loss, scores, ranks, tensors, log-probabilities).
 * ====================================================================================================== */
typedef struct vd_model vd_model;
typedef struct vd_model_params {   /* the `params` keys Model() consumes: opts.lua:15-40, train.lua:55-59 */
  int32_t vocabSize, embedSize, rnnHiddenSize, imgFeatureSize, imgSpatialSize, commonEmbeddingSize,
          numAttentionLayers, maxQuesCount, numOptions;
  float learningRate, lrDecayRate, minLRate;     /* opts.lua:35-38 */
  uint64_t seed;                                 /* dropout noise stream */
  int32_t lstmBf16;                              /* option recurrence: 0 = fp32 MFMA (a zeroed struct; the hosts' own default is 9); 1 = bf16 operands and compact bf16 state, plus bf16 operands in the encoder's recurrent products and dense weight gradients (configs[4] "bf16 LSTM step"); 9 = exact 3-way bf16 split of both operands, 9 products, in the two step kernels, the dWh contraction and the image attention's three dense products (fp32-grade; what bench.py measures and what -lstmPrecision defaults to in opts.py / lua/model.lua); 6 / 3 = fewer products (data only) */
  int32_t useStreams;                            /* 0 = everything on one stream (debug) */
  int32_t numLayers;                             /* -numLayers (opts.lua:27): lf-*, hre-* encoders and the gen decoder; <1 = 2 */
  int32_t imgEmbedSize;                          /* -imgEmbedSize (opts.lua:24): hre-ques-im-hist, hrea-ques-im-hist */
  float dropout;                                 /* -dropout (opts.lua:29): the fusion Dropout of lf-ques* */
} vd_model_params;
typedef struct vd_batch {          /* HOST pointers, dataloader layout (dataloader.lua:324-339, 378-475) */
  int32_t B, Tq, Th, To;           /* dialogs; trimmed question / history / option lengths (To: columns of options,
                                      or of option_in / option_out for gen retrieval) */
  const int32_t* ques_fwd;         /* [B*R x Tq] right-aligned, 0 = pad */
  const int32_t* hist;             /* [B*R x Th] right-aligned                      (encoders with `hist`) */
  const float* img_feat;           /* [B x S*S x C] attention encoders, [B x F] otherwise (encoders with `im`) */
  const int32_t* options;          /* [B*R x O x To] left-aligned                   (decoder disc) */
  const int32_t* answer_ind;       /* [B*R] 1-based, or NULL (test split) */
  int32_t Ta;                      /* trimmed answer length + 1 */
  const int32_t* answer_in;        /* [B*R x Ta] <START>+tokens, left-aligned       (decoder gen, training) */
  const int32_t* answer_out;       /* [B*R x Ta] tokens+<END> */
  const int32_t* option_in;        /* [B*R x O x To] <START>+tokens                 (decoder gen, retrieval) */
  const int32_t* option_out;       /* [B*R x O x To] tokens+<END> */
} vd_batch;
/* Model:__init (model.lua:10-63): parameter vectors (zeroed), optimiser state, streams */
int vd_model_create(const vd_model_params* p, const char* encoder, const char* decoder, vd_model** out);
void vd_model_destroy(vd_model* m);
/* wrapper:getParameters() (model.lua:55): flat layout = embed | encoder tensors | decoder tensors, every tensor
 * 16-byte aligned; tensor i = name, element offset, rows x cols */
int64_t vd_model_num_tensors(const vd_model* m);
int64_t vd_model_flat_size(const vd_model* m);
int vd_model_tensor_info(const vd_model* m, int64_t i, char* name64, int64_t* offset, int64_t* rows, int64_t* cols);
/* device pointers of wrapperW, wrapperdW and the Adam moments (for a host-side RCCL all-reduce of wrapperdW) */
int vd_model_flat_pointers(vd_model* m, float** W, float** dW, float** adam_m, float** adam_v);
void* vd_model_stream(vd_model* m);                       /* the main hipStream_t */
/* data-parallel bucketing (new relative to the single-GPU reference): flat element range [lo, hi) of the
 * encoder's own tensors, and "make `stream` wait until those gradients of the last forward_backward are final" -- under
 * a disc decoder that is the end of the encoder backward, well before the option-LSTM backward ends */
int vd_model_encoder_range(const vd_model* m, int64_t* lo, int64_t* hi);
int vd_model_wait_encoder_grads(vd_model* m, void* stream);
/* ---- data-parallel gradient exchange inside the library (csrc/comm.hip).  The reference is
 * single-GPU (train.lua:19); a host that wants N GPUs runs one process per GPU and makes three extra calls: rank 0
 * creates the 128-byte rendezvous token and the host hands it to the peers by any channel it has; every rank joins the
 * RCCL communicator on its current device; then once per step, between forward_backward and update(1/world),
 * vd_model_allreduce_grads sums wrapperdW over all ranks in two buckets -- the encoder's own tensors on a library-owned
 * communication stream underneath the option-LSTM backward, the shared embedding + decoder tensors behind the step --
 * and makes the main stream wait for both.  RCCL (librccl.so.1) is loaded at run time on first use. ---- */
int vd_comm_unique_id(void* out128);                      /* ncclGetUniqueId: 128 bytes */
int vd_comm_init(int rank, int world, const void* id128); /* ncclCommInitRank on the current device (collective) */
int vd_comm_info(int* rank, int* world);                  /* world = 0: no communicator */
int vd_comm_destroy(void);
/* no collective, no device: can this process load RCCL?  version = NCCL_VERSION_CODE of the loaded library (0 = unknown).
 * Hosts exchange the answer over their own channel before any rank enters vd_comm_unique_id / vd_comm_init. */
int vd_comm_available(int* version);
/* the last vd_model_allreduce_grads: floats in bucket 1 (encoder tensors, reduced under the decoder's backward) and bucket 2
 * (embedding + decoder), whether bucket 1 was issued early, calls since vd_comm_init */
int vd_comm_stats(int64_t* bucket1_floats, int64_t* bucket2_floats, int* overlapped, int64_t* calls);
/* the last vd_model_allreduce_grads: ms between "bucket 1 summed over the ranks" (communication stream) and "backward ended" (main
 * stream); positive = hidden under the decoder's backward with that much to spare, negative = exposed.  Waits for both events. */
int vd_comm_overlap_ms(float* lead_ms);
int vd_model_allreduce_grads(vd_model* m);                /* enqueue only; every rank, once per step */
int vd_model_init_params(vd_model* m, uint64_t seed);     /* library-default init */
int vd_model_set_tensor(vd_model* m, const char* name, const float* host, int64_t n);   /* wrapperW:copy(...) */
int vd_model_get_tensor(vd_model* m, const char* name, int which /*0 W, 1 dW, 2 m, 3 v*/, float* host, int64_t n);
int vd_model_set_training(vd_model* m, int on);           /* wrapper:training() / :evaluate() (model.lua:57,111) */
int vd_model_set_dropout_mask(vd_model* m, const char* site, const uint8_t* host_keep, int64_t n); /* NULL host = clear */
/* batch re-layout + upload (model.lua:255-294, dataloader.lua:410-475), asynchronous, double-buffered */
int vd_model_upload_batch(vd_model* m, const vd_batch* host_batch);
/* Model:forwardBackward on the uploaded batch (model.lua:249-342); zeroes the gradients first unless only_forward */
int vd_model_forward_backward(vd_model* m, int only_forward);
int vd_model_loss(vd_model* m, float* loss);              /* curLoss of the last forward (waits for it): disc = mean
                                                             cross-entropy, gen = summed NLL (model.lua:309-311,330) */
/* Model:retrieveBatch up to the option scores (model.lua:344-425): disc = scores of a forward pass, gen = candidate
 * log-likelihoods (utils.computeLhood, utils.lua:86-102); read with vd_model_scores / vd_model_ranks */
int vd_model_retrieve(vd_model* m);
/* wrapperdW*gscale -> clamp(-5,5) -> adam -> lr decay (model.lua:96-105; optim_updates.lua:62-91) */
int vd_model_update(vd_model* m, float gscale);
int vd_model_learning_rate(vd_model* m, double* lr, int set);
/* Model:generateAnswers, device side (model.lua:432-613; generative decoder).  The host keeps the candidate
 * bookkeeping exactly as the reference's Lua does; each call advances all live hypotheses together.
 *   encode        encoder forward of the uploaded batch = forwardBackward(batch, true, true) (model.lua:464)
 *   decode_begin  hiddenBeams (model.lua:478-503): hypothesis i starts from the encoder state of QA round rounds[i] (0-based)
 *   decode_step   one decoder step (model.lua:518-522, 590-596): tokens[n] in, log-probabilities [n x vocabSize] out (host)
 *   decode_select hypothesis i continues from the stepped state of hypothesis src[i] (model.lua:560-575); slots >= n_keep
 *                 keep their previous state; sampling passes the identity */
int vd_model_encode(vd_model* m);
int vd_model_decode_begin(vd_model* m, const int32_t* rounds, int n);
int vd_model_decode_step(vd_model* m, const int32_t* tokens, float* host_logprobs);
int vd_model_decode_select(vd_model* m, const int32_t* src, int n_keep);
int vd_model_scores(vd_model* m, float* host_scores, int64_t n);        /* [N x O] of the last forward / retrieve */
int vd_model_ranks(vd_model* m, int use_gt, int32_t* host_ranks);       /* utils.computeRanks (utils.lua:106-128) */
/* decoder disc: rows the option LSTM executed for the batch of the LAST STEP (before any step: of the uploaded batch) vs the N * O
 * candidates they stand for -- the upload encodes every DISTINCT candidate row once (decoders/disc.lua:4-15: the encoding depends on
 * the tokens only).  In a pipelined loop this is the batch that was stepped, not the one prefetched behind it. */
int vd_model_option_rows(vd_model* m, int64_t* executed, int64_t* total);
int vd_model_family_ms(vd_model* m, float* ms3);          /* device ms of option-LSTM fwd, bwd, dWh in the last step */
int vd_model_synchronize(vd_model* m);

#ifdef __cplusplus
}
#endif
#endif /* VISDIAL_HIP_H */
