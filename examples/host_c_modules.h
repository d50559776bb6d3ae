/* host_c_modules.h -- what the C plug-in examples share: the operator-level entry points of include/visdial_hip.h resolved with
 * dlsym (a host without a tensor library: vd_malloc + host pointers), and module objects with forward / backward over flat
 * parameter / gradient vectors -- SeqLSTM, Linear (+ Tanh) -- exactly the nn modules the reference's plug-in files build
 * (encoders/lf-ques.lua:6-33, encoders/mn-att-ques-im-hist.lua:21-106, decoders/disc.lua:3-38).  lua/vdnn.lua is the same
 * objects in Lua; these are the executable stand-in (no Lua interpreter exists in the build container or on the GPU box). */
#ifndef HOST_C_MODULES_H
#define HOST_C_MODULES_H
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"

static void* g_lib;
static const char* (*p_last_error)(void);
#define FN(ret, name, args) static ret(*name) args;
FN(int, p_set_device, (int))
FN(int, p_malloc, (void**, int64_t))
FN(int, p_memset, (void*, int, int64_t, void*))
FN(int, p_h2d, (void*, const void*, int64_t, void*))
FN(int, p_d2h, (void*, const void*, int64_t, void*))
FN(int, p_sync, (void*))
FN(int, p_gemm_nt, (const float*, int64_t, const float*, int64_t, const float*, float*, int64_t, int, int, int, int, int, void*))
FN(int, p_gemm_nn, (const float*, int64_t, const float*, int64_t, const float*, float*, int64_t, int, int, int, int, void*))
FN(int, p_gemm_tn_acc, (const float*, int64_t, const float*, int64_t, float*, int64_t, int, int, int, int, void*))
FN(int, p_colsum_acc, (const float*, int64_t, int, int, float*, void*))
FN(int, p_lstm_forward, (const float*, int64_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int, void*))
FN(int, p_lstm_backward, (const float*, float*, const float*, const float*, const float*, const float*, const float*, float*, float*, const float*, float*, int, int, int, int, void*))
FN(int, p_embed_gather, (const float*, const int32_t*, const uint8_t*, float*, int64_t, int, float, void*))
FN(int, p_embed_scatter_acc, (float*, const int32_t*, const uint8_t*, const float*, int64_t, int, float, void*))
FN(int, p_token_sort, (const int32_t*, int64_t, int, int32_t*, int32_t*, int32_t*, void*))
FN(int, p_segment_rowsum_acc, (const float*, int64_t, const int32_t*, const int32_t*, int64_t, int, float*, int64_t, void*))
FN(int, p_tanh_backward, (const float*, const float*, float*, int64_t, void*))
FN(int, p_score_ce, (const float*, const float*, const int32_t*, float*, float*, float*, float*, int, int, int, float, void*))
FN(int, p_ranks, (const float*, int32_t*, int, int, void*))
FN(int, p_clamp_adam, (float*, float*, float*, float*, int64_t, float, float, float, float, float, float, void*))
FN(int, p_hrea_attention_forward, (const float*, const float*, const float*, float*, float*, int, int, int, void*))
FN(int, p_hrea_attention_backward, (const float*, const float*, const float*, float*, float*, float*, int, int, int, void*))
FN(int, p_rowdot_forward, (const float*, const float*, const float*, float*, int, int, void*))
FN(int, p_rowdot_backward, (const float*, const float*, const float*, float*, float*, float*, int, int, void*))
FN(int, p_mask_time_forward, (const float*, const int32_t*, float*, int, int, int, void*))
FN(int, p_mask_time_backward, (const float*, const int32_t*, float*, int, int, int, void*))
FN(int, p_copy_2d, (float*, int64_t, const float*, int64_t, int64_t, int64_t, void*))
FN(int, p_logsoftmax_nll, (float*, int64_t, int64_t, int, const int32_t*, const int32_t*, float*, int, void*))
FN(int, p_dropout_mask, (uint8_t*, int64_t, uint64_t, float, void*))
FN(int, p_dropout_apply, (const float*, const uint8_t*, float*, int64_t, float, void*))
FN(int, p_axpby, (const float*, const float*, float*, int64_t, float, float, void*))
FN(int, p_mn_attention_forward, (const float*, const float*, const uint8_t*, float*, float*, int, int, int, void*))
FN(int, p_mn_attention_backward, (const float*, const float*, const float*, const float*, float*, float*, int, int, int, void*))
FN(int, p_img_common_forward, (const float*, const uint8_t*, const float*, const float*, const float*, const uint8_t*, float*, int, int, int, int, int, float, void*))
FN(int, p_img_att_forward, (const float*, const float*, const float*, const float*, const uint8_t*, const float*, float*, float*, int, int, int, int, int, float, void*))
FN(int, p_img_att_backward, (float*, const float*, const float*, const uint8_t*, const uint8_t*, const float*, const float*, float*, float*, float*, float*, int, int, int, int, int, float, void*))
FN(int, p_img_tr_backward, (const float*, const float*, const float*, const float*, const uint8_t*, float*, int, int, int, int, int, float, void*))
FN(int, p_img_common_wgrad, (const float*, const float*, const uint8_t*, float*, int, int, int, int, int, float, void*))

#define LOAD(var, sym) do { *(void**)(&var) = dlsym(g_lib, sym); if (!var) { fprintf(stderr, "missing symbol %s\n", sym); exit(2); } } while (0)
#define CHECK(call) do { int rc__ = (call); if (rc__ != 0) { fprintf(stderr, "%s:%d: %s failed (%d): %s\n", __FILE__, __LINE__, #call, rc__, p_last_error()); exit(3); } } while (0)

__attribute__((unused)) static float* dev_floats(int64_t n) {            /* torch.CudaTensor(n):zero() */
  void* p = NULL;
  CHECK(p_malloc(&p, (n ? n : 4) * 4));
  CHECK(p_memset(p, 0, (n ? n : 4) * 4, NULL));
  return (float*)p;
}
__attribute__((unused)) static int32_t* dev_ints_from(const int32_t* host, int64_t n) {
  void* p = NULL;
  CHECK(p_malloc(&p, n * 4));
  CHECK(p_h2d(p, host, n * 4, NULL));
  return (int32_t*)p;
}
__attribute__((unused)) static int64_t align4(int64_t n) { return (n + 3) / 4 * 4; }      /* every tensor 16-byte aligned inside the flat vectors */

/* ---- module objects: parameter views into the flat vectors + saved activations ------------------------------------------------ */
typedef struct {               /* nn.SeqLSTM(D, H) (encoders/lf-ques.lua:18-24, decoders/disc.lua:4, decoders/gen.lua:17-22) */
  int D, H, T, N;
  float *W, *b, *dW, *db;      /* W = [Wx ; Wh]: [(D+H) x 4H], gate columns i,f,o,g */
  const float* x;              /* [T*N x D] input of the last forward */
  float *gates, *h, *c;        /* .output = h [T x N x H], .cell = c */
  /* the state hand-off fields of Element-Research's SeqLSTM that decoders/gen.lua:30-60 reads and writes (all [N x H], nullable) */
  const float *userPrevOutput, *userPrevCell;       /* initial h / c of the next forward (consumed by it) */
  const float *gradPrevOutput, *userNextGradCell;   /* extra gradient into the LAST step's h / c of the next backward (consumed by it) */
  float *userGradPrevOutput, *userGradPrevCell;     /* out: gradients w.r.t. the initial h / c */
  const float *h0, *c0;                             /* what the last forward started from */
} SeqLSTM;

__attribute__((unused)) static void lstm_forward(SeqLSTM* l, const float* x, int T, int N, const int32_t* tok_mask) {
  const int H = l->H;
  l->x = x; l->T = T; l->N = N;
  l->gates = dev_floats((int64_t)T * N * 4 * H); l->h = dev_floats((int64_t)T * N * H); l->c = dev_floats((int64_t)T * N * H);
  /* hoisted input projection x*Wx + b straight into the gates buffer, then the recurrence in place (maskZero via tok_mask) */
  CHECK(p_gemm_nn(x, l->D, l->W, 4 * H, l->b, l->gates, 4 * H, T * N, 4 * H, l->D, 0, NULL));
  l->h0 = l->userPrevOutput; l->c0 = l->userPrevCell;           /* consumed once, like the reference's module */
  l->userPrevOutput = l->userPrevCell = NULL;
  if ((l->h0 == NULL) != (l->c0 == NULL)) { fprintf(stderr, "SeqLSTM: userPrevOutput and userPrevCell go together\n"); exit(3); }
  CHECK(p_lstm_forward(l->gates, (int64_t)N * 4 * H, 4 * H, NULL, tok_mask, l->W + (int64_t)l->D * 4 * H, l->h0, l->c0, l->gates, l->h, l->c,
                       T, N, H, 0, NULL));
}
/* returns dx [T*N x D] (or NULL); accumulates dW, db */
__attribute__((unused)) static float* lstm_backward(SeqLSTM* l, const float* dh_seq, const float* dh_last, int need_dx) {
  const int H = l->H, T = l->T, N = l->N;
  const int64_t TN = (int64_t)T * N;
  float* dc = dev_floats((int64_t)N * H);
  if (l->gradPrevOutput) {                                      /* gen.lua:49-51: the decoder's gradient w.r.t. this layer's final h */
    if (!dh_last) dh_last = l->gradPrevOutput;
    else {
      float* t = dev_floats((int64_t)N * H);
      CHECK(p_axpby(dh_last, l->gradPrevOutput, t, (int64_t)N * H, 1.f, 1.f, NULL));
      dh_last = t;
    }
  }
  const float* dc_last = l->userNextGradCell;
  l->gradPrevOutput = l->userNextGradCell = NULL;
  float* dh0 = l->h0 ? dev_floats((int64_t)N * H) : NULL;
  CHECK(p_lstm_backward(l->W + (int64_t)l->D * 4 * H, l->gates, l->c, l->c0, dh_seq, dh_last, dc_last, dc, dh0, NULL, NULL, T, N, H, 0, NULL));
  l->userGradPrevOutput = dh0;
  l->userGradPrevCell = l->h0 ? dc : NULL;
  float* dWh = l->dW + (int64_t)l->D * 4 * H;                 /* da now lives in l->gates */
  if (T > 1) CHECK(p_gemm_tn_acc(l->h, H, l->gates + (int64_t)N * 4 * H, 4 * H, dWh, 4 * H, H, 4 * H, (T - 1) * N, 0, NULL));
  if (l->h0) CHECK(p_gemm_tn_acc(l->h0, H, l->gates, 4 * H, dWh, 4 * H, H, 4 * H, N, 0, NULL));   /* step 0 multiplied the initial state */
  CHECK(p_colsum_acc(l->gates, 4 * H, (int)TN, 4 * H, l->db, NULL));
  CHECK(p_gemm_tn_acc(l->x, l->D, l->gates, 4 * H, l->dW, 4 * H, l->D, 4 * H, (int)TN, 0, NULL));
  if (!need_dx) return NULL;
  float* dx = dev_floats(TN * l->D);
  CHECK(p_gemm_nt(l->gates, 4 * H, l->W, 4 * H, NULL, dx, l->D, (int)TN, l->D, 4 * H, VD_ACT_NONE, 0, NULL));   /* da * Wx^T */
  return dx;
}

typedef struct {               /* nn.Linear(n_in, n_out) [+ nn.Tanh] (encoders/lf-ques.lua:29-31; mn-att:64-65,77,88,106) */
  int n_in, n_out, M;
  float *W, *b, *dW, *db;
  const float* x;
  float* y;
  int no_tanh;                 /* 1 = plain nn.Linear (mn-att:88 ques_common) */
} LinearTanh;

__attribute__((unused)) static float* linear_forward(LinearTanh* l, const float* x, int M) {
  l->x = x; l->M = M;
  l->y = dev_floats((int64_t)M * l->n_out);
  CHECK(p_gemm_nt(x, l->n_in, l->W, l->n_in, l->b, l->y, l->n_out, M, l->n_out, l->n_in, l->no_tanh ? VD_ACT_NONE : VD_ACT_TANH, 0, NULL));
  return l->y;
}
/* returns dx [M x n_in] (NULL when !need_dx); accumulates dW, db */
__attribute__((unused)) static float* linear_backward_ex(LinearTanh* l, const float* dy, int need_dx) {
  const int M = l->M;
  const float* dpre = dy;
  if (!l->no_tanh) {
    float* t = dev_floats((int64_t)M * l->n_out);
    CHECK(p_tanh_backward(dy, l->y, t, (int64_t)M * l->n_out, NULL));
    dpre = t;
  }
  CHECK(p_gemm_tn_acc(dpre, l->n_out, l->x, l->n_in, l->dW, l->n_in, l->n_out, l->n_in, M, 0, NULL));
  CHECK(p_colsum_acc(dpre, l->n_out, M, l->n_out, l->db, NULL));
  if (!need_dx) return NULL;
  float* dx = dev_floats((int64_t)M * l->n_in);
  CHECK(p_gemm_nn(dpre, l->n_out, l->W, l->n_in, NULL, dx, l->n_in, M, l->n_in, l->n_out, 0, NULL));
  return dx;
}
__attribute__((unused)) static float* linear_backward(LinearTanh* l, const float* dy) { return linear_backward_ex(l, dy, 1); }

static void load_entry_points(const char* path) {
  g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!g_lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); exit(2); }
  LOAD(p_last_error, "vd_last_error"); LOAD(p_set_device, "vd_set_device"); LOAD(p_malloc, "vd_malloc"); LOAD(p_memset, "vd_memset");
  LOAD(p_h2d, "vd_memcpy_h2d"); LOAD(p_d2h, "vd_memcpy_d2h"); LOAD(p_sync, "vd_stream_synchronize"); LOAD(p_gemm_nt, "vd_gemm_nt");
  LOAD(p_gemm_nn, "vd_gemm_nn"); LOAD(p_gemm_tn_acc, "vd_gemm_tn_acc"); LOAD(p_colsum_acc, "vd_colsum_acc");
  LOAD(p_lstm_forward, "vd_lstm_forward"); LOAD(p_lstm_backward, "vd_lstm_backward"); LOAD(p_embed_gather, "vd_embed_gather");
  LOAD(p_embed_scatter_acc, "vd_embed_scatter_acc"); LOAD(p_token_sort, "vd_token_sort"); LOAD(p_segment_rowsum_acc, "vd_segment_rowsum_acc");
  LOAD(p_tanh_backward, "vd_tanh_backward"); LOAD(p_score_ce, "vd_score_ce"); LOAD(p_clamp_adam, "vd_clamp_adam"); LOAD(p_ranks, "vd_ranks");
  LOAD(p_hrea_attention_forward, "vd_hrea_attention_forward"); LOAD(p_hrea_attention_backward, "vd_hrea_attention_backward");
  LOAD(p_rowdot_forward, "vd_rowdot_forward"); LOAD(p_rowdot_backward, "vd_rowdot_backward");
  LOAD(p_mask_time_forward, "vd_mask_time_forward"); LOAD(p_mask_time_backward, "vd_mask_time_backward");
  LOAD(p_copy_2d, "vd_copy_2d"); LOAD(p_logsoftmax_nll, "vd_logsoftmax_nll");
  LOAD(p_dropout_mask, "vd_dropout_mask"); LOAD(p_dropout_apply, "vd_dropout_apply"); LOAD(p_axpby, "vd_axpby");
  LOAD(p_mn_attention_forward, "vd_mn_attention_forward"); LOAD(p_mn_attention_backward, "vd_mn_attention_backward");
  LOAD(p_img_common_forward, "vd_img_common_forward"); LOAD(p_img_att_forward, "vd_img_att_forward");
  LOAD(p_img_att_backward, "vd_img_att_backward"); LOAD(p_img_tr_backward, "vd_img_tr_backward");
  LOAD(p_img_common_wgrad, "vd_img_common_wgrad");
  CHECK(p_set_device(0));
}

/* decoders/disc.lua as one module object: shared embedding table x option LSTM (decoder:forward / :backward of model.lua:329,335) */
typedef struct {
  int V, E, H, To, NO;
  float *emb, *demb, *W, *b, *dW, *db;   /* opt.W = [Wx ; Wh] [(E+H) x 4H], opt.b [4H] */
  const int32_t* opts;                   /* device, time-major [To x NO] */
  float *table, *gates, *h, *c;
} DiscDecoder;

/* the 100 clones are one batch of N*O sequences; embed -> x*Wx + b is a gather from the table Emb*Wx + b (no dropout on option
 * embeddings, disc.lua:12-14); returns optH [NO x H] = the state after the last step (disc.lua:16-20) */
__attribute__((unused)) static const float* disc_forward(DiscDecoder* d, const int32_t* opts, int To, int NO) {
  const int H = d->H, E = d->E, V = d->V;
  d->opts = opts; d->To = To; d->NO = NO;
  d->table = dev_floats((int64_t)(V + 1) * 4 * H);
  CHECK(p_gemm_nn(d->emb, E, d->W, 4 * H, d->b, d->table, 4 * H, V + 1, 4 * H, E, 0, NULL));
  d->gates = dev_floats((int64_t)To * NO * 4 * H); d->h = dev_floats((int64_t)To * NO * H); d->c = dev_floats((int64_t)To * NO * H);
  CHECK(p_lstm_forward(d->table, 0, 4 * H, opts, NULL, d->W + (int64_t)E * 4 * H, NULL, NULL, d->gates, d->h, d->c, To, NO, H, 0, NULL));
  return d->h + (int64_t)(To - 1) * NO * H;
}
__attribute__((unused)) static void disc_backward(DiscDecoder* d, const float* d_optH) {
  const int H = d->H, E = d->E, V = d->V, To = d->To, NO = d->NO;
  float* dc = dev_floats((int64_t)NO * H);
  CHECK(p_lstm_backward(d->W + (int64_t)E * 4 * H, d->gates, d->c, NULL, NULL, d_optH, NULL, dc, NULL, NULL, NULL, To, NO, H, 0, NULL));
  if (To > 1) CHECK(p_gemm_tn_acc(d->h, H, d->gates + (int64_t)NO * 4 * H, 4 * H, d->dW + (int64_t)E * 4 * H, 4 * H, H, 4 * H, (To - 1) * NO, 0, NULL));
  void *offs = NULL, *work = NULL, *perm = NULL;                  /* gradient of the gathered table: counting sort + segmented row sum */
  CHECK(p_malloc(&offs, (int64_t)(V + 2) * 4)); CHECK(p_malloc(&work, (int64_t)2 * (V + 1) * 4)); CHECK(p_malloc(&perm, (int64_t)To * NO * 4));
  float* dtab = dev_floats((int64_t)(V + 1) * 4 * H);
  CHECK(p_token_sort(d->opts, (int64_t)To * NO, V + 1, (int32_t*)offs, (int32_t*)work, (int32_t*)perm, NULL));
  CHECK(p_segment_rowsum_acc(d->gates, 4 * H, d->opts, (const int32_t*)perm, (int64_t)To * NO, 4 * H, dtab, 4 * H, NULL));
  CHECK(p_colsum_acc(dtab, 4 * H, V + 1, 4 * H, d->db, NULL));
  CHECK(p_gemm_tn_acc(d->emb, E, dtab, 4 * H, d->dW, 4 * H, E, 4 * H, V + 1, 0, NULL));
  CHECK(p_gemm_nt(dtab, 4 * H, d->W, 4 * H, NULL, d->demb, E, V + 1, E, 4 * H, VD_ACT_NONE, 2, NULL));       /* dEmb += dTable * Wx^T */
}

/* [rows x T] row-major tokens -> device int32 [T x rows] time-major (model.lua:255-257: view(-1, T):t()) */
__attribute__((unused)) static int32_t* time_major(const int32_t* host, int rows, int T) {
  int32_t* t = (int32_t*)malloc((size_t)rows * T * 4);
  for (int n = 0; n < rows; ++n) for (int k = 0; k < T; ++k) t[(size_t)k * rows + n] = host[(size_t)n * T + k];
  int32_t* d = dev_ints_from(t, (int64_t)rows * T);
  free(t);
  return d;
}
#endif
