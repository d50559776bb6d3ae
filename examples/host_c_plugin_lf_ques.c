/* host_c_plugin_lf_ques.c -- an encoder / decoder PLUG-IN PAIR written against the OPERATOR-LEVEL C ABI by a host that has no
 * tensor library: encoders/lf-ques.lua + decoders/disc.lua + the criterion and optimiser of model.lua, composed call for call from
 * the entry points of include/visdial_hip.h -- module objects (LookupTable, SeqLSTM, Linear) with :forward / :backward exactly like
 * the nn modules the reference's plug-in files build (encoders/lf-ques.lua:6-33, decoders/disc.lua:3-38, model.lua:297-337), flat
 * parameter / gradient vectors like wrapper:getParameters() (model.lua:55).  It is what a Lua plug-in file does with LuaJIT's ffi
 * (vd_malloc, host pointers, one call per module) and is the executable stand-in for it: no Lua interpreter exists in the build
 * container or on the GPU box.  tests/test_abi_c_host.py builds it with gcc and compares its loss, every gradient tensor and the
 * post-Adam parameters with the library's own model-level implementation of the same pair (NativeModel) on the same batch.
 *
 *   gcc -O2 -std=c99 -I include examples/host_c_plugin_lf_ques.c -ldl -lm -o host_c_plugin
 *   ./host_c_plugin <libvisdial_hip.so> <in.bin> <out.bin>
 * in.bin : int32 {V, E, H, B, R, O, Tq, To}, then the parameters in getParameters() order as float32 (embed [(V+1) x E],
 *          ques1.W [(E+H) x 4H], ques1.b [4H], ques2.W [2H x 4H], ques2.b, fuse.W [H x H], fuse.b [H], opt.W [(E+H) x 4H], opt.b),
 *          ques_fwd [B*R x Tq] int32, options [B*R*O x To] int32, answer_ind [B*R] int32 (1-based).
 * out.bin: float32 loss, then the flat gradient, then the flat parameters after clamp(-5,5) + adam.
 * Dropout is off (wrapper:evaluate()): the comparison side does the same. */
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
FN(int, p_clamp_adam, (float*, float*, float*, float*, int64_t, float, float, float, float, float, float, void*))

#define LOAD(var, sym) do { *(void**)(&var) = dlsym(g_lib, sym); if (!var) { fprintf(stderr, "missing symbol %s\n", sym); exit(2); } } while (0)
#define CHECK(call) do { int rc__ = (call); if (rc__ != 0) { fprintf(stderr, "%s:%d: %s failed (%d): %s\n", __FILE__, __LINE__, #call, rc__, p_last_error()); exit(3); } } while (0)

static float* dev_floats(int64_t n) {            /* torch.CudaTensor(n):zero() */
  void* p = NULL;
  CHECK(p_malloc(&p, (n ? n : 4) * 4));
  CHECK(p_memset(p, 0, (n ? n : 4) * 4, NULL));
  return (float*)p;
}
static int32_t* dev_ints_from(const int32_t* host, int64_t n) {
  void* p = NULL;
  CHECK(p_malloc(&p, n * 4));
  CHECK(p_h2d(p, host, n * 4, NULL));
  return (int32_t*)p;
}
static int64_t align4(int64_t n) { return (n + 3) / 4 * 4; }      /* every tensor 16-byte aligned inside the flat vectors */

/* ---- module objects: parameter views into the flat vectors + saved activations ------------------------------------------------ */
typedef struct {               /* nn.SeqLSTM(D, H) (encoders/lf-ques.lua:18-24, decoders/disc.lua:4) */
  int D, H, T, N;
  float *W, *b, *dW, *db;      /* W = [Wx ; Wh]: [(D+H) x 4H], gate columns i,f,o,g */
  const float* x;              /* [T*N x D] input of the last forward */
  float *gates, *h, *c;
} SeqLSTM;

static void lstm_forward(SeqLSTM* l, const float* x, int T, int N, const int32_t* tok_mask) {
  const int H = l->H;
  l->x = x; l->T = T; l->N = N;
  l->gates = dev_floats((int64_t)T * N * 4 * H); l->h = dev_floats((int64_t)T * N * H); l->c = dev_floats((int64_t)T * N * H);
  /* hoisted input projection x*Wx + b straight into the gates buffer, then the recurrence in place (maskZero via tok_mask) */
  CHECK(p_gemm_nn(x, l->D, l->W, 4 * H, l->b, l->gates, 4 * H, T * N, 4 * H, l->D, 0, NULL));
  CHECK(p_lstm_forward(l->gates, (int64_t)N * 4 * H, 4 * H, NULL, tok_mask, l->W + (int64_t)l->D * 4 * H, NULL, NULL, l->gates, l->h, l->c,
                       T, N, H, 0, NULL));
}
/* returns dx [T*N x D] (or NULL); accumulates dW, db */
static float* lstm_backward(SeqLSTM* l, const float* dh_seq, const float* dh_last, int need_dx) {
  const int H = l->H, T = l->T, N = l->N;
  const int64_t TN = (int64_t)T * N;
  float* dc = dev_floats((int64_t)N * H);
  CHECK(p_lstm_backward(l->W + (int64_t)l->D * 4 * H, l->gates, l->c, NULL, dh_seq, dh_last, NULL, dc, NULL, NULL, NULL, T, N, H, 0, NULL));
  float* dWh = l->dW + (int64_t)l->D * 4 * H;                 /* da now lives in l->gates */
  if (T > 1) CHECK(p_gemm_tn_acc(l->h, H, l->gates + (int64_t)N * 4 * H, 4 * H, dWh, 4 * H, H, 4 * H, (T - 1) * N, 0, NULL));
  CHECK(p_colsum_acc(l->gates, 4 * H, (int)TN, 4 * H, l->db, NULL));
  CHECK(p_gemm_tn_acc(l->x, l->D, l->gates, 4 * H, l->dW, 4 * H, l->D, 4 * H, (int)TN, 0, NULL));
  if (!need_dx) return NULL;
  float* dx = dev_floats(TN * l->D);
  CHECK(p_gemm_nt(l->gates, 4 * H, l->W, 4 * H, NULL, dx, l->D, (int)TN, l->D, 4 * H, VD_ACT_NONE, 0, NULL));   /* da * Wx^T */
  return dx;
}

typedef struct {               /* nn.Linear(n_in, n_out) + nn.Tanh (encoders/lf-ques.lua:29-31) */
  int n_in, n_out, M;
  float *W, *b, *dW, *db;
  const float* x;
  float* y;
} LinearTanh;

static float* linear_forward(LinearTanh* l, const float* x, int M) {
  l->x = x; l->M = M;
  l->y = dev_floats((int64_t)M * l->n_out);
  CHECK(p_gemm_nt(x, l->n_in, l->W, l->n_in, l->b, l->y, l->n_out, M, l->n_out, l->n_in, VD_ACT_TANH, 0, NULL));
  return l->y;
}
static float* linear_backward(LinearTanh* l, const float* dy) {
  const int M = l->M;
  float* dpre = dev_floats((int64_t)M * l->n_out);
  CHECK(p_tanh_backward(dy, l->y, dpre, (int64_t)M * l->n_out, NULL));
  CHECK(p_gemm_tn_acc(dpre, l->n_out, l->x, l->n_in, l->dW, l->n_in, l->n_out, l->n_in, M, 0, NULL));
  CHECK(p_colsum_acc(dpre, l->n_out, M, l->n_out, l->db, NULL));
  float* dx = dev_floats((int64_t)M * l->n_in);
  CHECK(p_gemm_nn(dpre, l->n_out, l->W, l->n_in, NULL, dx, l->n_in, M, l->n_in, l->n_out, 0, NULL));
  return dx;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  g_lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!g_lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  LOAD(p_last_error, "vd_last_error"); LOAD(p_set_device, "vd_set_device"); LOAD(p_malloc, "vd_malloc"); LOAD(p_memset, "vd_memset");
  LOAD(p_h2d, "vd_memcpy_h2d"); LOAD(p_d2h, "vd_memcpy_d2h"); LOAD(p_sync, "vd_stream_synchronize"); LOAD(p_gemm_nt, "vd_gemm_nt");
  LOAD(p_gemm_nn, "vd_gemm_nn"); LOAD(p_gemm_tn_acc, "vd_gemm_tn_acc"); LOAD(p_colsum_acc, "vd_colsum_acc");
  LOAD(p_lstm_forward, "vd_lstm_forward"); LOAD(p_lstm_backward, "vd_lstm_backward"); LOAD(p_embed_gather, "vd_embed_gather");
  LOAD(p_embed_scatter_acc, "vd_embed_scatter_acc"); LOAD(p_token_sort, "vd_token_sort"); LOAD(p_segment_rowsum_acc, "vd_segment_rowsum_acc");
  LOAD(p_tanh_backward, "vd_tanh_backward"); LOAD(p_score_ce, "vd_score_ce"); LOAD(p_clamp_adam, "vd_clamp_adam");
  CHECK(p_set_device(0));

  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[8];
  if (fread(hd, 4, 8, f) != 8) return 4;
  const int V = hd[0], E = hd[1], H = hd[2], B = hd[3], R = hd[4], O = hd[5], Tq = hd[6], To = hd[7];
  const int N = B * R, NO = N * O;

  /* ---- wrapper:getParameters(): flat W / dW (+ Adam m, v), every tensor 16-byte aligned (model.lua:55) ---- */
  const int64_t sizes[9] = {(int64_t)(V + 1) * E, (int64_t)(E + H) * 4 * H, 4 * H, (int64_t)2 * H * 4 * H, 4 * H, (int64_t)H * H, H,
                            (int64_t)(E + H) * 4 * H, 4 * H};
  int64_t off[10];
  off[0] = 0;
  for (int i = 0; i < 9; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[9];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < 9; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  int32_t* q_host = (int32_t*)malloc((size_t)N * Tq * 4);
  int32_t* o_host = (int32_t*)malloc((size_t)NO * To * 4);
  int32_t* a_host = (int32_t*)malloc((size_t)N * 4);
  if (fread(q_host, 4, (size_t)N * Tq, f) != (size_t)N * Tq || fread(o_host, 4, (size_t)NO * To, f) != (size_t)NO * To ||
      fread(a_host, 4, (size_t)N, f) != (size_t)N) return 4;
  fclose(f);

  /* ---- input re-layout of model.lua:255-294: time-major token matrices, 0-based targets ---- */
  int32_t* qt = (int32_t*)malloc((size_t)N * Tq * 4);
  for (int n = 0; n < N; ++n) for (int t = 0; t < Tq; ++t) qt[(size_t)t * N + n] = q_host[(size_t)n * Tq + t];
  int32_t* ot = (int32_t*)malloc((size_t)NO * To * 4);
  for (int n = 0; n < NO; ++n) for (int t = 0; t < To; ++t) ot[(size_t)t * NO + n] = o_host[(size_t)n * To + t];
  for (int n = 0; n < N; ++n) a_host[n] -= 1;
  int32_t *ques = dev_ints_from(qt, (int64_t)N * Tq), *opts = dev_ints_from(ot, (int64_t)NO * To), *gt = dev_ints_from(a_host, N);

  /* ---- the plug-in pair: encoders/lf-ques.lua (wordEmbed, 2 x SeqLSTM, Linear + Tanh) and decoders/disc.lua (shared table, one SeqLSTM) ---- */
  float *emb = Wf + off[0], *demb = Gf + off[0];
  SeqLSTM ques1 = {E, H, 0, 0, Wf + off[1], Wf + off[2], Gf + off[1], Gf + off[2], NULL, NULL, NULL, NULL};
  SeqLSTM ques2 = {H, H, 0, 0, Wf + off[3], Wf + off[4], Gf + off[3], Gf + off[4], NULL, NULL, NULL, NULL};
  LinearTanh fuse = {H, H, 0, Wf + off[5], Wf + off[6], Gf + off[5], Gf + off[6], NULL, NULL};
  float *Wopt = Wf + off[7], *bopt = Wf + off[8], *dWopt = Gf + off[7], *dbopt = Gf + off[8];

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() (model.lua:68) */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero zeroes the pad row on every forward */

  /* encoder:forward(inputs)  (model.lua:297; encoders/lf-ques.lua) */
  float* qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  lstm_forward(&ques1, qx, Tq, N, ques);
  lstm_forward(&ques2, ques1.h, Tq, N, ques);
  const float* q_last = ques2.h + (int64_t)(Tq - 1) * N * H;     /* nn.Select(1, -1) */
  float* encOut = linear_forward(&fuse, q_last, N);             /* Dropout is the identity in evaluate() */

  /* decoder:forward({options, encOut})  (model.lua:329; decoders/disc.lua): the 100 clones are one batch of N*O sequences;
   * embed -> x*Wx + b is a gather from the table Emb*Wx + b (no dropout on option embeddings, disc.lua:12-14) */
  float* table = dev_floats((int64_t)(V + 1) * 4 * H);
  CHECK(p_gemm_nn(emb, E, Wopt, 4 * H, bopt, table, 4 * H, V + 1, 4 * H, E, 0, NULL));
  float *og = dev_floats((int64_t)To * NO * 4 * H), *oh = dev_floats((int64_t)To * NO * H), *oc = dev_floats((int64_t)To * NO * H);
  CHECK(p_lstm_forward(table, 0, 4 * H, opts, NULL, Wopt + (int64_t)E * 4 * H, NULL, NULL, og, oh, oc, To, NO, H, 0, NULL));
  const float* optH = oh + (int64_t)(To - 1) * NO * H;

  /* criterion:forward + :backward (model.lua:330,334): nn.MM + CrossEntropyCriterion and both gradients in one kernel */
  float *scores = dev_floats((int64_t)N * O), *loss_rows = dev_floats(N), *d_optH = dev_floats((int64_t)NO * H), *d_enc = dev_floats((int64_t)N * H);
  CHECK(p_score_ce(optH, encOut, gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, NULL));

  /* decoder:backward (model.lua:335) */
  float* odc = dev_floats((int64_t)NO * H);
  CHECK(p_lstm_backward(Wopt + (int64_t)E * 4 * H, og, oc, NULL, NULL, d_optH, NULL, odc, NULL, NULL, NULL, To, NO, H, 0, NULL));
  if (To > 1) CHECK(p_gemm_tn_acc(oh, H, og + (int64_t)NO * 4 * H, 4 * H, dWopt + (int64_t)E * 4 * H, 4 * H, H, 4 * H, (To - 1) * NO, 0, NULL));
  void *offs = NULL, *work = NULL, *perm = NULL;                  /* gradient of the gathered table: counting sort + segmented row sum */
  CHECK(p_malloc(&offs, (int64_t)(V + 2) * 4)); CHECK(p_malloc(&work, (int64_t)2 * (V + 1) * 4)); CHECK(p_malloc(&perm, (int64_t)To * NO * 4));
  float* dtab = dev_floats((int64_t)(V + 1) * 4 * H);
  CHECK(p_token_sort(opts, (int64_t)To * NO, V + 1, (int32_t*)offs, (int32_t*)work, (int32_t*)perm, NULL));
  CHECK(p_segment_rowsum_acc(og, 4 * H, opts, (const int32_t*)perm, (int64_t)To * NO, 4 * H, dtab, 4 * H, NULL));
  CHECK(p_colsum_acc(dtab, 4 * H, V + 1, 4 * H, dbopt, NULL));
  CHECK(p_gemm_tn_acc(emb, E, dtab, 4 * H, dWopt, 4 * H, E, 4 * H, V + 1, 0, NULL));
  CHECK(p_gemm_nt(dtab, 4 * H, Wopt, 4 * H, NULL, demb, E, V + 1, E, 4 * H, VD_ACT_NONE, 2, NULL));       /* dEmb += dTable * Wx^T */

  /* encoder:backward(inputs, gradEncOut)  (model.lua:337) */
  float* dq_last = linear_backward(&fuse, d_enc);
  float* dh1_seq = lstm_backward(&ques2, NULL, dq_last, 1);      /* gradient arrives at the last step only */
  float* dqx = lstm_backward(&ques1, dh1_seq, NULL, 1);
  CHECK(p_embed_scatter_acc(demb, ques, NULL, dqx, (int64_t)Tq * N, E, 1.f, NULL));

  /* curLoss, then wrapperdW:clamp(-5,5) + adam(wrapperW, wrapperdW, optims) (model.lua:96-99; optim_updates.lua:62-91), t = 1 */
  float* lr_host = (float*)malloc((size_t)N * 4);
  CHECK(p_sync(NULL));
  CHECK(p_d2h(lr_host, loss_rows, (int64_t)N * 4, NULL));
  double loss = 0;
  for (int n = 0; n < N; ++n) loss += lr_host[n];
  loss /= N;
  float* grad_host = (float*)malloc((size_t)numel * 4);
  CHECK(p_d2h(grad_host, Gf, numel * 4, NULL));
  const double lr = 1e-3, t = 1.0;
  const float step = (float)(lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
  CHECK(p_clamp_adam(Wf, Gf, Mf, Vf, numel, 1.0f, 5.0f, 0.9f, 0.999f, 1e-8f, step, NULL));
  CHECK(p_sync(NULL));
  CHECK(p_d2h(host, Wf, numel * 4, NULL));

  FILE* o = fopen(argv[3], "wb");
  if (!o) { perror(argv[3]); return 4; }
  const float lossf = (float)loss;
  fwrite(&lossf, 4, 1, o);
  for (int i = 0; i < 9; ++i) fwrite(grad_host + off[i], 4, (size_t)sizes[i], o);
  for (int i = 0; i < 9; ++i) fwrite(host + off[i], 4, (size_t)sizes[i], o);
  fclose(o);
  printf("lf-ques + disc through the operator-level ABI: loss %.6f, %lld parameters\n", loss, (long long)numel);
  return 0;
}
