/* host_c_plugin_mn_att.c -- the FLAGSHIP plug-in pair, encoders/mn-att-ques-im-hist.lua + decoders/disc.lua, written against the
 * OPERATOR-LEVEL C ABI by a host that has no tensor library: every node of the reference's nngraph (mn-att:21-106) is one module
 * object or one entry point of include/visdial_hip.h, in the order model.lua:297-337 drives them --
 *   shared LookupTableMaskZero -> Dropout(0.5) -> 2 x SeqLSTM:maskZero() per text branch -> Select(1,-1)          mn-att:21-45
 *   nn.MM(false,true) -> MaskSoftMax -> nn.MM -> Tanh(Linear(Dropout)) -> Tanh(Linear(hAttTr + q))                 mn-att:48-65
 *   img_tr = Dropout(Tanh(Linear(img))); p = SoftMax(Linear(Dropout(Tanh(Linear(img_tr) + Linear(u))))); u += p . img_tr   :68-104
 *   Tanh(Linear(Dropout(u)))                                                                                      mn-att:106
 * with flat parameter / gradient vectors like wrapper:getParameters() (model.lua:55).  lua/encoders/mn-att-ques-im-hist.lua is this
 * file in Lua (same objects, same calls, same order); this one is built with gcc and checked on the GPU: its loss, every gradient
 * tensor and the post-Adam parameters against the library's own model-level implementation of the pair (tests/test_abi_c_host.py).
 *
 *   gcc -O2 -std=c99 -I include -I examples examples/host_c_plugin_mn_att.c -ldl -lm -o host_c_plugin_mn_att
 *   ./host_c_plugin_mn_att <libvisdial_hip.so> <in.bin> <out.bin> [train]
 * in.bin : int32 {V, E, H, C, S, K, B, R, O, Tq, Th, To}, the 25 parameter tensors in getParameters() order as float32 (embed,
 *          hist1.W/.b, hist2.W/.b, ques1.W/.b, ques2.W/.b, mn1, mn2, img_proj, img_common, ques_common, att, out, opt), then
 *          ques_fwd [B*R x Tq], hist [B*R x Th] (int32, right-aligned), img_feat [B x S x S x C] float32, options [B*R*O x To],
 *          answer_ind [B*R] (1-based).
 * out.bin: float32 loss, the flat gradient, the flat parameters after clamp(-5,5) + adam.
 * Without `train` Dropout is the identity (wrapper:evaluate(); what the comparison side does); with it the seven Dropout(0.5) nodes
 * draw masks (vd_dropout_mask) and the masks ride through the fused kernels exactly as in the library. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

#define NT 25
static const float P5 = 0.5f, S5 = 2.0f;      /* the nngraph encoders hard-code Dropout(0.5) (mn-att:24,25,64,74,92,106) */
static int g_train = 0;
static uint64_t g_seed = 1234;

static uint8_t* drop_mask(int64_t n) {          /* nn.Dropout(0.5) noise of this forward; NULL = identity (evaluate) */
  if (!g_train) return NULL;
  void* p = NULL;
  CHECK(p_malloc(&p, n));
  CHECK(p_dropout_mask((uint8_t*)p, n, g_seed++, P5, NULL));
  return (uint8_t*)p;
}
static const float* drop_apply(const float* x, const uint8_t* mask, int64_t n) {   /* forward and backward of nn.Dropout */
  if (!mask) return x;
  float* y = dev_floats(n);
  CHECK(p_dropout_apply(x, mask, y, n, S5, NULL));
  return y;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin> [train]\n", argv[0]); return 1; }
  g_train = argc > 4 && strcmp(argv[4], "train") == 0;
  load_entry_points(argv[1]);

  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[12];
  if (fread(hd, 4, 12, f) != 12) return 4;
  const int V = hd[0], E = hd[1], H = hd[2], C = hd[3], S = hd[4], K = hd[5], B = hd[6], R = hd[7], O = hd[8], Tq = hd[9], Th = hd[10], To = hd[11];
  const int N = B * R, NO = N * O, S2 = S * S;

  /* ---- wrapper:getParameters(): flat W / dW (+ Adam m, v), every tensor 16-byte aligned (model.lua:55) ---- */
  enum { EMBED, HIST1_W, HIST1_B, HIST2_W, HIST2_B, QUES1_W, QUES1_B, QUES2_W, QUES2_B, MN1_W, MN1_B, MN2_W, MN2_B, IMGP_W, IMGP_B,
         IMGC_W, IMGC_B, QC_W, QC_B, ATT_W, ATT_B, OUT_W, OUT_B, OPT_W, OPT_B };
  const int64_t lstm1 = (int64_t)(E + H) * 4 * H, lstm2 = (int64_t)2 * H * 4 * H;
  const int64_t sizes[NT] = {(int64_t)(V + 1) * E, lstm1, 4 * H, lstm2, 4 * H, lstm1, 4 * H, lstm2, 4 * H, (int64_t)H * H, H, (int64_t)H * H, H,
                             (int64_t)H * C, H, (int64_t)K * H, K, (int64_t)K * H, K, K, 1, (int64_t)H * H, H, lstm1, 4 * H};
  int64_t off[NT + 1];
  off[0] = 0;
  for (int i = 0; i < NT; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[NT];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < NT; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  const size_t nq = (size_t)N * Tq, nh = (size_t)N * Th, ni = (size_t)B * S2 * C, no = (size_t)NO * To;
  int32_t *q_host = (int32_t*)malloc(nq * 4), *h_host = (int32_t*)malloc(nh * 4), *o_host = (int32_t*)malloc(no * 4), *a_host = (int32_t*)malloc((size_t)N * 4);
  float* i_host = (float*)malloc(ni * 4);
  if (fread(q_host, 4, nq, f) != nq || fread(h_host, 4, nh, f) != nh || fread(i_host, 4, ni, f) != ni || fread(o_host, 4, no, f) != no ||
      fread(a_host, 4, (size_t)N, f) != (size_t)N) return 4;
  fclose(f);

  /* ---- input re-layout of model.lua:255-294: time-major token matrices, ONE feature map per image (the 10x repeatTensor of
   *      model.lua:262-265 is folded into the attention kernels' loaders), the causal mask of the memory attention (1 = hidden,
   *      model.lua:280-294: round i attends to facts j <= i), 0-based targets ---- */
  for (int n = 0; n < N; ++n) a_host[n] -= 1;
  int32_t *ques = time_major(q_host, N, Tq), *hist = time_major(h_host, N, Th), *opts = time_major(o_host, NO, To), *gt = dev_ints_from(a_host, N);
  float* img = dev_floats((int64_t)ni);
  CHECK(p_h2d(img, i_host, (int64_t)ni * 4, NULL));
  uint8_t* mask_host = (uint8_t*)malloc((size_t)N * R);
  for (int b = 0; b < B; ++b) for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) mask_host[((size_t)b * R + i) * R + j] = j > i;
  void* att_mask = NULL;
  CHECK(p_malloc(&att_mask, (int64_t)N * R));
  CHECK(p_h2d(att_mask, mask_host, (int64_t)N * R, NULL));

  /* ---- module objects over the flat vectors ---- */
#define LSTM(D, w, b) {D, H, 0, 0, Wf + off[w], Wf + off[b], Gf + off[w], Gf + off[b], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL}
#define LIN(nin, nout, w, b, plain) {nin, nout, 0, Wf + off[w], Wf + off[b], Gf + off[w], Gf + off[b], NULL, NULL, plain}
  float *emb = Wf + off[EMBED], *demb = Gf + off[EMBED];
  SeqLSTM hist1 = LSTM(E, HIST1_W, HIST1_B), hist2 = LSTM(H, HIST2_W, HIST2_B), ques1 = LSTM(E, QUES1_W, QUES1_B), ques2 = LSTM(H, QUES2_W, QUES2_B);
  LinearTanh mn1 = LIN(H, H, MN1_W, MN1_B, 0), mn2 = LIN(H, H, MN2_W, MN2_B, 0), img_proj = LIN(C, H, IMGP_W, IMGP_B, 0);
  LinearTanh ques_common = LIN(H, K, QC_W, QC_B, 1), out = LIN(H, H, OUT_W, OUT_B, 0);
  DiscDecoder dec = {V, E, H, 0, 0, emb, demb, Wf + off[OPT_W], Wf + off[OPT_B], Gf + off[OPT_W], Gf + off[OPT_B], NULL, NULL, NULL, NULL, NULL};

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() (model.lua:68) */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero zeroes the pad row on every forward */

  /* ================= encoder:forward({ques, img, hist, mask})  (model.lua:297) ================= */
  /* text branches (mn-att:21-45): embedding + Dropout fused in the gather; maskZero via the token matrix */
  uint8_t *m_h = drop_mask((int64_t)Th * N * E), *m_q = drop_mask((int64_t)Tq * N * E);
  float *hx = dev_floats((int64_t)Th * N * E), *qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, hist, m_h, hx, (int64_t)Th * N, E, m_h ? S5 : 1.f, NULL));
  CHECK(p_embed_gather(emb, ques, m_q, qx, (int64_t)Tq * N, E, m_q ? S5 : 1.f, NULL));
  lstm_forward(&hist1, hx, Th, N, hist); lstm_forward(&hist2, hist1.h, Th, N, hist);
  lstm_forward(&ques1, qx, Tq, N, ques); lstm_forward(&ques2, ques1.h, Tq, N, ques);
  const float* h3 = hist2.h + (int64_t)(Th - 1) * N * H;          /* nn.Select(1, -1) */
  const float* q3 = ques2.h + (int64_t)(Tq - 1) * N * H;
  /* memory network over the dialog's facts (mn-att:48-65) */
  float *prob = dev_floats((int64_t)N * R), *hatt = dev_floats((int64_t)N * H);
  CHECK(p_mn_attention_forward(q3, h3, (const uint8_t*)att_mask, prob, hatt, B, R, H, NULL));
  uint8_t* m_hatt = drop_mask((int64_t)N * H);
  const float* hattTr = linear_forward(&mn1, drop_apply(hatt, m_hatt, (int64_t)N * H), N);
  float* s2 = dev_floats((int64_t)N * H);
  CHECK(p_axpby(hattTr, q3, s2, (int64_t)N * H, 1.f, 1.f, NULL));                                    /* nn.CAddTable */
  const float* qh2 = linear_forward(&mn2, s2, N);
  /* stacked attention over the S x S regions, one hop (mn-att:68-104): per-IMAGE projection, per-round Dropout masks in the loaders */
  const float* pre = linear_forward(&img_proj, img, B * S2);                                          /* Tanh(Linear(img)), pre-Dropout */
  uint8_t *m1 = drop_mask((int64_t)N * S2 * H), *m2 = drop_mask((int64_t)N * S2 * K);
  const float sc = m1 ? S5 : 1.f;
  const float* qc = linear_forward(&ques_common, qh2, N);                                             /* mn-att:88 */
  float *iqc = dev_floats((int64_t)N * S2 * K), *patt = dev_floats((int64_t)N * S2), *u1 = dev_floats((int64_t)N * H);
  CHECK(p_img_common_forward(pre, m1, Wf + off[IMGC_W], Wf + off[IMGC_B], qc, m2, iqc, N, R, S2, H, K, sc, NULL));       /* mn-att:83-92 */
  CHECK(p_img_att_forward(iqc, Wf + off[ATT_W], Wf + off[ATT_B], pre, m1, qh2, patt, u1, N, R, S2, H, K, sc, NULL));     /* mn-att:93-102 */
  uint8_t* m_u = drop_mask((int64_t)N * H);
  float* encOut = linear_forward(&out, drop_apply(u1, m_u, (int64_t)N * H), N);                        /* mn-att:106 */

  /* ================= decoder:forward, criterion, decoder:backward  (model.lua:329-335) ================= */
  const float* optH = disc_forward(&dec, opts, To, NO);
  float *scores = dev_floats((int64_t)N * O), *loss_rows = dev_floats(N), *d_optH = dev_floats((int64_t)NO * H), *d_enc = dev_floats((int64_t)N * H);
  CHECK(p_score_ce(optH, encOut, gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, NULL));
  disc_backward(&dec, d_optH);

  /* ================= encoder:backward(inputs, gradEncOut)  (model.lua:337) ================= */
  const float* du = drop_apply(linear_backward(&out, d_enc), m_u, (int64_t)N * H);      /* d att of the hop + its residual */
  float *dpre = dev_floats((int64_t)B * S2 * H), *dqc = dev_floats((int64_t)N * K), *work = dev_floats((int64_t)N * S2);
  CHECK(p_img_att_backward(iqc, Wf + off[ATT_W], pre, m1, m2, patt, du, Gf + off[ATT_W], Gf + off[ATT_B], dqc, work, N, R, S2, H, K, sc, NULL));
  const float* dz = iqc;                                                               /* iqc now holds dz */
  CHECK(p_colsum_acc(dz, K, N * S2, K, Gf + off[IMGC_B], NULL));
  CHECK(p_img_common_wgrad(dz, pre, m1, Gf + off[IMGC_W], N, R, S2, H, K, sc, NULL));
  CHECK(p_img_tr_backward(dz, Wf + off[IMGC_W], patt, du, m1, dpre, N, R, S2, H, K, sc, NULL));             /* += into dpre */
  const float* du_q = linear_backward(&ques_common, dqc);
  float* dqh2 = dev_floats((int64_t)N * H);
  CHECK(p_axpby(du_q, du, dqh2, (int64_t)N * H, 1.f, 1.f, NULL));                       /* residual CAddTable (mn-att:102) */
  linear_backward_ex(&img_proj, dpre, 0);                                              /* tanh' + dW, db of mn-att:77 */
  /* memory block */
  const float* ds2 = linear_backward(&mn2, dqh2);
  const float* dhatt = drop_apply(linear_backward(&mn1, ds2), m_hatt, (int64_t)N * H);
  float *dq_att = dev_floats((int64_t)N * H), *dh3 = dev_floats((int64_t)N * H), *dq3 = dev_floats((int64_t)N * H);
  CHECK(p_mn_attention_backward(q3, h3, prob, dhatt, dq_att, dh3, B, R, H, NULL));
  CHECK(p_axpby(dq_att, ds2, dq3, (int64_t)N * H, 1.f, 1.f, NULL));
  /* text branches: the gradient arrives at the last step of the top layers only */
  float* dh1_seq = lstm_backward(&hist2, NULL, dh3, 1);
  float* dhx = lstm_backward(&hist1, dh1_seq, NULL, 1);
  float* dq1_seq = lstm_backward(&ques2, NULL, dq3, 1);
  float* dqx = lstm_backward(&ques1, dq1_seq, NULL, 1);
  CHECK(p_embed_scatter_acc(demb, hist, m_h, dhx, (int64_t)Th * N, E, m_h ? S5 : 1.f, NULL));
  CHECK(p_embed_scatter_acc(demb, ques, m_q, dqx, (int64_t)Tq * N, E, m_q ? S5 : 1.f, NULL));

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
  for (int i = 0; i < NT; ++i) fwrite(grad_host + off[i], 4, (size_t)sizes[i], o);
  for (int i = 0; i < NT; ++i) fwrite(host + off[i], 4, (size_t)sizes[i], o);
  fclose(o);
  printf("mn-att-ques-im-hist + disc through the operator-level ABI (%s): loss %.6f, %lld parameters\n", g_train ? "training" : "evaluate",
         loss, (long long)numel);
  return 0;
}
