/* host_c_plugin_hre.c -- BASELINE.json configs[2], encoders/hre-ques-im-hist.lua (its image-less sibling hre-ques-hist.lua and its attention sibling hrea-ques-im-hist.lua) + decoders/disc.lua, written against the OPERATOR-LEVEL
 * C ABI by a host without a tensor library, module object by module object of the reference's graph (hre-ques-im-hist.lua:5-97) --
 *   history  : wordEmbed -> numLayers x SeqLSTM:maskZero() -> Select(1,-1)                                              hre:28-41
 *   question : JoinTable{wordEmbed(ques), MaskTime(Linear(F, imgEmbedSize)(img))} -> numLayers x SeqLSTM:maskZero() -> Select   hre:43-82
 *   dialog   : JoinTable{question state, history state} viewed [B x R x 2H], transposed to round-major, SeqLSTM(2H, H) over the R
 *              rounds of every dialog (no maskZero), transposed back                                                    hre:84-95
 * and decoders/disc.lua (shared embedding table x option LSTM) + CrossEntropy + clamp / adam.  lua/encoders/hre-ques-im-hist.lua is this
 * file in Lua; this one is built with gcc and checked on the GPU against the library's model-level implementation of the pair
 * (tests/test_abi_c_host.py).
 *
 *   ./host_c_plugin_hre <libvisdial_hip.so> <in.bin> <out.bin>
 * hrea-ques-im-hist.lua:83-131: every question attends over the history states of the rounds up to its own (two Linear(H, 1) scores,
 * MaskFuture, ReplaceZero(-inf), SoftMax, weighted sum) and the attended history replaces the history state in front of the dialog LSTM.
 * in.bin : int32 {V, E, H, F, DI, B, R, O, Tq, Th, To, useIm, attention}, the 15 (13 without the image part, 19 with attention) parameter tensors in getParameters() order as float32 (embed, hist1.W/.b,
 *          hist2.W/.b, img_embed.W [DI x F]/.b, ques1.W [(E+DI+H) x 4H]/.b, ques2.W/.b, dialog.W [(2H+H) x 4H]/.b, opt.W/.b), ques_fwd
 *          [B*R x Tq], img_feat [B x F] float32, hist [B*R x Th], options [B*R*O x To], answer_ind [B*R] (1-based).
 * out.bin: float32 loss, the flat gradient, the flat parameters after clamp(-5,5) + adam.   Dropout off (wrapper:evaluate()). */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

#define NT 19
#define NL 2

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  load_entry_points(argv[1]);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[13];
  if (fread(hd, 4, 13, f) != 13) return 4;
  const int V = hd[0], E = hd[1], H = hd[2], F = hd[3], B = hd[5], R = hd[6], O = hd[7], Tq = hd[8], Th = hd[9], To = hd[10], useIm = hd[11],
            attention = hd[12];
  const int DI = useIm ? hd[4] : 0;
  const int N = B * R, NO = N * O, DQ = E + DI;

  const int64_t l2 = (int64_t)2 * H * 4 * H;
  int64_t sizes[NT];
  int nt = 0, H1W, H2W, IEW = -1, Q1W, Q2W, AQW = -1, AHW = -1, DLW, OPW;
  sizes[nt++] = (int64_t)(V + 1) * E;                                             /* embed */
  H1W = nt; sizes[nt++] = (int64_t)(E + H) * 4 * H; sizes[nt++] = 4 * H; H2W = nt; sizes[nt++] = l2; sizes[nt++] = 4 * H;
  if (useIm) { IEW = nt; sizes[nt++] = (int64_t)DI * F; sizes[nt++] = DI; }
  Q1W = nt; sizes[nt++] = (int64_t)(DQ + H) * 4 * H; sizes[nt++] = 4 * H; Q2W = nt; sizes[nt++] = l2; sizes[nt++] = 4 * H;
  if (attention) { AQW = nt; sizes[nt++] = H; sizes[nt++] = 1; AHW = nt; sizes[nt++] = H; sizes[nt++] = 1; }     /* two nn.Linear(H, 1) (hrea:83-85) */
  DLW = nt; sizes[nt++] = (int64_t)(2 * H + H) * 4 * H; sizes[nt++] = 4 * H;
  OPW = nt; sizes[nt++] = (int64_t)(E + H) * 4 * H; sizes[nt++] = 4 * H;
  int64_t off[NT + 1];
  off[0] = 0;
  for (int i = 0; i < nt; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[nt];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < nt; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  const size_t nq = (size_t)N * Tq, nh = (size_t)N * Th, ni = useIm ? (size_t)B * F : 0, no = (size_t)NO * To;
  int32_t *q_host = (int32_t*)malloc(nq * 4), *h_host = (int32_t*)malloc(nh * 4), *o_host = (int32_t*)malloc(no * 4), *a_host = (int32_t*)malloc((size_t)N * 4);
  float* i_host = (float*)malloc((ni ? ni : 1) * 4);
  if (fread(q_host, 4, nq, f) != nq || (useIm && fread(i_host, 4, ni, f) != ni) || fread(h_host, 4, nh, f) != nh || fread(o_host, 4, no, f) != no ||
      fread(a_host, 4, (size_t)N, f) != (size_t)N) return 4;
  fclose(f);

  /* ---- input re-layout of model.lua:252-279 + the row permutations of nn.View / nn.Transpose (hre:88-93) ---- */
  for (int n = 0; n < N; ++n) a_host[n] -= 1;
  int32_t *ques = time_major(q_host, N, Tq), *hist = time_major(h_host, N, Th), *opts = time_major(o_host, NO, To), *gt = dev_ints_from(a_host, N);
  float* img = dev_floats((int64_t)ni);
  if (useIm) CHECK(p_h2d(img, i_host, (int64_t)ni * 4, NULL));
  int32_t *rep_h = (int32_t*)malloc((size_t)N * 4), *rb_h = (int32_t*)malloc((size_t)N * 4), *n_h = (int32_t*)malloc((size_t)N * 4);
  for (int n = 0; n < N; ++n) {
    rep_h[n] = n / R;                       /* image row of QA round n = b * R + r */
    rb_h[n] = (n % B) * R + n / B;          /* round-major row r * B + b  <-  dialog-major row b * R + r */
    n_h[n] = (n % R) * B + n / R;           /* and back */
  }
  int32_t *rep = dev_ints_from(rep_h, N), *to_rb = dev_ints_from(rb_h, N), *to_n = dev_ints_from(n_h, N);

#define LSTM(D, w, b) {D, H, 0, 0, Wf + off[w], Wf + off[b], Gf + off[w], Gf + off[b], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL}
  float *emb = Wf + off[0], *demb = Gf + off[0];
  SeqLSTM hist_rnn[NL] = {LSTM(E, H1W, H1W + 1), LSTM(H, H2W, H2W + 1)};
  SeqLSTM ques_rnn[NL] = {LSTM(DQ, Q1W, Q1W + 1), LSTM(H, Q2W, Q2W + 1)};
  SeqLSTM dialog = LSTM(2 * H, DLW, DLW + 1);
  const int iw = useIm ? IEW : 0;                                                                  /* (unused without the image part) */
  LinearTanh img_embed = {F, DI, 0, Wf + off[iw], Wf + off[iw + 1], Gf + off[iw], Gf + off[iw + 1], NULL, NULL, 1};   /* plain nn.Linear (hre:46) */
  DiscDecoder dec = {V, E, H, 0, 0, emb, demb, Wf + off[OPW], Wf + off[OPW + 1], Gf + off[OPW], Gf + off[OPW + 1], NULL, NULL, NULL, NULL, NULL};

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero: pad row */

  /* ================= encoder:forward({ques, img, hist})  (model.lua:297) ================= */
  float* hx = dev_floats((int64_t)Th * N * E);
  CHECK(p_embed_gather(emb, hist, NULL, hx, (int64_t)Th * N, E, 1.f, NULL));
  const float* x = hx;
  for (int l = 0; l < NL; ++l) { lstm_forward(&hist_rnn[l], x, Th, N, hist); x = hist_rnn[l].h; }
  const float* hh = hist_rnn[NL - 1].h + (int64_t)(Th - 1) * N * H;
  /* question branch: the image embedding is repeated over the time steps of its round, zero at pad steps (MaskTime), and joined to the
   * word embedding column-wise */
  float* qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  const float* qcat = qx;                                                                        /* hre-ques-hist.lua: the word embedding alone */
  if (useIm) {
    float *img_rep = dev_floats((int64_t)N * F), *xi = dev_floats((int64_t)Tq * N * DI), *cat = dev_floats((int64_t)Tq * N * DQ);
    CHECK(p_embed_gather(img, rep, NULL, img_rep, N, F, 1.f, NULL));
    const float* imgE = linear_forward(&img_embed, img_rep, N);                                   /* hre:43-48 */
    CHECK(p_mask_time_forward(imgE, ques, xi, Tq, N, DI, NULL));                                  /* hre:50-53 */
    CHECK(p_copy_2d(cat, DQ, qx, E, (int64_t)Tq * N, E, NULL));                                   /* nn.JoinTable(2, 2) */
    CHECK(p_copy_2d(cat + E, DQ, xi, DI, (int64_t)Tq * N, DI, NULL));
    qcat = cat;
  }
  x = qcat;
  for (int l = 0; l < NL; ++l) { lstm_forward(&ques_rnn[l], x, Tq, N, ques); x = ques_rnn[l].h; }
  const float* hq = ques_rnn[NL - 1].h + (int64_t)(Tq - 1) * N * H;
  /* dialog-level recurrence over the rounds (hre:84-95): rows to round-major, JoinTable{question, history}, SeqLSTM(2H, H), rows back */
  const float *first = hq, *second = hh;
  float *sq = NULL, *sh = NULL, *P = NULL;
  if (attention) {                                                                               /* hrea:83-131; JoinTable{attended history, question} */
    sq = dev_floats(N); sh = dev_floats(N); P = dev_floats((int64_t)N * R);
    float* att = dev_floats((int64_t)N * H);
    CHECK(p_rowdot_forward(hq, Wf + off[AQW], Wf + off[AQW + 1], sq, N, H, NULL));
    CHECK(p_rowdot_forward(hh, Wf + off[AHW], Wf + off[AHW + 1], sh, N, H, NULL));
    CHECK(p_hrea_attention_forward(sq, sh, hh, P, att, B, R, H, NULL));
    first = att; second = hq;
  }
  float *f_rb = dev_floats((int64_t)N * H), *s_rb = dev_floats((int64_t)N * H), *dcat = dev_floats((int64_t)N * 2 * H);
  CHECK(p_embed_gather(first, to_rb, NULL, f_rb, N, H, 1.f, NULL));
  CHECK(p_embed_gather(second, to_rb, NULL, s_rb, N, H, 1.f, NULL));
  CHECK(p_copy_2d(dcat, 2 * H, f_rb, H, N, H, NULL));
  CHECK(p_copy_2d(dcat + H, 2 * H, s_rb, H, N, H, NULL));
  lstm_forward(&dialog, dcat, R, B, NULL);
  float* encOut = dev_floats((int64_t)N * H);
  CHECK(p_embed_gather(dialog.h, to_n, NULL, encOut, N, H, 1.f, NULL));

  /* ================= decoder:forward, criterion, decoder:backward  (model.lua:329-335) ================= */
  const float* optH = disc_forward(&dec, opts, To, NO);
  float *scores = dev_floats((int64_t)N * O), *loss_rows = dev_floats(N), *d_optH = dev_floats((int64_t)NO * H), *d_enc = dev_floats((int64_t)N * H);
  CHECK(p_score_ce(optH, encOut, gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, NULL));
  disc_backward(&dec, d_optH);

  /* ================= encoder:backward(inputs, gradEncOut)  (model.lua:337) ================= */
  float* g_rb = dev_floats((int64_t)N * H);
  CHECK(p_embed_gather(d_enc, to_rb, NULL, g_rb, N, H, 1.f, NULL));
  float* ddcat = lstm_backward(&dialog, g_rb, NULL, 1);                                         /* [R*B x 2H]: the gradient arrives at every round */
  float *df_rb = dev_floats((int64_t)N * H), *ds_rb = dev_floats((int64_t)N * H), *dq = dev_floats((int64_t)N * H), *dh = dev_floats((int64_t)N * H);
  CHECK(p_copy_2d(df_rb, H, ddcat, 2 * H, N, H, NULL));
  CHECK(p_copy_2d(ds_rb, H, ddcat + H, 2 * H, N, H, NULL));
  CHECK(p_embed_gather(df_rb, to_n, NULL, dq, N, H, 1.f, NULL));
  CHECK(p_embed_gather(ds_rb, to_n, NULL, dh, N, H, 1.f, NULL));
  if (attention) {                                               /* dq holds d attended history, dh holds d question state (JoinTable order) */
    float *dsq = dev_floats(N), *dsh = dev_floats(N), *dh_att = dev_floats((int64_t)N * H), *dq_s = dev_floats((int64_t)N * H), *dh_s = dev_floats((int64_t)N * H);
    CHECK(p_hrea_attention_backward(hh, P, dq, dsq, dsh, dh_att, B, R, H, NULL));
    CHECK(p_rowdot_backward(hq, Wf + off[AQW], dsq, Gf + off[AQW], Gf + off[AQW + 1], dq_s, N, H, NULL));
    CHECK(p_rowdot_backward(hh, Wf + off[AHW], dsh, Gf + off[AHW], Gf + off[AHW + 1], dh_s, N, H, NULL));
    float *dq2 = dev_floats((int64_t)N * H), *dh2 = dev_floats((int64_t)N * H);
    CHECK(p_axpby(dh, dq_s, dq2, (int64_t)N * H, 1.f, 1.f, NULL));
    CHECK(p_axpby(dh_att, dh_s, dh2, (int64_t)N * H, 1.f, 1.f, NULL));
    dq = dq2; dh = dh2;
  }
  float* dhx = lstm_backward(&hist_rnn[NL - 1], NULL, dh, 1);
  for (int l = NL - 2; l >= 0; --l) dhx = lstm_backward(&hist_rnn[l], dhx, NULL, 1);
  CHECK(p_embed_scatter_acc(demb, hist, NULL, dhx, (int64_t)Th * N, E, 1.f, NULL));
  float* dqcat = lstm_backward(&ques_rnn[NL - 1], NULL, dq, 1);
  for (int l = NL - 2; l >= 0; --l) dqcat = lstm_backward(&ques_rnn[l], dqcat, NULL, 1);
  if (useIm) {
    float *dqx = dev_floats((int64_t)Tq * N * E), *dxi = dev_floats((int64_t)Tq * N * DI), *dimgE = dev_floats((int64_t)N * DI);
    CHECK(p_copy_2d(dqx, E, dqcat, DQ, (int64_t)Tq * N, E, NULL));                                /* JoinTable backward */
    CHECK(p_copy_2d(dxi, DI, dqcat + E, DQ, (int64_t)Tq * N, DI, NULL));
    CHECK(p_embed_scatter_acc(demb, ques, NULL, dqx, (int64_t)Tq * N, E, 1.f, NULL));
    CHECK(p_mask_time_backward(dxi, ques, dimgE, Tq, N, DI, NULL));
    linear_backward_ex(&img_embed, dimgE, 0);
  } else {
    CHECK(p_embed_scatter_acc(demb, ques, NULL, dqcat, (int64_t)Tq * N, E, 1.f, NULL));
  }

  /* curLoss, then wrapperdW:clamp(-5,5) + adam (model.lua:96-99; optim_updates.lua:62-91), t = 1 */
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
  for (int i = 0; i < nt; ++i) fwrite(grad_host + off[i], 4, (size_t)sizes[i], o);
  for (int i = 0; i < nt; ++i) fwrite(host + off[i], 4, (size_t)sizes[i], o);
  fclose(o);
  printf("%s + disc through the operator-level ABI: loss %.6f, %lld parameters\n", attention ? "hrea-ques-im-hist" : useIm ? "hre-ques-im-hist" : "hre-ques-hist", loss, (long long)numel);
  return 0;
}
