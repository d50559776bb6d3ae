/* host_c_plugin_graph.c -- the three nngraph SIBLINGS of the flagship encoder, each + decoders/disc.lua, written against the OPERATOR-LEVEL
 * C ABI by a host without a tensor library (the flagship itself: host_c_plugin_mn_att.c):
 *   variant 1  encoders/mn-ques-hist.lua        text branches -> memory network over the dialog's facts
 *   variant 2  encoders/mn-ques-im-hist.lua     query = Tanh(Linear(JoinTable{question state, image feature})) -> memory network   (:47-48)
 *   variant 3  encoders/lf-att-ques-im-hist.lua u = Tanh(Linear(JoinTable{question state, history state})) -> stacked image attention -> out (:43-86)
 * Same blocks as the flagship, node for node: shared LookupTableMaskZero -> 2 x SeqLSTM:maskZero() per text branch -> Select(1,-1);
 * nn.MM -> MaskSoftMax -> nn.MM -> Tanh(Linear) -> Tanh(Linear(hAttTr + query)); img_tr / img_common / ques_common / att / out.
 * lua/encoders/{mn-ques-hist, mn-ques-im-hist, lf-att-ques-im-hist}.lua are this file in Lua, one variant each; this one is built with gcc
 * and checked on the GPU against the library's model-level implementation of every pair (tests/test_abi_c_host.py).
 *
 *   ./host_c_plugin_graph <libvisdial_hip.so> <in.bin> <out.bin>
 * in.bin : int32 {variant, V, E, H, C, S, K, F, B, R, O, Tq, Th, To}, the parameter tensors in getParameters() order as float32 (embed,
 *          hist1.W/.b, hist2.W/.b, ques1.W/.b, ques2.W/.b, [qi.W/.b | qh.W/.b], [mn1, mn2], [img_proj, img_common, ques_common, att, out],
 *          opt.W/.b), ques_fwd [B*R x Tq], hist [B*R x Th], img_feat (variant 2: [B x F]; variant 3: [B x S x S x C]; variant 1: none)
 *          float32, options [B*R*O x To], answer_ind [B*R] (1-based).
 * out.bin: float32 loss, the flat gradient, the flat parameters after clamp(-5,5) + adam.   Dropout off (wrapper:evaluate()). */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

#define NT 27

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  load_entry_points(argv[1]);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[14];
  if (fread(hd, 4, 14, f) != 14) return 4;
  const int variant = hd[0], V = hd[1], E = hd[2], H = hd[3], C = hd[4], S = hd[5], K = hd[6], F = hd[7], B = hd[8], R = hd[9], O = hd[10],
            Tq = hd[11], Th = hd[12], To = hd[13];
  const int N = B * R, NO = N * O, S2 = S * S;
  const int has_qi = variant == 2, has_qh = variant == 3, has_mem = variant != 3, has_san = variant == 3;

  /* ---- wrapper:getParameters() ---- */
  const int64_t lstm1 = (int64_t)(E + H) * 4 * H, lstm2 = (int64_t)2 * H * 4 * H;
  int64_t sizes[NT], off[NT + 1];
  int nt = 0, H1, H2, Q1, Q2, QI = -1, QH = -1, MN1 = -1, MN2 = -1, IMGP = -1, IMGC = -1, QC = -1, ATT = -1, OUT = -1, OPT;
  sizes[nt++] = (int64_t)(V + 1) * E;
  H1 = nt; sizes[nt++] = lstm1; sizes[nt++] = 4 * H; H2 = nt; sizes[nt++] = lstm2; sizes[nt++] = 4 * H;
  Q1 = nt; sizes[nt++] = lstm1; sizes[nt++] = 4 * H; Q2 = nt; sizes[nt++] = lstm2; sizes[nt++] = 4 * H;
  if (has_qi) { QI = nt; sizes[nt++] = (int64_t)H * (H + F); sizes[nt++] = H; }
  if (has_qh) { QH = nt; sizes[nt++] = (int64_t)H * 2 * H; sizes[nt++] = H; }
  if (has_mem) { MN1 = nt; sizes[nt++] = (int64_t)H * H; sizes[nt++] = H; MN2 = nt; sizes[nt++] = (int64_t)H * H; sizes[nt++] = H; }
  if (has_san) {
    IMGP = nt; sizes[nt++] = (int64_t)H * C; sizes[nt++] = H; IMGC = nt; sizes[nt++] = (int64_t)K * H; sizes[nt++] = K;
    QC = nt; sizes[nt++] = (int64_t)K * H; sizes[nt++] = K; ATT = nt; sizes[nt++] = K; sizes[nt++] = 1; OUT = nt; sizes[nt++] = (int64_t)H * H; sizes[nt++] = H;
  }
  OPT = nt; sizes[nt++] = lstm1; sizes[nt++] = 4 * H;
  off[0] = 0;
  for (int i = 0; i < nt; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[nt];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < nt; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  const size_t nq = (size_t)N * Tq, nh = (size_t)N * Th, no = (size_t)NO * To;
  const size_t ni = has_qi ? (size_t)B * F : has_san ? (size_t)B * S2 * C : 0;
  int32_t *q_host = (int32_t*)malloc(nq * 4), *h_host = (int32_t*)malloc(nh * 4), *o_host = (int32_t*)malloc(no * 4), *a_host = (int32_t*)malloc((size_t)N * 4);
  float* i_host = (float*)malloc((ni ? ni : 1) * 4);
  if (fread(q_host, 4, nq, f) != nq || fread(h_host, 4, nh, f) != nh || (ni && fread(i_host, 4, ni, f) != ni) || fread(o_host, 4, no, f) != no ||
      fread(a_host, 4, (size_t)N, f) != (size_t)N) return 4;
  fclose(f);

  /* ---- input re-layout of model.lua:255-294 ---- */
  for (int n = 0; n < N; ++n) a_host[n] -= 1;
  int32_t *ques = time_major(q_host, N, Tq), *hist = time_major(h_host, N, Th), *opts = time_major(o_host, NO, To), *gt = dev_ints_from(a_host, N);
  float* img = dev_floats((int64_t)ni);
  if (ni) CHECK(p_h2d(img, i_host, (int64_t)ni * 4, NULL));
  uint8_t* mask_host = (uint8_t*)malloc((size_t)N * R);
  for (int b = 0; b < B; ++b) for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) mask_host[((size_t)b * R + i) * R + j] = j > i;
  void* att_mask = NULL;
  CHECK(p_malloc(&att_mask, (int64_t)N * R));
  CHECK(p_h2d(att_mask, mask_host, (int64_t)N * R, NULL));

#define LSTM(D, w) {D, H, 0, 0, Wf + off[w], Wf + off[(w) + 1], Gf + off[w], Gf + off[(w) + 1], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL}
#define LIN(nin, nout, w, plain) {nin, nout, 0, Wf + off[(w) < 0 ? 0 : (w)], Wf + off[(w) < 0 ? 0 : (w) + 1], Gf + off[(w) < 0 ? 0 : (w)], Gf + off[(w) < 0 ? 0 : (w) + 1], NULL, NULL, plain}
  float *emb = Wf + off[0], *demb = Gf + off[0];
  SeqLSTM hist1 = LSTM(E, H1), hist2 = LSTM(H, H2), ques1 = LSTM(E, Q1), ques2 = LSTM(H, Q2);
  LinearTanh qi = LIN(H + F, H, QI, 0), qh = LIN(2 * H, H, QH, 0), mn1 = LIN(H, H, MN1, 0), mn2 = LIN(H, H, MN2, 0);
  LinearTanh img_proj = LIN(C, H, IMGP, 0), ques_common = LIN(H, K, QC, 1), out = LIN(H, H, OUT, 0);
  DiscDecoder dec = {V, E, H, 0, 0, emb, demb, Wf + off[OPT], Wf + off[OPT + 1], Gf + off[OPT], Gf + off[OPT + 1], NULL, NULL, NULL, NULL, NULL};

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero: pad row */

  /* ================= encoder:forward(inputs)  (model.lua:297) ================= */
  /* text branches: shared embedding, two LSTM layers per branch, last step */
  float *hx = dev_floats((int64_t)Th * N * E), *qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, hist, NULL, hx, (int64_t)Th * N, E, 1.f, NULL));
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  lstm_forward(&hist1, hx, Th, N, hist); lstm_forward(&hist2, hist1.h, Th, N, hist);
  lstm_forward(&ques1, qx, Tq, N, ques); lstm_forward(&ques2, ques1.h, Tq, N, ques);
  const float* h3 = hist2.h + (int64_t)(Th - 1) * N * H;          /* nn.Select(1, -1) */
  const float* q3 = ques2.h + (int64_t)(Tq - 1) * N * H;
  const float* query = q3;
  if (has_qi) {                                                   /* mn-ques-im-hist.lua:47-48: the image joins the question before the memory */
    int32_t* rep = (int32_t*)malloc((size_t)N * 4);
    for (int n = 0; n < N; ++n) rep[n] = n / R;
    float *img_rep = dev_floats((int64_t)N * F), *cat = dev_floats((int64_t)N * (H + F));
    CHECK(p_embed_gather(img, dev_ints_from(rep, N), NULL, img_rep, N, F, 1.f, NULL));
    CHECK(p_copy_2d(cat, H + F, q3, H, N, H, NULL));
    CHECK(p_copy_2d(cat + H, H + F, img_rep, F, N, F, NULL));
    query = linear_forward(&qi, cat, N);
  }
  const float* u = NULL;
  float *prob = NULL, *s2 = NULL;
  if (has_mem) {                                                  /* memory network over the dialog's facts (mn-ques-hist.lua:43-58) */
    prob = dev_floats((int64_t)N * R);
    float* hatt = dev_floats((int64_t)N * H);
    CHECK(p_mn_attention_forward(query, h3, (const uint8_t*)att_mask, prob, hatt, B, R, H, NULL));
    const float* hattTr = linear_forward(&mn1, hatt, N);
    s2 = dev_floats((int64_t)N * H);
    CHECK(p_axpby(hattTr, query, s2, (int64_t)N * H, 1.f, 1.f, NULL));                              /* nn.CAddTable */
    u = linear_forward(&mn2, s2, N);
  }
  if (has_qh) {                                                   /* lf-att-ques-im-hist.lua:43: question and history states fused */
    float* cat = dev_floats((int64_t)N * 2 * H);
    CHECK(p_copy_2d(cat, 2 * H, q3, H, N, H, NULL));
    CHECK(p_copy_2d(cat + H, 2 * H, h3, H, N, H, NULL));
    u = linear_forward(&qh, cat, N);
  }
  const float* encOut = u;
  const float* pre = NULL;
  float *iqc = NULL, *patt = NULL;
  if (has_san) {                                                  /* stacked attention over the S x S regions, one hop (lf-att:45-86) */
    pre = linear_forward(&img_proj, img, B * S2);
    const float* qc = linear_forward(&ques_common, u, N);
    iqc = dev_floats((int64_t)N * S2 * K); patt = dev_floats((int64_t)N * S2);
    float* u1 = dev_floats((int64_t)N * H);
    CHECK(p_img_common_forward(pre, NULL, Wf + off[IMGC], Wf + off[IMGC + 1], qc, NULL, iqc, N, R, S2, H, K, 1.f, NULL));
    CHECK(p_img_att_forward(iqc, Wf + off[ATT], Wf + off[ATT + 1], pre, NULL, u, patt, u1, N, R, S2, H, K, 1.f, NULL));
    encOut = linear_forward(&out, u1, N);
  }

  /* ================= decoder:forward, criterion, decoder:backward  (model.lua:329-335) ================= */
  const float* optH = disc_forward(&dec, opts, To, NO);
  float *scores = dev_floats((int64_t)N * O), *loss_rows = dev_floats(N), *d_optH = dev_floats((int64_t)NO * H), *d_enc = dev_floats((int64_t)N * H);
  CHECK(p_score_ce(optH, encOut, gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, NULL));
  disc_backward(&dec, d_optH);

  /* ================= encoder:backward(inputs, gradEncOut)  (model.lua:337) ================= */
  const float* du = d_enc;
  if (has_san) {
    const float* du1 = linear_backward(&out, d_enc);
    float *dpre = dev_floats((int64_t)B * S2 * H), *dqc = dev_floats((int64_t)N * K), *work = dev_floats((int64_t)N * S2);
    CHECK(p_img_att_backward(iqc, Wf + off[ATT], pre, NULL, NULL, patt, du1, Gf + off[ATT], Gf + off[ATT + 1], dqc, work, N, R, S2, H, K, 1.f, NULL));
    CHECK(p_colsum_acc(iqc, K, N * S2, K, Gf + off[IMGC + 1], NULL));                                /* iqc now holds dz */
    CHECK(p_img_common_wgrad(iqc, pre, NULL, Gf + off[IMGC], N, R, S2, H, K, 1.f, NULL));
    CHECK(p_img_tr_backward(iqc, Wf + off[IMGC], patt, du1, NULL, dpre, N, R, S2, H, K, 1.f, NULL));
    const float* du_q = linear_backward(&ques_common, dqc);
    float* dsum = dev_floats((int64_t)N * H);
    CHECK(p_axpby(du_q, du1, dsum, (int64_t)N * H, 1.f, 1.f, NULL));                                 /* residual CAddTable */
    linear_backward_ex(&img_proj, dpre, 0);
    du = dsum;
  }
  const float *dq3 = NULL, *dh3 = NULL;
  if (has_qh) {                                                   /* JoinTable backward */
    const float* dcat = linear_backward(&qh, du);
    float *a = dev_floats((int64_t)N * H), *b = dev_floats((int64_t)N * H);
    CHECK(p_copy_2d(a, H, dcat, 2 * H, N, H, NULL));
    CHECK(p_copy_2d(b, H, dcat + H, 2 * H, N, H, NULL));
    dq3 = a; dh3 = b;
  }
  if (has_mem) {
    const float* ds2 = linear_backward(&mn2, du);
    const float* dhatt = linear_backward(&mn1, ds2);
    float *dq_att = dev_floats((int64_t)N * H), *dh = dev_floats((int64_t)N * H), *dquery = dev_floats((int64_t)N * H);
    CHECK(p_mn_attention_backward(query, h3, prob, dhatt, dq_att, dh, B, R, H, NULL));
    CHECK(p_axpby(dq_att, ds2, dquery, (int64_t)N * H, 1.f, 1.f, NULL));
    dq3 = dquery; dh3 = dh;
    if (has_qi) {                                                 /* the question slice of JoinTable{question, image} (the image needs none) */
      const float* dcat = linear_backward(&qi, dquery);
      float* a = dev_floats((int64_t)N * H);
      CHECK(p_copy_2d(a, H, dcat, H + F, N, H, NULL));
      dq3 = a;
    }
  }
  float* dh1_seq = lstm_backward(&hist2, NULL, dh3, 1);
  float* dhx = lstm_backward(&hist1, dh1_seq, NULL, 1);
  float* dq1_seq = lstm_backward(&ques2, NULL, dq3, 1);
  float* dqx = lstm_backward(&ques1, dq1_seq, NULL, 1);
  CHECK(p_embed_scatter_acc(demb, hist, NULL, dhx, (int64_t)Th * N, E, 1.f, NULL));
  CHECK(p_embed_scatter_acc(demb, ques, NULL, dqx, (int64_t)Tq * N, E, 1.f, NULL));

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
  printf("%s + disc through the operator-level ABI: loss %.6f, %lld parameters\n",
         variant == 1 ? "mn-ques-hist" : variant == 2 ? "mn-ques-im-hist" : "lf-att-ques-im-hist", loss, (long long)numel);
  return 0;
}
