/* host_c_plugin_lf_ques_gen.c -- BASELINE.json configs[0], the reference's own CPU-runnable pair (`th train.lua -encoder lf-ques
 * -decoder gen -gpuid -1`), and configs[1] (`-encoder lf-ques-im-hist -decoder gen`), written against the OPERATOR-LEVEL C ABI by a host
 * without a tensor library: encoders/lf-ques.lua | lf-ques-im-hist.lua + decoders/gen.lua + the generative criterion and the optimiser of
 * model.lua, module object by module object --
 *   encoder  : wordEmbed -> numLayers x SeqLSTM:maskZero() -> Select(1,-1) -> Dropout -> Linear -> Tanh           (lf-ques.lua:6-33)
 *              lf-ques-im-hist: JoinTable{question state, image feature (repeated per round), history-LSTM state} in front of the
 *              Dropout -> Linear -> Tanh                                                               (lf-ques-im-hist.lua:19-58)
 *   connect  : decoder layer i starts from the encoder layer's final (h, c); the TOP layer's h from encOut       (gen.lua:30-42)
 *   decoder  : shared wordEmbed(answer_in) -> numLayers x SeqLSTM:maskZero() -> Linear(H, V) -> LogSoftMax       (gen.lua:8-27)
 *   criterion: SUM over non-pad targets of -log p(answer_out)                                                   (model.lua:32-36)
 *   backward : decoder:backward -> backwardConnect (cell / hidden gradients back into the encoder layers, gen.lua:45-60) ->
 *              encoder:backward(inputs, gradDecOut)                                                             (model.lua:318-324)
 * lua/decoders/gen.lua + lua/encoders/lf-ques.lua + lua/model_ops.lua are this file in Lua; this one is built with gcc and checked on
 * the GPU against the library's model-level implementation of the pair (tests/test_abi_c_host.py).
 *
 *   ./host_c_plugin_lf_ques_gen <libvisdial_hip.so> <in.bin> <out.bin>
 * in.bin : int32 {V, E, H, B, R, Tq, Ta, imHist (bit 0: image part, bit 1: history part), F, Th}, the parameter tensors in getParameters() order as float32 (embed, ques1.W/.b,
 *          ques2.W/.b, [hist1.W/.b, hist2.W/.b,] fuse.W/.b, dec1.W/.b, dec2.W/.b, vocab.W [V x H], vocab.b [V]), ques_fwd [B*R x Tq],
 *          [img_feat [B x F] float32, hist [B*R x Th],] answer_in [B*R x Ta], answer_out [B*R x Ta] (int32; answers left-aligned, 0 = pad).
 * out.bin: float32 loss (the SUM), the flat gradient, the flat parameters after clamp(-5,5) + adam.   Dropout off (wrapper:evaluate()). */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

#define NT 17                                  /* 13 tensors for lf-ques, 17 with the history stack */
#define NL 2                                   /* numLayers (opts.lua:22 default) */

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  load_entry_points(argv[1]);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[10];
  if (fread(hd, 4, 10, f) != 10) return 4;
  const int V = hd[0], E = hd[1], H = hd[2], B = hd[3], R = hd[4], Tq = hd[5], Ta = hd[6], imHist = hd[7], F = hd[8], Th = hd[9];
  const int useIm = imHist & 1, useHist = (imHist >> 1) & 1;      /* the capability flags the reference derives from the encoder NAME (opts.lua:54-59) */
  const int N = B * R, Vp = (V + 3) / 4 * 4, Dcat = H + (useIm ? F : 0) + (useHist ? H : 0);

  /* ---- wrapper:getParameters() ---- */
  const int64_t l1 = (int64_t)(E + H) * 4 * H, l2 = (int64_t)2 * H * 4 * H;
  int64_t sizes[NT], off[NT + 1];
  int nt = 0, Q1W, Q2W, H1W = -1, H2W = -1, FW, D1W, D2W, VW;
  sizes[nt++] = (int64_t)(V + 1) * E;                                             /* embed */
  Q1W = nt; sizes[nt++] = l1; sizes[nt++] = 4 * H; Q2W = nt; sizes[nt++] = l2; sizes[nt++] = 4 * H;
  if (useHist) { H1W = nt; sizes[nt++] = l1; sizes[nt++] = 4 * H; H2W = nt; sizes[nt++] = l2; sizes[nt++] = 4 * H; }
  FW = nt; sizes[nt++] = (int64_t)H * Dcat; sizes[nt++] = H;
  D1W = nt; sizes[nt++] = l1; sizes[nt++] = 4 * H; D2W = nt; sizes[nt++] = l2; sizes[nt++] = 4 * H;
  VW = nt; sizes[nt++] = (int64_t)V * H; sizes[nt++] = V;
  off[0] = 0;
  for (int i = 0; i < nt; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[nt];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < nt; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  const size_t nq = (size_t)N * Tq, na = (size_t)N * Ta, nh = useHist ? (size_t)N * Th : 0, ni = useIm ? (size_t)B * F : 0;
  int32_t *q_host = (int32_t*)malloc(nq * 4), *ai_host = (int32_t*)malloc(na * 4), *ao_host = (int32_t*)malloc(na * 4);
  int32_t* h_host = (int32_t*)malloc((nh ? nh : 1) * 4);
  float* i_host = (float*)malloc((ni ? ni : 1) * 4);
  if (fread(q_host, 4, nq, f) != nq) return 4;
  if (useIm && fread(i_host, 4, ni, f) != ni) return 4;
  if (useHist && fread(h_host, 4, nh, f) != nh) return 4;
  if (fread(ai_host, 4, na, f) != na || fread(ao_host, 4, na, f) != na) return 4;
  fclose(f);
  int32_t *ques = time_major(q_host, N, Tq), *ain = time_major(ai_host, N, Ta), *aout = time_major(ao_host, N, Ta);
  int32_t* hist = useHist ? time_major(h_host, N, Th) : NULL;
  float* img = dev_floats((int64_t)ni);                         /* one feature row per dialog (model.lua:266-270 repeats it per round) */
  if (useIm) CHECK(p_h2d(img, i_host, (int64_t)ni * 4, NULL));

#define LSTM(D, w, b) {D, H, 0, 0, Wf + off[w], Wf + off[b], Gf + off[w], Gf + off[b], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL}
  float *emb = Wf + off[0], *demb = Gf + off[0];
  SeqLSTM enc_rnn[NL] = {LSTM(E, Q1W, Q1W + 1), LSTM(H, Q2W, Q2W + 1)};          /* enc.rnnLayers (lf-ques.lua:18-24) */
  SeqLSTM hist_rnn[NL] = {LSTM(E, useHist ? H1W : Q1W, useHist ? H1W + 1 : Q1W + 1), LSTM(H, useHist ? H2W : Q2W, useHist ? H2W + 1 : Q2W + 1)};
  SeqLSTM dec_rnn[NL] = {LSTM(E, D1W, D1W + 1), LSTM(H, D2W, D2W + 1)};          /* dec.rnnLayers (gen.lua:17-22) */
  LinearTanh fuse = {Dcat, H, 0, Wf + off[FW], Wf + off[FW + 1], Gf + off[FW], Gf + off[FW + 1], NULL, NULL, 0};
  float *Wv = Wf + off[VW], *bv = Wf + off[VW + 1], *dWv = Gf + off[VW], *dbv = Gf + off[VW + 1];

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero: pad row */

  /* ================= encoder:forward({ques})  (model.lua:297) ================= */
  float* qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  const float* x = qx;
  for (int l = 0; l < NL; ++l) { lstm_forward(&enc_rnn[l], x, Tq, N, ques); x = enc_rnn[l].h; }
  const float* fuse_in = enc_rnn[NL - 1].h + (int64_t)(Tq - 1) * N * H;            /* nn.Select(1, -1) */
  if (useIm || useHist) {
    /* lf-ques-im-hist.lua:28-56 (lf-ques-im.lua, lf-ques-hist.lua: one of the two extra parts): the history stack on the concatenated
     * dialog, the image feature of the dialog repeated for its rounds (model.lua:266-270: a row gather here), nn.JoinTable of the parts */
    float* cat = dev_floats((int64_t)N * Dcat);
    int at = H;
    CHECK(p_copy_2d(cat, Dcat, fuse_in, H, N, H, NULL));
    if (useIm) {
      int32_t* rep = (int32_t*)malloc((size_t)N * 4);
      for (int n = 0; n < N; ++n) rep[n] = n / R;
      float* img_rep = dev_floats((int64_t)N * F);
      CHECK(p_embed_gather(img, dev_ints_from(rep, N), NULL, img_rep, N, F, 1.f, NULL));
      CHECK(p_copy_2d(cat + at, Dcat, img_rep, F, N, F, NULL));
      at += F;
    }
    if (useHist) {
      float* hx = dev_floats((int64_t)Th * N * E);
      CHECK(p_embed_gather(emb, hist, NULL, hx, (int64_t)Th * N, E, 1.f, NULL));
      x = hx;
      for (int l = 0; l < NL; ++l) { lstm_forward(&hist_rnn[l], x, Th, N, hist); x = hist_rnn[l].h; }
      CHECK(p_copy_2d(cat + at, Dcat, hist_rnn[NL - 1].h + (int64_t)(Th - 1) * N * H, H, N, H, NULL));
    }
    fuse_in = cat;
  }
  float* encOut = linear_forward(&fuse, fuse_in, N);

  /* ================= forwardConnect(enc, dec, encOut, seqLen)  (model.lua:300; gen.lua:30-42) ================= */
  for (int l = 0; l < NL; ++l) {
    dec_rnn[l].userPrevOutput = enc_rnn[l].h + (int64_t)(Tq - 1) * N * H;      /* enc.rnnLayers[l].output[seqLen] */
    dec_rnn[l].userPrevCell = enc_rnn[l].c + (int64_t)(Tq - 1) * N * H;        /* enc.rnnLayers[l].cell[seqLen] */
  }
  dec_rnn[NL - 1].userPrevOutput = encOut;                                     /* the top layer starts from the encoder output */

  /* ================= decoder:forward(answer_in)  (model.lua:313; gen.lua:8-27) ================= */
  const int64_t rows = (int64_t)Ta * N;
  float* ax = dev_floats(rows * E);
  CHECK(p_embed_gather(emb, ain, NULL, ax, rows, E, 1.f, NULL));
  x = ax;
  for (int l = 0; l < NL; ++l) { lstm_forward(&dec_rnn[l], x, Ta, N, ain); x = dec_rnn[l].h; }
  const float* dech = dec_rnn[NL - 1].h;
  float* logits = dev_floats(rows * Vp);                                       /* Linear(H, V); LogSoftMax is fused into the criterion */
  CHECK(p_gemm_nt(dech, H, Wv, H, bv, logits, Vp, (int)rows, V, H, VD_ACT_NONE, 0, NULL));

  /* ================= criterion:forward + :backward  (model.lua:314,318): logits -> d loss / d logits in place ================= */
  float* loss_rows = dev_floats(rows);
  CHECK(p_logsoftmax_nll(logits, Vp, rows, V, ain, aout, loss_rows, 1, NULL));
  float* dlog = logits;

  /* ================= decoder:backward(answer_in, gradCriterionOut)  (model.lua:319) ================= */
  CHECK(p_gemm_tn_acc(dlog, Vp, dech, H, dWv, H, V, H, (int)rows, 0, NULL));
  CHECK(p_colsum_acc(dlog, Vp, (int)rows, V, dbv, NULL));
  float* dh = dev_floats(rows * H);
  CHECK(p_gemm_nn(dlog, Vp, Wv, H, NULL, dh, H, (int)rows, H, V, 0, NULL));
  const float* dseq = dh;
  float* dax = NULL;
  for (int l = NL - 1; l >= 0; --l) { dax = lstm_backward(&dec_rnn[l], dseq, NULL, 1); dseq = dax; }
  CHECK(p_embed_scatter_acc(demb, ain, NULL, dax, rows, E, 1.f, NULL));

  /* ================= backwardConnect(enc, dec)  (model.lua:322; gen.lua:45-60) ================= */
  for (int l = 0; l < NL; ++l) {
    enc_rnn[l].userNextGradCell = dec_rnn[l].userGradPrevCell;
    if (l != NL - 1) enc_rnn[l].gradPrevOutput = dec_rnn[l].userGradPrevOutput;
  }
  const float* gradDecOut = dec_rnn[NL - 1].userGradPrevOutput;                /* d loss / d encOut */

  /* ================= encoder:backward(inputs, gradDecOut)  (model.lua:323) ================= */
  const float* dLast = linear_backward(&fuse, gradDecOut);
  if (useIm || useHist) {                                        /* JoinTable backward: the question and history slices (the image needs none) */
    float* dq = dev_floats((int64_t)N * H);
    CHECK(p_copy_2d(dq, H, dLast, Dcat, N, H, NULL));
    if (useHist) {
      float* dhl = dev_floats((int64_t)N * H);
      CHECK(p_copy_2d(dhl, H, dLast + H + (useIm ? F : 0), Dcat, N, H, NULL));
      float* dhx = lstm_backward(&hist_rnn[NL - 1], NULL, dhl, 1);
      for (int l = NL - 2; l >= 0; --l) dhx = lstm_backward(&hist_rnn[l], dhx, NULL, 1);
      CHECK(p_embed_scatter_acc(demb, hist, NULL, dhx, (int64_t)Th * N, E, 1.f, NULL));
    }
    dLast = dq;
  }
  float* dqx = lstm_backward(&enc_rnn[NL - 1], NULL, dLast, 1);
  for (int l = NL - 2; l >= 0; --l) dqx = lstm_backward(&enc_rnn[l], dqx, NULL, 1);
  CHECK(p_embed_scatter_acc(demb, ques, NULL, dqx, (int64_t)Tq * N, E, 1.f, NULL));

  /* curLoss = the SUM over tokens (SequencerCriterion(MaskZeroCriterion(ClassNLL, sizeAverage = false))), then clamp + adam, t = 1 */
  float* lr_host = (float*)malloc((size_t)rows * 4);
  CHECK(p_sync(NULL));
  CHECK(p_d2h(lr_host, loss_rows, rows * 4, NULL));
  double loss = 0;
  for (int64_t r = 0; r < rows; ++r) loss += lr_host[r];
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
  printf("%s + gen through the operator-level ABI: summed loss %.6f, %lld parameters\n", imHist == 3 ? "lf-ques-im-hist" : imHist == 1 ? "lf-ques-im" : imHist == 2 ? "lf-ques-hist" : "lf-ques", loss, (long long)numel);
  return 0;
}
