/* host_c_plugin_lf_ques_gen.c -- BASELINE.json configs[0], the reference's own CPU-runnable pair (`th train.lua -encoder lf-ques
 * -decoder gen -gpuid -1`), written against the OPERATOR-LEVEL C ABI by a host without a tensor library: encoders/lf-ques.lua +
 * decoders/gen.lua + the generative criterion and the optimiser of model.lua, module object by module object --
 *   encoder  : wordEmbed -> numLayers x SeqLSTM:maskZero() -> Select(1,-1) -> Dropout -> Linear -> Tanh           (lf-ques.lua:6-33)
 *   connect  : decoder layer i starts from the encoder layer's final (h, c); the TOP layer's h from encOut       (gen.lua:30-42)
 *   decoder  : shared wordEmbed(answer_in) -> numLayers x SeqLSTM:maskZero() -> Linear(H, V) -> LogSoftMax       (gen.lua:8-27)
 *   criterion: SUM over non-pad targets of -log p(answer_out)                                                   (model.lua:32-36)
 *   backward : decoder:backward -> backwardConnect (cell / hidden gradients back into the encoder layers, gen.lua:45-60) ->
 *              encoder:backward(inputs, gradDecOut)                                                             (model.lua:318-324)
 * lua/decoders/gen.lua + lua/encoders/lf-ques.lua + lua/model_ops.lua are this file in Lua; this one is built with gcc and checked on
 * the GPU against the library's model-level implementation of the pair (tests/test_abi_c_host.py).
 *
 *   ./host_c_plugin_lf_ques_gen <libvisdial_hip.so> <in.bin> <out.bin>
 * in.bin : int32 {V, E, H, B, R, Tq, Ta}, the 13 parameter tensors in getParameters() order as float32 (embed, ques1.W/.b, ques2.W/.b,
 *          fuse.W/.b, dec1.W/.b, dec2.W/.b, vocab.W [V x H], vocab.b [V]), ques_fwd [B*R x Tq], answer_in [B*R x Ta], answer_out [B*R x Ta]
 *          (int32; answers left-aligned, 0 = pad).
 * out.bin: float32 loss (the SUM), the flat gradient, the flat parameters after clamp(-5,5) + adam.   Dropout off (wrapper:evaluate()). */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

#define NT 13
#define NL 2                                   /* numLayers (opts.lua:22 default) */

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  load_entry_points(argv[1]);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 4; }
  int32_t hd[7];
  if (fread(hd, 4, 7, f) != 7) return 4;
  const int V = hd[0], E = hd[1], H = hd[2], B = hd[3], R = hd[4], Tq = hd[5], Ta = hd[6];
  const int N = B * R, Vp = (V + 3) / 4 * 4;

  /* ---- wrapper:getParameters() ---- */
  enum { EMBED, Q1W, Q1B, Q2W, Q2B, FW, FB, D1W, D1B, D2W, D2B, VW, VB };
  const int64_t l1 = (int64_t)(E + H) * 4 * H, l2 = (int64_t)2 * H * 4 * H;
  const int64_t sizes[NT] = {(int64_t)(V + 1) * E, l1, 4 * H, l2, 4 * H, (int64_t)H * H, H, l1, 4 * H, l2, 4 * H, (int64_t)V * H, V};
  int64_t off[NT + 1];
  off[0] = 0;
  for (int i = 0; i < NT; ++i) off[i + 1] = off[i] + align4(sizes[i]);
  const int64_t numel = off[NT];
  float *Wf = dev_floats(numel), *Gf = dev_floats(numel), *Mf = dev_floats(numel), *Vf = dev_floats(numel);
  float* host = (float*)calloc((size_t)numel, 4);
  for (int i = 0; i < NT; ++i)
    if (fread(host + off[i], 4, (size_t)sizes[i], f) != (size_t)sizes[i]) return 4;
  CHECK(p_h2d(Wf, host, numel * 4, NULL));
  const size_t nq = (size_t)N * Tq, na = (size_t)N * Ta;
  int32_t *q_host = (int32_t*)malloc(nq * 4), *ai_host = (int32_t*)malloc(na * 4), *ao_host = (int32_t*)malloc(na * 4);
  if (fread(q_host, 4, nq, f) != nq || fread(ai_host, 4, na, f) != na || fread(ao_host, 4, na, f) != na) return 4;
  fclose(f);
  int32_t *ques = time_major(q_host, N, Tq), *ain = time_major(ai_host, N, Ta), *aout = time_major(ao_host, N, Ta);

#define LSTM(D, w, b) {D, H, 0, 0, Wf + off[w], Wf + off[b], Gf + off[w], Gf + off[b], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL}
  float *emb = Wf + off[EMBED], *demb = Gf + off[EMBED];
  SeqLSTM enc_rnn[NL] = {LSTM(E, Q1W, Q1B), LSTM(H, Q2W, Q2B)};       /* enc.rnnLayers (lf-ques.lua:18-24) */
  SeqLSTM dec_rnn[NL] = {LSTM(E, D1W, D1B), LSTM(H, D2W, D2B)};       /* dec.rnnLayers (gen.lua:17-22) */
  LinearTanh fuse = {H, H, 0, Wf + off[FW], Wf + off[FB], Gf + off[FW], Gf + off[FB], NULL, NULL, 0};
  float *Wv = Wf + off[VW], *bv = Wf + off[VB], *dWv = Gf + off[VW], *dbv = Gf + off[VB];

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero: pad row */

  /* ================= encoder:forward({ques})  (model.lua:297) ================= */
  float* qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  const float* x = qx;
  for (int l = 0; l < NL; ++l) { lstm_forward(&enc_rnn[l], x, Tq, N, ques); x = enc_rnn[l].h; }
  float* encOut = linear_forward(&fuse, enc_rnn[NL - 1].h + (int64_t)(Tq - 1) * N * H, N);

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
  for (int i = 0; i < NT; ++i) fwrite(grad_host + off[i], 4, (size_t)sizes[i], o);
  for (int i = 0; i < NT; ++i) fwrite(host + off[i], 4, (size_t)sizes[i], o);
  fclose(o);
  printf("lf-ques + gen through the operator-level ABI: summed loss %.6f, %lld parameters\n", loss, (long long)numel);
  return 0;
}
