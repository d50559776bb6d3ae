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
 * out.bin: float32 loss, then the flat gradient, then the flat parameters after clamp(-5,5) + adam, then the rank of the ground-truth
 *          option of every round on the INITIAL parameters (Model:retrieveBatch).
 * Dropout is off (wrapper:evaluate()): the comparison side does the same. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "visdial_hip.h"
#include "host_c_modules.h"

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <lib.so> <in.bin> <out.bin>\n", argv[0]); return 1; }
  load_entry_points(argv[1]);

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
  for (int n = 0; n < N; ++n) a_host[n] -= 1;
  int32_t *ques = time_major(q_host, N, Tq), *opts = time_major(o_host, NO, To), *gt = dev_ints_from(a_host, N);

  /* ---- the plug-in pair: encoders/lf-ques.lua (wordEmbed, 2 x SeqLSTM, Linear + Tanh) and decoders/disc.lua (shared table, one SeqLSTM) ---- */
  float *emb = Wf + off[0], *demb = Gf + off[0];
  SeqLSTM ques1 = {E, H, 0, 0, Wf + off[1], Wf + off[2], Gf + off[1], Gf + off[2], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
  SeqLSTM ques2 = {H, H, 0, 0, Wf + off[3], Wf + off[4], Gf + off[3], Gf + off[4], NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
  LinearTanh fuse = {H, H, 0, Wf + off[5], Wf + off[6], Gf + off[5], Gf + off[6], NULL, NULL, 0};
  DiscDecoder dec = {V, E, H, 0, 0, emb, demb, Wf + off[7], Wf + off[8], Gf + off[7], Gf + off[8], NULL, NULL, NULL, NULL, NULL};

  CHECK(p_memset(Gf, 0, numel * 4, NULL));                      /* wrapper:zeroGradParameters() (model.lua:68) */
  CHECK(p_memset(emb, 0, (int64_t)E * 4, NULL));                /* LookupTableMaskZero zeroes the pad row on every forward */

  /* encoder:forward(inputs)  (model.lua:297; encoders/lf-ques.lua) */
  float* qx = dev_floats((int64_t)Tq * N * E);
  CHECK(p_embed_gather(emb, ques, NULL, qx, (int64_t)Tq * N, E, 1.f, NULL));
  lstm_forward(&ques1, qx, Tq, N, ques);
  lstm_forward(&ques2, ques1.h, Tq, N, ques);
  const float* q_last = ques2.h + (int64_t)(Tq - 1) * N * H;     /* nn.Select(1, -1) */
  float* encOut = linear_forward(&fuse, q_last, N);             /* Dropout is the identity in evaluate() */

  /* decoder:forward({options, encOut})  (model.lua:329; decoders/disc.lua) */
  const float* optH = disc_forward(&dec, opts, To, NO);

  /* criterion:forward + :backward (model.lua:330,334): nn.MM + CrossEntropyCriterion and both gradients in one kernel */
  float *scores = dev_floats((int64_t)N * O), *loss_rows = dev_floats(N), *d_optH = dev_floats((int64_t)NO * H), *d_enc = dev_floats((int64_t)N * H);
  CHECK(p_score_ce(optH, encOut, gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, NULL));
  /* Model:retrieveBatch (model.lua:344-430, disc branch): the same forward, then utils.computeRanks (utils.lua:106-128) on the scores --
   * the 1-based rank of the ground-truth option of every round */
  void* ranks_dev = NULL;
  CHECK(p_malloc(&ranks_dev, (int64_t)N * O * 4));
  CHECK(p_ranks(scores, (int32_t*)ranks_dev, N, O, NULL));

  /* decoder:backward (model.lua:335) */
  disc_backward(&dec, d_optH);

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
  int32_t* ranks_host = (int32_t*)malloc((size_t)N * O * 4);
  CHECK(p_d2h(ranks_host, ranks_dev, (int64_t)N * O * 4, NULL));
  for (int n = 0; n < N; ++n) { const float r = (float)ranks_host[(size_t)n * O + a_host[n]]; fwrite(&r, 4, 1, o); }   /* GT ranks (a_host is 0-based here) */
  fclose(o);
  printf("lf-ques + disc through the operator-level ABI: loss %.6f, %lld parameters\n", loss, (long long)numel);
  return 0;
}
