// Native step runtime, part 3: the decoder plug-ins (reference decoders/disc.lua, decoders/gen.lua) and the two
// branches of Model:forwardBackward / Model:retrieveBatch they select (model.lua:306-338, 344-430).  Mirrors
// visdial_amd/decoders/disc.py, gen.py and visdial_amd/model.py.  Included by runtime.hip only.
#pragma once
#include "rt_encoders.h"

namespace vdrt {

struct Decoder {
  virtual ~Decoder() {}
  virtual void declare(vd_model* m) = 0;
  // Model:forwardBackward on slot b: encoder + decoder forward, criterion, and (unless only_forward) both backwards
  virtual int forward_backward(vd_model* m, BatchSlot& b, bool only_forward) = 0;
  // Model:retrieveBatch up to the scores: leaves [N x O] option scores in m->scores
  virtual int retrieve(vd_model* m, BatchSlot& b) = 0;
  // Model:generateAnswers device steps (model.lua:432-613); generative decoder only
  virtual int gen_begin(vd_model*, const int32_t*, int) { return no_gen(); }
  virtual int gen_step(vd_model*, const int32_t*, float*) { return no_gen(); }
  virtual int gen_select(vd_model*, const int32_t*, int) { return no_gen(); }
  static int no_gen() {
    vd_set_error("sampling / beam search only for the generative decoder (model.lua:436-438)");
    return VD_ERR_STATE;
  }
};

inline int stage_loss(vd_model* m, const float* loss_rows, long n, bool is_sum, hipStream_t s) {
  if (m->loss_cap < n) {
    if (m->loss_host) VD_HIP(hipHostFree(m->loss_host));
    m->loss_host = nullptr;
    VD_HIP(hipHostMalloc((void**)&m->loss_host, (size_t)n * sizeof(float), hipHostMallocDefault));
    m->loss_cap = n;
  }
  m->loss_n = n;
  m->loss_is_sum = is_sum;
  VD_HIP(hipMemcpyAsync(m->loss_host, loss_rows, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
  VD_HIP(hipEventRecord(m->ev_loss, s));
  return VD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Discriminative decoder (decoders/disc.lua:3-38): the 100 candidate answers of a round share one embedding and one
// LSTM; option scores = <last hidden state, encoder output>; criterion = CrossEntropy over the options.
//   * 100 clones under nn.Concat(2) -> ONE recurrence over N*O rows
//   * LookupTable + Wx*x + b -> a [V+1 x 4H] table gathered by token id inside the step kernel
//   * nn.MM + CrossEntropyCriterion (+ their backward) -> vd_score_ce
// The option LSTM (a handful of big launches) runs on the main stream, the encoder (~200 small ones) beside it.
// ------------------------------------------------------------------------------------------------------------
#ifdef VD_PROBE_PHASES
inline hipEvent_t* probe_events() {
  static hipEvent_t e[6];
  static bool made = false;
  if (!made) { for (auto& x : e) (void)hipEventCreate(&x); made = true; }
  return e;
}
#define VD_PROBE_REC(i, stream) (void)hipEventRecord(probe_events()[i], stream)
#else
#define VD_PROBE_REC(i, stream)
#endif
struct Disc : Decoder {
  void declare(vd_model* m) override { add_lstm(m, "opt", m->p.embedSize, m->p.rnnHiddenSize); }
  int forward_backward(vd_model* m, BatchSlot& b, bool only_forward) override {
    VD_CHECK_ARG(b.opt.present, "decoder 'disc' needs batch.options");
    VD_CHECK_ARG(only_forward || b.has_gt, "training decoder 'disc' needs batch.answer_ind");
    // NO = rows the option LSTM executes: N * O, or the number of DISTINCT candidates when the upload de-duplicated them
    const int N = b.q.N, O = m->p.numOptions, NOfull = N * O, NO = b.opt.N, To = b.opt.T;
    const bool dedup = b.opt_uid != nullptr;
    VD_CHECK_ARG(dedup ? NO <= NOfull : NO == NOfull, "decoder 'disc': %d option rows for %d x %d candidates", NO, N, O);
    const long H = m->p.rnnHiddenSize, E = m->p.embedSize, V = m->p.vocabSize;
    hipStream_t s = m->s_main;
    hipStream_t se = side_stream(m, m->s_enc, s);
    hipStream_t st = side_stream(m, m->s_tab, s);
    float *table, *gates, *h, *c, *scores, *loss_rows;
    VD_TRY(ws_get(m, "opt.table", (size_t)(V + 1) * 4 * H, &table));
    VD_TRY(ws_get(m, "opt.gates", (size_t)To * NO * 4 * H, &gates));
    VD_TRY(ws_get(m, "opt.h", (size_t)To * NO * H, &h));
    VD_TRY(ws_get(m, "opt.c", (size_t)To * NO * H, &c));
    VD_TRY(ws_get(m, "opt.scores", (size_t)N * O, &scores));
    VD_TRY(ws_get(m, "crit.loss_rows", (size_t)N, &loss_rows));
    float* Wopt = Wp(m, "opt.W");
    const int flags = m->p.lstmBf16 == 1 ? VD_FLAG_BF16 : m->p.lstmBf16 == 9 ? VD_FLAG_SPLIT9 : m->p.lstmBf16 == 6 ? VD_FLAG_SPLIT6
                      : m->p.lstmBf16 == 3 ? VD_FLAG_SPLIT3 : 0;
    VD_TRY(fork_stream(m, s, se));
    float* enc_out = nullptr;
    VD_TRY(vd_gemm_nn(Wp(m, "embed"), E, Wopt, 4 * H, Wp(m, "opt.b"), table, 4 * H, (int)V + 1, (int)(4 * H), (int)E, 0, s));
    // bf16 pass at a throughput shape: COMPACT state (common.h) -- gates / da only as bf16 (in the first half of `gates`), the
    // projection table as bf16 rows, h as bf16 plus the last step's fp32 state; c stays fp32
    const bool c16 = (flags & VD_FLAG_BF16) && NO >= 2048 && H % 128 == 0;
    vd_bf16_bits *gates16 = reinterpret_cast<vd_bf16_bits*>(gates), *h16 = reinterpret_cast<vd_bf16_bits*>(h), *table16 = nullptr;
    float* h_last = nullptr;
    if (c16) {
      float* t16;
      VD_TRY(ws_get(m, "opt.table16", (size_t)(V + 1) * 2 * H, &t16));
      VD_TRY(ws_get(m, "opt.h_last", (size_t)NO * H, &h_last));
      table16 = reinterpret_cast<vd_bf16_bits*>(t16);
      VD_TRY(vd_f32_to_bf16(table, table16, (V + 1) * 4 * H, s));
    }
    VD_HIP(hipEventRecord(m->ev_prof[0], s));
    {
      VdRange r("disc: option LSTM forward");
      if (c16) VD_TRY(vd_lstm_forward_c16(table16, 4 * H, b.opt.tok, Wopt + E * 4 * H, gates16, h16, h_last, c, To, NO, (int)H, s));
      else VD_TRY(vd_lstm_forward(table, 0, 4 * H, b.opt.tok, nullptr, Wopt + E * 4 * H, nullptr, nullptr, gates, h, c, To, NO, (int)H, flags, s));
    }
    VD_HIP(hipEventRecord(m->ev_prof[1], s));
    {
      VdRange r("encoder forward");
      VD_TRY(m->enc->forward(m, se, b, &enc_out));                                 // model.lua:297
      VD_PROBE_REC(0, se);
    }
    VD_TRY(join_stream(m, se, s));
    // criterion (+ nn.MM backward) in one kernel (model.lua:330-335)
    const float* optH = c16 ? h_last : h + (long)(To - 1) * NO * H;
    float *d_optH = nullptr, *d_enc = nullptr, *d_optH_full = nullptr;
    if (dedup) {   // candidate (n, o) reads the state of its distinct row
      float* full;
      VD_TRY(ws_get(m, "opt.h_full", (size_t)NOfull * H, &full));
      VD_TRY(vd_embed_gather(optH, b.opt_uid, nullptr, full, NOfull, (int)H, 1.f, s));
      optH = full;
    }
    if (!only_forward) {
      VD_TRY(ws_get(m, "crit.d_optH", (size_t)NO * H, &d_optH));
      VD_TRY(ws_get(m, "crit.d_enc", (size_t)N * H, &d_enc));
      if (dedup) VD_TRY(ws_get(m, "crit.d_optH_full", (size_t)NOfull * H, &d_optH_full));
    }
    VD_TRY(vd_score_ce(optH, enc_out, b.gt, scores, loss_rows, dedup ? d_optH_full : d_optH, d_enc, N, O, (int)H, 1.0f / N, s));
    if (dedup && !only_forward) {   // the gradients of the copies of a distinct row add up
      VD_TRY(vd_memset(d_optH, 0, (long)NO * H * 4, s));
      VD_TRY(vd_embed_scatter_acc(d_optH, b.opt_uid, nullptr, d_optH_full, NOfull, (int)H, 1.f, s));
    }
    VD_TRY(stage_loss(m, loss_rows, N, false, s));
    m->scores = scores;
    m->prof_valid = !only_forward;
    if (only_forward) return VD_OK;
    // decoder backward on the main stream, encoder backward beside it (model.lua:335-337)
    VD_TRY(fork_stream(m, s, se));
    float *dc, *dtab;
    int32_t* perm = b.opt_sort_perm;               // sorted at upload time (runtime.hip)
    VD_TRY(ws_get(m, "opt.dc", (size_t)NO * H, &dc));
    VD_TRY(ws_get(m, "opt.dtable", (size_t)(V + 1) * 4 * H, &dtab));
    VD_CHECK_ARG(perm, "decoder 'disc': the batch slot carries no option-token sort");
    VD_TRY(fork_stream(m, s, st));
    VD_TRY(vd_memset(dtab, 0, (V + 1) * 4 * H * 4, st));
    VD_HIP(hipEventRecord(m->ev_prof[2], s));
    // off-chain parameter-gradient work of the encoder on its own (middle-priority) stream = its own hardware queue in a bf16 pass
    // (the encoder chain is that step's critical path: 12.34 -> 12.02 ms, profiles/r03_experiments.txt section 21); in fp32 the step
    // is work-conserving on the matrix pipe and it stays on the encoder stream.  Never joined into `se`: a wait there would be a
    // barrier packet in front of the table-gradient chain.
    m->wg_active = m->streams && m->s_wg && (flags & VD_FLAG_BF16) != 0;
    m->wg_used = false;
    auto enc_bwd = [&]() -> int {
      VdRange r("encoder backward");
      VD_TRY(m->enc->backward(m, se, b, d_enc));
      VD_PROBE_REC(1, se);
      if (m->wg_used) VD_PROBE_REC(2, m->s_wg);
      if (m->wg_used) {   // encoder tensors final = the chain on `se` AND the gradient work on s_wg
        VD_TRY(fork_stream(m, se, m->s_wg));
        VD_HIP(hipEventRecord(m->ev_enc_grads, m->s_wg));
      } else {
        VD_HIP(hipEventRecord(m->ev_enc_grads, se));                                // encoder tensors final (data-parallel bucket 1)
      }
      m->enc_grads_recorded = true;
      return VD_OK;
    };
    {
      VdRange r("disc: option LSTM backward");
      if (c16) VD_TRY(vd_lstm_backward_c16(Wopt + E * 4 * H, gates16, c, d_optH, dc, To, NO, (int)H, s));
      else VD_TRY(vd_lstm_backward(Wopt + E * 4 * H, gates, c, nullptr, nullptr, d_optH, nullptr, dc, nullptr, nullptr, nullptr, To, NO, (int)H, flags, s));
    }
    VD_HIP(hipEventRecord(m->ev_prof[3], s));
    VD_TRY(enc_bwd());       // enqueued behind the option recurrence: it runs beside it on the encoder stream
    // table gradient + its consumers beside the dWh contraction
    VdRange rwg("disc: table gradient + dWh");
    VD_TRY(fork_stream(m, s, st));
    if (c16) VD_TRY(vd_segment_rowsum_acc_bf16(gates16, 4 * H, b.opt.tok, perm, (long)To * NO, (int)(4 * H), dtab, 4 * H, st));
    else VD_TRY(vd_segment_rowsum_acc(gates, 4 * H, b.opt.tok, perm, (long)To * NO, (int)(4 * H), dtab, 4 * H, st));
    VD_TRY(vd_colsum_acc(dtab, 4 * H, (int)V + 1, (int)(4 * H), Gp(m, "opt.b"), st));
    VD_TRY(vd_gemm_tn_acc(Wp(m, "embed"), E, dtab, 4 * H, Gp(m, "opt.W"), 4 * H, (int)E, (int)(4 * H), (int)V + 1, 0, st));
    VD_HIP(hipEventRecord(m->ev_prof[4], s));
    if (To > 1 && c16)
      VD_TRY(vd_gemm_tn_acc_bf16(h16, gates16 + (long)NO * 4 * H, Gp(m, "opt.W") + E * 4 * H, 4 * H, (int)H, (int)(4 * H), (To - 1) * NO, s));
    else if (To > 1)
      VD_TRY(vd_gemm_tn_acc(h, H, gates + (long)NO * 4 * H, 4 * H, Gp(m, "opt.W") + E * 4 * H, 4 * H, (int)H, (int)(4 * H), (To - 1) * NO,
                            flags & (VD_FLAG_BF16 | VD_FLAG_SPLIT9), s));
    VD_HIP(hipEventRecord(m->ev_prof[5], s));
    // dEmb += dTable * Wx^T on the table stream, with float atomics: the SHARED embedding gradient has concurrent atomic
    // writers (the encoder's scatters), and the product is off the main stream's critical path this way
    VD_TRY(vd_gemm_nt(dtab, 4 * H, Wopt, 4 * H, nullptr, Gp(m, "embed"), E, (int)V + 1, (int)E, (int)(4 * H), VD_ACT_NONE, 2, st));
    VD_PROBE_REC(3, st);
    VD_TRY(join_stream(m, se, s));
    if (m->wg_used) VD_TRY(join_stream(m, m->s_wg, s));
    m->wg_active = m->wg_used = false;
    VD_TRY(join_stream(m, st, s));
    VD_PROBE_REC(4, s);
    return VD_OK;
  }
  int retrieve(vd_model* m, BatchSlot& b) override { return forward_backward(m, b, true); }   // model.lua:421-425
};

// ------------------------------------------------------------------------------------------------------------
// Generative decoder (decoders/gen.lua:3-68):
//   answer_in -> shared embedding -> numLayers x SeqLSTM(maskZero) -> Linear(H, V) -> LogSoftMax,
//   criterion = sum over non-pad steps of -log p(answer_out)            (model.lua:32-36, 306-324)
// The encoder's per-layer final (h, c) seed the decoder layers and the encoder output replaces the top layer's initial
// h (forwardConnect, gen.lua:30-42); backwardConnect hands the gradients w.r.t. those states back (gen.lua:45-60).
// ------------------------------------------------------------------------------------------------------------
struct Gen;
int Gen_begin(Gen* g, vd_model* m, const int32_t* rounds, int n);
int Gen_step(Gen* g, vd_model* m, const int32_t* tokens, float* host_logp);
int Gen_select(Gen* g, vd_model* m, const int32_t* src, int n_keep);

struct Gen : Decoder {
  int gen_begin(vd_model* m, const int32_t* r, int n) override { return Gen_begin(this, m, r, n); }
  int gen_step(vd_model* m, const int32_t* t, float* lp) override { return Gen_step(this, m, t, lp); }
  int gen_select(vd_model* m, const int32_t* src, int k) override { return Gen_select(this, m, src, k); }
  std::vector<SeqLSTM> rnn;
  long E = 0, H = 0, V = 0, Vp = 0;
  int gen_n = 0;                                     // live hypotheses of the running generation
  void declare(vd_model* m) override {
    E = m->p.embedSize; H = m->p.rnnHiddenSize; V = m->p.vocabSize; Vp = (V + 3) / 4 * 4;
    rnn.resize(m->p.numLayers);
    for (int l = 0; l < m->p.numLayers; ++l) {
      add_lstm(m, "dec" + std::to_string(l + 1), l == 0 ? E : H, H);              // gen.lua:17-22
      rnn[l].init("dec" + std::to_string(l + 1), l == 0 ? E : H, H);
    }
    add_linear(m, "vocab", H, V);                                                   // gen.lua:23
  }
  // gen.lua:30-42; `rep` (retrieval) replicates every encoder state row over the chunk's options
  int forwardConnect(vd_model* m, hipStream_t s, const float* encOut, int seqLen, const int32_t* rep, long rows) {
    auto put = [&](const float* x, const std::string& key, const float** dst) -> int {
      if (!rep) {
        *dst = x;
        return VD_OK;
      }
      float* r;
      VD_TRY(ws_get(m, key, (size_t)rows * H, &r));
      VD_TRY(vd_embed_gather(x, rep, nullptr, r, rows, (int)H, 1.f, s));
      *dst = r;
      return VD_OK;
    };
    std::vector<SeqLSTM>* layers = m->enc->rnnLayers();
    if (layers) {
      for (size_t i = 0; i < layers->size(); ++i) {
        VD_TRY(put((*layers)[i].out_at(seqLen - 1), "ret.h0_" + std::to_string(i), &rnn[i].userPrevOutput));
        VD_TRY(put((*layers)[i].cell_at(seqLen - 1), "ret.c0_" + std::to_string(i), &rnn[i].userPrevCell));
      }
      return put(encOut, "ret.enc", &rnn[layers->size() - 1].userPrevOutput);
    }
    return put(encOut, "ret.enc", &rnn.back().userPrevOutput);
  }
  // gen.lua:45-60: returns dL/d encOut
  const float* backwardConnect(vd_model* m) {
    std::vector<SeqLSTM>* layers = m->enc->rnnLayers();
    if (layers) {
      const size_t n = rnn.size();
      for (size_t i = 0; i < n; ++i) {
        (*layers)[i].userNextGradCell = rnn[i].userGradPrevCell;
        if (i != n - 1) (*layers)[i].gradPrevOutput = rnn[i].userGradPrevOutput;
      }
      return rnn[layers->size() - 1].userGradPrevOutput;
    }
    return rnn.back().userGradPrevOutput;
  }
  int forward_backward(vd_model* m, BatchSlot& b, bool only_forward) override {
    VD_CHECK_ARG(b.ain.present && b.aout.present, "decoder 'gen' needs batch.answer_in / answer_out");
    hipStream_t s = m->s_main;
    const int N = b.q.N, Ta = b.ain.T;
    const long rows = (long)Ta * N;
    float* encOut;
    {
      VdRange r("encoder forward");
      VD_TRY(m->enc->forward(m, s, b, &encOut));                                    // model.lua:297
    }
    VdRange rdec("gen: decoder forward + criterion + backward");
    VD_TRY(forwardConnect(m, s, encOut, m->enc->seqLen(b), nullptr, N));            // model.lua:300
    float *x, *h, *logits, *loss_rows;
    VD_TRY(ws_get(m, "dec.x", (size_t)rows * E, &x));
    VD_TRY(ws_get(m, "dec.logits", (size_t)rows * Vp, &logits));
    VD_TRY(ws_get(m, "dec.loss_rows", (size_t)rows, &loss_rows));
    VD_TRY(vd_embed_gather(Wp(m, "embed"), b.ain.tok, nullptr, x, rows, (int)E, 1.f, s));
    VD_TRY(lstm_stack_forward(m, s, rnn, {x}, Ta, N, b.ain.tok, &h));
    m->prof_valid = false;
    m->prof_hist = false;
    VD_HIP(hipEventRecord(m->ev_prof[4], s));          // family 3 of a gen pair: vocabulary projection + criterion + their gradients
    VD_TRY(vd_gemm_nt(h, H, Wp(m, "vocab.W"), H, Wp(m, "vocab.b"), logits, Vp, (int)rows, (int)V, (int)H, VD_ACT_NONE, 0, s));
    VD_TRY(vd_logsoftmax_nll(logits, Vp, rows, (int)V, b.ain.tok, b.aout.tok, loss_rows, only_forward ? 0 : 1, s));   // model.lua:309-311
    VD_TRY(stage_loss(m, loss_rows, rows, true, s));
    if (only_forward) return VD_OK;
    float* dh;                                                                      // logits now hold d loss / d logits
    VD_TRY(ws_get(m, "dec.dh", (size_t)rows * H, &dh));
    VD_TRY(vd_gemm_tn_acc(logits, Vp, h, H, Gp(m, "vocab.W"), H, (int)V, (int)H, (int)rows, 0, s));
    VD_TRY(vd_colsum_acc(logits, Vp, (int)rows, (int)V, Gp(m, "vocab.b"), s));
    VD_TRY(vd_gemm_nn(logits, Vp, Wp(m, "vocab.W"), H, nullptr, dh, H, (int)rows, (int)H, (int)V, 0, s));
    VD_HIP(hipEventRecord(m->ev_prof[5], s));
    std::vector<float*> dx;
    VD_TRY(lstm_stack_backward(m, s, rnn, nullptr, dh, &dx));                       // model.lua:316
    VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), b.ain.tok, nullptr, dx[0], rows, (int)E, 1.f, s));
    const float* gradDecOut = backwardConnect(m);                                   // model.lua:319
    VD_CHECK_ARG(gradDecOut, "backwardConnect produced no gradient");
    {
      VdRange r("encoder backward");
      VD_TRY(m->enc->backward(m, s, b, gradDecOut));                                // model.lua:322
    }
    VD_HIP(hipEventRecord(m->ev_enc_grads, s));
    m->enc_grads_recorded = true;
    m->prof_valid = m->prof_hist;       // [history branch fwd, history branch bwd, vocabulary family] (vd_model_family_ms)
    return VD_OK;
  }
  // Model:retrieveBatch gen branch (model.lua:392-420) + utils.computeLhood (utils.lua:86-102).  The reference loops
  // over the 100 options; here chunks of options are ONE decoder batch (rows = round x option) seeded by the replicated
  // encoder state, and the [rows x V] logits only ever exist for one chunk.
  int retrieve(vd_model* m, BatchSlot& b) override {
    VD_CHECK_ARG(b.oin.present && b.oout.present, "retrieval with decoder 'gen' needs batch.option_in / option_out");
    hipStream_t s = m->s_main;
    const int N = b.q.N, O = m->p.numOptions, T = b.oin.T;
    float* encOut;
    VD_TRY(m->enc->forward(m, s, b, &encOut));
    const int seqLen = m->enc->seqLen(b);
    float* lhood;
    VD_TRY(ws_get(m, "ret.lhood", (size_t)N * O, &lhood));
    const long per_opt = (long)T * N * Vp;
    const int oc = (int)std::max<long>(1, std::min<long>(O, (1L << 30) / std::max<long>(1, per_opt)));   // <= 4 GiB of logits
    for (int o0 = 0; o0 < O; o0 += oc) {
      const int C = std::min(O, o0 + oc) - o0;
      const long rows = (long)N * C;
      int32_t *cin, *cout, *idx;
      VD_TRY(ws_get(m, "ret.cin", (size_t)T * rows, &cin));
      VD_TRY(ws_get(m, "ret.cout", (size_t)T * rows, &cout));
      // [T*N x O] int32 -> columns [o0, o0+C): a strided dword copy
      VD_TRY(vd_copy_2d((float*)cin, C, (const float*)(b.oin.tok + o0), O, (long)T * N, C, s));
      VD_TRY(vd_copy_2d((float*)cout, C, (const float*)(b.oout.tok + o0), O, (long)T * N, C, s));
      std::vector<int32_t> hidx(rows);
      for (long r = 0; r < rows; ++r) hidx[r] = (int32_t)(r / C);
      VD_TRY(index_array(m, "idx.ret." + std::to_string(rows) + "." + std::to_string(C), hidx, &idx));
      VD_TRY(forwardConnect(m, s, encOut, seqLen, idx, rows));
      float *x, *h, *logits, *nll, *acc;
      VD_TRY(ws_get(m, "ret.x", (size_t)T * rows * E, &x));
      VD_TRY(ws_get(m, "ret.logits", (size_t)T * rows * Vp, &logits));
      VD_TRY(ws_get(m, "ret.nll", (size_t)T * rows, &nll));
      VD_TRY(ws_get(m, "ret.acc", (size_t)rows, &acc));
      VD_TRY(vd_embed_gather(Wp(m, "embed"), cin, nullptr, x, T * rows, (int)E, 1.f, s));
      VD_TRY(lstm_stack_forward(m, s, rnn, {x}, T, (int)rows, cin, &h));
      VD_TRY(vd_gemm_nt(h, H, Wp(m, "vocab.W"), H, Wp(m, "vocab.b"), logits, Vp, (int)(T * rows), (int)V, (int)H, VD_ACT_NONE, 0, s));
      VD_TRY(vd_logsoftmax_nll(logits, Vp, T * rows, (int)V, cin, cout, nll, 0, s));
      VD_TRY(vd_memset(acc, 0, rows * 4, s));
      VD_TRY(vd_colsum_acc(nll, rows, T, (int)rows, acc, s));                       // sum over time (utils.lua:98)
      VD_TRY(vd_copy_2d(lhood + o0, O, acc, C, N, C, s));
    }
    VD_TRY(vd_axpby(lhood, nullptr, lhood, (long)N * O, -1.f, 0.f, s));             // log-likelihood = -NLL
    m->scores = lhood;
    m->prof_valid = false;
    return VD_OK;
  }
};

// ---- Model:generateAnswers (model.lua:432-613): the device side of sampling / beam search.  The host keeps the
// candidate bookkeeping (as the reference does in Lua); one call = one decoder step for all live hypotheses.
inline int gen_rows(vd_model* m, const char* key, const int32_t* host, int n, int32_t** out) {
  VD_TRY(ws_get(m, key, (size_t)std::max(n, 1), out));
  VD_HIP(hipMemcpyAsync(*out, host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, m->s_main));
  VD_HIP(hipStreamSynchronize(m->s_main));           // `host` may be a temporary of the caller
  return VD_OK;
}
// hiddenBeams (model.lua:478-503): hypothesis i starts from the encoder state of QA round rounds[i]
inline int Gen_begin(Gen* g, vd_model* m, const int32_t* rounds, int n) {
  VD_CHECK_ARG(m->gen_enc_out && rounds && n > 0, "vd_model_decode_begin: call vd_model_encode first");
  for (int i = 0; i < n; ++i) VD_CHECK_ARG(rounds[i] >= 0 && rounds[i] < m->N, "vd_model_decode_begin: round %d out of range", rounds[i]);
  hipStream_t s = m->s_main;
  int32_t* idx;
  VD_TRY(gen_rows(m, "gen.idx", rounds, n, &idx));
  std::vector<SeqLSTM>* layers = m->enc->rnnLayers();
  const int L = (int)g->rnn.size(), seqLen = m->gen_seq_len;
  const long H = g->H;
  for (int l = 0; l < L; ++l) {
    float *h, *c, *hn, *cn;
    VD_TRY(ws_get(m, "gen.h" + std::to_string(l), (size_t)n * H, &h));
    VD_TRY(ws_get(m, "gen.c" + std::to_string(l), (size_t)n * H, &c));
    VD_TRY(ws_get(m, "gen.hn" + std::to_string(l), (size_t)n * H, &hn));
    VD_TRY(ws_get(m, "gen.cn" + std::to_string(l), (size_t)n * H, &cn));
    if (layers) {
      VD_CHECK_ARG(l < (int)layers->size(), "decoder has more layers than the encoder's recurrence");
      const float* hs = l == (int)layers->size() - 1 ? m->gen_enc_out : (*layers)[l].out_at(seqLen - 1);
      VD_TRY(vd_embed_gather(hs, idx, nullptr, h, n, (int)H, 1.f, s));
      VD_TRY(vd_embed_gather((*layers)[l].cell_at(seqLen - 1), idx, nullptr, c, n, (int)H, 1.f, s));
    } else {
      VD_TRY(vd_memset(c, 0, (long)n * H * 4, s));
      if (l == L - 1) VD_TRY(vd_embed_gather(m->gen_enc_out, idx, nullptr, h, n, (int)H, 1.f, s));
      else VD_TRY(vd_memset(h, 0, (long)n * H * 4, s));
    }
  }
  g->gen_n = n;
  return VD_OK;
}
// model.lua:518-522 / :590-596: one decoder step for the n live hypotheses -> log-probabilities [n x V] on the host
inline int Gen_step(Gen* g, vd_model* m, const int32_t* tokens, float* host_logp) {
  const int n = g->gen_n;
  VD_CHECK_ARG(n > 0 && tokens && host_logp, "vd_model_decode_step: call vd_model_decode_begin first");
  hipStream_t s = m->s_main;
  const long H = g->H, E = g->E, V = g->V, Vp = g->Vp;
  int32_t* tok;
  VD_TRY(gen_rows(m, "gen.tok", tokens, n, &tok));
  VD_TRY(vd_memset(Wp(m, "embed"), 0, E * 4, s));                                   // LookupTableMaskZero pad row
  const int L = (int)g->rnn.size();
  std::vector<float*> h(L), c(L), hn(L), cn(L);
  for (int l = 0; l < L; ++l) {
    VD_TRY(ws_get(m, "gen.h" + std::to_string(l), (size_t)n * H, &h[l]));
    VD_TRY(ws_get(m, "gen.c" + std::to_string(l), (size_t)n * H, &c[l]));
    VD_TRY(ws_get(m, "gen.hn" + std::to_string(l), (size_t)n * H, &hn[l]));
    VD_TRY(ws_get(m, "gen.cn" + std::to_string(l), (size_t)n * H, &cn[l]));
    g->rnn[l].userPrevOutput = h[l];
    g->rnn[l].userPrevCell = c[l];
  }
  float *x, *top, *logits;
  VD_TRY(ws_get(m, "gen1.x", (size_t)n * E, &x));
  VD_TRY(ws_get(m, "gen1.logits", (size_t)n * Vp, &logits));
  VD_TRY(vd_embed_gather(Wp(m, "embed"), tok, nullptr, x, n, (int)E, 1.f, s));
  VD_TRY(lstm_stack_forward(m, s, g->rnn, {x}, 1, n, tok, &top));
  VD_TRY(vd_gemm_nt(top, H, Wp(m, "vocab.W"), H, Wp(m, "vocab.b"), logits, Vp, n, (int)V, (int)H, VD_ACT_NONE, 0, s));
  VD_TRY(vd_log_softmax_rows(logits, Vp, n, (int)V, s));
  for (int l = 0; l < L; ++l) {                                                      // the stepped state (decoderConnect, gen.lua:63-68)
    VD_TRY(vd_memcpy_d2d(hn[l], g->rnn[l].out_at(0), (long)n * H * 4, s));
    VD_TRY(vd_memcpy_d2d(cn[l], g->rnn[l].cell_at(0), (long)n * H * 4, s));
  }
  VD_HIP(hipMemcpy2DAsync(host_logp, (size_t)V * 4, logits, (size_t)Vp * 4, (size_t)V * 4, (size_t)n, hipMemcpyDeviceToHost, s));
  VD_HIP(hipStreamSynchronize(s));
  // Sequencer(MaskZero(Linear)) + Sequencer(MaskZero(LogSoftMax)) (decoders/gen.lua:23-24): the row of a hypothesis whose token is 0 -- a
  // beam slot that was never filled (model.lua:560) -- is all ZEROS, not log_softmax(bias); its state is already zero (maskZero)
  for (int i = 0; i < n; ++i)
    if (tokens[i] == 0) memset(host_logp + (size_t)i * V, 0, (size_t)V * 4);
  return VD_OK;
}
// model.lua:560-575: hypothesis i continues from the stepped state of hypothesis src[i]; slots >= n_keep keep theirs
inline int Gen_select(Gen* g, vd_model* m, const int32_t* src, int n_keep) {
  const int n = g->gen_n;
  VD_CHECK_ARG(n > 0 && src && n_keep >= 0 && n_keep <= n, "vd_model_decode_select: bad arguments");
  for (int i = 0; i < n_keep; ++i) VD_CHECK_ARG(src[i] >= 0 && src[i] < n, "vd_model_decode_select: src[%d] = %d out of range", i, src[i]);
  if (n_keep == 0) return VD_OK;
  hipStream_t s = m->s_main;
  int32_t* idx;
  VD_TRY(gen_rows(m, "gen.idx", src, n_keep, &idx));
  const long H = g->H;
  for (size_t l = 0; l < g->rnn.size(); ++l) {
    float *h, *c, *hn, *cn;
    VD_TRY(ws_get(m, "gen.h" + std::to_string(l), (size_t)n * H, &h));
    VD_TRY(ws_get(m, "gen.c" + std::to_string(l), (size_t)n * H, &c));
    VD_TRY(ws_get(m, "gen.hn" + std::to_string(l), (size_t)n * H, &hn));
    VD_TRY(ws_get(m, "gen.cn" + std::to_string(l), (size_t)n * H, &cn));
    VD_TRY(vd_embed_gather(hn, idx, nullptr, h, n_keep, (int)H, 1.f, s));
    VD_TRY(vd_embed_gather(cn, idx, nullptr, c, n_keep, (int)H, 1.f, s));
  }
  return VD_OK;
}

inline std::unique_ptr<Decoder> make_decoder(const std::string& n) {
  if (n == "disc") return std::unique_ptr<Decoder>(new Disc());
  if (n == "gen") return std::unique_ptr<Decoder>(new Gen());
  return nullptr;
}

}  // namespace vdrt
