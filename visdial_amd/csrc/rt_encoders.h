// Native step runtime, part 2: the encoder plug-ins (reference encoders/<name>.lua, 11 files) as launch orders over the
// operator-level entry points.  Mirrors visdial_amd/encoders/_blocks.py, _late_fusion.py, _hre.py and the four
// nngraph encoder files.  Included by runtime.hip only.
#pragma once
#include "rt_core.h"

namespace vdrt {

struct Encoder {
  virtual ~Encoder() {}
  virtual void declare(vd_model* m) = 0;
  // returns the encoder output [N x H] (a workspace buffer); everything is enqueued on `s` and library side streams
  // that are joined back into `s` before returning
  virtual int forward(vd_model* m, hipStream_t s, BatchSlot& b, float** out) = 0;
  virtual int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* grad_out) = 0;
  // enc.rnnLayers of the Sequential encoders (read by decoders/gen.lua:31-35,46-52); null for the nngraph encoders
  virtual std::vector<SeqLSTM>* rnnLayers() { return nullptr; }
  virtual int seqLen(const BatchSlot& b) const { return b.q.T; }
};

// arithmetic of the encoder's recurrent products: bf16 operands in a bf16 pass (lstmPrecision = 'bf16', BASELINE configs[4] "bf16 LSTM
// step"), fp32 MFMA otherwise (the exact-split modes of the option recurrence leave the encoder on fp32: it is not on their critical path)
inline int tick_flags(const vd_model* m) { return m->p.lstmBf16 == 1 ? VD_FLAG_BF16 : 0; }

inline int causal_mask(vd_model* m, int B, int R, uint8_t** out) {
  // model.lua:280-294: mask[i][j] = 0 iff j <= i, tiled over the batch ([N x R] bytes, 1 = hidden)
  const std::string key = "causal." + std::to_string(B);
  DevBuf& d = m->ws[key];
  const size_t n = (size_t)B * R * R;
  if (!d.p) {
    std::vector<uint8_t> host(n);
    for (int r = 0; r < B * R; ++r)
      for (int j = 0; j < R; ++j) host[(size_t)r * R + j] = j > (r % R) ? 1 : 0;
    VD_HIP(hipMalloc(&d.p, n + 16));
    d.bytes = n;
    VD_HIP(hipMemcpy(d.p, host.data(), n, hipMemcpyHostToDevice));
  }
  *out = static_cast<uint8_t*>(d.p);
  return VD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// history + question branches of the nngraph encoders: shared embedding -> Dropout(0.5) -> 2 x SeqLSTM(maskZero) ->
// Select(1,-1) (mn-att:21-45).  Both stacks advance together as one skewed wavefront on length-sorted rows.
// ------------------------------------------------------------------------------------------------------------
struct TextBranches {
  SeqLSTM hist1, hist2, ques1, ques2;
  long E = 0, H = 0;
  uint8_t *m_q = nullptr, *m_h = nullptr;
  float *xs_q = nullptr, *xs_h = nullptr;
  static void declare(vd_model* m) {
    const long E = m->p.embedSize, H = m->p.rnnHiddenSize;
    add_lstm(m, "hist1", E, H); add_lstm(m, "hist2", H, H); add_lstm(m, "ques1", E, H); add_lstm(m, "ques2", H, H);
  }
  void init(vd_model* m) {
    E = m->p.embedSize; H = m->p.rnnHiddenSize;
    hist1.init("hist1", E, H); hist2.init("hist2", H, H); ques1.init("ques1", E, H); ques2.init("ques2", H, H);
  }
  // drop: the nn.Dropout(0.5) behind the shared embedding of the nngraph encoders (mn-att:24-25); the Sequential encoders have none
  static int prepare(vd_model* m, hipStream_t s, const SeqTok& ss, const std::string& tag, SeqLSTM& l1, SeqLSTM& l2, uint8_t** mask_out,
                     float** xs_out, bool drop = true) {
    const long E = l1.D, H = l1.H;
    const long TN = (long)ss.T * ss.N;
    uint8_t* mk = nullptr;
    if (drop) VD_TRY(drop_mask(m, tag + "_emb", (size_t)TN * E, 0.5f, s, &mk));
    float *x, *xs;
    VD_TRY(ws_get(m, tag + ".x", (size_t)TN * E, &x));
    VD_TRY(ws_get(m, tag + ".xs", (size_t)TN * E, &xs));
    VD_TRY(l1.alloc(m, ss.T, ss.N));
    VD_TRY(l2.alloc(m, ss.T, ss.N));
    l1.h0 = l1.c0 = l2.h0 = l2.c0 = nullptr;
    VD_TRY(vd_embed_gather(Wp(m, "embed"), ss.tok, mk, x, TN, (int)E, drop ? 2.0f : 1.0f, s));
    VD_TRY(vd_embed_gather(x, ss.fwd_idx, nullptr, xs, TN, (int)E, 1.0f, s));      // permute rows per step (length sort)
    VD_TRY(vd_gemm_nn(xs, E, l1.Wx(m), 4 * H, Wp(m, l1.name + ".b"), l1.gates, 4 * H, (int)TN, (int)(4 * H), (int)E, 0, s));
    // skipped (t, row) pairs must read as zeros: previous state of rows that become active later, da = 0 in the
    // weight-gradient contractions
    if (ss.sorted) {
      // only the skipped pairs, all six buffers in one launch: the active rows are written by the recurrence before anything reads them
      VdZeroSet z{{l1.gates, l1.h, l1.c, l2.h, l2.c, l2.gates}, {(int)(4 * H), (int)H, (int)H, (int)H, (int)H, (int)(4 * H)}, 6, 0};
      VD_TRY(vd_zero_inactive_multi(z, ss.nact_dev, ss.T, ss.N, s));
    } else {
      VD_TRY(vd_zero_inactive_rows(l1.gates, (int64_t)ss.N * 4 * H, 4 * H, (int)(4 * H), ss.nact_dev, ss.T, ss.N, s));
      VD_TRY(vd_memset(l1.h, 0, TN * H * 4, s));
      VD_TRY(vd_memset(l1.c, 0, TN * H * 4, s));
      VD_TRY(vd_memset(l2.h, 0, TN * H * 4, s));
      VD_TRY(vd_memset(l2.c, 0, TN * H * 4, s));
      VD_TRY(vd_memset(l2.gates, 0, TN * 4 * H * 4, s));
    }
    l1.xs = {xs};
    l2.xs = {l1.h};
    l1.rows = l2.rows = ss.sorted ? &ss : nullptr;
    *mask_out = mk;
    *xs_out = xs;
    return VD_OK;
  }
  static void fill_fwd(vd_model* m, const SeqTok& ss, SeqLSTM& l1, SeqLSTM& l2, vd_lstm2_fwd_t* o) {
    o->T = ss.T; o->N = ss.N;
    o->tok_mask = ss.tok_sorted;
    o->Wh1 = l1.Wh(m); o->Wx2 = l2.Wx(m); o->b2 = Wp(m, l2.name + ".b"); o->Wh2 = l2.Wh(m);
    o->gates1 = l1.gates; o->h1 = l1.h; o->c1 = l1.c; o->gates2 = l2.gates; o->h2 = l2.h; o->c2 = l2.c;
    o->nact = ss.nact.data();
  }
  int forward(vd_model* m, hipStream_t s, BatchSlot& b, float** q3, float** h3) {
    const int N = b.q.N;
    VD_TRY(prepare(m, s, b.h, "h", hist1, hist2, &m_h, &xs_h));
    VD_TRY(prepare(m, s, b.q, "q", ques1, ques2, &m_q, &xs_q));
    vd_lstm2_fwd_t fw[2];
    fill_fwd(m, b.h, hist1, hist2, &fw[0]);
    fill_fwd(m, b.q, ques1, ques2, &fw[1]);
    VD_TRY(vd_lstm2_forward_p(fw, 2, (int)H, tick_flags(m), s));
    float *hl, *ql;
    VD_TRY(ws_get(m, "h.last", (size_t)N * H, &hl));
    VD_TRY(ws_get(m, "q.last", (size_t)N * H, &ql));
    VD_TRY(vd_embed_gather(hist2.out_at(b.h.T - 1), b.h.inv, nullptr, hl, N, (int)H, 1.f, s));   // Select(1,-1), back to batch order
    VD_TRY(vd_embed_gather(ques2.out_at(b.q.T - 1), b.q.inv, nullptr, ql, N, (int)H, 1.f, s));
    *q3 = ql;
    *h3 = hl;
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* dq3, const float* dh3) {
    const int N = b.q.N;
    float *dh3s, *dq3s;
    VD_TRY(ws_get(m, "h.dlast", (size_t)N * H, &dh3s));
    VD_TRY(ws_get(m, "q.dlast", (size_t)N * H, &dq3s));
    VD_TRY(vd_embed_gather(dh3, b.h.perm, nullptr, dh3s, N, (int)H, 1.f, s));
    VD_TRY(vd_embed_gather(dq3, b.q.perm, nullptr, dq3s, N, (int)H, 1.f, s));
    vd_lstm2_bwd_t bw[2];
    SeqLSTM* L1[2] = {&hist1, &ques1};
    SeqLSTM* L2[2] = {&hist2, &ques2};
    const SeqTok* ss[2] = {&b.h, &b.q};
    const float* dl[2] = {dh3s, dq3s};
    for (int k = 0; k < 2; ++k) {
      const size_t TN = (size_t)ss[k]->T * N;
      float *dhseq, *dc1, *dc2;
      VD_TRY(ws_get(m, L1[k]->name + ".dhseq", TN * H, &dhseq));
      VD_TRY(ws_get(m, L1[k]->name + ".dc", (size_t)N * H, &dc1));
      VD_TRY(ws_get(m, L2[k]->name + ".dc", (size_t)N * H, &dc2));
      bw[k].T = ss[k]->T; bw[k].N = N;
      bw[k].Wh1 = L1[k]->Wh(m); bw[k].Wx2 = L2[k]->Wx(m); bw[k].Wh2 = L2[k]->Wh(m);
      bw[k].gates1 = L1[k]->gates; bw[k].c1 = L1[k]->c; bw[k].gates2 = L2[k]->gates; bw[k].c2 = L2[k]->c;
      bw[k].dh_last2 = dl[k];
      bw[k].dh1_seq = dhseq; bw[k].dc1 = dc1; bw[k].dc2 = dc2;
      bw[k].nact = ss[k]->nact.data();
    }
    VD_TRY(vd_lstm2_backward_p(bw, 2, (int)H, tick_flags(m), s));
    const uint8_t* mk[2] = {m_h, m_q};
    const char* tag[2] = {"h", "q"};
    hipStream_t sw;
    VD_TRY(wg_fork(m, s, &sw));                 // the recurrences are done: only parameter gradients are left
    for (int k = 0; k < 2; ++k) {
      const long TN = (long)ss[k]->T * N;
      std::vector<float*> dx;
      VD_TRY(L2[k]->param_grads(m, sw, {false}, nullptr));
      VD_TRY(L1[k]->param_grads(m, sw, {true}, &dx));
      float* dxo;
      VD_TRY(ws_get(m, std::string(tag[k]) + ".dxo", (size_t)TN * E, &dxo));
      VD_TRY(vd_embed_gather(dx[0], ss[k]->inv_idx, nullptr, dxo, TN, (int)E, 1.f, sw));     // back to batch order
      VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), ss[k]->tok, mk[k], dxo, TN, (int)E, 2.f, sw));
    }
    return VD_OK;
  }
};

// The history branch of the Sequential encoders (encoders/lf-ques-hist.lua:38-45, lf-ques-im-hist.lua:38-45, hre-*.lua:30-38):
// numLayers x SeqLSTM:maskZero() whose only consumer is Select(1,-1) of the top layer.  With the default two layers it runs like
// the text branches of the nngraph encoders: rows sorted by length at upload (right-aligned input: the leading pad steps of the
// shorter rows are skipped, maskZero's zero state is what they would produce), both layers as ONE skewed wavefront
// (vd_lstm2_forward / vd_lstm2_backward: T + 1 launches instead of 2 T), weight gradients over the non-pad (t, row) pairs.  At
// lf-ques-im-hist's Th <= 300 (the concatenated dialog, dataloader.lua:243-255) that branch IS the step (configs[1]).
struct HistWave {
  float *xs = nullptr;
  static bool usable(const BatchSlot& b, const std::vector<SeqLSTM>& hist) { return hist.size() == 2 && b.h.sorted; }
  int forward(vd_model* m, hipStream_t s, BatchSlot& b, SeqLSTM& l1, SeqLSTM& l2, const float** last) {
    const int N = b.h.N;
    const long H = l1.H;
    uint8_t* mk;
    VD_TRY(TextBranches::prepare(m, s, b.h, "h", l1, l2, &mk, &xs, false));
    vd_lstm2_fwd_t fw;
    TextBranches::fill_fwd(m, b.h, l1, l2, &fw);
    VD_TRY(vd_lstm2_forward_p(&fw, 1, (int)H, tick_flags(m), s));
    float* hl;
    VD_TRY(ws_get(m, "h.last", (size_t)N * H, &hl));
    VD_TRY(vd_embed_gather(l2.out_at(b.h.T - 1), b.h.inv, nullptr, hl, N, (int)H, 1.f, s));   // Select(1,-1), back to batch order
    *last = hl;
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, SeqLSTM& l1, SeqLSTM& l2, const float* dlast) {
    const int N = b.h.N;
    const long H = l1.H, E = l1.D;
    const size_t TN = (size_t)b.h.T * N;
    float *dls, *dhseq, *dc1, *dc2, *dxo;
    VD_TRY(ws_get(m, "h.dlast", (size_t)N * H, &dls));
    VD_TRY(vd_embed_gather(dlast, b.h.perm, nullptr, dls, N, (int)H, 1.f, s));
    VD_TRY(ws_get(m, l1.name + ".dhseq", TN * H, &dhseq));
    VD_TRY(ws_get(m, l1.name + ".dc", (size_t)N * H, &dc1));
    VD_TRY(ws_get(m, l2.name + ".dc", (size_t)N * H, &dc2));
    vd_lstm2_bwd_t bw;
    bw.T = b.h.T; bw.N = N;
    bw.Wh1 = l1.Wh(m); bw.Wx2 = l2.Wx(m); bw.Wh2 = l2.Wh(m);
    bw.gates1 = l1.gates; bw.c1 = l1.c; bw.gates2 = l2.gates; bw.c2 = l2.c;
    bw.dh_last2 = dls;
    bw.dh1_seq = dhseq; bw.dc1 = dc1; bw.dc2 = dc2;
    bw.nact = b.h.nact.data();
    VD_TRY(vd_lstm2_backward_p(&bw, 1, (int)H, tick_flags(m), s));
    std::vector<float*> dx;
    VD_TRY(l2.param_grads(m, s, {false}, nullptr));
    VD_TRY(l1.param_grads(m, s, {true}, &dx));
    VD_TRY(ws_get(m, "h.dxo", TN * E, &dxo));
    VD_TRY(vd_embed_gather(dx[0], b.h.inv_idx, nullptr, dxo, (long)TN, (int)E, 1.f, s));       // back to batch order
    return vd_embed_scatter_acc(Gp(m, "embed"), b.h.tok, nullptr, dxo, (long)TN, (int)E, 1.f, s);
  }
};

// nn.MM(false,true) -> MaskSoftMax -> nn.MM -> Tanh(Linear(Dropout)) -> Tanh(Linear(hAttTr + query)) (mn-att:48-65)
struct MemoryBlock {
  Linear mn1, mn2;
  long H = 0;
  int R = 0, N = 0, B = 0;
  const float *query = nullptr, *h3 = nullptr;
  float *prob = nullptr, *s2 = nullptr;
  uint8_t* m_hatt = nullptr;
  static void declare(vd_model* m) {
    const long H = m->p.rnnHiddenSize;
    add_linear(m, "mn1", H, H); add_linear(m, "mn2", H, H);
  }
  void init(vd_model* m) {
    H = m->p.rnnHiddenSize; R = m->p.maxQuesCount;
    mn1.init("mn1", H, H); mn2.init("mn2", H, H);
  }
  int forward(vd_model* m, hipStream_t s, const float* query_, const float* h3_, int N_, float** out) {
    query = query_; h3 = h3_; N = N_; B = N / R;
    uint8_t* mask;
    VD_TRY(causal_mask(m, B, R, &mask));
    float* hatt;
    VD_TRY(ws_get(m, "mn.prob", (size_t)N * R, &prob));
    VD_TRY(ws_get(m, "mn.hatt", (size_t)N * H, &hatt));
    VD_TRY(ws_get(m, "mn.s2", (size_t)N * H, &s2));
    VD_TRY(vd_mn_attention_forward(query, h3, mask, prob, hatt, B, R, (int)H, s));
    VD_TRY(drop_mask(m, "hatt", (size_t)N * H, 0.5f, s, &m_hatt));
    const float* hd;
    VD_TRY(dropout_fwd(m, "mn.hatt_d", hatt, m_hatt, 2.f, (long)N * H, s, &hd));
    float* hattTr;
    VD_TRY(mn1.forward(m, s, hd, N, true, &hattTr));
    VD_TRY(vd_axpby(hattTr, query, s2, (long)N * H, 1.f, 1.f, s));   // CAddTable
    return mn2.forward(m, s, s2, N, true, out);
  }
  int backward(vd_model* m, hipStream_t s, const float* dqh2, float** dquery, float** dh3) {
    float *ds2, *dhatt_d;
    VD_TRY(mn2.backward(m, s, dqh2, true, &ds2));
    VD_TRY(mn1.backward(m, s, ds2, true, &dhatt_d));
    const float* dha;
    VD_TRY(dropout_fwd(m, "mn.dhatt", dhatt_d, m_hatt, 2.f, (long)N * H, s, &dha));
    float *dq_att, *dh, *dq;
    VD_TRY(ws_get(m, "mn.dq", (size_t)N * H, &dq_att));
    VD_TRY(ws_get(m, "mn.dh", (size_t)N * H, &dh));
    VD_TRY(ws_get(m, "mn.dquery", (size_t)N * H, &dq));
    VD_TRY(vd_mn_attention_backward(query, h3, prob, dha, dq_att, dh, B, R, (int)H, s));
    VD_TRY(vd_axpby(dq_att, ds2, dq, (long)N * H, 1.f, 1.f, s));
    *dquery = dq;
    *dh3 = dh;
    return VD_OK;
  }
};

// Stacked attention over the S x S image regions + output layer (mn-att:68-106), numAttentionLayers hops, each with its
// OWN img_common / ques_common / att Linears and Dropout; `pre` = tanh(Linear(img)) once per IMAGE, per-round Dropout
// masks applied by the GEMM loaders, the 10x repeat (model.lua:262-265) never materialises.
struct SANBlock {
  Linear img_proj, out;
  std::vector<Linear> ques_common;
  long H = 0, C = 0, K = 0;
  int S2 = 0, R = 0, L = 1, N = 0;
  float* pre = nullptr;
  uint8_t *m1 = nullptr, *m_u = nullptr;
  std::vector<uint8_t*> m2;
  std::vector<const float*> u_in;
  std::vector<float*> iqc, patt;
  float sc = 1.f;
  float* xdrop = nullptr;   // split9 pass: dropout1(gather(pre)) [N*S2 x H], materialised once per step
  int iflags = 0;
  static void declare(vd_model* m) {
    const long H = m->p.rnnHiddenSize, C = m->p.imgFeatureSize, K = m->p.commonEmbeddingSize;
    add_linear(m, "img_proj", C, H);
    for (int i = 0; i < std::max(1, m->p.numAttentionLayers); ++i) {
      add_linear(m, "img_common" + hop_sfx(i), H, K);
      add_linear(m, "ques_common" + hop_sfx(i), H, K);
      add_linear(m, "att" + hop_sfx(i), K, 1);
    }
    add_linear(m, "out", H, H);
  }
  void init(vd_model* m) {
    H = m->p.rnnHiddenSize; C = m->p.imgFeatureSize; K = m->p.commonEmbeddingSize;
    S2 = m->p.imgSpatialSize * m->p.imgSpatialSize; R = m->p.maxQuesCount; L = std::max(1, m->p.numAttentionLayers);
    img_proj.init("img_proj", C, H); out.init("out", H, H);
    ques_common.resize(L);
    for (int i = 0; i < L; ++i) ques_common[i].init("ques_common" + hop_sfx(i), H, K);
  }
  // per-image projection + this step's dropout masks: independent of the text branches -> own stream
  int prefetch(vd_model* m, hipStream_t s, BatchSlot& b, int N_) {
    N = N_;
    hipStream_t si = side_stream(m, m->s_img, s);
    VD_TRY(fork_stream(m, s, si));
    VD_TRY(img_proj.forward(m, si, b.img, (long)b.B * S2, true, &pre));           // mn-att:74-78 (pre-dropout)
    VD_TRY(drop_mask(m, "img_tr", (size_t)N * S2 * H, 0.5f, si, &m1));
    m2.assign(L, nullptr);
    for (int i = 0; i < L; ++i) VD_TRY(drop_mask(m, "iqc" + hop_sfx(i), (size_t)N * S2 * K, 0.5f, si, &m2[i]));
    sc = m1 ? 2.f : 1.f;
    // split9 pass: the attention's dense products run on the exact split, from the materialised per-round image tensor (attention.hip)
    static const int img_split_on = [] { const char* e = getenv("VD_SPLIT_IMG"); return e ? atoi(e) : 1; }();     // (A/B switch)
    iflags = (m->p.lstmBf16 == 9 && img_split_on && (long)N * S2 >= 128 && H % 16 == 0 && K % 16 == 0) ? VD_FLAG_SPLIT9 : 0;
    xdrop = nullptr;
    if (iflags) {
      VD_TRY(ws_get(m, "att.xdrop", (size_t)N * S2 * H, &xdrop));
      VD_TRY(vd_img_drop_gather(pre, m1, xdrop, N, R, S2, (int)H, sc, si));
    }
    return VD_OK;
  }
  int forward(vd_model* m, hipStream_t s, const float* u0, float** y) {
    VD_TRY(join_stream(m, side_stream(m, m->s_img, s), s));
    u_in.assign(L, nullptr); iqc.assign(L, nullptr); patt.assign(L, nullptr);
    const float* u = u0;
    for (int i = 0; i < L; ++i) {
      const std::string sf = hop_sfx(i);
      float *qc, *u1;
      VD_TRY(ques_common[i].forward(m, s, u, N, false, &qc));                     // mn-att:88
      VD_TRY(ws_get(m, "att.iqc" + sf, (size_t)N * S2 * K, &iqc[i]));
      VD_TRY(ws_get(m, "att.p" + sf, (size_t)N * S2, &patt[i]));
      VD_TRY(ws_get(m, "att.u1" + sf, (size_t)N * H, &u1));
      VD_TRY(vd_img_common_forward_p(pre, m1, xdrop, Wp(m, "img_common" + sf + ".W"), Wp(m, "img_common" + sf + ".b"), qc, m2[i], iqc[i], N, R,
                                     S2, (int)H, (int)K, sc, iflags, s));         // mn-att:83-92
      VD_TRY(vd_img_att_forward(iqc[i], Wp(m, "att" + sf + ".W"), Wp(m, "att" + sf + ".b"), pre, m1, u, patt[i], u1, N, R, S2, (int)H,
                                (int)K, sc, s));                                  // mn-att:93-102
      u_in[i] = u;
      u = u1;
    }
    VD_TRY(drop_mask(m, "u", (size_t)N * H, 0.5f, s, &m_u));
    const float* ud;
    VD_TRY(dropout_fwd(m, "att.u1_d", u, m_u, 2.f, (long)N * H, s, &ud));
    return out.forward(m, s, ud, N, true, y);                                      // mn-att:106
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* grad_out, float** du0) {
    float* du1d;
    VD_TRY(out.backward(m, s, grad_out, true, &du1d));
    const float* dcur;
    VD_TRY(dropout_fwd(m, "att.du1", du1d, m_u, 2.f, (long)N * H, s, &dcur));
    float *dpre, *dscore;
    VD_TRY(ws_get(m, "att.dpre", (size_t)b.B * S2 * H, &dpre));
    VD_TRY(ws_get(m, "att.dscore", (size_t)N * S2, &dscore));
    VD_TRY(vd_memset(dpre, 0, (long)b.B * S2 * H * 4, s));
    for (int i = L - 1; i >= 0; --i) {
      const std::string sf = hop_sfx(i);
      float *dqc, *duq, *dun;
      VD_TRY(ws_get(m, "att.dqc" + sf, (size_t)N * K, &dqc));
      VD_TRY(ws_get(m, "att.du0" + sf, (size_t)N * H, &dun));
      VD_TRY(vd_img_att_backward(iqc[i], Wp(m, "att" + sf + ".W"), pre, m1, m2[i], patt[i], dcur, Gp(m, "att" + sf + ".W"),
                                 Gp(m, "att" + sf + ".b"), dqc, dscore, N, R, S2, (int)H, (int)K, sc, s));   // iqc now holds dz
      // dz is final: what follows on `sw` feeds parameter gradients only (the chain to the text branches continues with dqc)
      hipStream_t sw;
      VD_TRY(wg_fork(m, s, &sw));
      VD_TRY(vd_colsum_acc(iqc[i], K, N * S2, (int)K, Gp(m, "img_common" + sf + ".b"), sw));
      VD_TRY(vd_img_common_wgrad_p(iqc[i], pre, m1, xdrop, Gp(m, "img_common" + sf + ".W"), N, R, S2, (int)H, (int)K, sc, iflags, sw));
      VD_TRY(vd_img_tr_backward_p(iqc[i], Wp(m, "img_common" + sf + ".W"), patt[i], dcur, m1, dpre, N, R, S2, (int)H, (int)K, sc, iflags, sw));
      VD_TRY(ques_common[i].backward(m, s, dqc, true, &duq));
      VD_TRY(vd_axpby(duq, dcur, dun, (long)N * H, 1.f, 1.f, s));                 // residual CAddTable (mn-att:102)
      dcur = dun;
    }
    hipStream_t sw;
    VD_TRY(wg_fork(m, s, &sw));
    VD_TRY(img_proj.backward(m, sw, dpre, false, nullptr));                       // tanh' + dW, db of mn-att:77
    *du0 = const_cast<float*>(dcur);
    return VD_OK;
  }
};

// ------------------------------------------------------------------------------------------------------------
// the four nngraph encoders: mn-att-ques-im-hist, mn-ques-hist, mn-ques-im-hist, lf-att-ques-im-hist
// ------------------------------------------------------------------------------------------------------------
struct GraphEncoder : Encoder {
  bool memory_ = false, san_ = false, qi_ = false, qh_ = false;
  TextBranches text;
  MemoryBlock memory;
  SANBlock san;
  CatLinear qi, qh;
  explicit GraphEncoder(const std::string& name) {
    memory_ = name.rfind("mn", 0) == 0;
    san_ = name.find("att") != std::string::npos;
    qi_ = name == "mn-ques-im-hist";
    qh_ = name == "lf-att-ques-im-hist";
  }
  void declare(vd_model* m) override {
    const long H = m->p.rnnHiddenSize, F = m->p.imgFeatureSize;
    TextBranches::declare(m);
    if (qi_) add_linear(m, "qi", H + F, H);
    if (qh_) add_linear(m, "qh", 2 * H, H);
    if (memory_) MemoryBlock::declare(m);
    if (san_) SANBlock::declare(m);
    text.init(m);
    if (memory_) memory.init(m);
    if (san_) san.init(m);
    if (qi_) qi.init("qi", {H, F}, H);
    if (qh_) qh.init("qh", {H, H}, H);
  }
  int forward(vd_model* m, hipStream_t s, BatchSlot& b, float** out) override {
    const int N = b.q.N, R = m->p.maxQuesCount;
    if (san_) VD_TRY(san.prefetch(m, s, b, N));
    float *q3, *h3;
    VD_TRY(text.forward(m, s, b, &q3, &h3));
    float* u = nullptr;
    if (qh_) {
      VD_TRY(qh.forward(m, s, {q3, h3}, N, nullptr, 1.f, &u));                     // lf-att:43
    } else {
      const float* query = q3;
      if (qi_) {                                                                   // mn-ques-im-hist.lua:47-48
        int32_t* rep;
        float *img_rep, *qp;
        VD_TRY(round_index(m, N, R, &rep));
        VD_TRY(ws_get(m, "img.rep", (size_t)N * m->p.imgFeatureSize, &img_rep));
        VD_TRY(vd_embed_gather(b.img, rep, nullptr, img_rep, N, m->p.imgFeatureSize, 1.f, s));
        VD_TRY(qi.forward(m, s, {q3, img_rep}, N, nullptr, 1.f, &qp));
        query = qp;
      }
      VD_TRY(memory.forward(m, s, query, h3, N, &u));                              // mn-att:48-65
    }
    if (san_) return san.forward(m, s, u, out);                                    // mn-att:68-106
    *out = u;
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* grad_out) override {
    const float* du = grad_out;
    if (san_) {
      float* t;
      VD_TRY(san.backward(m, s, b, grad_out, &t));
      du = t;
    }
    const float *dq3, *dh3;
    if (qh_) {
      std::vector<float*> g;
      VD_TRY(qh.backward(m, s, du, {}, &g));
      dq3 = g[0];
      dh3 = g[1];
    } else {
      float *dquery, *dh;
      VD_TRY(memory.backward(m, s, du, &dquery, &dh));
      dq3 = dquery;
      dh3 = dh;
      if (qi_) {
        std::vector<float*> g;
        VD_TRY(qi.backward(m, s, dquery, {true, false}, &g));
        dq3 = g[0];
      }
    }
    return text.backward(m, s, b, dq3, dh3);
  }
};

// ------------------------------------------------------------------------------------------------------------
// Late fusion (encoders/lf-ques.lua, lf-ques-im.lua, lf-ques-hist.lua, lf-ques-im-hist.lua):
//   [quesLSTM last step ; img ; histLSTM last step] -> Dropout(p) -> Linear(., H) -> Tanh
// ------------------------------------------------------------------------------------------------------------
struct LateFusion : Encoder {
  bool use_im, use_hist;
  std::vector<SeqLSTM> rnn, hist;
  HistWave wave;
  CatLinear fuse;
  long E = 0, H = 0, F = 0;
  float pdrop = 0.5f;
  LateFusion(bool im, bool hs) : use_im(im), use_hist(hs) {}
  void declare(vd_model* m) override {
    E = m->p.embedSize; H = m->p.rnnHiddenSize; F = m->p.imgFeatureSize; pdrop = m->p.dropout;
    const int L = m->p.numLayers;
    rnn.resize(L);
    for (int l = 0; l < L; ++l) {
      add_lstm(m, "ques" + std::to_string(l + 1), l == 0 ? E : H, H);              // lf-ques-im-hist.lua:19-26
      rnn[l].init("ques" + std::to_string(l + 1), l == 0 ? E : H, H);
    }
    if (use_hist) {
      hist.resize(L);
      for (int l = 0; l < L; ++l) {
        add_lstm(m, "hist" + std::to_string(l + 1), l == 0 ? E : H, H);            // :38-45
        hist[l].init("hist" + std::to_string(l + 1), l == 0 ? E : H, H);
      }
    }
    std::vector<long> dims = {H};
    if (use_im) dims.push_back(F);
    if (use_hist) dims.push_back(H);
    long D = 0;
    for (long d : dims) D += d;
    add_linear(m, "fuse", D, H);                                                   // :58
    fuse.init("fuse", dims, H);
  }
  std::vector<SeqLSTM>* rnnLayers() override { return &rnn; }
  int forward(vd_model* m, hipStream_t s, BatchSlot& b, float** out) override {
    const int N = b.q.N, Tq = b.q.T, R = m->p.maxQuesCount;
    hipStream_t sh = side_stream(m, m->s_hist, s);
    const float* hh_last = nullptr;
    if (use_hist) {
      VD_TRY(fork_stream(m, s, sh));
      const int Th = b.h.T;
      const bool prof = m->dec_name == "gen";     // (disc pairs: ev_prof belongs to the option-LSTM families)
      if (prof) VD_HIP(hipEventRecord(m->ev_prof[0], sh));
      if (HistWave::usable(b, hist)) {
        VD_TRY(wave.forward(m, sh, b, hist[0], hist[1], &hh_last));
      } else {
        float *hx, *ht;
        VD_TRY(ws_get(m, "h.x", (size_t)Th * N * E, &hx));
        VD_TRY(vd_embed_gather(Wp(m, "embed"), b.h.tok, nullptr, hx, (long)Th * N, (int)E, 1.f, sh));
        VD_TRY(lstm_stack_forward(m, sh, hist, {hx}, Th, N, b.h.tok, &ht));
        hh_last = hist.back().out_at(Th - 1);
      }
      if (prof) VD_HIP(hipEventRecord(m->ev_prof[1], sh));
    }
    float *qx, *qt;
    VD_TRY(ws_get(m, "q.x", (size_t)Tq * N * E, &qx));
    VD_TRY(vd_embed_gather(Wp(m, "embed"), b.q.tok, nullptr, qx, (long)Tq * N, (int)E, 1.f, s));
    VD_TRY(lstm_stack_forward(m, s, rnn, {qx}, Tq, N, b.q.tok, &qt));
    std::vector<const float*> parts = {rnn.back().out_at(Tq - 1)};
    if (use_im) {
      int32_t* rep;
      float* img_rep;
      VD_TRY(round_index(m, N, R, &rep));
      VD_TRY(ws_get(m, "img.rep", (size_t)N * F, &img_rep));
      VD_TRY(vd_embed_gather(b.img, rep, nullptr, img_rep, N, (int)F, 1.f, s));    // repeat per round
      parts.push_back(img_rep);
    }
    if (use_hist) {
      VD_TRY(join_stream(m, sh, s));
      parts.push_back(hh_last);
    }
    uint8_t* mk;
    VD_TRY(drop_mask(m, "fuse", (size_t)N * fuse.D, pdrop, s, &mk));
    return fuse.forward(m, s, parts, N, mk, pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f, out);
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* grad_out) override {
    std::vector<bool> need = {true};
    if (use_im) need.push_back(false);
    if (use_hist) need.push_back(true);
    std::vector<float*> g;
    VD_TRY(fuse.backward(m, s, grad_out, need, &g));
    hipStream_t sh = side_stream(m, m->s_hist, s);
    if (use_hist) {
      VD_TRY(fork_stream(m, s, sh));
      const bool prof = m->dec_name == "gen";
      if (prof) VD_HIP(hipEventRecord(m->ev_prof[2], sh));
      if (HistWave::usable(b, hist)) {
        VD_TRY(wave.backward(m, sh, b, hist[0], hist[1], g.back()));
      } else {
        std::vector<float*> dx;
        VD_TRY(lstm_stack_backward(m, sh, hist, g.back(), nullptr, &dx));
        VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), b.h.tok, nullptr, dx[0], (long)b.h.T * b.h.N, (int)E, 1.f, sh));
      }
      if (prof) VD_HIP(hipEventRecord(m->ev_prof[3], sh));
      m->prof_hist = prof;
    }
    std::vector<float*> dx;
    VD_TRY(lstm_stack_backward(m, s, rnn, g[0], nullptr, &dx));
    VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), b.q.tok, nullptr, dx[0], (long)b.q.T * b.q.N, (int)E, 1.f, s));
    if (use_hist) VD_TRY(join_stream(m, sh, s));
    return VD_OK;
  }
};

// ------------------------------------------------------------------------------------------------------------
// Hierarchical recurrent encoders (encoders/hre-ques-hist.lua, hre-ques-im-hist.lua, hrea-ques-im-hist.lua)
// ------------------------------------------------------------------------------------------------------------
struct Hre : Encoder {
  bool use_im, attention;
  std::vector<SeqLSTM> rnn, hist;
  HistWave wave;
  SeqLSTM dialog;
  Linear img_embed;
  long E = 0, H = 0, F = 0, DI = 0;
  int R = 0;
  float *hh = nullptr, *hq = nullptr, *sq = nullptr, *sh_ = nullptr, *P = nullptr;
  uint8_t* m_img = nullptr;
  Hre(bool im, bool att) : use_im(im), attention(att) {}
  void declare(vd_model* m) override {
    E = m->p.embedSize; H = m->p.rnnHiddenSize; F = m->p.imgFeatureSize; DI = use_im ? m->p.imgEmbedSize : 0; R = m->p.maxQuesCount;
    const int L = m->p.numLayers;
    hist.resize(L); rnn.resize(L);
    for (int l = 0; l < L; ++l) {
      add_lstm(m, "hist" + std::to_string(l + 1), l == 0 ? E : H, H);
      hist[l].init("hist" + std::to_string(l + 1), l == 0 ? E : H, H);
    }
    if (use_im) {
      add_linear(m, "img_embed", F, DI);
      img_embed.init("img_embed", F, DI);
    }
    for (int l = 0; l < L; ++l) {
      const long D = l == 0 ? E + DI : H;
      add_lstm(m, "ques" + std::to_string(l + 1), D, H);
      rnn[l].init("ques" + std::to_string(l + 1), D, H, (l == 0 && use_im) ? std::vector<long>{E, DI} : std::vector<long>{});
    }
    if (attention) {
      add_linear(m, "att_q", H, 1);
      add_linear(m, "att_h", H, 1);
    }
    add_lstm(m, "dialog", 2 * H, H);
    dialog.init("dialog", 2 * H, H, {H, H});
  }
  std::vector<SeqLSTM>* rnnLayers() override { return &rnn; }
  // round-major <-> dialog-major row permutations (nn.View + nn.Transpose({1,2}), hre:88-93)
  int indices(vd_model* m, int N, int32_t** rep, int32_t** to_rb, int32_t** to_n) {
    const int B = N / R;
    std::vector<int32_t> a(N), c(N);
    for (int n = 0; n < N; ++n) {
      a[n] = (n % B) * R + n / B;   // row r*B+b  <- n = b*R + r
      c[n] = (n % R) * B + n / R;   // row b*R+r  <- r*B + b
    }
    VD_TRY(round_index(m, N, R, rep));
    VD_TRY(index_array(m, "idx.to_rb." + std::to_string(N), a, to_rb));
    return index_array(m, "idx.to_n." + std::to_string(N), c, to_n);
  }
  int forward(vd_model* m, hipStream_t s, BatchSlot& b, float** out) override {
    const int N = b.q.N, Tq = b.q.T, Th = b.h.T, B = N / R;
    int32_t *rep, *to_rb, *to_n;
    VD_TRY(indices(m, N, &rep, &to_rb, &to_n));
    hipStream_t sh = side_stream(m, m->s_hist, s);
    VD_TRY(fork_stream(m, s, sh));
    if (HistWave::usable(b, hist)) {
      const float* last;
      VD_TRY(wave.forward(m, sh, b, hist[0], hist[1], &last));
      hh = const_cast<float*>(last);
    } else {
      float *hx, *ht;
      VD_TRY(ws_get(m, "h.x", (size_t)Th * N * E, &hx));
      VD_TRY(vd_embed_gather(Wp(m, "embed"), b.h.tok, nullptr, hx, (long)Th * N, (int)E, 1.f, sh));
      VD_TRY(lstm_stack_forward(m, sh, hist, {hx}, Th, N, b.h.tok, &ht));
      hh = hist.back().out_at(Th - 1);
    }
    float* qx;
    VD_TRY(ws_get(m, "q.x", (size_t)Tq * N * E, &qx));
    VD_TRY(vd_embed_gather(Wp(m, "embed"), b.q.tok, nullptr, qx, (long)Tq * N, (int)E, 1.f, s));
    std::vector<const float*> x = {qx};
    if (use_im) {
      float *img_rep, *imgE, *xi;
      VD_TRY(ws_get(m, "img.rep", (size_t)N * F, &img_rep));
      VD_TRY(vd_embed_gather(b.img, rep, nullptr, img_rep, N, (int)F, 1.f, s));
      m_img = nullptr;
      if (attention) VD_TRY(drop_mask(m, "img", (size_t)N * F, 0.5f, s, &m_img));          // hrea:47
      const float* img_in;
      VD_TRY(dropout_fwd(m, "img.in", img_rep, m_img, 2.f, (long)N * F, s, &img_in));
      VD_TRY(img_embed.forward(m, s, img_in, N, false, &imgE));                           // hre:43-48
      VD_TRY(ws_get(m, "q.ximg", (size_t)Tq * N * DI, &xi));
      VD_TRY(vd_mask_time_forward(imgE, b.q.tok, xi, Tq, N, (int)DI, s));                 // MaskTime hre:50-53
      x.push_back(xi);
    }
    float* qt;
    VD_TRY(lstm_stack_forward(m, s, rnn, x, Tq, N, b.q.tok, &qt));
    hq = rnn.back().out_at(Tq - 1);
    VD_TRY(join_stream(m, sh, s));
    const float *first = hq, *second = hh;
    if (attention) {                                                                       // hrea:83-131
      float* att;
      VD_TRY(ws_get(m, "hrea.sq", (size_t)N, &sq));
      VD_TRY(ws_get(m, "hrea.sh", (size_t)N, &sh_));
      VD_TRY(ws_get(m, "hrea.P", (size_t)N * R, &P));
      VD_TRY(ws_get(m, "hrea.att", (size_t)N * H, &att));
      VD_TRY(vd_rowdot_forward(hq, Wp(m, "att_q.W"), Wp(m, "att_q.b"), sq, N, (int)H, s));
      VD_TRY(vd_rowdot_forward(hh, Wp(m, "att_h.W"), Wp(m, "att_h.b"), sh_, N, (int)H, s));
      VD_TRY(vd_hrea_attention_forward(sq, sh_, hh, P, att, B, R, (int)H, s));
      first = att;                                                                         // concat4: {att, ques}
      second = hq;
    }
    float *f_rb, *s_rb, *d, *o;
    VD_TRY(ws_get(m, "hre.f_rb", (size_t)N * H, &f_rb));
    VD_TRY(ws_get(m, "hre.s_rb", (size_t)N * H, &s_rb));
    VD_TRY(ws_get(m, "hre.out", (size_t)N * H, &o));
    VD_TRY(vd_embed_gather(first, to_rb, nullptr, f_rb, N, (int)H, 1.f, s));
    VD_TRY(vd_embed_gather(second, to_rb, nullptr, s_rb, N, (int)H, 1.f, s));
    VD_TRY(dialog.forward(m, s, {f_rb, s_rb}, R, B, nullptr, &d));                        // hre:90-94 (no maskZero)
    VD_TRY(vd_embed_gather(d, to_n, nullptr, o, N, (int)H, 1.f, s));
    *out = o;
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, BatchSlot& b, const float* grad_out) override {
    const int N = b.q.N, Tq = b.q.T, B = N / R;
    int32_t *rep, *to_rb, *to_n;
    VD_TRY(indices(m, N, &rep, &to_rb, &to_n));
    float *g_rb, *dfirst, *dsecond;
    VD_TRY(ws_get(m, "hre.g_rb", (size_t)N * H, &g_rb));
    VD_TRY(ws_get(m, "hre.dfirst", (size_t)N * H, &dfirst));
    VD_TRY(ws_get(m, "hre.dsecond", (size_t)N * H, &dsecond));
    VD_TRY(vd_embed_gather(grad_out, to_rb, nullptr, g_rb, N, (int)H, 1.f, s));
    std::vector<float*> dd;
    VD_TRY(dialog.backward(m, s, g_rb, nullptr, {}, &dd));
    VD_TRY(vd_embed_gather(dd[0], to_n, nullptr, dfirst, N, (int)H, 1.f, s));
    VD_TRY(vd_embed_gather(dd[1], to_n, nullptr, dsecond, N, (int)H, 1.f, s));
    const float *dq = dfirst, *dh = dsecond;
    if (attention) {
      float *dsq, *dsh, *dh_att, *dq_s, *dh_s, *dqq, *dhh;
      VD_TRY(ws_get(m, "hrea.dsq", (size_t)N, &dsq));
      VD_TRY(ws_get(m, "hrea.dsh", (size_t)N, &dsh));
      VD_TRY(ws_get(m, "hrea.dh", (size_t)N * H, &dh_att));
      VD_TRY(ws_get(m, "hrea.dq_s", (size_t)N * H, &dq_s));
      VD_TRY(ws_get(m, "hrea.dh_s", (size_t)N * H, &dh_s));
      VD_TRY(ws_get(m, "hrea.dq", (size_t)N * H, &dqq));
      VD_TRY(ws_get(m, "hrea.dhh", (size_t)N * H, &dhh));
      VD_TRY(vd_hrea_attention_backward(hh, P, dfirst, dsq, dsh, dh_att, B, R, (int)H, s));
      VD_TRY(vd_rowdot_backward(hq, Wp(m, "att_q.W"), dsq, Gp(m, "att_q.W"), Gp(m, "att_q.b"), dq_s, N, (int)H, s));
      VD_TRY(vd_rowdot_backward(hh, Wp(m, "att_h.W"), dsh, Gp(m, "att_h.W"), Gp(m, "att_h.b"), dh_s, N, (int)H, s));
      VD_TRY(vd_axpby(dsecond, dq_s, dqq, (long)N * H, 1.f, 1.f, s));
      VD_TRY(vd_axpby(dh_att, dh_s, dhh, (long)N * H, 1.f, 1.f, s));
      dq = dqq;
      dh = dhh;
    }
    hipStream_t sh = side_stream(m, m->s_hist, s);
    VD_TRY(fork_stream(m, s, sh));
    if (HistWave::usable(b, hist)) {
      VD_TRY(wave.backward(m, sh, b, hist[0], hist[1], dh));
    } else {
      std::vector<float*> dx;
      VD_TRY(lstm_stack_backward(m, sh, hist, dh, nullptr, &dx));
      VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), b.h.tok, nullptr, dx[0], (long)b.h.T * N, (int)E, 1.f, sh));
    }
    std::vector<float*> dx;
    VD_TRY(lstm_stack_backward(m, s, rnn, dq, nullptr, &dx));
    // the two scatters into the shared embedding gradient use float atomics: concurrent streams are fine
    VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), b.q.tok, nullptr, dx[0], (long)Tq * N, (int)E, 1.f, s));
    if (use_im) {
      float* dimgE;
      VD_TRY(ws_get(m, "q.dimgE", (size_t)N * DI, &dimgE));
      VD_TRY(vd_mask_time_backward(dx[1], b.q.tok, dimgE, Tq, N, (int)DI, s));
      // (hrea: the Dropout in front of the image Linear only scales its input; its gradient is not needed)
      VD_TRY(img_embed.backward(m, s, dimgE, false, nullptr));
    }
    return join_stream(m, sh, s);
  }
};

inline std::unique_ptr<Encoder> make_encoder(const std::string& n) {
  if (n == "mn-att-ques-im-hist" || n == "mn-ques-hist" || n == "mn-ques-im-hist" || n == "lf-att-ques-im-hist")
    return std::unique_ptr<Encoder>(new GraphEncoder(n));
  if (n == "lf-ques") return std::unique_ptr<Encoder>(new LateFusion(false, false));
  if (n == "lf-ques-im") return std::unique_ptr<Encoder>(new LateFusion(true, false));
  if (n == "lf-ques-hist") return std::unique_ptr<Encoder>(new LateFusion(false, true));
  if (n == "lf-ques-im-hist") return std::unique_ptr<Encoder>(new LateFusion(true, true));
  if (n == "hre-ques-hist") return std::unique_ptr<Encoder>(new Hre(false, false));
  if (n == "hre-ques-im-hist") return std::unique_ptr<Encoder>(new Hre(true, false));
  if (n == "hrea-ques-im-hist") return std::unique_ptr<Encoder>(new Hre(true, true));
  return nullptr;
}

}  // namespace vdrt
