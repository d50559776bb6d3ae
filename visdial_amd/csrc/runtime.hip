// Native step runtime: the model-level half of the C ABI (include/visdial_hip.h, "model-level entry points").
//
// The operator-level entry points replace single nn / rnn module calls; a host that composes them (Python,
// LuaJIT-FFI) must also own everything that makes the training step fast: stream fork/join, the skewed two-layer
// wavefront, the host-side length sort, workspace reuse, launch order.  This file puts that orchestration behind
// the ABI for the headline pair of the reference -- encoder `mn-att-ques-im-hist`
// (encoders/mn-att-ques-im-hist.lua:5-115) + decoder `disc` (decoders/disc.lua:3-38) -- as the proxy object a Lua
// `model.lua` drives: vd_model_create ~ Model:__init (model.lua:10-63), vd_model_upload_batch ~ the batch
// re-layout + :cuda() copies (model.lua:255-294, dataloader.lua:410-475), vd_model_forward_backward ~
// Model:forwardBackward (model.lua:249-342), vd_model_update ~ clamp + adam + lr decay (model.lua:96-105),
// vd_model_ranks ~ Model:retrieveBatch + utils.computeRanks (model.lua:344-430, utils.lua:106-128).
// No arithmetic lives here: every tensor op is one of the operator-level launches of this same library.
//
// Streams: main (option LSTM, criterion, optimiser), enc (encoder chains), img (per-image projection + masks),
// tab (token sort + table gradient), copy (H2D uploads of the NEXT batch; two batch slots).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/visdial_hip.h"
#include "common.h"

#define VD_TRY(expr)                  \
  do {                                \
    const int rc__ = (expr);          \
    if (rc__ != VD_OK) return rc__;   \
  } while (0)

namespace {

struct Tensor {
  std::string name;
  long off, rows, cols;
  int kind;  // 0 embed, 1 lstm weight, 2 lstm bias, 3 linear weight, 4 linear bias
  long numel() const { return rows * cols; }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// length-sorted view of a right-aligned [T x N] token matrix (host metadata + device copies)
struct SeqSort {
  int T = 0, N = 0;
  std::vector<int32_t> nact;               // host [T]
  int32_t *tok = nullptr, *tok_sorted = nullptr, *fwd_idx = nullptr, *inv_idx = nullptr, *perm = nullptr, *inv = nullptr,
          *nact_dev = nullptr;             // device
};

struct BatchSlot {
  int B = 0, Tq = 0, Th = 0, To = 0;
  SeqSort q, h;
  float* img = nullptr;        // [B*S2 x C]
  int32_t* options = nullptr;  // [To x N*O] time-major
  int32_t* gt = nullptr;       // [N] 0-based
  std::vector<int32_t> gt_host;
  hipEvent_t ready = nullptr;           // recorded on the copy stream when the upload has landed
  hipEvent_t done = nullptr;            // recorded on the main stream behind the last reader of this slot
  bool used = false;
  std::map<std::string, DevBuf> bufs;   // per-slot device buffers
  std::map<std::string, DevBuf> pinned; // per-slot pinned staging
};

struct EncState {  // what the encoder backward needs from the forward of the same step
  uint8_t *m_q = nullptr, *m_h = nullptr, *m_hatt = nullptr, *m1 = nullptr, *m_u = nullptr;
  std::vector<uint8_t*> m2;
  float *xs_q = nullptr, *xs_h = nullptr;
  float sc = 1.f;
};

}  // namespace

struct vd_model {
  vd_model_params p;
  std::vector<Tensor> spec;
  std::map<std::string, int> index;
  long numel = 0;
  float *W = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  std::map<std::string, DevBuf> ws;
  std::map<std::string, DevBuf> ext_masks;
  hipStream_t s_main = nullptr, s_enc = nullptr, s_img = nullptr, s_tab = nullptr, s_copy = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  hipEvent_t ev_loss = nullptr, ev_prof[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool training = true, streams = true;
  long step = 0;
  int adam_t = 0;
  double lr = 1e-3;
  BatchSlot slot[2];
  int cur = -1, uploaded = -1;
  float* loss_host = nullptr;  // pinned [N]
  int loss_n = 0;
  uint8_t* causal = nullptr;   // [N x R] bytes, 1 = hidden
  int causal_B = 0;
  // saved by forward for backward
  int N = 0, NO = 0;
  EncState enc;
};

namespace {

int dev_get(std::map<std::string, DevBuf>& m, const std::string& key, size_t bytes, void** out) {
  DevBuf& b = m[key];
  if (b.bytes < bytes) {
    if (b.p) VD_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    VD_HIP(hipMalloc(&b.p, bytes ? bytes : 16));
    b.bytes = bytes ? bytes : 16;
  }
  *out = b.p;
  return VD_OK;
}
int pin_get(std::map<std::string, DevBuf>& m, const std::string& key, size_t bytes, void** out) {
  DevBuf& b = m[key];
  if (b.bytes < bytes) {
    if (b.p) VD_HIP(hipHostFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    VD_HIP(hipHostMalloc(&b.p, bytes ? bytes : 16, hipHostMallocDefault));
    b.bytes = bytes ? bytes : 16;
  }
  *out = b.p;
  return VD_OK;
}

template <class T>
int ws_get(vd_model* m, const char* key, size_t count, T** out) {
  void* p = nullptr;
  VD_TRY(dev_get(m->ws, key, count * sizeof(T), &p));
  *out = static_cast<T*>(p);
  return VD_OK;
}

void add_lstm(vd_model* m, const char* name, long D, long H) {
  m->spec.push_back(Tensor{std::string(name) + ".W", 0, D + H, 4 * H, 1});
  m->spec.push_back(Tensor{std::string(name) + ".b", 0, 1, 4 * H, 2});
}
void add_linear(vd_model* m, const char* name, long in, long out) {
  m->spec.push_back(Tensor{std::string(name) + ".W", 0, out, in, 3});
  m->spec.push_back(Tensor{std::string(name) + ".b", 0, 1, out, 4});
}
std::string hop_sfx(int i) { return i == 0 ? std::string() : std::to_string(i + 1); }

float* Wp(vd_model* m, const std::string& n) { return m->W + m->spec[m->index.at(n)].off; }
float* Gp(vd_model* m, const std::string& n) { return m->G + m->spec[m->index.at(n)].off; }

// fork: `side` waits for everything enqueued on `from` so far; join: `to` waits for `side`
int fork_stream(vd_model* m, hipStream_t from, hipStream_t side) {
  if (side == from) return VD_OK;
  hipEvent_t e = m->ev_pool[m->ev_next++ % m->ev_pool.size()];
  VD_HIP(hipEventRecord(e, from));
  VD_HIP(hipStreamWaitEvent(side, e, 0));
  return VD_OK;
}
int join_stream(vd_model* m, hipStream_t side, hipStream_t to) { return fork_stream(m, side, to); }

// nn.Dropout keep-mask for a call site (null in evaluate mode); external masks pin the noise for parity runs
int drop_mask(vd_model* m, const char* site, size_t numel, hipStream_t s, uint8_t** out) {
  *out = nullptr;
  if (!m->training) return VD_OK;
  auto it = m->ext_masks.find(site);
  if (it != m->ext_masks.end()) {
    VD_CHECK_ARG(it->second.bytes >= numel, "dropout mask '%s' holds %zu bytes, the batch needs %zu", site, it->second.bytes,
                 numel);
    *out = static_cast<uint8_t*>(it->second.p);
    return VD_OK;
  }
  if (!m->ext_masks.empty()) {
    vd_set_error("external dropout masks are set but none for site '%s'", site);
    return VD_ERR_STATE;
  }
  uint8_t* buf = nullptr;
  VD_TRY(ws_get(m, (std::string("dropmask.") + site).c_str(), numel + 4, &buf));
  uint32_t h = 2166136261u;  // FNV-1a of the site name: one independent stream per (seed, step, site)
  for (const char* c = site; *c; ++c) h = (h ^ (uint8_t)*c) * 16777619u;
  const uint64_t seed = ((uint64_t)((m->p.seed * 1000003ull + (uint64_t)m->step) & 0xffffffffull) << 32) | h;
  VD_TRY(vd_dropout_mask(buf, (int64_t)numel, seed, 0.5f, s));
  *out = buf;
  return VD_OK;
}

// nn.Linear forward (+ optional tanh) / backward helpers on named parameters
int linear_fwd(vd_model* m, const std::string& name, const float* x, long M, long in, long out, float* y, bool tanh_, hipStream_t s) {
  return vd_gemm_nt(x, in, Wp(m, name + ".W"), in, Wp(m, name + ".b"), y, out, (int)M, (int)out, (int)in, tanh_ ? VD_ACT_TANH : VD_ACT_NONE, 0,
                    s);
}
// dy = gradient w.r.t. the output; with tanh_ the pre-activation gradient is formed in `dpre` first
int linear_bwd(vd_model* m, const std::string& name, const float* x, const float* y, const float* dy, float* dpre, long M, long in,
               long out, float* dx, bool tanh_, hipStream_t s) {
  const float* d = dy;
  if (tanh_) {
    VD_TRY(vd_tanh_backward(dy, y, dpre, M * out, s));
    d = dpre;
  }
  VD_TRY(vd_gemm_tn_acc(d, out, x, in, Gp(m, name + ".W"), in, (int)out, (int)in, (int)M, 0, s));
  VD_TRY(vd_colsum_acc(d, out, (int)M, (int)out, Gp(m, name + ".b"), s));
  if (dx) VD_TRY(vd_gemm_nn(d, out, Wp(m, name + ".W"), in, nullptr, dx, in, (int)M, (int)in, (int)out, 0, s));
  return VD_OK;
}

int build_seqsort(vd_model* m, BatchSlot& sl, SeqSort& ss, const char* tag, const int32_t* rows_major, int N, int T, hipStream_t s) {
  // rows_major: [N x T] right-aligned (the dataloader's layout).  Host side: time-major transpose, lengths, stable
  // sort by decreasing length, per-step active-row counts, gather indices (nn.py SeqSort).
  ss.T = T;
  ss.N = N;
  const size_t TN = (size_t)T * N;
  int32_t* stage = nullptr;
  VD_TRY(pin_get(sl.pinned, std::string(tag) + ".stage", (4 * TN + 2 * (size_t)N + T) * sizeof(int32_t), (void**)&stage));
  int32_t* tok = stage;
  int32_t* tok_sorted = stage + TN;
  int32_t* fwd_idx = stage + 2 * TN;
  int32_t* inv_idx = stage + 3 * TN;
  int32_t* perm = stage + 4 * TN;
  int32_t* inv = perm + N;
  int32_t* nact = inv + N;
  std::vector<int> len(N);
  for (int n = 0; n < N; ++n) {
    int l = 0;
    for (int t = 0; t < T; ++t) {
      const int32_t v = rows_major[(size_t)n * T + t];
      tok[(size_t)t * N + n] = v;
      l += v != 0;
    }
    len[n] = l;
  }
  std::vector<int> order(N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
  for (int i = 0; i < N; ++i) {
    perm[i] = order[i];
    inv[order[i]] = i;
  }
  ss.nact.assign(T, 0);
  for (int t = 0; t < T; ++t) {
    int c = 0;
    for (int n = 0; n < N; ++n) c += len[n] >= T - t;
    ss.nact[t] = c;
    nact[t] = c;
    for (int i = 0; i < N; ++i) {
      tok_sorted[(size_t)t * N + i] = tok[(size_t)t * N + perm[i]];
      fwd_idx[(size_t)t * N + i] = t * N + perm[i];
      inv_idx[(size_t)t * N + i] = t * N + inv[i];
    }
  }
  int32_t* dev = nullptr;
  const size_t total = 4 * TN + 2 * (size_t)N + T;
  VD_TRY(dev_get(sl.bufs, std::string(tag) + ".dev", total * sizeof(int32_t), (void**)&dev));
  VD_HIP(hipMemcpyAsync(dev, stage, total * sizeof(int32_t), hipMemcpyHostToDevice, s));
  ss.tok = dev;
  ss.tok_sorted = dev + TN;
  ss.fwd_idx = dev + 2 * TN;
  ss.inv_idx = dev + 3 * TN;
  ss.perm = dev + 4 * TN;
  ss.inv = ss.perm + N;
  ss.nact_dev = ss.inv + N;
  return VD_OK;
}

struct Dims {
  int B, R, N, O, NO, E, H, K, C, S2, V, L;
};
Dims dims(const vd_model* m, const BatchSlot& b) {
  Dims d;
  d.B = b.B;
  d.R = m->p.maxQuesCount;
  d.N = d.B * d.R;
  d.O = m->p.numOptions;
  d.NO = d.N * d.O;
  d.E = m->p.embedSize;
  d.H = m->p.rnnHiddenSize;
  d.K = m->p.commonEmbeddingSize;
  d.C = m->p.imgFeatureSize;
  d.S2 = m->p.imgSpatialSize * m->p.imgSpatialSize;
  d.V = m->p.vocabSize;
  d.L = m->p.numAttentionLayers > 0 ? m->p.numAttentionLayers : 1;
  return d;
}

// ------------------------------------------------------------------------------------------------------------
// encoder forward (encoders/mn-att-ques-im-hist.lua:21-106) on stream s (+ the img side stream)
// ------------------------------------------------------------------------------------------------------------
int text_branch_prepare(vd_model* m, const Dims& d, const SeqSort& ss, const char* tag, const char* l1, hipStream_t s, uint8_t** mask_out,
                        float** xs_out) {
  // shared LookupTableMaskZero -> Dropout(0.5) (mn-att:21,24-25), rows permuted per step into length-sorted order,
  // hoisted layer-1 input projection written into the gates buffer
  const long TN = (long)ss.T * ss.N;
  const std::string t(tag);
  uint8_t* mk = nullptr;
  VD_TRY(drop_mask(m, (t + "_emb").c_str(), (size_t)TN * d.E, s, &mk));
  float *x, *xs, *gates1, *h1, *c1, *gates2, *h2, *c2;
  VD_TRY(ws_get(m, (t + ".x").c_str(), (size_t)TN * d.E, &x));
  VD_TRY(ws_get(m, (t + ".xs").c_str(), (size_t)TN * d.E, &xs));
  VD_TRY(ws_get(m, (std::string(l1) + "1.gates").c_str(), (size_t)TN * 4 * d.H, &gates1));
  VD_TRY(ws_get(m, (std::string(l1) + "1.h").c_str(), (size_t)TN * d.H, &h1));
  VD_TRY(ws_get(m, (std::string(l1) + "1.c").c_str(), (size_t)TN * d.H, &c1));
  VD_TRY(ws_get(m, (std::string(l1) + "2.gates").c_str(), (size_t)TN * 4 * d.H, &gates2));
  VD_TRY(ws_get(m, (std::string(l1) + "2.h").c_str(), (size_t)TN * d.H, &h2));
  VD_TRY(ws_get(m, (std::string(l1) + "2.c").c_str(), (size_t)TN * d.H, &c2));
  VD_TRY(vd_embed_gather(Wp(m, "embed"), ss.tok, mk, x, TN, d.E, 2.0f, s));
  VD_TRY(vd_embed_gather(x, ss.fwd_idx, nullptr, xs, TN, d.E, 1.0f, s));
  const std::string w1 = std::string(l1) + "1";
  VD_TRY(vd_gemm_nn(xs, d.E, Wp(m, w1 + ".W"), 4L * d.H, Wp(m, w1 + ".b"), gates1, 4L * d.H, (int)TN, 4 * d.H, d.E, 0, s));
  // skipped (t, row) pairs must read as zeros
  VD_TRY(vd_zero_inactive_rows(gates1, (int64_t)ss.N * 4 * d.H, 4L * d.H, 4 * d.H, ss.nact_dev, ss.T, ss.N, s));
  VD_TRY(vd_memset(h1, 0, TN * d.H * 4, s));
  VD_TRY(vd_memset(c1, 0, TN * d.H * 4, s));
  VD_TRY(vd_memset(h2, 0, TN * d.H * 4, s));
  VD_TRY(vd_memset(c2, 0, TN * d.H * 4, s));
  VD_TRY(vd_memset(gates2, 0, TN * 4 * d.H * 4, s));
  *mask_out = mk;
  *xs_out = xs;
  return VD_OK;
}

int lstm2_desc_fwd(vd_model* m, const Dims& d, const SeqSort& ss, const char* l, vd_lstm2_fwd_t* o) {
  const std::string p(l);
  float *g1, *h1, *c1, *g2, *h2, *c2;
  const size_t TN = (size_t)ss.T * ss.N;
  VD_TRY(ws_get(m, (p + "1.gates").c_str(), TN * 4 * d.H, &g1));
  VD_TRY(ws_get(m, (p + "1.h").c_str(), TN * d.H, &h1));
  VD_TRY(ws_get(m, (p + "1.c").c_str(), TN * d.H, &c1));
  VD_TRY(ws_get(m, (p + "2.gates").c_str(), TN * 4 * d.H, &g2));
  VD_TRY(ws_get(m, (p + "2.h").c_str(), TN * d.H, &h2));
  VD_TRY(ws_get(m, (p + "2.c").c_str(), TN * d.H, &c2));
  o->T = ss.T;
  o->N = ss.N;
  o->tok_mask = ss.tok_sorted;
  o->Wh1 = Wp(m, p + "1.W") + (long)d.E * 4 * d.H;
  o->Wx2 = Wp(m, p + "2.W");
  o->b2 = Wp(m, p + "2.b");
  o->Wh2 = Wp(m, p + "2.W") + (long)d.H * 4 * d.H;
  o->gates1 = g1; o->h1 = h1; o->c1 = c1; o->gates2 = g2; o->h2 = h2; o->c2 = c2;
  o->nact = ss.nact.data();
  return VD_OK;
}

int encoder_forward(vd_model* m, BatchSlot& b, EncState& st, float** enc_out, hipStream_t s) {
  const Dims d = dims(m, b);
  const int N = d.N, H = d.H, K = d.K, S2 = d.S2, R = d.R;
  // --- image branch prefetch (img stream): per-IMAGE projection + this step's masks (mn-att:74-78)
  hipStream_t si = m->streams ? m->s_img : s;
  VD_TRY(fork_stream(m, s, si));
  float* pre;
  VD_TRY(ws_get(m, "img_proj.y", (size_t)d.B * S2 * H, &pre));
  VD_TRY(linear_fwd(m, "img_proj", b.img, (long)d.B * S2, d.C, H, pre, true, si));
  VD_TRY(drop_mask(m, "img_tr", (size_t)N * S2 * H, si, &st.m1));
  st.m2.assign(d.L, nullptr);
  for (int i = 0; i < d.L; ++i) VD_TRY(drop_mask(m, ("iqc" + hop_sfx(i)).c_str(), (size_t)N * S2 * K, si, &st.m2[i]));
  st.sc = st.m1 ? 2.f : 1.f;
  // --- text branches: both 2-layer stacks advance as one skewed wavefront (mn-att:27-45)
  VD_TRY(text_branch_prepare(m, d, b.h, "h", "hist", s, &st.m_h, &st.xs_h));
  VD_TRY(text_branch_prepare(m, d, b.q, "q", "ques", s, &st.m_q, &st.xs_q));
  vd_lstm2_fwd_t fw[2];
  VD_TRY(lstm2_desc_fwd(m, d, b.h, "hist", &fw[0]));
  VD_TRY(lstm2_desc_fwd(m, d, b.q, "ques", &fw[1]));
  VD_TRY(vd_lstm2_forward(fw, 2, H, s));
  float *h3, *q3;
  VD_TRY(ws_get(m, "h.last", (size_t)N * H, &h3));
  VD_TRY(ws_get(m, "q.last", (size_t)N * H, &q3));
  VD_TRY(vd_embed_gather(fw[0].h2 + (long)(b.h.T - 1) * N * H, b.h.inv, nullptr, h3, N, H, 1.f, s));   // Select(1,-1), batch order
  VD_TRY(vd_embed_gather(fw[1].h2 + (long)(b.q.T - 1) * N * H, b.q.inv, nullptr, q3, N, H, 1.f, s));
  // --- memory attention + the two Tanh(Linear) (mn-att:48-65)
  float *prob, *hatt, *hatt_d, *hattTr, *s2, *qh2;
  VD_TRY(ws_get(m, "mn.prob", (size_t)N * R, &prob));
  VD_TRY(ws_get(m, "mn.hatt", (size_t)N * H, &hatt));
  VD_TRY(ws_get(m, "mn.hatt_d", (size_t)N * H, &hatt_d));
  VD_TRY(ws_get(m, "mn1.y", (size_t)N * H, &hattTr));
  VD_TRY(ws_get(m, "mn.s2", (size_t)N * H, &s2));
  VD_TRY(ws_get(m, "mn2.y", (size_t)N * H, &qh2));
  VD_TRY(vd_mn_attention_forward(q3, h3, m->causal, prob, hatt, d.B, R, H, s));
  VD_TRY(drop_mask(m, "hatt", (size_t)N * H, s, &st.m_hatt));
  const float* hd = hatt;
  if (st.m_hatt) {
    VD_TRY(vd_dropout_apply(hatt, st.m_hatt, hatt_d, (long)N * H, 2.f, s));
    hd = hatt_d;
  }
  VD_TRY(linear_fwd(m, "mn1", hd, N, H, H, hattTr, true, s));
  VD_TRY(vd_axpby(hattTr, q3, s2, (long)N * H, 1.f, 1.f, s));                 // CAddTable
  VD_TRY(linear_fwd(m, "mn2", s2, N, H, H, qh2, true, s));
  // --- SAN image attention, L hops (mn-att:82-104) + output layer (mn-att:106)
  VD_TRY(join_stream(m, si, s));
  const float* u = qh2;
  for (int i = 0; i < d.L; ++i) {
    const std::string sf = hop_sfx(i);
    float *qc, *iqc, *patt, *u1;
    VD_TRY(ws_get(m, ("ques_common" + sf + ".y").c_str(), (size_t)N * K, &qc));
    VD_TRY(ws_get(m, ("att.iqc" + sf).c_str(), (size_t)N * S2 * K, &iqc));
    VD_TRY(ws_get(m, ("att.p" + sf).c_str(), (size_t)N * S2, &patt));
    VD_TRY(ws_get(m, ("att.u1" + sf).c_str(), (size_t)N * H, &u1));
    VD_TRY(linear_fwd(m, "ques_common" + sf, u, N, H, K, qc, false, s));
    VD_TRY(vd_img_common_forward(pre, st.m1, Wp(m, "img_common" + sf + ".W"), Wp(m, "img_common" + sf + ".b"), qc, st.m2[i], iqc, N, R, S2, H,
                                 K, st.sc, s));
    VD_TRY(vd_img_att_forward(iqc, Wp(m, "att" + sf + ".W"), Wp(m, "att" + sf + ".b"), pre, st.m1, u, patt, u1, N, R, S2, H, K, st.sc, s));
    u = u1;
  }
  VD_TRY(drop_mask(m, "u", (size_t)N * H, s, &st.m_u));
  float *u_d, *out;
  VD_TRY(ws_get(m, "att.u1_d", (size_t)N * H, &u_d));
  VD_TRY(ws_get(m, "out.y", (size_t)N * H, &out));
  const float* ud = u;
  if (st.m_u) {
    VD_TRY(vd_dropout_apply(u, st.m_u, u_d, (long)N * H, 2.f, s));
    ud = u_d;
  }
  VD_TRY(linear_fwd(m, "out", ud, N, H, H, out, true, s));
  *enc_out = out;
  return VD_OK;
}

// weight / bias / input gradients of one SeqLSTM layer from da (held in its gates buffer after BPTT)
int lstm_param_grads(vd_model* m, const std::string& name, const float* x, long D, const float* da, const float* h, int T, int N, int H,
                     float* dx, hipStream_t s) {
  const long TN = (long)T * N;
  float* dW = Gp(m, name + ".W");
  if (T > 1) VD_TRY(vd_gemm_tn_acc(h, H, da + (long)N * 4 * H, 4L * H, dW + D * 4 * H, 4L * H, H, 4 * H, (T - 1) * N, 0, s));
  VD_TRY(vd_colsum_acc(da, 4L * H, (int)TN, 4 * H, Gp(m, name + ".b"), s));
  VD_TRY(vd_gemm_tn_acc(x, D, da, 4L * H, dW, 4L * H, (int)D, 4 * H, (int)TN, 0, s));
  if (dx) VD_TRY(vd_gemm_nt(da, 4L * H, Wp(m, name + ".W"), 4L * H, nullptr, dx, D, (int)TN, (int)D, 4 * H, VD_ACT_NONE, 0, s));
  return VD_OK;
}

int encoder_backward(vd_model* m, BatchSlot& b, EncState& st, const float* grad_out, hipStream_t s) {
  const Dims d = dims(m, b);
  const int N = d.N, H = d.H, K = d.K, S2 = d.S2, R = d.R, E = d.E;
  auto W = [&](const char* k, size_t n, float** p) { return ws_get(m, k, n, p); };
  float *pre, *out, *u_d, *qh2, *s2, *hattTr, *hatt, *hatt_d, *prob, *q3, *h3;
  VD_TRY(W("img_proj.y", (size_t)d.B * S2 * H, &pre));
  VD_TRY(W("out.y", (size_t)N * H, &out));
  VD_TRY(W("att.u1_d", (size_t)N * H, &u_d));
  VD_TRY(W("mn2.y", (size_t)N * H, &qh2));
  VD_TRY(W("mn.s2", (size_t)N * H, &s2));
  VD_TRY(W("mn1.y", (size_t)N * H, &hattTr));
  VD_TRY(W("mn.hatt", (size_t)N * H, &hatt));
  VD_TRY(W("mn.hatt_d", (size_t)N * H, &hatt_d));
  VD_TRY(W("mn.prob", (size_t)N * R, &prob));
  VD_TRY(W("q.last", (size_t)N * H, &q3));
  VD_TRY(W("h.last", (size_t)N * H, &h3));
  // --- output layer + dropout
  float *dpre_o, *du1d, *du;
  VD_TRY(W("out.dpre", (size_t)N * H, &dpre_o));
  VD_TRY(W("out.dx", (size_t)N * H, &du1d));
  VD_TRY(W("att.du1", (size_t)N * H, &du));
  const float* u_last;
  {
    float* t;
    VD_TRY(W(("att.u1" + hop_sfx(d.L - 1)).c_str(), (size_t)N * H, &t));
    u_last = t;
  }
  VD_TRY(linear_bwd(m, "out", st.m_u ? u_d : u_last, out, grad_out, dpre_o, N, H, H, du1d, true, s));
  const float* dcur = du1d;
  if (st.m_u) {
    VD_TRY(vd_dropout_apply(du1d, st.m_u, du, (long)N * H, 2.f, s));
    dcur = du;
  }
  // --- attention hops in reverse
  float *dpre, *dscore;
  VD_TRY(W("att.dpre", (size_t)d.B * S2 * H, &dpre));
  VD_TRY(W("att.dscore", (size_t)N * S2, &dscore));
  VD_TRY(vd_memset(dpre, 0, (long)d.B * S2 * H * 4, s));
  for (int i = d.L - 1; i >= 0; --i) {
    const std::string sf = hop_sfx(i);
    float *iqc, *patt, *dqc, *duq, *dun, *qc;
    VD_TRY(W(("att.iqc" + sf).c_str(), (size_t)N * S2 * K, &iqc));
    VD_TRY(W(("att.p" + sf).c_str(), (size_t)N * S2, &patt));
    VD_TRY(W(("att.dqc" + sf).c_str(), (size_t)N * K, &dqc));
    VD_TRY(W(("ques_common" + sf + ".dx").c_str(), (size_t)N * H, &duq));
    VD_TRY(W(("att.du0" + sf).c_str(), (size_t)N * H, &dun));
    VD_TRY(W(("ques_common" + sf + ".y").c_str(), (size_t)N * K, &qc));
    const float* u_in;
    if (i == 0) u_in = qh2;
    else {
      float* t;
      VD_TRY(W(("att.u1" + hop_sfx(i - 1)).c_str(), (size_t)N * H, &t));
      u_in = t;
    }
    VD_TRY(vd_img_att_backward(iqc, Wp(m, "att" + sf + ".W"), pre, st.m1, st.m2[i], patt, dcur, Gp(m, "att" + sf + ".W"), Gp(m, "att" + sf + ".b"),
                               dqc, dscore, N, R, S2, H, K, st.sc, s));   // iqc now holds dz
    VD_TRY(vd_colsum_acc(iqc, K, N * S2, K, Gp(m, "img_common" + sf + ".b"), s));
    VD_TRY(vd_img_common_wgrad(iqc, pre, st.m1, Gp(m, "img_common" + sf + ".W"), N, R, S2, H, K, st.sc, s));
    VD_TRY(vd_img_tr_backward(iqc, Wp(m, "img_common" + sf + ".W"), patt, dcur, st.m1, dpre, N, R, S2, H, K, st.sc, s));
    VD_TRY(linear_bwd(m, "ques_common" + sf, u_in, qc, dqc, nullptr, N, H, K, duq, false, s));
    VD_TRY(vd_axpby(duq, dcur, dun, (long)N * H, 1.f, 1.f, s));           // residual CAddTable (mn-att:102)
    dcur = dun;
  }
  {
    float* dpp;
    VD_TRY(W("img_proj.dpre", (size_t)d.B * S2 * H, &dpp));
    VD_TRY(linear_bwd(m, "img_proj", b.img, pre, dpre, dpp, (long)d.B * S2, d.C, H, nullptr, true, s));
  }
  // --- memory block (mn-att:48-65)
  float *t1, *ds2, *t2, *dhatt_d, *dhatt, *dq_att, *dh3, *dq3;
  VD_TRY(W("mn2.dpre", (size_t)N * H, &t1));
  VD_TRY(W("mn2.dx", (size_t)N * H, &ds2));
  VD_TRY(W("mn1.dpre", (size_t)N * H, &t2));
  VD_TRY(W("mn1.dx", (size_t)N * H, &dhatt_d));
  VD_TRY(W("mn.dhatt", (size_t)N * H, &dhatt));
  VD_TRY(W("mn.dq", (size_t)N * H, &dq_att));
  VD_TRY(W("mn.dh", (size_t)N * H, &dh3));
  VD_TRY(W("mn.dquery", (size_t)N * H, &dq3));
  VD_TRY(linear_bwd(m, "mn2", s2, qh2, dcur, t1, N, H, H, ds2, true, s));
  VD_TRY(linear_bwd(m, "mn1", st.m_hatt ? hatt_d : hatt, hattTr, ds2, t2, N, H, H, dhatt_d, true, s));
  const float* dha = dhatt_d;
  if (st.m_hatt) {
    VD_TRY(vd_dropout_apply(dhatt_d, st.m_hatt, dhatt, (long)N * H, 2.f, s));
    dha = dhatt;
  }
  VD_TRY(vd_mn_attention_backward(q3, h3, prob, dha, dq_att, dh3, d.B, R, H, s));
  VD_TRY(vd_axpby(dq_att, ds2, dq3, (long)N * H, 1.f, 1.f, s));
  // --- text branches: fused two-layer BPTT for both stacks, then weight gradients and the embedding scatter
  float *dh3s, *dq3s;
  VD_TRY(W("h.dlast", (size_t)N * H, &dh3s));
  VD_TRY(W("q.dlast", (size_t)N * H, &dq3s));
  VD_TRY(vd_embed_gather(dh3, b.h.perm, nullptr, dh3s, N, H, 1.f, s));
  VD_TRY(vd_embed_gather(dq3, b.q.perm, nullptr, dq3s, N, H, 1.f, s));
  vd_lstm2_bwd_t bw[2];
  const SeqSort* ss[2] = {&b.h, &b.q};
  const char* nm[2] = {"hist", "ques"};
  const float* dl[2] = {dh3s, dq3s};
  float *g1[2], *h1[2], *g2[2], *h2[2];
  for (int k = 0; k < 2; ++k) {
    const std::string p(nm[k]);
    const size_t TN = (size_t)ss[k]->T * N;
    float *c1, *c2, *dhseq, *dc1, *dc2;
    VD_TRY(W((p + "1.gates").c_str(), TN * 4 * H, &g1[k]));
    VD_TRY(W((p + "1.h").c_str(), TN * H, &h1[k]));
    VD_TRY(W((p + "1.c").c_str(), TN * H, &c1));
    VD_TRY(W((p + "2.gates").c_str(), TN * 4 * H, &g2[k]));
    VD_TRY(W((p + "2.h").c_str(), TN * H, &h2[k]));
    VD_TRY(W((p + "2.c").c_str(), TN * H, &c2));
    VD_TRY(W((p + "1.dhseq").c_str(), TN * H, &dhseq));
    VD_TRY(W((p + "1.dc").c_str(), (size_t)N * H, &dc1));
    VD_TRY(W((p + "2.dc").c_str(), (size_t)N * H, &dc2));
    bw[k].T = ss[k]->T;
    bw[k].N = N;
    bw[k].Wh1 = Wp(m, p + "1.W") + (long)E * 4 * H;
    bw[k].Wx2 = Wp(m, p + "2.W");
    bw[k].Wh2 = Wp(m, p + "2.W") + (long)H * 4 * H;
    bw[k].gates1 = g1[k]; bw[k].c1 = c1; bw[k].gates2 = g2[k]; bw[k].c2 = c2;
    bw[k].dh_last2 = dl[k];
    bw[k].dh1_seq = dhseq; bw[k].dc1 = dc1; bw[k].dc2 = dc2;
    bw[k].nact = ss[k]->nact.data();
  }
  VD_TRY(vd_lstm2_backward(bw, 2, H, s));
  for (int k = 0; k < 2; ++k) {
    const std::string p(nm[k]);
    const int T = ss[k]->T;
    const long TN = (long)T * N;
    float *dx1, *dxo;
    VD_TRY(W((p + "1.dx").c_str(), (size_t)TN * E, &dx1));
    VD_TRY(W((std::string(k == 0 ? "h" : "q") + ".dxo").c_str(), (size_t)TN * E, &dxo));
    VD_TRY(lstm_param_grads(m, p + "2", h1[k], H, g2[k], h2[k], T, N, H, nullptr, s));
    VD_TRY(lstm_param_grads(m, p + "1", k == 0 ? st.xs_h : st.xs_q, E, g1[k], h1[k], T, N, H, dx1, s));
    VD_TRY(vd_embed_gather(dx1, ss[k]->inv_idx, nullptr, dxo, TN, E, 1.f, s));   // back to batch order
    VD_TRY(vd_embed_scatter_acc(Gp(m, "embed"), ss[k]->tok, k == 0 ? st.m_h : st.m_q, dxo, TN, E, 2.f, s));
  }
  return VD_OK;
}

}  // namespace

extern "C" {

int vd_model_create(const vd_model_params* p, const char* encoder, const char* decoder, vd_model** out) {
  VD_CHECK_ARG(p && encoder && decoder && out, "vd_model_create: null argument");
  if (strcmp(encoder, "mn-att-ques-im-hist") != 0 || strcmp(decoder, "disc") != 0) {
    vd_set_error("vd_model_create: the native step runtime covers encoder 'mn-att-ques-im-hist' + decoder 'disc' (got '%s' + '%s'); "
                 "other pairs run through the operator-level ABI (visdial_amd/ Python host)", encoder, decoder);
    return VD_ERR_ARG;
  }
  VD_CHECK_ARG(p->rnnHiddenSize % 32 == 0 && p->embedSize % 4 == 0 && p->commonEmbeddingSize % 4 == 0 && p->imgFeatureSize % 4 == 0,
               "vd_model_create: rnnHiddenSize must be a multiple of 32, embed/common/image sizes multiples of 4");
  vd_model* m = new vd_model();
  m->p = *p;
  if (m->p.numAttentionLayers < 1) m->p.numAttentionLayers = 1;
  m->lr = p->learningRate;
  m->streams = p->useStreams != 0;
  const long E = p->embedSize, H = p->rnnHiddenSize, K = p->commonEmbeddingSize, C = p->imgFeatureSize;
  m->spec.push_back(Tensor{"embed", 0, p->vocabSize + 1, E, 0});
  add_lstm(m, "hist1", E, H); add_lstm(m, "hist2", H, H); add_lstm(m, "ques1", E, H); add_lstm(m, "ques2", H, H);
  add_linear(m, "mn1", H, H); add_linear(m, "mn2", H, H);
  add_linear(m, "img_proj", C, H);
  for (int i = 0; i < m->p.numAttentionLayers; ++i) {
    add_linear(m, ("img_common" + hop_sfx(i)).c_str(), H, K);
    add_linear(m, ("ques_common" + hop_sfx(i)).c_str(), H, K);
    add_linear(m, ("att" + hop_sfx(i)).c_str(), K, 1);
  }
  add_linear(m, "out", H, H);
  add_lstm(m, "opt", E, H);
  long off = 0;
  for (size_t i = 0; i < m->spec.size(); ++i) {
    m->spec[i].off = off;
    m->index[m->spec[i].name] = (int)i;
    off += (m->spec[i].numel() + 3) / 4 * 4;  // every tensor 16-byte aligned (same layout as visdial_amd/params.py)
  }
  m->numel = off;
  auto fail = [&](int rc) {
    vd_model_destroy(m);
    return rc;
  };
  for (float** v : {&m->W, &m->G, &m->M, &m->V}) {
    if (hipMalloc((void**)v, off * sizeof(float)) != hipSuccess) {
      vd_set_error("vd_model_create: hipMalloc of %ld floats failed", off);
      return fail(VD_ERR_HIP);
    }
    (void)hipMemset(*v, 0, off * sizeof(float));
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // numerically lower = higher priority
  if (hipStreamCreateWithPriority(&m->s_main, hipStreamNonBlocking, lo) != hipSuccess) return fail(VD_ERR_HIP);
  for (hipStream_t* s : {&m->s_enc, &m->s_img, &m->s_tab, &m->s_copy})
    if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) != hipSuccess) return fail(VD_ERR_HIP);
  m->ev_pool.resize(64);
  for (auto& e : m->ev_pool)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
  if (hipEventCreateWithFlags(&m->ev_loss, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
  for (auto& e : m->ev_prof)
    if (hipEventCreate(&e) != hipSuccess) return fail(VD_ERR_HIP);
  for (auto& sl : m->slot)
    if (hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess)
      return fail(VD_ERR_HIP);
  *out = m;
  return VD_OK;
}

void vd_model_destroy(vd_model* m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  for (float* v : {m->W, m->G, m->M, m->V})
    if (v) (void)hipFree(v);
  for (auto& kv : m->ws)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& kv : m->ext_masks)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& sl : m->slot) {
    for (auto& kv : sl.bufs)
      if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : sl.pinned)
      if (kv.second.p) (void)hipHostFree(kv.second.p);
    if (sl.ready) (void)hipEventDestroy(sl.ready);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (m->loss_host) (void)hipHostFree(m->loss_host);
  if (m->causal) (void)hipFree(m->causal);
  for (auto& e : m->ev_pool)
    if (e) (void)hipEventDestroy(e);
  if (m->ev_loss) (void)hipEventDestroy(m->ev_loss);
  for (auto& e : m->ev_prof)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t s : {m->s_main, m->s_enc, m->s_img, m->s_tab, m->s_copy})
    if (s) (void)hipStreamDestroy(s);
  delete m;
}

int64_t vd_model_num_tensors(const vd_model* m) { return m ? (int64_t)m->spec.size() : 0; }
int64_t vd_model_flat_size(const vd_model* m) { return m ? m->numel : 0; }

int vd_model_tensor_info(const vd_model* m, int64_t i, char* name64, int64_t* offset, int64_t* rows, int64_t* cols) {
  VD_CHECK_ARG(m && i >= 0 && i < (int64_t)m->spec.size(), "vd_model_tensor_info: bad index");
  const Tensor& t = m->spec[i];
  if (name64) {
    strncpy(name64, t.name.c_str(), 63);
    name64[63] = 0;
  }
  if (offset) *offset = t.off;
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return VD_OK;
}

int vd_model_flat_pointers(vd_model* m, float** W, float** dW, float** adam_m, float** adam_v) {
  VD_CHECK_ARG(m, "vd_model_flat_pointers: null model");
  if (W) *W = m->W;
  if (dW) *dW = m->G;
  if (adam_m) *adam_m = m->M;
  if (adam_v) *adam_v = m->V;
  return VD_OK;
}

void* vd_model_stream(vd_model* m) { return m ? (void*)m->s_main : nullptr; }

// Library-default initialisation (weight-init.lua is a no-op in the reference): SeqLSTM weight ~
// N(0, 1/sqrt(D+H)), bias 0 with the forget gate at 1; Linear weight and bias ~ U(+-1/sqrt(in)); LookupTable ~ N(0,1)
// with the pad row zero.
int vd_model_init_params(vd_model* m, uint64_t seed) {
  VD_CHECK_ARG(m, "vd_model_init_params: null model");
  std::mt19937_64 rng(seed);
  std::vector<float> host((size_t)m->numel, 0.f);
  const long H = m->p.rnnHiddenSize;
  for (const Tensor& t : m->spec) {
    float* w = host.data() + t.off;
    const long n = t.numel();
    if (t.kind == 0) {
      std::normal_distribution<float> nd(0.f, 1.f);
      for (long i = t.cols; i < n; ++i) w[i] = nd(rng);
    } else if (t.kind == 1) {
      std::normal_distribution<float> nd(0.f, 1.f / sqrtf((float)t.rows));
      for (long i = 0; i < n; ++i) w[i] = nd(rng);
    } else if (t.kind == 2) {
      for (long i = H; i < 2 * H; ++i) w[i] = 1.f;
    } else {
      const long in = t.kind == 3 ? t.cols : m->spec[m->index.at(t.name.substr(0, t.name.size() - 2) + ".W")].cols;
      const float s = 1.f / sqrtf((float)in);
      std::uniform_real_distribution<float> ud(-s, s);
      for (long i = 0; i < n; ++i) w[i] = ud(rng);
    }
  }
  VD_HIP(hipMemcpy(m->W, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return VD_OK;
}

int vd_model_set_tensor(vd_model* m, const char* name, const float* host, int64_t n) {
  VD_CHECK_ARG(m && name && host && m->index.count(name), "vd_model_set_tensor: unknown tensor '%s'", name ? name : "(null)");
  const Tensor& t = m->spec[m->index.at(name)];
  VD_CHECK_ARG(n == t.numel(), "vd_model_set_tensor: '%s' holds %ld values, got %ld", name, t.numel(), (long)n);
  VD_HIP(hipStreamSynchronize(m->s_main));
  VD_HIP(hipMemcpy(m->W + t.off, host, n * sizeof(float), hipMemcpyHostToDevice));
  return VD_OK;
}

int vd_model_get_tensor(vd_model* m, const char* name, int which, float* host, int64_t n) {
  VD_CHECK_ARG(m && name && host && m->index.count(name), "vd_model_get_tensor: unknown tensor '%s'", name ? name : "(null)");
  const Tensor& t = m->spec[m->index.at(name)];
  VD_CHECK_ARG(n == t.numel() && which >= 0 && which <= 3, "vd_model_get_tensor: bad size / selector");
  VD_HIP(hipStreamSynchronize(m->s_main));
  float* base[4] = {m->W, m->G, m->M, m->V};
  VD_HIP(hipMemcpy(host, base[which] + t.off, n * sizeof(float), hipMemcpyDeviceToHost));
  return VD_OK;
}

int vd_model_set_training(vd_model* m, int on) {
  VD_CHECK_ARG(m, "vd_model_set_training: null model");
  m->training = on != 0;
  return VD_OK;
}

// pin the nn.Dropout noise of one call site (q_emb, h_emb, hatt, img_tr, iqc[, iqc2..], u); host == NULL clears all
int vd_model_set_dropout_mask(vd_model* m, const char* site, const uint8_t* host, int64_t n) {
  VD_CHECK_ARG(m, "vd_model_set_dropout_mask: null model");
  VD_HIP(hipStreamSynchronize(m->s_main));
  if (!host) {
    for (auto& kv : m->ext_masks)
      if (kv.second.p) VD_HIP(hipFree(kv.second.p));
    m->ext_masks.clear();
    return VD_OK;
  }
  VD_CHECK_ARG(site && n > 0, "vd_model_set_dropout_mask: bad arguments");
  void* p = nullptr;
  VD_TRY(dev_get(m->ext_masks, site, (size_t)n, &p));
  VD_HIP(hipMemcpy(p, host, (size_t)n, hipMemcpyHostToDevice));
  return VD_OK;
}

// Upload a batch in the dataloader's layout (dataloader.lua:324-339,378-475) into the free slot, asynchronously on the
// copy stream: it may be called while the previous step is still executing.  Host buffers are consumed before return
// (staged into pinned memory), the device copy completes in the background.
int vd_model_upload_batch(vd_model* m, const vd_batch* hb) {
  VD_CHECK_ARG(m && hb && hb->ques_fwd && hb->hist && hb->img_feat && hb->options && hb->B > 0 && hb->Tq > 0 && hb->Th > 0 && hb->To > 0,
               "vd_model_upload_batch: missing field");
  BatchSlot& sl = m->slot[m->cur < 0 ? 0 : (m->cur ^ 1)];   // the slot the running step does not read
  hipStream_t s = m->s_copy;
  // the step that last read this slot may still be executing (the host runs ahead of the device)
  if (sl.used) VD_HIP(hipStreamWaitEvent(s, sl.done, 0));
  sl.B = hb->B; sl.Tq = hb->Tq; sl.Th = hb->Th; sl.To = hb->To;
  const Dims d = dims(m, sl);
  VD_TRY(build_seqsort(m, sl, sl.q, "q", hb->ques_fwd, d.N, hb->Tq, s));
  VD_TRY(build_seqsort(m, sl, sl.h, "h", hb->hist, d.N, hb->Th, s));
  // image features [B x S2 x C]
  const size_t img_n = (size_t)d.B * d.S2 * d.C;
  float* ip = nullptr;
  VD_TRY(pin_get(sl.pinned, "img.stage", img_n * sizeof(float), (void**)&ip));
  memcpy(ip, hb->img_feat, img_n * sizeof(float));
  VD_TRY(dev_get(sl.bufs, "img", img_n * sizeof(float), (void**)&sl.img));
  VD_HIP(hipMemcpyAsync(sl.img, ip, img_n * sizeof(float), hipMemcpyHostToDevice, s));
  // options [N x O x To] -> time-major [To x N*O]
  const size_t on = (size_t)d.NO * hb->To;
  int32_t* op = nullptr;
  VD_TRY(pin_get(sl.pinned, "opt.stage", (on + d.N) * sizeof(int32_t), (void**)&op));
  for (long r = 0; r < d.NO; ++r)
    for (int t = 0; t < hb->To; ++t) op[(size_t)t * d.NO + r] = hb->options[(size_t)r * hb->To + t];
  sl.gt_host.assign(d.N, 0);
  for (int n = 0; n < d.N; ++n) {
    const int g = hb->answer_ind ? hb->answer_ind[n] - 1 : 0;   // 1-based on disk (prepro.py:169)
    VD_CHECK_ARG(!hb->answer_ind || (g >= 0 && g < d.O), "vd_model_upload_batch: answer_ind[%d] = %d out of 1..%d", n, g + 1, d.O);
    sl.gt_host[n] = g;
    op[on + n] = g;
  }
  VD_TRY(dev_get(sl.bufs, "opt", (on + d.N) * sizeof(int32_t), (void**)&sl.options));
  VD_HIP(hipMemcpyAsync(sl.options, op, (on + d.N) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  sl.gt = sl.options + on;
  VD_HIP(hipEventRecord(sl.ready, s));
  m->uploaded = (int)(&sl - m->slot);
  return VD_OK;
}

// Model:forwardBackward on the uploaded batch (model.lua:249-342, disc branch :326-338).  Enqueues only.
int vd_model_forward_backward(vd_model* m, int only_forward) {
  VD_CHECK_ARG(m && m->uploaded >= 0, "vd_model_forward_backward: no batch uploaded");
  m->cur = m->uploaded;
  BatchSlot& b = m->slot[m->cur];
  const Dims d = dims(m, b);
  const int N = d.N, NO = d.NO, H = d.H, E = d.E, V = d.V, To = b.To, O = d.O;
  m->N = N;
  m->NO = NO;
  hipStream_t s = m->s_main;
  hipStream_t se = m->streams ? m->s_enc : s;
  hipStream_t st = m->streams ? m->s_tab : s;
  VD_HIP(hipStreamWaitEvent(s, b.ready, 0));
  // causal mask of the memory attention (model.lua:280-294), cached per batch size
  if (m->causal_B != d.B) {
    std::vector<uint8_t> host((size_t)N * d.R);
    for (int n = 0; n < N; ++n)
      for (int j = 0; j < d.R; ++j) host[(size_t)n * d.R + j] = j > (n % d.R) ? 1 : 0;
    if (m->causal) VD_HIP(hipFree(m->causal));
    VD_HIP(hipMalloc((void**)&m->causal, host.size()));
    VD_HIP(hipMemcpy(m->causal, host.data(), host.size(), hipMemcpyHostToDevice));
    m->causal_B = d.B;
  }
  if (!only_forward) VD_TRY(vd_memset(m->G, 0, m->numel * 4, s));              // wrapper:zeroGradParameters (model.lua:68)
  VD_TRY(vd_memset(Wp(m, "embed"), 0, (long)E * 4, s));                        // LookupTableMaskZero zeroes the pad row
  // ---- decoder forward first on the main stream (a handful of big launches), encoder on the side stream
  float *table, *gates, *h, *c, *scores, *loss_rows;
  VD_TRY(ws_get(m, "opt.table", (size_t)(V + 1) * 4 * H, &table));
  VD_TRY(ws_get(m, "opt.gates", (size_t)To * NO * 4 * H, &gates));
  VD_TRY(ws_get(m, "opt.h", (size_t)To * NO * H, &h));
  VD_TRY(ws_get(m, "opt.c", (size_t)To * NO * H, &c));
  VD_TRY(ws_get(m, "opt.scores", (size_t)N * O, &scores));
  VD_TRY(ws_get(m, "crit.loss_rows", (size_t)N, &loss_rows));
  float* Wopt = Wp(m, "opt.W");
  const int flags = m->p.lstmBf16 ? VD_FLAG_BF16 : 0;
  VD_TRY(fork_stream(m, s, se));
  VD_TRY(vd_gemm_nn(Wp(m, "embed"), E, Wopt, 4L * H, Wp(m, "opt.b"), table, 4L * H, V + 1, 4 * H, E, 0, s));   // table = Emb*Wx + b
  VD_HIP(hipEventRecord(m->ev_prof[0], s));
  VD_TRY(vd_lstm_forward(table, 0, 4L * H, b.options, nullptr, Wopt + (long)E * 4 * H, nullptr, nullptr, gates, h, c, To, NO, H, flags, s));
  VD_HIP(hipEventRecord(m->ev_prof[1], s));
  float* enc_out = nullptr;
  VD_TRY(encoder_forward(m, b, m->enc, &enc_out, se));
  VD_TRY(join_stream(m, se, s));
  // ---- criterion (+ nn.MM backward) in one kernel (model.lua:330-335)
  const float* optH = h + (long)(To - 1) * NO * H;
  float *d_optH = nullptr, *d_enc = nullptr;
  if (!only_forward) {
    VD_TRY(ws_get(m, "crit.d_optH", (size_t)NO * H, &d_optH));
    VD_TRY(ws_get(m, "crit.d_enc", (size_t)N * H, &d_enc));
  }
  VD_TRY(vd_score_ce(optH, enc_out, b.gt, scores, loss_rows, d_optH, d_enc, N, O, H, 1.0f / N, s));
  if (m->loss_n < N) {
    if (m->loss_host) VD_HIP(hipHostFree(m->loss_host));
    VD_HIP(hipHostMalloc((void**)&m->loss_host, (size_t)N * sizeof(float), hipHostMallocDefault));
    m->loss_n = N;
  }
  VD_HIP(hipMemcpyAsync(m->loss_host, loss_rows, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, s));
  VD_HIP(hipEventRecord(m->ev_loss, s));
  b.used = true;
  if (only_forward) {
    VD_HIP(hipEventRecord(b.done, s));
    return VD_OK;
  }
  // ---- decoder backward on the main stream, encoder backward on the side stream (model.lua:335-337)
  VD_TRY(fork_stream(m, s, se));
  float *dc, *dtab;
  int32_t *offset, *work, *perm;
  VD_TRY(ws_get(m, "opt.dc", (size_t)NO * H, &dc));
  VD_TRY(ws_get(m, "opt.dtable", (size_t)(V + 1) * 4 * H, &dtab));
  VD_TRY(ws_get(m, "opt.sort_off", (size_t)V + 2, &offset));
  VD_TRY(ws_get(m, "opt.sort_work", (size_t)2 * (V + 1), &work));
  VD_TRY(ws_get(m, "opt.sort_perm", (size_t)To * NO, &perm));
  VD_TRY(fork_stream(m, s, st));
  VD_TRY(vd_token_sort(b.options, (long)To * NO, V + 1, offset, work, perm, st));
  VD_TRY(vd_memset(dtab, 0, (long)(V + 1) * 4 * H * 4, st));
  VD_HIP(hipEventRecord(m->ev_prof[2], s));
  VD_TRY(vd_lstm_backward(Wopt + (long)E * 4 * H, gates, c, nullptr, nullptr, d_optH, nullptr, dc, nullptr, nullptr, nullptr, To, NO, H, flags, s));
  VD_HIP(hipEventRecord(m->ev_prof[3], s));
  VD_TRY(encoder_backward(m, b, m->enc, d_enc, se));
  // table gradient + its consumers beside the dWh contraction
  VD_TRY(fork_stream(m, s, st));
  VD_TRY(vd_segment_rowsum_acc(gates, 4L * H, b.options, perm, (long)To * NO, 4 * H, dtab, 4L * H, st));
  VD_TRY(vd_colsum_acc(dtab, 4L * H, V + 1, 4 * H, Gp(m, "opt.b"), st));
  VD_TRY(vd_gemm_tn_acc(Wp(m, "embed"), E, dtab, 4L * H, Gp(m, "opt.W"), 4L * H, E, 4 * H, V + 1, 0, st));
  VD_HIP(hipEventRecord(m->ev_prof[4], s));
  if (To > 1)
    VD_TRY(vd_gemm_tn_acc(h, H, gates + (long)NO * 4 * H, 4L * H, Gp(m, "opt.W") + (long)E * 4 * H, 4L * H, H, 4 * H, (To - 1) * NO, flags, s));
  VD_HIP(hipEventRecord(m->ev_prof[5], s));
  VD_TRY(join_stream(m, se, s));
  VD_TRY(join_stream(m, st, s));
  // dEmb += dTable * Wx^T: non-atomic read-modify-write of the SHARED embedding gradient, after every other writer
  VD_TRY(vd_gemm_nt(dtab, 4L * H, Wopt, 4L * H, nullptr, Gp(m, "embed"), E, V + 1, E, 4 * H, VD_ACT_NONE, 1, s));
  VD_HIP(hipEventRecord(b.done, s));
  return VD_OK;
}

// waits for the loss of the last vd_model_forward_backward (mean cross-entropy over the rounds)
int vd_model_loss(vd_model* m, float* loss) {
  VD_CHECK_ARG(m && loss && m->cur >= 0, "vd_model_loss: nothing to read");
  VD_HIP(hipEventSynchronize(m->ev_loss));
  double s = 0;
  for (int i = 0; i < m->N; ++i) s += m->loss_host[i];
  *loss = (float)(s / m->N);
  return VD_OK;
}

// [gradient already reduced by the caller] -> clamp(-5, 5) -> adam -> lr decay (model.lua:96-105; optim_updates.lua:62-91)
int vd_model_update(vd_model* m, float gscale) {
  VD_CHECK_ARG(m, "vd_model_update: null model");
  m->adam_t += 1;
  const double t = m->adam_t;
  const float step = (float)(m->lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
  VD_TRY(vd_clamp_adam(m->W, m->G, m->M, m->V, m->numel, gscale, 5.0f, 0.9f, 0.999f, 1e-8f, step, m->s_main));
  if (m->lr > m->p.minLRate) m->lr *= m->p.lrDecayRate;
  m->step += 1;
  return VD_OK;
}

int vd_model_learning_rate(vd_model* m, double* lr, int set) {
  VD_CHECK_ARG(m && lr, "vd_model_learning_rate: null");
  if (set) m->lr = *lr;
  else *lr = m->lr;
  return VD_OK;
}

// scores [N x O] of the last forward (host buffer)
int vd_model_scores(vd_model* m, float* host, int64_t n) {
  VD_CHECK_ARG(m && host && m->cur >= 0 && n == (int64_t)m->N * m->p.numOptions, "vd_model_scores: bad size");
  float* scores;
  VD_TRY(ws_get(m, "opt.scores", (size_t)n, &scores));
  VD_HIP(hipStreamSynchronize(m->s_main));
  VD_HIP(hipMemcpy(host, scores, n * sizeof(float), hipMemcpyDeviceToHost));
  return VD_OK;
}

// utils.computeRanks on the scores of the last forward: use_gt -> ranks_out[N] (rank of the ground-truth option),
// else ranks_out[N x O] (rank of every option), 1-based (utils.lua:106-128)
int vd_model_ranks(vd_model* m, int use_gt, int32_t* ranks_out) {
  VD_CHECK_ARG(m && ranks_out && m->cur >= 0, "vd_model_ranks: nothing to rank");
  const int N = m->N, O = m->p.numOptions;
  float* scores;
  int32_t* ranks;
  VD_TRY(ws_get(m, "opt.scores", (size_t)N * O, &scores));
  VD_TRY(ws_get(m, "opt.ranks", (size_t)N * O, &ranks));
  VD_TRY(vd_ranks(scores, ranks, N, O, m->s_main));
  std::vector<int32_t> host((size_t)N * O);
  VD_HIP(hipMemcpyAsync(host.data(), ranks, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m->s_main));
  VD_HIP(hipStreamSynchronize(m->s_main));
  if (use_gt) {
    const BatchSlot& b = m->slot[m->cur];
    for (int n = 0; n < N; ++n) ranks_out[n] = host[(size_t)n * O + b.gt_host[n]];
  } else {
    memcpy(ranks_out, host.data(), host.size() * sizeof(int32_t));
  }
  return VD_OK;
}

// device time of the three option-LSTM kernel families inside the last forward_backward, ms: [fwd, bwd, dWh]
int vd_model_family_ms(vd_model* m, float* ms3) {
  VD_CHECK_ARG(m && ms3, "vd_model_family_ms: null");
  VD_HIP(hipStreamSynchronize(m->s_main));
  VD_HIP(hipEventElapsedTime(&ms3[0], m->ev_prof[0], m->ev_prof[1]));
  VD_HIP(hipEventElapsedTime(&ms3[1], m->ev_prof[2], m->ev_prof[3]));
  VD_HIP(hipEventElapsedTime(&ms3[2], m->ev_prof[4], m->ev_prof[5]));
  return VD_OK;
}

int vd_model_synchronize(vd_model* m) {
  VD_CHECK_ARG(m, "vd_model_synchronize: null model");
  for (hipStream_t s : {m->s_copy, m->s_enc, m->s_img, m->s_tab, m->s_main}) VD_HIP(hipStreamSynchronize(s));
  return VD_OK;
}

}  // extern "C"
