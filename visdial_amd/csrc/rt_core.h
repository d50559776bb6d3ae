// Native step runtime, part 1: the model object, memory / stream helpers and the host-side building blocks
// (nn.Linear, CatLinear, nn.SeqLSTM with its hand-off fields, LSTM stacks).  Included by runtime.hip only.
//
// The building blocks mirror the Python host one to one (visdial_amd/nn.py, encoders/_blocks.py): they own no
// arithmetic -- every forward / backward is an ordered list of operator-level launches of this same library -- only
// buffers, shapes and launch order.  The parity tests pin both hosts against the same oracle.
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/visdial_hip.h"
#include "common.h"

// lstm.hip: the two-layer wavefront with the pass's arithmetic (flags & VD_FLAG_BF16: bf16 operands on the ticks' MFMAs in a bf16 pass;
// the C-ABI entry points vd_lstm2_forward / vd_lstm2_backward are these with flags = 0)
int vd_lstm2_forward_p(const vd_lstm2_fwd_t* st, int nstacks, int H, int flags, hipStream_t stream);
int vd_lstm2_backward_p(const vd_lstm2_bwd_t* st, int nstacks, int H, int flags, hipStream_t stream);

// attention.hip: the image attention's dense products with a pass's arithmetic (flags & VD_FLAG_SPLIT9: on the exact split, from the
// materialised dropped image tensor `xdrop`; else the C-ABI entry points)
int vd_img_drop_gather(const float* pre, const uint8_t* mask1, float* xdrop, int N, int R, int S2, int H, float scale, hipStream_t stream);
int vd_img_common_forward_p(const float* pre, const uint8_t* mask1, const float* xdrop, const float* Wc, const float* bc, const float* qc,
                            const uint8_t* mask2, float* iqc, int N, int R, int S2, int H, int Kc, float scale, int flags, hipStream_t stream);
int vd_img_tr_backward_p(const float* dz, const float* Wc, const float* p, const float* datt, const uint8_t* mask1, float* dpre, int N, int R,
                         int S2, int H, int Kc, float scale, int flags, hipStream_t stream);
int vd_img_common_wgrad_p(const float* dz, const float* pre, const uint8_t* mask1, const float* xdrop, float* dWc, int N, int R, int S2, int H,
                          int Kc, float scale, int flags, hipStream_t stream);

#define VD_TRY(expr)                  \
  do {                                \
    const int rc__ = (expr);          \
    if (rc__ != VD_OK) return rc__;   \
  } while (0)

namespace vdrt {

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

// time-major [T x N] token matrix on the device (+ length-sort metadata for the nngraph encoders' text branches)
struct SeqTok {
  int T = 0, N = 0;
  bool present = false, sorted = false;
  int32_t* tok = nullptr;
  std::vector<int32_t> nact;  // host [T]
  int32_t *tok_sorted = nullptr, *fwd_idx = nullptr, *inv_idx = nullptr, *perm = nullptr, *inv = nullptr, *nact_dev = nullptr;
  // the non-pad (t, row) pairs as linear row indices of the sorted [T*N x .] tensors, in step order; act1 / prev1 =
  // the pairs with t >= 1 and their previous-step rows (weight gradients contract these only)
  int32_t *act = nullptr, *act1 = nullptr, *prev1 = nullptr;
  int n_act = 0, n_act1 = 0;
};

struct BatchSlot {
  int B = 0;
  SeqTok q, h, opt, ain, aout, oin, oout;
  float* img = nullptr;
  int32_t* gt = nullptr;  // [N] 0-based
  std::vector<int32_t> gt_host;
  bool has_gt = false;
  // option de-duplication (decoder disc): the option LSTM encodes every DISTINCT candidate once (decoders/disc.lua:4-15 --
  // the encoding depends on the tokens only); opt_uid[n * O + o] = its row among the opt.N unique rows, or null
  int32_t* opt_uid = nullptr;
  int opt_total = 0;           // N * O
  // counting sort of the option tokens (the table gradient's row order), done on the copy stream at upload time: it depends on the
  // batch alone, and inside the step it stood in front of the encoder backward in the side streams' in-order hardware queue
  int32_t *opt_sort_off = nullptr, *opt_sort_perm = nullptr;
  hipEvent_t ready = nullptr;  // recorded on the copy stream when the upload has landed
  hipEvent_t done = nullptr;   // recorded on the main stream behind the last reader of this slot
  bool used = false;      // a step has read this slot (done is recorded)
  bool uploaded = false;  // ready is recorded (an upload has been issued into this slot)
  std::map<std::string, DevBuf> bufs, pinned;
};

struct Encoder;
struct Decoder;

}  // namespace vdrt

struct vd_model {
  vd_model_params p;
  std::string enc_name, dec_name;
  std::vector<vdrt::Tensor> spec;
  std::map<std::string, int> index;
  long numel = 0;
  float *W = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  std::map<std::string, vdrt::DevBuf> ws;
  std::map<std::string, vdrt::DevBuf> ext_masks;
  hipStream_t s_main = nullptr, s_enc = nullptr, s_img = nullptr, s_hist = nullptr, s_tab = nullptr, s_copy = nullptr;
  // parameter-gradient work of the nngraph encoders that nothing downstream in the backward pass waits for (image-attention
  // weight gradients, the recurrences' weight / input gradients): middle priority = its own hardware queue beside the in-order
  // queue of the other side streams (runtime.hip).  wg_active: the current backward uses it; wg_used: it holds un-joined work
  hipStream_t s_wg = nullptr;
  bool wg_active = false, wg_used = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  hipEvent_t ev_enc_grads = nullptr;  // recorded behind the encoder backward: its gradient tensors are final
  hipEvent_t ev_updated = nullptr;    // recorded behind the optimiser launch: the prefetch upload's kernels (the option tokens'
                                      // counting sort) wait for it instead of sharing HBM with clamp_adam (65 -> 104 us beside them)
  bool updated_recorded = false;
  bool enc_grads_recorded = false;
  hipEvent_t ev_loss = nullptr, ev_prof[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool training = true, streams = true, prof_valid = false;
  long step = 0;
  int adam_t = 0;
  double lr = 1e-3;
  vdrt::BatchSlot slot[2];
  int cur = -1, uploaded = -1;
  float* loss_host = nullptr;  // pinned
  long loss_cap = 0, loss_n = 0;
  bool loss_is_sum = false;
  std::unique_ptr<vdrt::Encoder> enc;
  std::unique_ptr<vdrt::Decoder> dec;
  float* scores = nullptr;  // [N x O] of the last forward / retrieval
  float* gen_enc_out = nullptr;  // encoder output of the last vd_model_encode (generation)
  int gen_seq_len = 0;
  int N = 0, O = 0;
  // capability flags from the encoder NAME (opts.lua:54-67)
  bool use_im = false, use_hist = false, is_att = false, is_graph = false;
  bool prof_hist = false;   // ev_prof[0..3] bracket the history branch of a Sequential encoder (gen pairs: vd_model_family_ms)
  ~vd_model();
};

namespace vdrt {

// ------------------------------------------------------------------------------------------------------------
// memory / stream helpers
// ------------------------------------------------------------------------------------------------------------
inline int dev_get(std::map<std::string, DevBuf>& m, const std::string& key, size_t bytes, void** out) {
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
inline int pin_get(std::map<std::string, DevBuf>& m, const std::string& key, size_t bytes, void** out) {
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
int ws_get(vd_model* m, const std::string& key, size_t count, T** out) {
  void* p = nullptr;
  VD_TRY(dev_get(m->ws, key, count * sizeof(T), &p));
  *out = static_cast<T*>(p);
  return VD_OK;
}

inline void add_lstm(vd_model* m, const std::string& name, long D, long H) {
  m->spec.push_back(Tensor{name + ".W", 0, D + H, 4 * H, 1});
  m->spec.push_back(Tensor{name + ".b", 0, 1, 4 * H, 2});
}
inline void add_linear(vd_model* m, const std::string& name, long in, long out) {
  m->spec.push_back(Tensor{name + ".W", 0, out, in, 3});
  m->spec.push_back(Tensor{name + ".b", 0, 1, out, 4});
}
inline std::string hop_sfx(int i) { return i == 0 ? std::string() : std::to_string(i + 1); }

inline float* Wp(vd_model* m, const std::string& n) { return m->W + m->spec[m->index.at(n)].off; }
inline float* Gp(vd_model* m, const std::string& n) { return m->G + m->spec[m->index.at(n)].off; }

// fork: `side` waits for everything enqueued on `from` so far; join: `to` waits for `side`
inline int fork_stream(vd_model* m, hipStream_t from, hipStream_t side) {
  if (side == from) return VD_OK;
  hipEvent_t e = m->ev_pool[m->ev_next++ % m->ev_pool.size()];
  VD_HIP(hipEventRecord(e, from));
  VD_HIP(hipStreamWaitEvent(side, e, 0));
  return VD_OK;
}
inline int join_stream(vd_model* m, hipStream_t side, hipStream_t to) { return fork_stream(m, side, to); }
inline hipStream_t side_stream(vd_model* m, hipStream_t wanted, hipStream_t cur) { return m->streams ? wanted : cur; }
// the stream for off-chain parameter-gradient work enqueued after everything on `cur` so far (cur itself when the feature is off)
inline int wg_fork(vd_model* m, hipStream_t cur, hipStream_t* out) {
  *out = cur;
  if (!m->wg_active || !m->s_wg) return VD_OK;
  VD_TRY(fork_stream(m, cur, m->s_wg));
  m->wg_used = true;
  *out = m->s_wg;
  return VD_OK;
}

// nn.Dropout keep-mask for a call site (null in evaluate mode or p == 0); external masks pin the noise for parity runs
inline int drop_mask(vd_model* m, const std::string& site, size_t numel, float p, hipStream_t s, uint8_t** out) {
  *out = nullptr;
  if (!m->training || p <= 0.f) return VD_OK;
  auto it = m->ext_masks.find(site);
  if (it != m->ext_masks.end()) {
    VD_CHECK_ARG(it->second.bytes >= numel, "dropout mask '%s' holds %zu bytes, the batch needs %zu", site.c_str(),
                 it->second.bytes, numel);
    *out = static_cast<uint8_t*>(it->second.p);
    return VD_OK;
  }
  if (!m->ext_masks.empty()) {
    vd_set_error("external dropout masks are set but none for site '%s'", site.c_str());
    return VD_ERR_STATE;
  }
  uint8_t* buf = nullptr;
  VD_TRY(ws_get(m, "dropmask." + site, numel + 4, &buf));
  uint32_t h = 2166136261u;  // FNV-1a of the site name: one independent stream per (seed, step, site)
  for (const char c : site) h = (h ^ (uint8_t)c) * 16777619u;
  const uint64_t seed = ((uint64_t)((m->p.seed * 1000003ull + (uint64_t)m->step) & 0xffffffffull) << 32) | h;
  VD_TRY(vd_dropout_mask(buf, (int64_t)numel, seed, p, s));
  *out = buf;
  return VD_OK;
}
// y = dropout(x) (x itself when mask is null)
inline int dropout_fwd(vd_model* m, const std::string& key, const float* x, const uint8_t* mask, float scale, long n, hipStream_t s,
                       const float** y) {
  *y = x;
  if (!mask) return VD_OK;
  float* buf;
  VD_TRY(ws_get(m, key, (size_t)n, &buf));
  VD_TRY(vd_dropout_apply(x, mask, buf, n, scale, s));
  *y = buf;
  return VD_OK;
}

// small int32 index arrays cached on the device (row permutations / per-round image index)
inline int index_array(vd_model* m, const std::string& key, const std::vector<int32_t>& host, int32_t** out) {
  DevBuf& b = m->ws[key];
  if (b.bytes != host.size() * sizeof(int32_t) || !b.p) {
    if (b.p) VD_HIP(hipFree(b.p));
    b.p = nullptr;
    VD_HIP(hipMalloc(&b.p, host.size() * sizeof(int32_t) + 16));
    b.bytes = host.size() * sizeof(int32_t);
    VD_HIP(hipMemcpy(b.p, host.data(), b.bytes, hipMemcpyHostToDevice));
  }
  *out = static_cast<int32_t*>(b.p);
  return VD_OK;
}
inline int round_index(vd_model* m, int N, int R, int32_t** out) {  // row n of a per-round tensor -> its image row n / R
  std::vector<int32_t> h(N);
  for (int n = 0; n < N; ++n) h[n] = n / R;
  return index_array(m, "idx.rep." + std::to_string(N) + "." + std::to_string(R), h, out);
}

// ------------------------------------------------------------------------------------------------------------
// nn.Linear (+ optional fused nn.Tanh)                                                  visdial_amd/nn.py:Linear
// ------------------------------------------------------------------------------------------------------------
struct Linear {
  std::string name;
  long in = 0, out = 0;
  const float* x = nullptr;
  float* y = nullptr;
  long M = 0;
  bool tanh_ = false;
  void init(const std::string& n, long i, long o) { name = n; in = i; out = o; }
  int forward(vd_model* m, hipStream_t s, const float* x_, long M_, bool tanh, float** y_out) {
    x = x_; M = M_; tanh_ = tanh;
    VD_TRY(ws_get(m, name + ".y", (size_t)M * out, &y));
    VD_TRY(vd_gemm_nt(x, in, Wp(m, name + ".W"), in, Wp(m, name + ".b"), y, out, (int)M, (int)out, (int)in,
                      tanh ? VD_ACT_TANH : VD_ACT_NONE, 0, s));
    if (y_out) *y_out = y;
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, const float* dy, bool need_dx, float** dx_out) {
    const float* d = dy;
    if (tanh_) {
      float* dpre;
      VD_TRY(ws_get(m, name + ".dpre", (size_t)M * out, &dpre));
      VD_TRY(vd_tanh_backward(dy, y, dpre, M * out, s));
      d = dpre;
    }
    VD_TRY(vd_gemm_tn_acc(d, out, x, in, Gp(m, name + ".W"), in, (int)out, (int)in, (int)M, 0, s));
    VD_TRY(vd_colsum_acc(d, out, (int)M, (int)out, Gp(m, name + ".b"), s));
    if (dx_out) *dx_out = nullptr;
    if (need_dx) {
      float* dx;
      VD_TRY(ws_get(m, name + ".dx", (size_t)M * in, &dx));
      VD_TRY(vd_gemm_nn(d, out, Wp(m, name + ".W"), in, nullptr, dx, in, (int)M, (int)in, (int)out, 0, s));
      if (dx_out) *dx_out = dx;
    }
    return VD_OK;
  }
};

// Tanh(Linear(JoinTable(parts))) with optional Dropout on the joined vector            _blocks.py:CatLinear
struct CatLinear {
  std::string name;
  std::vector<long> dims;
  long D = 0, H = 0, N = 0;
  Linear lin;
  const uint8_t* mask = nullptr;
  float scale = 1.f;
  void init(const std::string& n, const std::vector<long>& d, long H_) {
    name = n; dims = d; H = H_; D = 0;
    for (long v : d) D += v;
    lin.init(n, D, H_);
  }
  int forward(vd_model* m, hipStream_t s, const std::vector<const float*>& parts, long N_, const uint8_t* mask_, float scale_,
              float** y) {
    N = N_; mask = mask_; scale = scale_;
    float* cat;
    VD_TRY(ws_get(m, name + ".cat", (size_t)N * D, &cat));
    long off = 0;
    for (size_t i = 0; i < dims.size(); ++i) {
      VD_TRY(vd_copy_2d(cat + off, D, parts[i], dims[i], N, dims[i], s));
      off += dims[i];
    }
    const float* catd;
    VD_TRY(dropout_fwd(m, name + ".in", cat, mask, scale, N * D, s, &catd));
    return lin.forward(m, s, catd, N, true, y);
  }
  // per-part gradients ([N x d_i] contiguous; null where not needed)
  int backward(vd_model* m, hipStream_t s, const float* dy, const std::vector<bool>& need, std::vector<float*>* out) {
    float* dcatd;
    VD_TRY(lin.backward(m, s, dy, true, &dcatd));
    const float* dcat;
    VD_TRY(dropout_fwd(m, name + ".din", dcatd, mask, scale, N * D, s, &dcat));
    out->assign(dims.size(), nullptr);
    long off = 0;
    for (size_t i = 0; i < dims.size(); ++i) {
      if (need.empty() || need[i]) {
        float* g;
        VD_TRY(ws_get(m, name + ".dpart" + std::to_string(i), (size_t)N * dims[i], &g));
        VD_TRY(vd_copy_2d(g, dims[i], dcat + off, D, N, dims[i], s));
        (*out)[i] = g;
      }
      off += dims[i];
    }
    return VD_OK;
  }
};

// ------------------------------------------------------------------------------------------------------------
// nn.SeqLSTM(D, H)[:maskZero()] over a whole time-major sequence                        visdial_amd/nn.py:SeqLSTM
// The input may arrive as several column blocks (`parts`): x*Wx is then the sum of the per-block products against
// the matching row blocks of Wx (nn.JoinTable(-1) in front of an LSTM folded away).  State hand-off fields carry the
// reference's names (rnn SeqLSTM; used by decoders/gen.lua:30-60).
// ------------------------------------------------------------------------------------------------------------
struct SeqLSTM {
  std::string name;
  long D = 0, H = 0;
  std::vector<long> parts;
  const float *userPrevOutput = nullptr, *userPrevCell = nullptr, *userNextGradCell = nullptr, *gradPrevOutput = nullptr;
  float *userGradPrevOutput = nullptr, *userGradPrevCell = nullptr;
  int T = 0, N = 0;
  std::vector<const float*> xs;
  const int32_t* tok_mask = nullptr;
  const float *h0 = nullptr, *c0 = nullptr;
  float *gates = nullptr, *h = nullptr, *c = nullptr;
  const SeqTok* rows = nullptr;   // set for length-sorted stacks: weight gradients skip the pad (t, row) pairs

  void init(const std::string& n, long D_, long H_, const std::vector<long>& p = {}) {
    name = n; D = D_; H = H_;
    parts = p.empty() ? std::vector<long>{D_} : p;
  }
  float* Wx(vd_model* m) const { return Wp(m, name + ".W"); }
  float* Wh(vd_model* m) const { return Wp(m, name + ".W") + D * 4 * H; }
  float* out_at(int t) const { return h + (long)t * N * H; }    // .output[t]
  float* cell_at(int t) const { return c + (long)t * N * H; }   // .cell[t]

  int alloc(vd_model* m, int T_, int N_) {
    T = T_; N = N_;
    VD_TRY(ws_get(m, name + ".gates", (size_t)T * N * 4 * H, &gates));
    VD_TRY(ws_get(m, name + ".h", (size_t)T * N * H, &h));
    VD_TRY(ws_get(m, name + ".c", (size_t)T * N * H, &c));
    return VD_OK;
  }
  int forward(vd_model* m, hipStream_t s, const std::vector<const float*>& x, int T_, int N_, const int32_t* tok, float** h_out) {
    h0 = userPrevOutput; c0 = userPrevCell;
    userPrevOutput = userPrevCell = nullptr;                       // consumed once (rnn semantics)
    if (h0 && !c0) {
      float* z;
      VD_TRY(ws_get(m, name + ".c0zero", (size_t)N_ * H, &z));
      VD_TRY(vd_memset(z, 0, (long)N_ * H * 4, s));
      c0 = z;
    }
    if (c0 && !h0) {
      float* z;
      VD_TRY(ws_get(m, name + ".h0zero", (size_t)N_ * H, &z));
      VD_TRY(vd_memset(z, 0, (long)N_ * H * 4, s));
      h0 = z;
    }
    VD_TRY(alloc(m, T_, N_));
    xs = x; tok_mask = tok; rows = nullptr;
    // hoisted input projection, written straight into the gates buffer (the recurrence runs in place)
    long roff = 0;
    for (size_t i = 0; i < parts.size(); ++i) {
      VD_TRY(vd_gemm_nn(xs[i], parts[i], Wx(m) + roff * 4 * H, 4 * H, i == 0 ? Wp(m, name + ".b") : nullptr, gates, 4 * H,
                        T * N, (int)(4 * H), (int)parts[i], i > 0 ? 1 : 0, s));
      roff += parts[i];
    }
    VD_TRY(vd_lstm_forward(gates, (int64_t)N * 4 * H, 4 * H, nullptr, tok, Wh(m), h0, c0, gates, h, c, T, N, (int)H, 0, s));
    if (h_out) *h_out = h;
    return VD_OK;
  }
  // weight / bias / input gradients from da (held in the gates buffer after BPTT)
  int param_grads(vd_model* m, hipStream_t s, const std::vector<bool>& need_dx, std::vector<float*>* dxs) {
    const long TN = (long)T * N;
    float* dW = Gp(m, name + ".W");
    float* dWh = dW + D * 4 * H;
    const bool by_rows = rows && rows->act;
    // Where the shape fits the k-major LDS-DMA contraction (M, N multiples of 128: the recurrent weights and the
    // layer-2 input weights at H = 512) the DENSE product over all T*N rows on that kernel beats the index-list
    // contraction of the non-pad pairs: the pad pairs hold da = 0 (rt_encoders.h zero-fills them), the kernel streams at
    // the option dWh kernel's rate, and its workgroups live ~0.1 ms instead of ~0.8 ms -- the index-list kernel is a chain
    // of dependent (row index -> row) loads per K tile whose long-lived workgroups take the third slot of a third of the
    // CUs away from the option-LSTM backward kernels for most of their run.
    auto dense_fits = [&](long Mrows, long K) { return Mrows % 128 == 0 && (4 * H) % 128 == 0 && K >= 1024; };
    // bf16 pass (lstmPrecision = 'bf16', configs[4]): the dense contractions whose shape the bf16 kernel takes (256-row tiles) round both
    // fp32 operands to bf16 in registers and multiply on the bf16 MFMA (gemm_ops.hip: 77 vs 167 us at K = 8 000)
    auto wg_flags = [&](long Mrows) { return m->p.lstmBf16 == 1 && Mrows % 256 == 0 ? VD_FLAG_BF16 : 0; };
    if (by_rows && !(T > 1 && dense_fits(H, (long)(T - 1) * N))) {
      if (rows->n_act1 > 0)
        VD_TRY(vd_gemm_tn_rows_acc(h, H, rows->prev1, gates, 4 * H, rows->act1, dWh, 4 * H, (int)H, (int)(4 * H), rows->n_act1, s));
    } else if (T > 1) {
      VD_TRY(vd_gemm_tn_acc(h, H, gates + (long)N * 4 * H, 4 * H, dWh, 4 * H, (int)H, (int)(4 * H), (T - 1) * N, wg_flags(H), s));
    }
    if (h0) VD_TRY(vd_gemm_tn_acc(h0, H, gates, 4 * H, dWh, 4 * H, (int)H, (int)(4 * H), N, 0, s));
    VD_TRY(vd_colsum_acc(gates, 4 * H, (int)TN, (int)(4 * H), Gp(m, name + ".b"), s));
    if (dxs) dxs->assign(parts.size(), nullptr);
    long roff = 0;
    for (size_t i = 0; i < parts.size(); ++i) {
      if (by_rows && !dense_fits(parts[i], TN)) {
        if (rows->n_act > 0)
          VD_TRY(vd_gemm_tn_rows_acc(xs[i], parts[i], rows->act, gates, 4 * H, rows->act, dW + roff * 4 * H, 4 * H, (int)parts[i],
                                     (int)(4 * H), rows->n_act, s));
      } else {
        VD_TRY(vd_gemm_tn_acc(xs[i], parts[i], gates, 4 * H, dW + roff * 4 * H, 4 * H, (int)parts[i], (int)(4 * H), (int)TN, wg_flags(parts[i]), s));
      }
      if (need_dx.empty() || need_dx[i]) {
        float* dx;
        VD_TRY(ws_get(m, name + ".dx" + std::to_string(i), (size_t)TN * parts[i], &dx));
        VD_TRY(vd_gemm_nt(gates, 4 * H, Wx(m) + roff * 4 * H, 4 * H, nullptr, dx, parts[i], (int)TN, (int)parts[i], (int)(4 * H),
                          VD_ACT_NONE, 0, s));
        if (dxs) (*dxs)[i] = dx;
      }
      roff += parts[i];
    }
    return VD_OK;
  }
  int backward(vd_model* m, hipStream_t s, const float* dh_seq, const float* dh_last, const std::vector<bool>& need_dx,
               std::vector<float*>* dxs) {
    if (gradPrevOutput) {
      if (!dh_last) dh_last = gradPrevOutput;
      else {
        float* t;
        VD_TRY(ws_get(m, name + ".dhl", (size_t)N * H, &t));
        VD_TRY(vd_axpby(dh_last, gradPrevOutput, t, (long)N * H, 1.f, 1.f, s));
        dh_last = t;
      }
    }
    const float* dc_last = userNextGradCell;
    gradPrevOutput = userNextGradCell = nullptr;
    float *dc, *dh0 = nullptr;
    VD_TRY(ws_get(m, name + ".dc", (size_t)N * H, &dc));
    if (h0) VD_TRY(ws_get(m, name + ".dh0", (size_t)N * H, &dh0));
    VD_TRY(vd_lstm_backward(Wh(m), gates, c, c0, dh_seq, dh_last, dc_last, dc, dh0, nullptr, nullptr, T, N, (int)H, 0, s));
    userGradPrevOutput = dh0;
    userGradPrevCell = h0 ? dc : nullptr;
    return param_grads(m, s, need_dx, dxs);
  }
};

// numLayers x SeqLSTM(maskZero) (encoders/lf-ques.lua:18-24)
inline int lstm_stack_forward(vd_model* m, hipStream_t s, std::vector<SeqLSTM>& layers, const std::vector<const float*>& x, int T,
                              int N, const int32_t* tok, float** h_top) {
  float* h = nullptr;
  VD_TRY(layers[0].forward(m, s, x, T, N, tok, &h));
  for (size_t l = 1; l < layers.size(); ++l) VD_TRY(layers[l].forward(m, s, {h}, T, N, tok, &h));
  *h_top = h;
  return VD_OK;
}
inline int lstm_stack_backward(vd_model* m, hipStream_t s, std::vector<SeqLSTM>& layers, const float* dh_last_top,
                               const float* dh_seq_top, std::vector<float*>* dxs) {
  const float* dseq = dh_seq_top;
  for (int i = (int)layers.size() - 1; i >= 0; --i) {
    std::vector<float*> d;
    VD_TRY(layers[i].backward(m, s, dseq, i == (int)layers.size() - 1 ? dh_last_top : nullptr, {}, &d));
    if (i > 0) dseq = d[0];
    else if (dxs) *dxs = d;
  }
  return VD_OK;
}

}  // namespace vdrt
