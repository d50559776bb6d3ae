// Native step runtime: the model-level half of the C ABI (include/visdial_hip.h, "model-level entry points").
//
// The operator-level entry points replace single nn / rnn module calls; a host that composes them (Python,
// LuaJIT-FFI) must also own everything that makes the training step fast: stream fork/join, the skewed two-layer
// wavefront, the host-side length sort, workspace reuse, launch order.  This file puts that orchestration behind the
// ABI for every encoder / decoder plug-in pair of the reference (encoders/*.lua x decoders/{disc,gen}.lua) as the proxy
// object a Lua `model.lua` drives: vd_model_create ~ Model:__init (model.lua:10-63), vd_model_upload_batch ~ the batch
// re-layout + :cuda() copies (model.lua:255-294, dataloader.lua:410-475), vd_model_forward_backward ~
// Model:forwardBackward (model.lua:249-342), vd_model_update ~ clamp + adam + lr decay (model.lua:96-105),
// vd_model_retrieve / vd_model_ranks ~ Model:retrieveBatch + utils.computeRanks (model.lua:344-430, utils.lua:106-128).
// No arithmetic lives here: every tensor op is one of the operator-level launches of this same library.
//   rt_core.h      model object, helpers, nn.Linear / CatLinear / nn.SeqLSTM building blocks
//   rt_encoders.h  the 11 encoder plug-ins        rt_decoders.h  disc, gen + the forwardBackward / retrieve branches
//
// Streams: main (option LSTM, criterion, decoder, optimiser), enc (encoder chains under a disc decoder), img (per-image
// projection + masks; also the history branch of lf-* / hre-*), tab (token sort + table gradient), copy (H2D uploads
// of the NEXT batch; two batch slots).
#include "rt_decoders.h"

using namespace vdrt;

vd_model::~vd_model() {}

namespace {

// [N x T] dataloader rows -> time-major [T x N] on the device; with `sorted` also the length-sort metadata of
// nn.py:SeqSort (stable sort by decreasing length, per-step active-row counts, gather indices)
int upload_tokens(BatchSlot& sl, SeqTok& ss, const std::string& tag, const int32_t* rows_major, int N, int T, bool sorted, hipStream_t s) {
  ss.T = T;
  ss.N = N;
  ss.present = true;
  ss.sorted = sorted;
  const size_t TN = (size_t)T * N;
  const size_t total = sorted ? 7 * TN + 2 * (size_t)N + T : TN;   // (+ up to 3 x T*N row-list entries)
  int32_t* stage = nullptr;
  VD_TRY(pin_get(sl.pinned, tag + ".stage", total * sizeof(int32_t), (void**)&stage));
  int32_t* tok = stage;
  std::vector<int> len(sorted ? N : 0);
  // A row's length is counted FROM ITS FIRST NON-ZERO TOKEN: the wavefront skips a row only at the steps before that token, where
  // maskZero() keeps the state at zero (exact), and applies the per-token mask inside the active span -- so a row that is not the
  // dataloader's right-aligned layout (left-aligned, interior zeros: reachable through the generic C ABI) is encoded exactly like the
  // masked per-step stack does.  For right-aligned rows this is the count of non-zero tokens.
  for (int n = 0; n < N; ++n) {
    int first = T;
    for (int t = 0; t < T; ++t) {
      const int32_t v = rows_major[(size_t)n * T + t];
      tok[(size_t)t * N + n] = v;
      if (v != 0 && first == T) first = t;
    }
    if (sorted) len[n] = T - first;
  }
  if (sorted) {
    int32_t* tok_sorted = stage + TN;
    int32_t* fwd_idx = stage + 2 * TN;
    int32_t* inv_idx = stage + 3 * TN;
    int32_t* perm = stage + 4 * TN;
    int32_t* inv = perm + N;
    int32_t* nact = inv + N;
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
    // non-pad (t, row) pairs of the sorted layout, in step order (rows [0, nact[t]) of step t)
    int32_t* act = nact + T;
    int na = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < ss.nact[t]; ++i) act[na++] = t * N + i;
    const int n0 = T > 0 ? ss.nact[0] : 0;
    int32_t* prev1 = act + TN;
    for (int k = n0; k < na; ++k) prev1[k - n0] = act[k] - N;
    ss.n_act = na;
    ss.n_act1 = na - n0;
  }
  int32_t* dev = nullptr;
  VD_TRY(dev_get(sl.bufs, tag + ".dev", total * sizeof(int32_t), (void**)&dev));
  VD_HIP(hipMemcpyAsync(dev, stage, total * sizeof(int32_t), hipMemcpyHostToDevice, s));
  ss.tok = dev;
  if (sorted) {
    ss.tok_sorted = dev + TN;
    ss.fwd_idx = dev + 2 * TN;
    ss.inv_idx = dev + 3 * TN;
    ss.perm = dev + 4 * TN;
    ss.inv = ss.perm + N;
    ss.nact_dev = ss.inv + N;
    ss.act = ss.nact_dev + T;
    ss.act1 = ss.act + (T > 0 ? ss.nact[0] : 0);
    ss.prev1 = ss.act + TN;
  }
  return VD_OK;
}

bool has(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

}  // namespace

extern "C" {

int vd_model_create(const vd_model_params* p, const char* encoder, const char* decoder, vd_model** out) {
  VD_CHECK_ARG(p && encoder && decoder && out, "vd_model_create: null argument");
  std::unique_ptr<Encoder> enc = make_encoder(encoder);
  std::unique_ptr<Decoder> dec = make_decoder(decoder);
  if (!enc || !dec) {
    vd_set_error("vd_model_create: unknown plug-in pair '%s' + '%s' (encoders: lf-ques, lf-ques-im, lf-ques-hist, lf-ques-im-hist, "
                 "lf-att-ques-im-hist, hre-ques-hist, hre-ques-im-hist, hrea-ques-im-hist, mn-ques-hist, mn-ques-im-hist, "
                 "mn-att-ques-im-hist; decoders: disc, gen)", encoder, decoder);
    return VD_ERR_ARG;
  }
  const std::string en(encoder);
  const bool use_im = has(en, "im"), is_att = has(en, "att") && !has(en, "hrea");
  VD_CHECK_ARG(p->rnnHiddenSize > 0 && p->rnnHiddenSize % 32 == 0 && p->embedSize > 0 && p->embedSize % 4 == 0 && p->vocabSize > 0 &&
                   p->maxQuesCount > 0 && p->numOptions > 0,
               "vd_model_create: rnnHiddenSize must be a positive multiple of 32, embedSize of 4; vocabSize, maxQuesCount, numOptions > 0");
  // the projection table [V+1 x 4H] of the option / decoder recurrences is gathered with 32-bit byte offsets (common.h)
  VD_CHECK_ARG(((long)p->vocabSize + 1) * 4 * p->rnnHiddenSize * 4 < (1L << 32),
               "vd_model_create: (vocabSize + 1) x 4 x rnnHiddenSize floats exceed 4 GB");
  VD_CHECK_ARG(!use_im || (p->imgFeatureSize > 0 && p->imgFeatureSize % 4 == 0), "vd_model_create: imgFeatureSize must be a positive multiple of 4");
  VD_CHECK_ARG(!is_att || (p->commonEmbeddingSize > 0 && p->commonEmbeddingSize % 4 == 0 && p->imgSpatialSize > 0),
               "vd_model_create: commonEmbeddingSize must be a positive multiple of 4 and imgSpatialSize > 0 for attention encoders");
  VD_CHECK_ARG(!(has(en, "hre") && use_im) || (p->imgEmbedSize > 0 && p->imgEmbedSize % 4 == 0),
               "vd_model_create: imgEmbedSize must be a positive multiple of 4 for hre*-ques-im-hist");
  VD_CHECK_ARG(p->dropout >= 0.f && p->dropout < 1.f, "vd_model_create: dropout must lie in [0, 1)");
  vd_model* m = new vd_model();
  m->p = *p;
  if (m->p.numAttentionLayers < 1 || has(en, "lf-att")) m->p.numAttentionLayers = 1;   // (lf-att-ques-im-hist.lua:49 hard-codes one hop)
  if (m->p.numLayers < 1) m->p.numLayers = 2;        // opts.lua:27
  m->enc_name = encoder;
  m->dec_name = decoder;
  m->use_im = use_im;
  m->use_hist = has(en, "hist");
  m->is_att = is_att;
  m->is_graph = en.rfind("mn", 0) == 0 || en.rfind("lf-att", 0) == 0;
  m->lr = p->learningRate;
  m->streams = p->useStreams != 0;
  m->enc = std::move(enc);
  m->dec = std::move(dec);
  // wrapper = Sequential(encoder, decoder) -> getParameters (model.lua:45-55): embed | encoder tensors | decoder tensors
  m->spec.push_back(Tensor{"embed", 0, p->vocabSize + 1, p->embedSize, 0});
  m->enc->declare(m);
  m->dec->declare(m);
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
  // throughput work (option LSTM, criterion, optimiser) at the LEAST priority, the latency-bound encoder chains and
  // uploads at the GREATEST: their small workgroups take free slots ahead of the next big-kernel workgroup
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (hipStreamCreateWithPriority(&m->s_main, hipStreamNonBlocking, least) != hipSuccess) return fail(VD_ERR_HIP);
  for (hipStream_t* s : {&m->s_enc, &m->s_img, &m->s_copy})
    if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) != hipSuccess) return fail(VD_ERR_HIP);
  // the table-gradient stream shares the `greatest` class (one hardware queue per priority class): on a queue of its own its
  // HBM-bound row sum would start beside the MFMA-bound dWh contraction and cost 0.5 ms per step (profiles/r03_experiments.txt)
  const int tab_prio = greatest;
  if (hipStreamCreateWithPriority(&m->s_tab, hipStreamNonBlocking, tab_prio) != hipSuccess) return fail(VD_ERR_HIP);
  if (least - greatest >= 2 &&
      hipStreamCreateWithPriority(&m->s_wg, hipStreamNonBlocking, (least + greatest) / 2) != hipSuccess)
    return fail(VD_ERR_HIP);
  // No encoder uses both side branches, so the history branch of lf-* / hre-* shares the image-branch stream: HIP
  // multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4), and a sixth stream put the
  // table-gradient stream on the main stream's queue (measured: +0.75 ms per headline step).  Hosts that own the
  // process should go further and export GPU_MAX_HW_QUEUES=1 before HIP initialises (bench.py does): with every stream
  // of the step multiplexed onto ONE hardware queue the cross-stream event waits resolve inside the command processor
  // and the headline step is 0.75-1.0 ms (3-4 %) faster (profiles/r02_hw_queues.txt).
  m->s_hist = m->s_img;
  m->ev_pool.resize(64);
  for (auto& e : m->ev_pool)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
  if (hipEventCreateWithFlags(&m->ev_loss, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
  if (hipEventCreateWithFlags(&m->ev_enc_grads, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
  if (hipEventCreateWithFlags(&m->ev_updated, hipEventDisableTiming) != hipSuccess) return fail(VD_ERR_HIP);
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
  for (auto& e : m->ev_pool)
    if (e) (void)hipEventDestroy(e);
  if (m->ev_loss) (void)hipEventDestroy(m->ev_loss);
  if (m->ev_enc_grads) (void)hipEventDestroy(m->ev_enc_grads);
  if (m->ev_updated) (void)hipEventDestroy(m->ev_updated);
  for (auto& e : m->ev_prof)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t s : {m->s_main, m->s_enc, m->s_img, m->s_tab, m->s_copy, m->s_wg})
    if (s) {
      (void)hipStreamDestroy(s);
    }
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

// Data-parallel gradient bucketing: the encoder's own tensors occupy the flat element range [lo, hi)
// (everything between the shared embedding and the decoder's tensors).  Under a `disc` decoder they are final when
// the encoder backward ends on its side stream, long before the option-LSTM backward does: a host makes its
// communication stream wait for that point and all-reduces the range underneath the rest of the step.
int vd_model_encoder_range(const vd_model* m, int64_t* lo, int64_t* hi) {
  VD_CHECK_ARG(m && lo && hi, "vd_model_encoder_range: null");
  const char* first_dec = m->dec_name == "disc" ? "opt.W" : "dec1.W";
  *lo = m->spec[1].off;
  *hi = m->spec[m->index.at(first_dec)].off;
  return VD_OK;
}
int vd_model_wait_encoder_grads(vd_model* m, void* stream) {
  VD_CHECK_ARG(m && m->enc_grads_recorded, "vd_model_wait_encoder_grads: no backward pass has been enqueued");
  VD_HIP(hipStreamWaitEvent((hipStream_t)stream, m->ev_enc_grads, 0));
  return VD_OK;
}

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

// pin the nn.Dropout noise of one call site (q_emb, h_emb, hatt, img_tr, iqc[, iqc2..], u; fuse; img); host == NULL
// clears all
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
// (staged into pinned memory), the device copy completes in the background.  Which fields are read follows the
// plug-in pair: hist / img_feat by the encoder name (opts.lua:54-67), options by `disc`, answer_in/out (training) and
// option_in/out (retrieval) by `gen`; fields the pair does not use may be NULL.
int vd_model_upload_batch(vd_model* m, const vd_batch* hb) {
  VD_CHECK_ARG(m && hb && hb->ques_fwd && hb->B > 0 && hb->Tq > 0, "vd_model_upload_batch: missing ques_fwd / B / Tq");
  VD_CHECK_ARG(!m->use_hist || (hb->hist && hb->Th > 0), "vd_model_upload_batch: encoder '%s' needs hist", m->enc_name.c_str());
  VD_CHECK_ARG(!m->use_im || hb->img_feat, "vd_model_upload_batch: encoder '%s' needs img_feat", m->enc_name.c_str());
  const bool disc = m->dec_name == "disc";
  VD_CHECK_ARG(!disc || (hb->options && hb->To > 0), "vd_model_upload_batch: decoder 'disc' needs options");
  // (decoder gen: answer_in/answer_out for training, option_in/option_out for retrieval, neither for generation)
  VdRange r("vd_model_upload_batch");
  BatchSlot& sl = m->slot[m->cur < 0 ? 0 : (m->cur ^ 1)];   // the slot the running step does not read
  hipStream_t s = m->s_copy;
  // the step that last read this slot may still be executing (the host runs ahead of the device)
  if (sl.used) VD_HIP(hipStreamWaitEvent(s, sl.done, 0));
  // ... and the previous upload INTO this slot may still be queued behind that wait, reading the slot's pinned staging
  // buffers: a host that runs two or more steps ahead without reading a loss (forward_backward loops, deferred loss)
  // would otherwise overwrite them under the copy.  In the pipelined training loop this returns at once.
  if (sl.uploaded) VD_HIP(hipEventSynchronize(sl.ready));
  sl.uploaded = true;
  sl.B = hb->B;
  const int R = m->p.maxQuesCount, N = hb->B * R, O = m->p.numOptions;
  const long NO = (long)N * O;
  for (SeqTok* t : {&sl.q, &sl.h, &sl.opt, &sl.ain, &sl.aout, &sl.oin, &sl.oout}) t->present = false;
  VD_TRY(upload_tokens(sl, sl.q, "q", hb->ques_fwd, N, hb->Tq, m->is_graph, s));
  // the history branch of the Sequential encoders runs as a length-sorted two-layer wavefront too (rt_encoders.h: HistWave)
  const bool hist_wave = !m->is_graph && m->p.numLayers == 2;
  if (m->use_hist) VD_TRY(upload_tokens(sl, sl.h, "h", hb->hist, N, hb->Th, m->is_graph || hist_wave, s));
  if (m->use_im) {
    // [B x S*S x C] for the attention encoders (model.lua:262-265 keeps one map per image), [B x F] otherwise
    const size_t img_n = (size_t)hb->B * (m->is_att ? (size_t)m->p.imgSpatialSize * m->p.imgSpatialSize : 1) * m->p.imgFeatureSize;
    float* ip = nullptr;
    VD_TRY(pin_get(sl.pinned, "img.stage", img_n * sizeof(float), (void**)&ip));
    memcpy(ip, hb->img_feat, img_n * sizeof(float));
    VD_TRY(dev_get(sl.bufs, "img", img_n * sizeof(float), (void**)&sl.img));
    VD_HIP(hipMemcpyAsync(sl.img, ip, img_n * sizeof(float), hipMemcpyHostToDevice, s));
  }
  if (disc) {
    // Candidate answers repeat inside a batch on real VisDial (every round's 100 options draw from one answer pool: the
    // popular answers appear in most rounds), and the option encoding depends on the tokens alone (decoders/disc.lua:4-15):
    // encode each DISTINCT row once and address it through opt_uid (first-occurrence order; exact -- duplicates share
    // one forward value and their gradients add).  Synthetic batches have no repeats: the plain path, no extra kernels.
    sl.opt_uid = nullptr;
    sl.opt_total = (int)NO;
    const int To = hb->To;
    bool dedup = false;
    if (NO > 1) {
      std::vector<int32_t> uid((size_t)NO), uniq;
      uniq.reserve((size_t)NO * To);
      const size_t cap = (size_t)1 << (64 - __builtin_clzll((unsigned long long)(2 * NO)));   // power of two >= 2 NO
      std::vector<int32_t> table(cap, -1);
      int U = 0;
      for (long r = 0; r < NO; ++r) {
        const int32_t* row = hb->options + (size_t)r * To;
        uint64_t h = 1469598103934665603ull;
        for (int t = 0; t < To; ++t) h = (h ^ (uint32_t)row[t]) * 1099511628211ull;
        size_t pos = (size_t)(h ^ (h >> 29)) & (cap - 1);
        for (;;) {
          const int32_t u = table[pos];
          if (u < 0) {
            table[pos] = U;
            uid[r] = U++;
            uniq.insert(uniq.end(), row, row + To);
            break;
          }
          if (memcmp(uniq.data() + (size_t)u * To, row, (size_t)To * sizeof(int32_t)) == 0) {
            uid[r] = u;
            break;
          }
          pos = (pos + 1) & (cap - 1);
        }
      }
      if ((double)U <= 0.95 * (double)NO) {     // worth two extra [N*O x H] passes (gather / scatter-add)
        dedup = true;
        VD_TRY(upload_tokens(sl, sl.opt, "opt", uniq.data(), U, To, false, s));
        int32_t* up = nullptr;
        VD_TRY(pin_get(sl.pinned, "opt.uid.stage", (size_t)NO * sizeof(int32_t), (void**)&up));
        memcpy(up, uid.data(), (size_t)NO * sizeof(int32_t));
        VD_TRY(dev_get(sl.bufs, "opt.uid", (size_t)NO * sizeof(int32_t), (void**)&sl.opt_uid));
        VD_HIP(hipMemcpyAsync(sl.opt_uid, up, (size_t)NO * sizeof(int32_t), hipMemcpyHostToDevice, s));
      }
    }
    if (!dedup) VD_TRY(upload_tokens(sl, sl.opt, "opt", hb->options, (int)NO, hb->To, false, s));   // [N x O x To] -> [To x N*O]
    sl.opt_sort_off = sl.opt_sort_perm = nullptr;
    {   // counting sort of the option tokens on the copy stream (the table gradient's row order depends on the batch alone)
      const long V1 = (long)m->p.vocabSize + 1, n = (long)sl.opt.T * sl.opt.N;
      int32_t* work;
      VD_TRY(dev_get(sl.bufs, "opt.sort_off", (size_t)(V1 + 1) * sizeof(int32_t), (void**)&sl.opt_sort_off));
      VD_TRY(dev_get(sl.bufs, "opt.sort_work", (size_t)2 * V1 * sizeof(int32_t), (void**)&work));
      VD_TRY(dev_get(sl.bufs, "opt.sort_perm", (size_t)n * sizeof(int32_t), (void**)&sl.opt_sort_perm));
      VD_TRY(vd_token_sort(sl.opt.tok, n, (int)V1, sl.opt_sort_off, work, sl.opt_sort_perm, s));
    }
  }
  if (!disc && hb->answer_in && hb->answer_out) {
    VD_TRY(upload_tokens(sl, sl.ain, "ain", hb->answer_in, N, hb->Ta, false, s));
    VD_TRY(upload_tokens(sl, sl.aout, "aout", hb->answer_out, N, hb->Ta, false, s));
  }
  if (!disc && hb->option_in && hb->option_out) {                                               // model.lua:393-399
    VD_TRY(upload_tokens(sl, sl.oin, "oin", hb->option_in, (int)NO, hb->To, false, s));
    VD_TRY(upload_tokens(sl, sl.oout, "oout", hb->option_out, (int)NO, hb->To, false, s));
  }
  sl.has_gt = hb->answer_ind != nullptr;
  sl.gt_host.assign(N, 0);
  int32_t* gp = nullptr;
  VD_TRY(pin_get(sl.pinned, "gt.stage", (size_t)N * sizeof(int32_t), (void**)&gp));
  for (int n = 0; n < N; ++n) {
    const int g = hb->answer_ind ? hb->answer_ind[n] - 1 : 0;   // 1-based on disk (prepro.py:169)
    VD_CHECK_ARG(!hb->answer_ind || (g >= 0 && g < O), "vd_model_upload_batch: answer_ind[%d] = %d out of 1..%d", n, g + 1, O);
    sl.gt_host[n] = g;
    gp[n] = g;
  }
  VD_TRY(dev_get(sl.bufs, "gt", (size_t)N * sizeof(int32_t), (void**)&sl.gt));
  VD_HIP(hipMemcpyAsync(sl.gt, gp, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, s));
  VD_HIP(hipEventRecord(sl.ready, s));
  m->uploaded = (int)(&sl - m->slot);
  return VD_OK;
}

static int begin_step(vd_model* m, bool zero_grads, BatchSlot** out) {
  VD_CHECK_ARG(m && m->uploaded >= 0, "no batch uploaded");
  m->cur = m->uploaded;
  m->enc_grads_recorded = false;
  BatchSlot& b = m->slot[m->cur];
  m->N = b.q.N;
  m->O = m->p.numOptions;
  hipStream_t s = m->s_main;
  VD_HIP(hipStreamWaitEvent(s, b.ready, 0));
  if (zero_grads) VD_TRY(vd_memset(m->G, 0, m->numel * 4, s));                  // wrapper:zeroGradParameters (model.lua:68)
  VD_TRY(vd_memset(Wp(m, "embed"), 0, (long)m->p.embedSize * 4, s));            // LookupTableMaskZero zeroes the pad row
  b.used = true;
  *out = &b;
  return VD_OK;
}

// Model:forwardBackward on the uploaded batch (model.lua:249-342).  Enqueues only.
int vd_model_forward_backward(vd_model* m, int only_forward) {
  VdRange r(only_forward ? "vd_model_forward" : "vd_model_forward_backward");
  BatchSlot* b = nullptr;
  VD_TRY(begin_step(m, !only_forward, &b));
  const int rc = m->dec->forward_backward(m, *b, only_forward != 0);
  VD_HIP(hipEventRecord(b->done, m->s_main));
  return rc;
}

// Model:retrieveBatch up to the option scores (model.lua:344-425): disc = option scores of a forward pass, gen = the
// log-likelihood of every candidate under the decoder.  Read them with vd_model_scores / vd_model_ranks.
int vd_model_retrieve(vd_model* m) {
  VdRange r("vd_model_retrieve");
  BatchSlot* b = nullptr;
  VD_TRY(begin_step(m, false, &b));
  const int rc = m->dec->retrieve(m, *b);
  VD_HIP(hipEventRecord(b->done, m->s_main));
  return rc;
}

// Model:generateAnswers, device side (model.lua:432-613).  vd_model_encode = `forwardBackward(batch, true, true)`
// (model.lua:464): encoder forward of the uploaded batch, state kept for vd_model_decode_begin.
int vd_model_encode(vd_model* m) {
  BatchSlot* b = nullptr;
  VD_TRY(begin_step(m, false, &b));
  m->gen_enc_out = nullptr;
  float* out = nullptr;
  const int rc = m->enc->forward(m, m->s_main, *b, &out);
  VD_HIP(hipEventRecord(b->done, m->s_main));
  if (rc == VD_OK) {
    m->gen_enc_out = out;
    m->gen_seq_len = m->enc->seqLen(*b);
  }
  return rc;
}
int vd_model_decode_begin(vd_model* m, const int32_t* rounds, int n) {
  VD_CHECK_ARG(m, "vd_model_decode_begin: null model");
  return m->dec->gen_begin(m, rounds, n);
}
int vd_model_decode_step(vd_model* m, const int32_t* tokens, float* host_logprobs) {
  VD_CHECK_ARG(m, "vd_model_decode_step: null model");
  return m->dec->gen_step(m, tokens, host_logprobs);
}
int vd_model_decode_select(vd_model* m, const int32_t* src, int n_keep) {
  VD_CHECK_ARG(m, "vd_model_decode_select: null model");
  return m->dec->gen_select(m, src, n_keep);
}

// waits for the loss of the last vd_model_forward_backward: disc = mean cross-entropy over the rounds, gen = summed
// NLL over the non-pad answer tokens (SequencerCriterion of ClassNLLCriterion, model.lua:32-36)
int vd_model_loss(vd_model* m, float* loss) {
  VD_CHECK_ARG(m && loss && m->cur >= 0 && m->loss_n > 0, "vd_model_loss: nothing to read");
  VD_HIP(hipEventSynchronize(m->ev_loss));
  double s = 0;
  for (long i = 0; i < m->loss_n; ++i) s += m->loss_host[i];
  *loss = (float)(m->loss_is_sum ? s : s / (double)m->loss_n);
  return VD_OK;
}

// [gradient already reduced by the caller] -> clamp(-5, 5) -> adam -> lr decay (model.lua:96-105; optim_updates.lua:62-91)
int vd_model_update(vd_model* m, float gscale) {
  VD_CHECK_ARG(m, "vd_model_update: null model");
  VdRange r("vd_model_update: clamp + adam");
  m->adam_t += 1;
  const double t = m->adam_t;
  const float step = (float)(m->lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
  VD_TRY(vd_clamp_adam(m->W, m->G, m->M, m->V, m->numel, gscale, 5.0f, 0.9f, 0.999f, 1e-8f, step, m->s_main));
  VD_HIP(hipEventRecord(m->ev_updated, m->s_main));
  m->updated_recorded = true;
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

// scores [N x O] of the last forward / retrieval (host buffer)
int vd_model_scores(vd_model* m, float* host, int64_t n) {
  VD_CHECK_ARG(m && host && m->cur >= 0 && m->scores && n == (int64_t)m->N * m->O, "vd_model_scores: no scores / bad size");
  VD_HIP(hipStreamSynchronize(m->s_main));
  VD_HIP(hipMemcpy(host, m->scores, n * sizeof(float), hipMemcpyDeviceToHost));
  return VD_OK;
}

// utils.computeRanks on the scores of the last forward / retrieval: use_gt -> ranks_out[N] (rank of the ground-truth
// option), else ranks_out[N x O] (rank of every option), 1-based (utils.lua:106-128)
int vd_model_ranks(vd_model* m, int use_gt, int32_t* ranks_out) {
  VD_CHECK_ARG(m && ranks_out && m->cur >= 0 && m->scores, "vd_model_ranks: nothing to rank");
  const int N = m->N, O = m->O;
  const BatchSlot& b = m->slot[m->cur];
  VD_CHECK_ARG(!use_gt || b.has_gt, "vd_model_ranks: the batch carries no answer_ind");
  int32_t* ranks;
  VD_TRY(ws_get(m, "opt.ranks", (size_t)N * O, &ranks));
  VD_TRY(vd_ranks(m->scores, ranks, N, O, m->s_main));
  std::vector<int32_t> host((size_t)N * O);
  VD_HIP(hipMemcpyAsync(host.data(), ranks, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m->s_main));
  VD_HIP(hipStreamSynchronize(m->s_main));
  if (use_gt) {
    for (int n = 0; n < N; ++n) ranks_out[n] = host[(size_t)n * O + b.gt_host[n]];
  } else {
    memcpy(ranks_out, host.data(), host.size() * sizeof(int32_t));
  }
  return VD_OK;
}

// device time of three kernel families inside the last training step, ms.  `disc` pairs: the option LSTM [fwd, bwd, dWh];
// `gen` pairs over a Sequential encoder with a history branch: [history branch fwd, history branch bwd, vocabulary projection +
// criterion + their gradients]; zeros when the last call was not such a step
int vd_model_family_ms(vd_model* m, float* ms3) {
  VD_CHECK_ARG(m && ms3, "vd_model_family_ms: null");
  ms3[0] = ms3[1] = ms3[2] = 0.f;
  if (!m->prof_valid) return VD_OK;
  VD_HIP(hipStreamSynchronize(m->s_main));
  VD_HIP(hipEventElapsedTime(&ms3[0], m->ev_prof[0], m->ev_prof[1]));
  VD_HIP(hipEventElapsedTime(&ms3[1], m->ev_prof[2], m->ev_prof[3]));
  VD_HIP(hipEventElapsedTime(&ms3[2], m->ev_prof[4], m->ev_prof[5]));
#ifdef VD_PROBE_PHASES
  { float t[12]; hipEvent_t* pe = vdrt::probe_events();
    for (int i = 1; i < 6; ++i) (void)hipEventElapsedTime(&t[i], m->ev_prof[0], m->ev_prof[i]);
    for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&t[6 + i], m->ev_prof[0], pe[i]);
    fprintf(stderr, "PHASES (ms after option-fwd start) main: fwd end %.2f | bwd %.2f -> %.2f | dWh %.2f -> %.2f | all joined %.2f  ||  enc fwd end %.2f | enc bwd chain end %.2f | "
            "wg chain end %.2f | table chain end %.2f\n", t[1], t[2], t[3], t[4], t[5], t[10], t[6], t[7], t[8], t[9]); }
#endif
  return VD_OK;
}

// rows the option LSTM executed for the batch of the LAST STEP (vd_model_forward_backward / vd_model_retrieve) vs the N * O candidates
// they stand for (decoder disc); before any step: of the uploaded batch.  In a pipelined loop that is NOT the prefetched batch.
int vd_model_option_rows(vd_model* m, int64_t* executed, int64_t* total) {
  VD_CHECK_ARG(m && executed && total && (m->cur >= 0 || m->uploaded >= 0), "vd_model_option_rows: no batch uploaded");
  const BatchSlot& b = m->slot[m->cur >= 0 ? m->cur : m->uploaded];
  *executed = b.opt.present ? b.opt.N : 0;
  *total = b.opt.present ? (b.opt_total ? b.opt_total : b.opt.N) : 0;
  return VD_OK;
}

int vd_model_synchronize(vd_model* m) {
  VD_CHECK_ARG(m, "vd_model_synchronize: null model");
  for (hipStream_t s : {m->s_copy, m->s_enc, m->s_img, m->s_tab, m->s_wg, m->s_main})
    if (s) VD_HIP(hipStreamSynchronize(s));
  return VD_OK;
}

}  // extern "C"
