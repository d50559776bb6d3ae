// Discriminative decoder head: option scoring (decoders/disc.lua:22-29 nn.MM + Squeeze),
// nn.CrossEntropyCriterion (model.lua:37-38,330,334) and the rank computation of
// utils.lua:106-128 (computeRanks), all as wave-reduction kernels (HBM-bound: the
// [N x O x H] option encodings are read exactly once in forward and once in backward).
#include "common.h"

#define MAX_OPT 128

// One workgroup per QA round n.
//   score[o] = <optH[n,o,:], enc[n,:]> ; loss_n = logsumexp(score) - score[gt]
//   if train: ds = (softmax - onehot) * gscale ; dOptH[n,o,:] = ds[o]*enc[n,:] ; dEnc[n,:] = sum_o ds[o]*optH[n,o,:]
__global__ void __launch_bounds__(256)
score_ce_kernel(const float* __restrict__ optH, const float* __restrict__ enc, const int* __restrict__ gt,
                float* __restrict__ scores, float* __restrict__ loss_rows, float* __restrict__ dOptH,
                float* __restrict__ dEnc, int O, int H, float gscale) {
  __shared__ float sc[MAX_OPT];
  __shared__ float ds[MAX_OPT];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* e = enc + (long)n * H;
  const float* oh = optH + (long)n * O * H;
  for (int o = wave; o < O; o += 4) {
    float a = 0.f;
    for (int k = lane * 4; k < H; k += 256) {
      const float4 x = *reinterpret_cast<const float4*>(oh + (long)o * H + k);
      const float4 y = *reinterpret_cast<const float4*>(e + k);
      a += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    a = wave_sum(a);
    if (lane == 0) {
      sc[o] = a;
      scores[(long)n * O + o] = a;
    }
  }
  __syncthreads();
  if (gt == nullptr) return;
  if (wave == 0) {
    float mx = -INFINITY;
    for (int o = lane; o < O; o += 64) mx = fmaxf(mx, sc[o]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int o = lane; o < O; o += 64) sum += expf(sc[o] - mx);
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    const int g = gt[n];
    if (lane == 0) loss_rows[n] = lse - sc[g];
    for (int o = lane; o < O; o += 64) ds[o] = (expf(sc[o] - lse) - (o == g ? 1.f : 0.f)) * gscale;
  }
  if (dOptH == nullptr) return;
  __syncthreads();
  for (int k = tid * 4; k < H; k += 1024) {
    const float4 ev = *reinterpret_cast<const float4*>(e + k);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int o = 0; o < O; ++o) {
      const float d = ds[o];
      const float4 x = *reinterpret_cast<const float4*>(oh + (long)o * H + k);
      acc.x += d * x.x;
      acc.y += d * x.y;
      acc.z += d * x.z;
      acc.w += d * x.w;
      *reinterpret_cast<float4*>(dOptH + ((long)n * O + o) * H + k) =
          make_float4(d * ev.x, d * ev.y, d * ev.z, d * ev.w);
    }
    *reinterpret_cast<float4*>(dEnc + (long)n * H + k) = acc;
  }
}

// rank[n,o] = 1 + #{j : s[j] > s[o]  or (s[j] == s[o] and j < o)}   (descending sort position)
__global__ void __launch_bounds__(128)
ranks_kernel(const float* __restrict__ scores, int* __restrict__ ranks, int O) {
  __shared__ float sc[MAX_OPT];
  const int n = blockIdx.x, o = threadIdx.x;
  if (o < O) sc[o] = scores[(long)n * O + o];
  __syncthreads();
  if (o >= O) return;
  const float s = sc[o];
  int r = 1;
  for (int j = 0; j < O; ++j) r += (sc[j] > s) || (sc[j] == s && j < o);
  ranks[(long)n * O + o] = r;
}

// Generative head, per (t, n) row of logits [rows x ld] (V valid columns):
//   nn.Sequencer(nn.MaskZero(nn.LogSoftMax(), 1)) + SequencerCriterion(MaskZeroCriterion(ClassNLL(sum)))
//   (decoders/gen.lua:23-24, model.lua:33-36): rows whose decoder input token is 0 contribute nothing.
//   loss_rows[r] = logsumexp(logits[r]) - logits[r, target-1]   (target ids are 1-based vocabulary ids)
//   logits[r, :] <- (softmax - onehot) (the gradient, unscaled: sizeAverage = false), zeros for pad rows.
__global__ void __launch_bounds__(256)
logsoftmax_nll_kernel(float* __restrict__ logits, long ld, int V, const int* __restrict__ tok_in,
                      const int* __restrict__ target, float* __restrict__ loss_rows, int write_grad) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* row = logits + r * ld;
  if (write_grad && tid < ld - V) row[V + tid] = 0.f;  // keep the alignment pad columns finite (K-padding rule)
  // masked rows: a pad decoder input (MaskZero, gen.lua:23-24) or a pad TARGET -- utils.computeLhood masks on
  // words == 0 (utils.lua:86-102): an empty candidate has option_in = <START>,0.. and option_out = 0.. (processOptions
  // writes no <END> for length 0), so its first row has a non-pad input and target 0; it must contribute 0, not
  // read column -1
  if (tok_in[r] == 0 || target[r] <= 0) {
    if (tid == 0) loss_rows[r] = 0.f;
    if (write_grad)
      for (int c = tid; c < V; c += 256) row[c] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += expf(row[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float lse = mx + logf(red[4] + red[5] + red[6] + red[7]);
  const int tgt = target[r] - 1;
  if (tid == 0) loss_rows[r] = lse - row[tgt];
  if (write_grad) {
    __syncthreads();
    for (int c = tid; c < V; c += 256) row[c] = expf(row[c] - lse) - (c == tgt ? 1.f : 0.f);
  }
}

// in-place log-softmax of `rows` rows of V logits (decoders/gen.lua:24 at sampling / beam-search time)
__global__ void __launch_bounds__(256)
log_softmax_rows_kernel(float* __restrict__ x, long ld, int V) {
  __shared__ float red[8];
  float* row = x + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += expf(row[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float lse = mx + logf(red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < V; c += 256) row[c] -= lse;
}

extern "C" {

int vd_log_softmax_rows(float* x, int64_t ld, int64_t rows, int V, void* stream) {
  VD_CHECK_ARG(x && rows >= 0 && V >= 1 && ld >= V, "vd_log_softmax_rows: bad args");
  if (rows == 0) return VD_OK;
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (long)ld, V);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_logsoftmax_nll(float* logits, int64_t ld, int64_t rows, int V, const int32_t* tok_in,
                      const int32_t* target, float* loss_rows, int write_grad, void* stream) {
  VD_CHECK_ARG(logits && tok_in && target && loss_rows && rows >= 0 && V >= 1 && ld >= V,
               "vd_logsoftmax_nll: bad args");
  if (rows == 0) return VD_OK;
  hipLaunchKernelGGL(logsoftmax_nll_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits,
                     (long)ld, V, tok_in, target, loss_rows, write_grad);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// gt: 0-based ground-truth option per round, or null for scores only.
// dOptH/dEnc: null for forward-only.  gscale = 1/N_global (CrossEntropyCriterion sizeAverage).
int vd_score_ce(const float* optH, const float* enc, const int32_t* gt, float* scores, float* loss_rows,
                float* dOptH, float* dEnc, int N, int O, int H, float gscale, void* stream) {
  VD_CHECK_ARG(optH && enc && scores && N >= 0 && O >= 1 && O <= MAX_OPT && H % 4 == 0,
               "vd_score_ce: bad args (O=%d must be <= %d)", O, MAX_OPT);
  VD_CHECK_ARG(gt == nullptr || loss_rows != nullptr, "vd_score_ce: loss_rows required with gt");
  VD_CHECK_ARG((dOptH == nullptr) == (dEnc == nullptr), "vd_score_ce: dOptH and dEnc go together");
  VD_CHECK_ARG(dOptH == nullptr || gt != nullptr, "vd_score_ce: gradients need gt");
  if (N == 0) return VD_OK;
  hipLaunchKernelGGL(score_ce_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, optH, enc, gt, scores,
                     loss_rows, dOptH, dEnc, O, H, gscale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_ranks(const float* scores, int32_t* ranks, int N, int O, void* stream) {
  VD_CHECK_ARG(scores && ranks && N >= 0 && O >= 1 && O <= MAX_OPT, "vd_ranks: bad args");
  if (N == 0) return VD_OK;
  hipLaunchKernelGGL(ranks_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, scores, ranks, O);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

}  // extern "C"
