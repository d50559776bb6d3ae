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

extern "C" {

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
