// Memory-network attention over dialog history and the 1-hop SAN image attention of
// encoders/mn-att-ques-im-hist.lua (reference lines 48-62 and 68-104), plus
// model_utils/MaskSoftMax.lua semantics (mask byte 1 = hidden; hidden scores := -9999999).
//
// MI355X mapping:
//  * memory attention: one workgroup per dialog; the R x R scores, the masked softmax and
//    the weighted sums are wavefront reductions over H.
//  * image attention: the per-round image tensor img_tr = dropout(tanh(Linear(img))) is
//    NEVER materialised per round: the per-image pre-dropout map `pre` [B*S2 x H] stays
//    L2/MALL-resident and the GEMM operand loader applies the per-round dropout mask
//    (SURVEY.md H4).  img_common + ques_common + tanh + dropout run in the MFMA epilogue;
//    score/softmax(196)/weighted-sum is one wave-reduction kernel per QA round (K7).
#include "gemm_core.h"
#include "split_core.h"

// =====================================================================================
// Memory-network attention
// =====================================================================================
#define MN_MAX_R 16

// One 16-wave workgroup per dialog.  The R question states and R facts of the dialog (2 x 20 KB) are
// staged in LDS with one coalesced pass, so the R*R dot products and the R*H weighted sums never touch
// global memory again: the kernel is a handful of dependent global round trips instead of ~60.
#define MN_THREADS 1024
__global__ void __launch_bounds__(MN_THREADS)
mn_att_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Hm, const uint8_t* __restrict__ mask,
                  float* __restrict__ P, float* __restrict__ hAtt, int R, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // q [R*H] | h [R*H]
  __shared__ float S[MN_MAX_R * MN_MAX_R];
  float* q = lds;
  float* h = lds + R * H;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = MN_THREADS / 64;
  for (int i = tid * 4; i < R * H; i += MN_THREADS * 4) {
    *reinterpret_cast<float4*>(q + i) = *reinterpret_cast<const float4*>(Q + (long)b * R * H + i);
    *reinterpret_cast<float4*>(h + i) = *reinterpret_cast<const float4*>(Hm + (long)b * R * H + i);
  }
  __syncthreads();
  for (int p = wave; p < R * R; p += nw) {
    const int i = p / R, j = p % R;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s += q[i * H + k] * h[j * H + k];
    s = wave_sum(s);
    if (lane == 0) S[i * MN_MAX_R + j] = s;
  }
  __syncthreads();
  if (tid < R) {
    const int i = tid;
    const uint8_t* mrow = mask + ((long)b * R + i) * R;
    float mx = -INFINITY;
    for (int j = 0; j < R; ++j) {
      float v = mrow[j] ? -9999999.f : S[i * MN_MAX_R + j];
      S[i * MN_MAX_R + j] = v;
      mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int j = 0; j < R; ++j) {
      const float e = expf(S[i * MN_MAX_R + j] - mx);
      S[i * MN_MAX_R + j] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    for (int j = 0; j < R; ++j) {
      const float pv = S[i * MN_MAX_R + j] * inv;
      S[i * MN_MAX_R + j] = pv;
      P[((long)b * R + i) * R + j] = pv;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < R * H; idx += MN_THREADS) {
    const int i = idx / H, k = idx % H;
    float a = 0.f;
    for (int j = 0; j < R; ++j) a += S[i * MN_MAX_R + j] * h[j * H + k];
    hAtt[(long)b * R * H + idx] = a;
  }
}

__global__ void __launch_bounds__(MN_THREADS)
mn_att_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Hm, const float* __restrict__ P,
                  const float* __restrict__ dhAtt, float* __restrict__ dQ, float* __restrict__ dHm, int R,
                  int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // q | h | dhAtt, [R*H] each
  __shared__ float Ps[MN_MAX_R * MN_MAX_R];
  __shared__ float dS[MN_MAX_R * MN_MAX_R];
  float* q = lds;
  float* h = lds + R * H;
  float* da = lds + 2 * R * H;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = MN_THREADS / 64;
  for (int i = tid * 4; i < R * H; i += MN_THREADS * 4) {
    *reinterpret_cast<float4*>(q + i) = *reinterpret_cast<const float4*>(Q + (long)b * R * H + i);
    *reinterpret_cast<float4*>(h + i) = *reinterpret_cast<const float4*>(Hm + (long)b * R * H + i);
    *reinterpret_cast<float4*>(da + i) = *reinterpret_cast<const float4*>(dhAtt + (long)b * R * H + i);
  }
  __syncthreads();
  for (int p = wave; p < R * R; p += nw) {
    const int i = p / R, j = p % R;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s += da[i * H + k] * h[j * H + k];
    s = wave_sum(s);
    if (lane == 0) {
      dS[i * MN_MAX_R + j] = s;  // dP for now
      Ps[i * MN_MAX_R + j] = P[((long)b * R + i) * R + j];
    }
  }
  __syncthreads();
  if (tid < R) {
    const int i = tid;
    float dot = 0.f;
    for (int j = 0; j < R; ++j) dot += Ps[i * MN_MAX_R + j] * dS[i * MN_MAX_R + j];
    for (int j = 0; j < R; ++j) dS[i * MN_MAX_R + j] = Ps[i * MN_MAX_R + j] * (dS[i * MN_MAX_R + j] - dot);
  }
  __syncthreads();
  for (int idx = tid; idx < R * H; idx += MN_THREADS) {
    const int i = idx / H, k = idx % H;  // i doubles as the fact index j for dHm
    float aq = 0.f, ah = 0.f;
    for (int j = 0; j < R; ++j) {
      aq += dS[i * MN_MAX_R + j] * h[j * H + k];
      ah += Ps[j * MN_MAX_R + i] * da[j * H + k] + dS[j * MN_MAX_R + i] * q[j * H + k];
    }
    dQ[(long)b * R * H + idx] = aq;
    dHm[(long)b * R * H + idx] = ah;
  }
}

// =====================================================================================
// History attention of encoders/hrea-ques-im-hist.lua:83-131: additive scores
//   A[i,j] = sq[i] + sh[j]  (nn.Linear(H,1) on question / history states, Replicate + CAddTable),
//   MaskFuture (j > i -> 0), ReplaceZero(-inf) (every exact 0 -> -inf), SoftMax over j,
//   att[i,:] = sum_j P[i,j] * h[j,:]   (CMulTable + Sum(3)).   One workgroup per dialog.
// =====================================================================================
__global__ void __launch_bounds__(256)
hrea_att_fwd_kernel(const float* __restrict__ sq, const float* __restrict__ sh, const float* __restrict__ Hm,
                    float* __restrict__ P, float* __restrict__ att, int R, int H) {
  __shared__ float S[MN_MAX_R * MN_MAX_R];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* h = Hm + (long)b * R * H;
  if (tid < R) {
    const int i = tid;
    float mx = -INFINITY;
    for (int j = 0; j < R; ++j) {
      float v = (j > i) ? 0.f : sq[b * R + i] + sh[b * R + j];
      v = (v == 0.f) ? -INFINITY : v;
      S[i * MN_MAX_R + j] = v;
      mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int j = 0; j < R; ++j) {
      const float e = expf(S[i * MN_MAX_R + j] - mx);
      S[i * MN_MAX_R + j] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    for (int j = 0; j < R; ++j) {
      const float pv = S[i * MN_MAX_R + j] * inv;
      S[i * MN_MAX_R + j] = pv;
      P[((long)b * R + i) * R + j] = pv;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < R * H; idx += 256) {
    const int i = idx / H, k = idx % H;
    float a = 0.f;
    for (int j = 0; j <= i; ++j) a += S[i * MN_MAX_R + j] * h[j * H + k];
    att[(long)b * R * H + idx] = a;
  }
}

__global__ void __launch_bounds__(256)
hrea_att_bwd_kernel(const float* __restrict__ Hm, const float* __restrict__ P, const float* __restrict__ datt,
                    float* __restrict__ dsq, float* __restrict__ dsh, float* __restrict__ dHm, int R, int H) {
  __shared__ float Ps[MN_MAX_R * MN_MAX_R];
  __shared__ float dA[MN_MAX_R * MN_MAX_R];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* h = Hm + (long)b * R * H;
  const float* da = datt + (long)b * R * H;
  for (int p = wave; p < R * R; p += 4) {
    const int i = p / R, j = p % R;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s += da[i * H + k] * h[j * H + k];
    s = wave_sum(s);
    if (lane == 0) {
      dA[i * MN_MAX_R + j] = s;  // dP
      Ps[i * MN_MAX_R + j] = P[((long)b * R + i) * R + j];
    }
  }
  __syncthreads();
  if (tid < R) {
    const int i = tid;
    float dot = 0.f;
    for (int j = 0; j < R; ++j) dot += Ps[i * MN_MAX_R + j] * dA[i * MN_MAX_R + j];
    float rs = 0.f;
    for (int j = 0; j < R; ++j) {
      const float v = Ps[i * MN_MAX_R + j] * (dA[i * MN_MAX_R + j] - dot);
      dA[i * MN_MAX_R + j] = v;
      rs += v;
    }
    dsq[b * R + i] = rs;
  }
  __syncthreads();
  if (tid < R) {
    const int j = tid;
    float cs = 0.f;
    for (int i = 0; i < R; ++i) cs += dA[i * MN_MAX_R + j];
    dsh[b * R + j] = cs;
  }
  for (int idx = tid; idx < R * H; idx += 256) {
    const int j = idx / H, k = idx % H;
    float a = 0.f;
    for (int i = j; i < R; ++i) a += Ps[i * MN_MAX_R + j] * da[i * H + k];
    dHm[(long)b * R * H + idx] = a;
  }
}

// nn.Linear(H, 1): out[n] = <x[n,:], w> + b, and its backward (dw, db accumulated; dx written)
__global__ void __launch_bounds__(256)
rowdot_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                  float* __restrict__ out, int N, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < H; k += 64) s += x[(long)n * H + k] * w[k];
  s = wave_sum(s);
  if (lane == 0) out[n] = s + bias[0];
}
__global__ void __launch_bounds__(256)
rowdot_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dout,
                  float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dx, int N, int H) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= H) return;
  const float wk = w[k];
  float a = 0.f, bsum = 0.f;
  for (int n = 0; n < N; ++n) {
    const float d = dout[n];
    a += d * x[(long)n * H + k];
    dx[(long)n * H + k] = d * wk;
    bsum += d;
  }
  unsafeAtomicAdd(dw + k, a);
  if (k == 0) unsafeAtomicAdd(db, bsum);
}

// =====================================================================================
// Image attention
// =====================================================================================
// Row (n, s) of the per-round image tensor, read from the per-image map with the round's
// dropout mask applied on the fly.
struct SrcImgDropRow {
  static constexpr bool KMAJOR = false;
  static constexpr bool PLAIN = false;  // two streams (values + mask bytes): generic per-tile addressing
  const float* pre;      // [Bi*S2 x H]
  const uint8_t* mask;   // [N*S2 x H] or null
  int H, S2, R;
  float scale;
  __device__ __forceinline__ float4 ld4(int row, int k) const {
    const int n = row / S2, s = row - n * S2;
    float4 v = *reinterpret_cast<const float4*>(pre + ((long)(n / R) * S2 + s) * H + k);
    if (mask) {
      const uint32_t m = *reinterpret_cast<const uint32_t*>(mask + (long)row * H + k);
      v.x = (m & 0xff) ? v.x * scale : 0.f;
      v.y = (m & 0xff00) ? v.y * scale : 0.f;
      v.z = (m & 0xff0000) ? v.z * scale : 0.f;
      v.w = (m & 0xff000000u) ? v.w * scale : 0.f;
    }
    return v;
  }
};
// same tensor as a k-major operand: "r" runs over H (contiguous), "k" over rows (n, s).
struct SrcImgDropK {
  static constexpr bool KMAJOR = true;
  static constexpr bool PLAIN = false;
  const float* pre;
  const uint8_t* mask;
  int H, S2, R;
  float scale;
  __device__ __forceinline__ float4 ld4(int r, int row) const {
    SrcImgDropRow s{pre, mask, H, S2, R, scale};
    return s.ld4(row, r);
  }
};

// iqc = dropout2( tanh( acc + bc[col] + qc[n, col] ) )
struct EpiImgCommon {
  float* iqc;            // [N*S2 x Kc]
  const float* bc;       // [Kc]
  const float* qc;       // [N x Kc]
  const uint8_t* mask2;  // [N*S2 x Kc] or null
  int S2;
  float scale;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int col0, int lane, int M,
                                             int N, float* /*scr*/ = nullptr) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + j * 32 + (lane & 31);
      if (col >= N) continue;
      const float bv = bc[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + mfma_row(r, lane);
        if (row >= M) continue;
        const int n = row / S2;
        float v = tanhf(acc[j][r] + bv + qc[(long)n * N + col]);
        if (mask2) v = mask2[(long)row * N + col] ? v * scale : 0.f;
        iqc[(long)row * N + col] = v;
      }
    }
  }
};

// d(pre)[image row, col] += (acc + p[row]*datt[n, col]) * mask1*scale   (sum over the R rounds)
struct EpiImgTrBwd {
  float* dpre;           // [Bi*S2 x H], atomically accumulated
  const float* p;        // [N*S2]
  const float* datt;     // [N x H]
  const uint8_t* mask1;  // [N*S2 x H] or null
  int S2, R;
  float scale;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int col0, int lane, int M,
                                             int N, float* /*scr*/ = nullptr) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + j * 32 + (lane & 31);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + mfma_row(r, lane);
        if (row >= M) continue;
        const int n = row / S2, s = row - n * S2;
        float v = acc[j][r] + p[row] * datt[(long)n * N + col];
        if (mask1) v = mask1[(long)row * N + col] ? v * scale : 0.f;
        unsafeAtomicAdd(dpre + ((long)(n / R) * S2 + s) * N + col, v);
      }
    }
  }
};

// ---------------------------------------------------------------------------
// SAN attention head.  Forward = two kernels:
//   img_att_score_kernel  (one 16-wave workgroup per QA round n):
//       score[s] = <iqc[n,s,:], wa> + ba ; p[n,:] = softmax_s(score)
//   img_att_wsum_kernel   (grid N x H/128): u1[n,h] = u0[n,h] + sum_s p[n,s]*img_tr[n,s,h]
// The first version ran everything as ONE 256-thread workgroup per round (200 workgroups, each thread
// walking the 196 regions serially): 160 us alone, ~0.6 ms under the option LSTM.  Both kernels now keep
// >= 8 independent 16-byte loads per lane in flight and spread a round over several workgroups.
// ---------------------------------------------------------------------------
#define IMG_ATT_WAVES 16
__device__ __forceinline__ float block_reduce(float v, float* red, int wave, int lane, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();  // red may still be read from a previous reduction
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < IMG_ATT_WAVES; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ void __launch_bounds__(IMG_ATT_WAVES * 64)
img_att_score_kernel(const float* __restrict__ iqc, const float* __restrict__ wa, const float* __restrict__ ba,
                     float* __restrict__ p_out, int S2, int Kc) {
  extern __shared__ float sc[];  // [S2]
  __shared__ float red[IMG_ATT_WAVES];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* iq = iqc + (long)n * S2 * Kc;
  const float b0 = ba[0];
  for (int s = wave; s < S2; s += IMG_ATT_WAVES) {
    float a = 0.f;
    for (int k = lane * 4; k < Kc; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(iq + (long)s * Kc + k);
      const float4 w = *reinterpret_cast<const float4*>(wa + k);
      a += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
    }
    a = wave_sum(a);
    if (lane == 0) sc[s] = a + b0;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int s = tid; s < S2; s += IMG_ATT_WAVES * 64) mx = fmaxf(mx, sc[s]);
  mx = block_reduce(mx, red, wave, lane, true);
  float sum = 0.f;
  for (int s = tid; s < S2; s += IMG_ATT_WAVES * 64) {
    const float e = expf(sc[s] - mx);
    sc[s] = e;
    sum += e;
  }
  sum = block_reduce(sum, red, wave, lane, false);
  const float inv = 1.f / sum;
  for (int s = tid; s < S2; s += IMG_ATT_WAVES * 64) p_out[(long)n * S2 + s] = sc[s] * inv;
}

// 256 threads = 8 region groups x 32 lanes; lane l owns 4 consecutive channels of a 128-channel tile.
__global__ void __launch_bounds__(256)
img_att_wsum_kernel(const float* __restrict__ pre, const uint8_t* __restrict__ mask1,
                    const float* __restrict__ p, const float* __restrict__ u0, float* __restrict__ u1, int S2,
                    int R, int H, float scale) {
  extern __shared__ float sh[];  // p [S2] | partial sums [8][128]
  float* ps = sh;
  float* part = sh + ((S2 + 3) & ~3);
  const int n = blockIdx.x, tid = threadIdx.x, g = tid >> 5, l = tid & 31;
  const int h = blockIdx.y * 128 + l * 4;
  for (int s = tid; s < S2; s += 256) ps[s] = p[(long)n * S2 + s];
  __syncthreads();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (h < H) {
    const float* pr = pre + (long)(n / R) * S2 * H + h;
    const uint8_t* m1 = mask1 ? mask1 + (long)n * S2 * H + h : nullptr;
#pragma unroll 4
    for (int s = g; s < S2; s += 8) {
      float4 v = *reinterpret_cast<const float4*>(pr + (long)s * H);
      if (m1) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(m1 + (long)s * H);
        v.x = (m & 0xffu) ? v.x * scale : 0.f;
        v.y = (m & 0xff00u) ? v.y * scale : 0.f;
        v.z = (m & 0xff0000u) ? v.z * scale : 0.f;
        v.w = (m & 0xff000000u) ? v.w * scale : 0.f;
      }
      const float w = ps[s];
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
  }
  *reinterpret_cast<float4*>(part + g * 128 + l * 4) = acc;
  __syncthreads();
  if (g == 0 && h < H) {
    float4 t = *reinterpret_cast<const float4*>(u0 + (long)n * H + h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(part + q * 128 + l * 4);
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    *reinterpret_cast<float4*>(u1 + (long)n * H + h) = t;
  }
}

// ---------------------------------------------------------------------------
// Backward of the attention head + of the tanh/dropout epilogue of img_common.  Per QA round n:
//   dp[s] = <datt[n,:], img_tr[n,s,:]> ; dscore = p*(dp - <p,dp>)            (img_att_dscore_kernel)
//   dwa += sum_s dscore[s]*iqc[n,s,:] ; dba += sum_s dscore[s]
//   dz[n,s,k] = dscore[s]*wa[k]*scale2*mask2*(1 - tanh^2)   (written over iqc in place)
//   dqc[n,k] = sum_s dz[n,s,k]                                               (img_att_dz_kernel)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(IMG_ATT_WAVES * 64)
img_att_dscore_kernel(const float* __restrict__ pre, const uint8_t* __restrict__ mask1,
                      const float* __restrict__ p, const float* __restrict__ datt, float* __restrict__ dscore,
                      float* __restrict__ dba, int S2, int R, int H, float scale) {
  extern __shared__ float sc[];  // dp [S2]
  __shared__ float red[IMG_ATT_WAVES];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pr = pre + (long)(n / R) * S2 * H;
  const uint8_t* m1 = mask1 ? mask1 + (long)n * S2 * H : nullptr;
  const float* da = datt + (long)n * H;
  for (int s = wave; s < S2; s += IMG_ATT_WAVES) {
    float a = 0.f;
    for (int h = lane * 4; h < H; h += 256) {
      float4 v = *reinterpret_cast<const float4*>(pr + (long)s * H + h);
      if (m1) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(m1 + (long)s * H + h);
        v.x = (m & 0xffu) ? v.x * scale : 0.f;
        v.y = (m & 0xff00u) ? v.y * scale : 0.f;
        v.z = (m & 0xff0000u) ? v.z * scale : 0.f;
        v.w = (m & 0xff000000u) ? v.w * scale : 0.f;
      }
      const float4 d = *reinterpret_cast<const float4*>(da + h);
      a += d.x * v.x + d.y * v.y + d.z * v.z + d.w * v.w;
    }
    a = wave_sum(a);
    if (lane == 0) sc[s] = a;
  }
  __syncthreads();
  float dot = 0.f;
  for (int s = tid; s < S2; s += IMG_ATT_WAVES * 64) dot += p[(long)n * S2 + s] * sc[s];
  dot = block_reduce(dot, red, wave, lane, false);
  float dsum = 0.f;
  for (int s = tid; s < S2; s += IMG_ATT_WAVES * 64) {
    const float ds = p[(long)n * S2 + s] * (sc[s] - dot);
    dscore[(long)n * S2 + s] = ds;
    dsum += ds;
  }
  dsum = block_reduce(dsum, red, wave, lane, false);
  if (tid == 0) unsafeAtomicAdd(dba, dsum);
}

__global__ void __launch_bounds__(256)
img_att_dz_kernel(float* __restrict__ iqc, const float* __restrict__ wa, const uint8_t* __restrict__ mask2,
                  const float* __restrict__ dscore, float* __restrict__ dwa, float* __restrict__ dqc, int S2,
                  int Kc, float scale) {
  extern __shared__ float sh[];  // dscore [S2] | partial sums [2][8][128]
  float* dsv = sh;
  float* part = sh + ((S2 + 3) & ~3);
  const int n = blockIdx.x, tid = threadIdx.x, g = tid >> 5, l = tid & 31;
  const int k = blockIdx.y * 128 + l * 4;
  for (int s = tid; s < S2; s += 256) dsv[s] = dscore[(long)n * S2 + s];
  __syncthreads();
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), aq = aw;
  if (k < Kc) {
    float* iq = iqc + (long)n * S2 * Kc + k;
    const uint8_t* m2 = mask2 ? mask2 + (long)n * S2 * Kc + k : nullptr;
    const float inv_scale = m2 ? 1.f / scale : 1.f;
    const float sc2 = m2 ? scale : 1.f;
    float4 w = *reinterpret_cast<const float4*>(wa + k);
    w.x *= sc2; w.y *= sc2; w.z *= sc2; w.w *= sc2;
#pragma unroll 4
    for (int s = g; s < S2; s += 8) {
      const float4 y = *reinterpret_cast<const float4*>(iq + (long)s * Kc);
      const uint32_t m = m2 ? *reinterpret_cast<const uint32_t*>(m2 + (long)s * Kc) : 0xffffffffu;
      const float ds = dsv[s];
      aw.x += ds * y.x; aw.y += ds * y.y; aw.z += ds * y.z; aw.w += ds * y.w;
      float4 dz;
      { const float t = y.x * inv_scale; dz.x = (m & 0xffu) ? ds * w.x * (1.f - t * t) : 0.f; }
      { const float t = y.y * inv_scale; dz.y = (m & 0xff00u) ? ds * w.y * (1.f - t * t) : 0.f; }
      { const float t = y.z * inv_scale; dz.z = (m & 0xff0000u) ? ds * w.z * (1.f - t * t) : 0.f; }
      { const float t = y.w * inv_scale; dz.w = (m & 0xff000000u) ? ds * w.w * (1.f - t * t) : 0.f; }
      *reinterpret_cast<float4*>(iq + (long)s * Kc) = dz;
      aq.x += dz.x; aq.y += dz.y; aq.z += dz.z; aq.w += dz.w;
    }
  }
  *reinterpret_cast<float4*>(part + g * 128 + l * 4) = aw;
  *reinterpret_cast<float4*>(part + 1024 + g * 128 + l * 4) = aq;
  __syncthreads();
  if (g == 0 && k < Kc) {
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f), tq = tw;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(part + q * 128 + l * 4);
      const float4 b = *reinterpret_cast<const float4*>(part + 1024 + q * 128 + l * 4);
      tw.x += a.x; tw.y += a.y; tw.z += a.z; tw.w += a.w;
      tq.x += b.x; tq.y += b.y; tq.z += b.z; tq.w += b.w;
    }
    *reinterpret_cast<float4*>(dqc + (long)n * Kc + k) = tq;
    unsafeAtomicAdd(dwa + k, tw.x);
    unsafeAtomicAdd(dwa + k + 1, tw.y);
    unsafeAtomicAdd(dwa + k + 2, tw.z);
    unsafeAtomicAdd(dwa + k + 3, tw.w);
  }
}

using CfgBigDB = GemmCfg<4, 1, 4, 32>;
using CfgBig = GemmCfg<4, 1, 4, 16, 0, 3>;

extern "C" {

int vd_mn_attention_forward(const float* Q, const float* Hm, const uint8_t* mask, float* P, float* hAtt, int B,
                            int R, int H, void* stream) {
  VD_CHECK_ARG(Q && Hm && mask && P && hAtt && B >= 0 && R >= 1 && R <= MN_MAX_R && H > 0 && H % 4 == 0,
               "vd_mn_attention_forward: bad args (R=%d must be <= %d, H %% 4 == 0)", R, MN_MAX_R);
  if (B == 0) return VD_OK;
  const size_t lds = (size_t)2 * R * H * sizeof(float);
  VD_CHECK_ARG(lds <= 64 * 1024, "vd_mn_attention_forward: R*H = %d does not fit the LDS staging", R * H);
  hipLaunchKernelGGL(mn_att_fwd_kernel, dim3(B), dim3(MN_THREADS), lds, (hipStream_t)stream, Q, Hm, mask, P, hAtt,
                     R, H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_mn_attention_backward(const float* Q, const float* Hm, const float* P, const float* dhAtt, float* dQ,
                             float* dHm, int B, int R, int H, void* stream) {
  VD_CHECK_ARG(Q && Hm && P && dhAtt && dQ && dHm && B >= 0 && R >= 1 && R <= MN_MAX_R && H > 0 && H % 4 == 0,
               "vd_mn_attention_backward: bad args");
  if (B == 0) return VD_OK;
  const size_t lds = (size_t)3 * R * H * sizeof(float);
  VD_CHECK_ARG(lds <= 64 * 1024, "vd_mn_attention_backward: R*H = %d does not fit the LDS staging", R * H);
  hipLaunchKernelGGL(mn_att_bwd_kernel, dim3(B), dim3(MN_THREADS), lds, (hipStream_t)stream, Q, Hm, P, dhAtt, dQ,
                     dHm, R, H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_hrea_attention_forward(const float* sq, const float* sh, const float* Hm, float* P, float* att, int B, int R,
                              int H, void* stream) {
  VD_CHECK_ARG(sq && sh && Hm && P && att && B >= 0 && R >= 1 && R <= MN_MAX_R && H > 0,
               "vd_hrea_attention_forward: bad args (R=%d must be <= %d)", R, MN_MAX_R);
  if (B == 0) return VD_OK;
  hipLaunchKernelGGL(hrea_att_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sq, sh, Hm, P, att, R, H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_hrea_attention_backward(const float* Hm, const float* P, const float* datt, float* dsq, float* dsh,
                               float* dHm, int B, int R, int H, void* stream) {
  VD_CHECK_ARG(Hm && P && datt && dsq && dsh && dHm && B >= 0 && R >= 1 && R <= MN_MAX_R && H > 0,
               "vd_hrea_attention_backward: bad args");
  if (B == 0) return VD_OK;
  hipLaunchKernelGGL(hrea_att_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Hm, P, datt, dsq, dsh, dHm,
                     R, H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_rowdot_forward(const float* x, const float* w, const float* bias, float* out, int N, int H, void* stream) {
  VD_CHECK_ARG(x && w && bias && out && N >= 0 && H > 0, "vd_rowdot_forward: bad args");
  if (N == 0) return VD_OK;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(vd_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, N,
                     H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_rowdot_backward(const float* x, const float* w, const float* dout, float* dw, float* db, float* dx, int N,
                       int H, void* stream) {
  VD_CHECK_ARG(x && w && dout && dw && db && dx && N >= 0 && H > 0, "vd_rowdot_backward: bad args");
  if (N == 0) return VD_OK;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(vd_cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, x, w, dout, dw,
                     db, dx, N, H);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// iqc[(n,s), :] = drop2(tanh(img_tr[(n,s), :] * Wc^T + bc + qc[n, :]))
int vd_img_common_forward(const float* pre, const uint8_t* mask1, const float* Wc, const float* bc,
                          const float* qc, const uint8_t* mask2, float* iqc, int N, int R, int S2, int H, int Kc,
                          float scale, void* stream) {
  VD_CHECK_ARG(pre && Wc && bc && qc && iqc && N >= 0 && R >= 1 && S2 >= 1 && H % 4 == 0 && Kc % 4 == 0,
               "vd_img_common_forward: bad args");
  SrcImgDropRow a{pre, mask1, H, S2, R, scale};
  SrcRow b{Wc, H};
  EpiImgCommon e{iqc, bc, qc, mask2, S2, scale};
  return launch_gemm<CfgBig>(N * S2, Kc, H, 1, a, b, e, (hipStream_t)stream);
}

int vd_img_att_forward(const float* iqc, const float* wa, const float* ba, const float* pre,
                       const uint8_t* mask1, const float* u0, float* p, float* u1, int N, int R, int S2, int H,
                       int Kc, float scale, void* stream) {
  VD_CHECK_ARG(iqc && wa && ba && pre && u0 && p && u1 && N >= 0 && Kc % 4 == 0 && H % 4 == 0,
               "vd_img_att_forward: bad args");
  if (N == 0) return VD_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(img_att_score_kernel, dim3(N), dim3(IMG_ATT_WAVES * 64), S2 * sizeof(float), s, iqc, wa, ba,
                     p, S2, Kc);
  VD_LAUNCH_CHECK();
  const size_t lds = (((S2 + 3) & ~3) + 8 * 128) * sizeof(float);
  hipLaunchKernelGGL(img_att_wsum_kernel, dim3(N, vd_cdiv(H, 128)), dim3(256), lds, s, pre, mask1, p, u0, u1, S2,
                     R, H, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// iqc is overwritten by dz.  dwa/dba are accumulated, dqc is written.  work: [N x S2] floats (dscore).
int vd_img_att_backward(float* iqc_dz, const float* wa, const float* pre, const uint8_t* mask1,
                        const uint8_t* mask2, const float* p, const float* datt, float* dwa, float* dba,
                        float* dqc, float* work, int N, int R, int S2, int H, int Kc, float scale, void* stream) {
  VD_CHECK_ARG(iqc_dz && wa && pre && p && datt && dwa && dba && dqc && work && N >= 0 && Kc % 4 == 0 &&
                   H % 4 == 0,
               "vd_img_att_backward: bad args");
  if (N == 0) return VD_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(img_att_dscore_kernel, dim3(N), dim3(IMG_ATT_WAVES * 64), S2 * sizeof(float), s, pre, mask1,
                     p, datt, work, dba, S2, R, H, scale);
  VD_LAUNCH_CHECK();
  const size_t lds = (((S2 + 3) & ~3) + 2 * 8 * 128) * sizeof(float);
  hipLaunchKernelGGL(img_att_dz_kernel, dim3(N, vd_cdiv(Kc, 128)), dim3(256), lds, s, iqc_dz, wa, mask2, work,
                     dwa, dqc, S2, Kc, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// dpre[(b,s), :] += sum_{rounds} (dz[(n,s), :] * Wc + p[n,s]*datt[n, :]) * mask1*scale
int vd_img_tr_backward(const float* dz, const float* Wc, const float* p, const float* datt,
                       const uint8_t* mask1, float* dpre, int N, int R, int S2, int H, int Kc, float scale,
                       void* stream) {
  VD_CHECK_ARG(dz && Wc && p && datt && dpre && N >= 0 && H % 4 == 0 && Kc % 4 == 0,
               "vd_img_tr_backward: bad args");
  SrcRow a{dz, Kc};
  SrcK b{Wc, H};  // B[k][h] = Wc[k][h]
  EpiImgTrBwd e{dpre, p, datt, mask1, S2, R, scale};
  return launch_gemm<CfgBig>(N * S2, H, Kc, 1, a, b, e, (hipStream_t)stream);
}

// dWc[Kc x H] += dz^T * img_tr
int vd_img_common_wgrad(const float* dz, const float* pre, const uint8_t* mask1, float* dWc, int N, int R,
                        int S2, int H, int Kc, float scale, void* stream) {
  VD_CHECK_ARG(dz && pre && dWc && N >= 0 && H % 4 == 0 && Kc % 4 == 0, "vd_img_common_wgrad: bad args");
  if (N == 0) return VD_OK;
  SrcK a{dz, Kc};
  SrcImgDropK b{pre, mask1, H, S2, R, scale};
  EpiAtomic<4> e{dWc, H};
  const int K = N * S2;
  const long tiles = (long)vd_cdiv(Kc, CfgBig::BM) * vd_cdiv(H, CfgBig::BN);
  long splits = vd_cdiv(1024, tiles);
  const long max_splits = vd_cdiv(K, 4 * CfgBig::BK);
  if (splits > max_splits) splits = max_splits;
  return launch_gemm<CfgBig>(Kc, H, K, (int)splits, a, b, e, (hipStream_t)stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------------
// The three dense [N*S2 x 512 x 512] products of the image attention on the EXACT SPLIT (split_core.h), for a split9 pass of the model-level
// runtime (every fp32 GEMM with K >= 512 that shares the chip with the option recurrence costs 9/16 of the matrix-pipe
// time this way).  The split kernels move their operands by LDS-DMA, which cannot apply a dropout mask on the fly, so the per-round image
// tensor img_tr = dropout(gather(pre)) IS materialised once per step (`xdrop` [N*S2 x H], 80 MB at the headline shape, written by one
// streaming kernel and read by the forward product and the weight gradient); the weights go through three bf16 planes in the stream's
// scratch.  Same epilogues as the fp32 kernels (EpiImgCommon, EpiImgTrBwd), same results to fp32 rounding; no LDS bank conflicts (the
// register-staged BK = 16 kernels they replace carried 18-31 %, profiles/r05_lds_conflicts_bench.txt).
// ---------------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) img_drop_gather_kernel(const float* __restrict__ pre, const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                              long n4, int H4, int S2, int R, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long row = i / H4;
  const int k4 = (int)(i - row * H4);
  const long n = row / S2, sidx = row - n * S2;
  float4 v = reinterpret_cast<const float4*>(pre)[((n / R) * S2 + sidx) * H4 + k4];
  if (mask) {
    const uint32_t mk = reinterpret_cast<const uint32_t*>(mask)[i];
    v.x = (mk & 0xff) ? v.x * scale : 0.f;
    v.y = (mk & 0xff00) ? v.y * scale : 0.f;
    v.z = (mk & 0xff0000) ? v.z * scale : 0.f;
    v.w = (mk & 0xff000000u) ? v.w * scale : 0.f;
  }
  reinterpret_cast<float4*>(out)[i] = v;
}
__global__ void __launch_bounds__(256) transpose32_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];      // dst[c][r] = src[r][c]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    tile[ty + i * 8][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = tile[tx][ty + i * 8];
  }
}

static bool img_split_ok(int flags, long rows, int H, int Kc) {
  return (flags & VD_FLAG_SPLIT9) && rows >= 128 && H % 16 == 0 && Kc % 16 == 0 && H % 4 == 0 && rows * (long)(H > Kc ? H : Kc) * 4 < (1L << 32);
}

// xdrop[(n, s), :] = dropout1(pre[(n / R, s), :]) -- once per step, shared by every attention hop's forward product and weight gradient
int vd_img_drop_gather(const float* pre, const uint8_t* mask1, float* xdrop, int N, int R, int S2, int H, float scale, hipStream_t stream) {
  VD_CHECK_ARG(pre && xdrop && N >= 0 && R >= 1 && S2 >= 1 && H % 4 == 0, "vd_img_drop_gather: bad args");
  const long n4 = (long)N * S2 * (H / 4);
  if (n4 == 0) return VD_OK;
  hipLaunchKernelGGL(img_drop_gather_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, pre, mask1, xdrop, n4, H / 4, S2, R, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_img_common_forward_p(const float* pre, const uint8_t* mask1, const float* xdrop, const float* Wc, const float* bc, const float* qc,
                            const uint8_t* mask2, float* iqc, int N, int R, int S2, int H, int Kc, float scale, int flags, hipStream_t stream) {
  if (!xdrop || !img_split_ok(flags, (long)N * S2, H, Kc))
    return vd_img_common_forward(pre, mask1, Wc, bc, qc, mask2, iqc, N, R, S2, H, Kc, scale, stream);
  VdStreamScratch scr;
  if (int rc = vd_stream_scratch(stream, (size_t)Kc * H * 6, 0, &scr)) return rc;
  vd_bf16_bits* W3 = reinterpret_cast<vd_bf16_bits*>(scr.wht);
  if (int rc = weights_to_bf16x3(Wc, W3, (long)Kc * H, stream)) return rc;                // Wc [Kc x H]: rows are k-contiguous as they lie
  EpiImgCommon e{iqc, bc, qc, mask2, S2, scale};
  return launch_gemm_split<9>(N * S2, Kc, H, xdrop, (long)H, W3, (long)H, (long)Kc * H, e, stream);
}

int vd_img_tr_backward_p(const float* dz, const float* Wc, const float* p, const float* datt, const uint8_t* mask1, float* dpre, int N, int R,
                         int S2, int H, int Kc, float scale, int flags, hipStream_t stream) {
  if (!img_split_ok(flags, (long)N * S2, H, Kc)) return vd_img_tr_backward(dz, Wc, p, datt, mask1, dpre, N, R, S2, H, Kc, scale, stream);
  VdStreamScratch scr;
  if (int rc = vd_stream_scratch(stream, (size_t)Kc * H * 10, 0, &scr)) return rc;
  float* WcT = scr.wht;                                                                    // [H x Kc]: B^T of dz * Wc, rows k-contiguous
  vd_bf16_bits* W3 = reinterpret_cast<vd_bf16_bits*>(scr.wht + (size_t)Kc * H);
  hipLaunchKernelGGL(transpose32_kernel, dim3(vd_cdiv(H, 32), vd_cdiv(Kc, 32)), dim3(256), 0, stream, Wc, WcT, Kc, H);
  VD_LAUNCH_CHECK();
  if (int rc = weights_to_bf16x3(WcT, W3, (long)Kc * H, stream)) return rc;
  EpiImgTrBwd e{dpre, p, datt, mask1, S2, R, scale};
  return launch_gemm_split<9>(N * S2, H, Kc, dz, (long)Kc, W3, (long)Kc, (long)Kc * H, e, stream);
}

int vd_img_common_wgrad_p(const float* dz, const float* pre, const uint8_t* mask1, const float* xdrop, float* dWc, int N, int R, int S2, int H,
                          int Kc, float scale, int flags, hipStream_t stream) {
  const long K = (long)N * S2;
  if (!xdrop || !(flags & VD_FLAG_SPLIT9) || Kc % SplitTnCfg::BM != 0 || H % SplitTnCfg::BN != 0 || K < 1024)
    return vd_img_common_wgrad(dz, pre, mask1, dWc, N, R, S2, H, Kc, scale, stream);
  const int K1 = (int)(K & ~15L);
  if (int rc = launch_gemm_split_tn<9>(Kc, H, K1, dz, (long)Kc, xdrop, (long)H, dWc, (long)H, stream)) return rc;
  if (K1 == K) return VD_OK;
  SrcK a{dz + (long)K1 * Kc, Kc}, b{xdrop + (long)K1 * H, H};                              // the last < 16 rows on the fp32 MFMA
  EpiAtomic<4> e{dWc, H};
  return launch_gemm<CfgBig>(Kc, H, (int)(K - K1), 1, a, b, e, stream);
}
