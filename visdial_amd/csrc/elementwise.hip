// HBM-bound helper kernels: embedding gather / gradient scatter, dropout masks, small
// pointwise glue, token counting-sort + segmented row sum (option-table gradient), and the
// fused clamp + Adam update.  All are plain coalesced float4 / wave-per-row kernels.
#include <stdlib.h>

#include "common.h"

// ------------------------------------------------------------------ dropout masks
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (uint32_t)x;
}

// mask[i] = 1 with probability (1-p).  Counter-based: (seed, i) -> bit, replayable.
__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, long n, uint64_t seed, float p) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const uint32_t thr = (uint32_t)(p * 16777216.0f);
  uint8_t m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t h = mix32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(i4 + k));
    m[k] = ((h >> 8) >= thr) ? 1 : 0;
  }
  if (i4 + 3 < n) {
    *reinterpret_cast<uint32_t*>(mask + i4) = m[0] | (m[1] << 8) | (m[2] << 16) | ((uint32_t)m[3] << 24);
  } else {
    for (int k = 0; k < 4 && i4 + k < n; ++k) mask[i4 + k] = m[k];
  }
}

// y = x * mask * scale   (nn.Dropout forward in training mode and its backward)
__global__ void dropout_apply_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                     float* __restrict__ y, long n, float scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = mask[i] ? x[i] * scale : 0.f;
}

// dx = dy * (1 - y^2)
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float t = y[i];
    dx[i] = dy[i] * (1.f - t * t);
  }
}

// c = alpha*a + beta*b
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                             long n, float alpha, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// ------------------------------------------------------------------ embedding
// out[r, :] = Emb[tok[r], :] * (mask ? mask*scale : 1).  One thread per float4.
// (nn.LookupTableMaskZero: token 0 -> table row 0, which the host keeps at zero.)
__global__ void embed_gather_kernel(const float* __restrict__ emb, const int* __restrict__ tok,
                                    const uint8_t* __restrict__ mask, float* __restrict__ out, long rows, int E,
                                    float scale) {
  const int e4 = E >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * e4) return;
  const long r = idx / e4;
  const int q = (int)(idx - r * e4);
  float4 v = *reinterpret_cast<const float4*>(emb + (long)tok[r] * E + q * 4);
  if (mask) {
    const uint32_t m = *reinterpret_cast<const uint32_t*>(mask + r * E + q * 4);
    v.x = (m & 0xff) ? v.x * scale : 0.f;
    v.y = (m & 0xff00) ? v.y * scale : 0.f;
    v.z = (m & 0xff0000) ? v.z * scale : 0.f;
    v.w = (m & 0xff000000u) ? v.w * scale : 0.f;
  }
  *reinterpret_cast<float4*>(out + r * E + q * 4) = v;
}

// dEmb[tok[r], :] += dX[r, :] * mask * scale
__global__ void embed_scatter_kernel(float* __restrict__ demb, const int* __restrict__ tok,
                                     const uint8_t* __restrict__ mask, const float* __restrict__ dx, long rows,
                                     int E, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * E) return;
  const long r = idx / E;
  const int e = (int)(idx - r * E);
  float v = dx[idx];
  if (mask) v = mask[idx] ? v * scale : 0.f;
  if (v != 0.f) unsafeAtomicAdd(demb + (long)tok[r] * E + e, v);
}

// ------------------------------------------------------------------ MaskTime (model_utils/MaskTime.lua:12-40)
// forward: out[t, n, :] = tok[t, n] != 0 ? feat[n, :] : 0      (replicate over time, zero at pads)
__global__ void mask_time_fwd_kernel(const float* __restrict__ feat, const int* __restrict__ tok,
                                     float* __restrict__ out, int T, int N, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * N * D;
  if (idx >= total) return;
  const long tn = idx / D;
  const int d = (int)(idx - tn * D);
  const int n = (int)(tn % N);
  out[idx] = tok[tn] != 0 ? feat[(long)n * D + d] : 0.f;
}
// backward: dfeat[n, :] = sum_t (tok[t, n] != 0) * dout[t, n, :]
__global__ void mask_time_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ tok,
                                     float* __restrict__ dfeat, int T, int N, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * D) return;
  const int n = (int)(idx / D);
  const int d = (int)(idx - (long)n * D);
  float s = 0.f;
  for (int t = 0; t < T; ++t)
    if (tok[(long)t * N + n] != 0) s += dout[((long)t * N + n) * D + d];
  dfeat[idx] = s;
}

// zero rows [nact[t], N) of every time slice of a [T x N x ld] buffer (pad rows of length-sorted
// right-aligned sequences: they are skipped by the recurrence kernels and must read as zeros)
__global__ void zero_inactive_rows_kernel(float* __restrict__ buf, long tstride, long ld, int ncols,
                                          const int* __restrict__ nact, int N) {
  const int t = blockIdx.y;
  const int first = nact[t];
  const long total = (long)(N - first) * (ncols >> 2);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / (ncols >> 2);
    const int q = (int)(idx - r * (ncols >> 2));
    *reinterpret_cast<float4*>(buf + t * tstride + (first + r) * ld + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// the same for up to 6 buffers of one [T x N] row layout in ONE launch (the step runtime's encoder stacks: gates1, h1, c1, h2, c2,
// gates2 -- it used to memset five of them completely, 55 % of those bytes only to be overwritten by the recurrence)
__global__ void zero_inactive_multi_kernel(VdZeroSet z, const int* __restrict__ nact, int N) {
  const int t = blockIdx.y;
  const int first = nact[t];
  const long per_row = z.quads_per_row;                       // float4s per row over all buffers
  const long total = (long)(N - first) * per_row;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / per_row;
    int q = (int)(idx - r * per_row);
    int b = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (b == i && i + 1 < z.n && q >= (z.ncols[i] >> 2)) { q -= z.ncols[i] >> 2; b = i + 1; }
    *reinterpret_cast<float4*>(z.buf[b] + ((long)t * N + first + r) * z.ncols[b] + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ------------------------------------------------------------------ token counting sort
// Wave-aggregated atomics: lanes holding the same token as the wave's first active lane are
// counted with one atomic (the pad token dominates option batches: ~50% of all ids are 0).
__device__ __forceinline__ int wave_agg_atomic_inc(int* base, int tok, bool active) {
  // returns this lane's slot (old counter value + rank among equal-token lanes)
  int result = 0;
  unsigned long long todo = __ballot(active);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int ltok = __shfl(tok, leader, 64);
    const unsigned long long same = __ballot(active && tok == ltok) & todo;
    int basev = 0;
    if (lane == leader) basev = atomicAdd(base + ltok, __popcll(same));
    basev = __shfl(basev, leader, 64);
    if ((same >> lane) & 1ULL) result = basev + __popcll(same & ((1ULL << lane) - 1ULL));
    todo &= ~same;
  }
  return result;
}

__global__ void tok_count_kernel(const int* __restrict__ tok, long n, int* __restrict__ count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  wave_agg_atomic_inc(count, active ? tok[i] : 0, active);
}

// exclusive scan of count[0..V) into offset[0..V]; cursor := offset.  Single block.
__global__ void tok_scan_kernel(const int* __restrict__ count, int V, int* __restrict__ offset,
                                int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int per = (V + nt - 1) / nt;
  const int b = tid * per, e = min(V, b + per);
  int s = 0;
  for (int i = b; i < e; ++i) s += count[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < nt; ++i) {
      const int t = part[i];
      part[i] = run;
      run += t;
    }
    offset[V] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int i = b; i < e; ++i) {
    offset[i] = run;
    cursor[i] = run;
    run += count[i];
  }
}

__global__ void tok_fill_kernel(const int* __restrict__ tok, long n, int* __restrict__ cursor,
                                int* __restrict__ perm) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  const int p = wave_agg_atomic_inc(cursor, active ? tok[i] : 0, active);
  if (active) perm[p] = (int)i;
}

// out[tok, :] += sum over rows r with tok[r] == tok of X[r, :].  perm lists the rows sorted by
// token.  Each block walks CHUNK consecutive sorted rows, keeps the running sum of the current
// token in registers (one float4 per thread per 1024 columns) and flushes with atomics when the
// token changes, so a long run (the pad token) is split over many blocks.
// X16 = true: X holds bf16 rows (the compact da of a bf16 pass; ldx in bf16 elements), sums still fp32
template <int CHUNK, bool X16 = false>
__global__ void __launch_bounds__(256)
segment_rowsum_kernel(const float* __restrict__ X, long ldx, const int* __restrict__ tok,
                      const int* __restrict__ perm, long n, int ncol, float* __restrict__ out, long ldo) {
  __shared__ int srow[CHUNK], stok[CHUNK];
  const long p0 = (long)blockIdx.x * CHUNK;
  if (p0 >= n) return;
  const int cnt = (int)min((long)CHUNK, n - p0);
  // stage the chunk's row ids / tokens once (removes two dependent loads from every row visit)
  if ((int)threadIdx.x < cnt) {
    const int r = perm[p0 + threadIdx.x];
    srow[threadIdx.x] = r;
    stok[threadIdx.x] = tok[r];
  }
  __syncthreads();
  if constexpr (X16) {
    // bf16 rows: TWO column groups of 4 per thread and row visit (c and c + 1024: the two 8-byte loads of a row are independent, so the
    // same four rows in flight carry twice the bytes; the flush keeps the fp32 variant's pattern -- lane-contiguous 16-byte groups.
    // 16-byte loads of 8 adjacent columns were measured too: 0.80 vs 0.47 ms, the stride-8 atomics of the flush cost more than the loads gain)
    const vd_bf16_bits* X16p = reinterpret_cast<const vd_bf16_bits*>(X);
    for (int c = threadIdx.x * 4; c < ncol; c += blockDim.x * 8) {
      const int c2 = c + blockDim.x * 4;
      const bool two = c2 < ncol;
      float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
      auto flush = [&](int token) {
        float* o = out + (long)token * ldo + c;
        unsafeAtomicAdd(o, acc0.x); unsafeAtomicAdd(o + 1, acc0.y); unsafeAtomicAdd(o + 2, acc0.z); unsafeAtomicAdd(o + 3, acc0.w);
        if (two) {
          float* o2 = out + (long)token * ldo + c2;
          unsafeAtomicAdd(o2, acc1.x); unsafeAtomicAdd(o2 + 1, acc1.y); unsafeAtomicAdd(o2 + 2, acc1.z); unsafeAtomicAdd(o2 + 3, acc1.w);
        }
        acc0 = acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
      };
      int cur = stok[0];
      int j = 0;
      while (j < cnt) {
        uint2 w0[4], w1[4];
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = min(j + u, cnt - 1);
          t[u] = stok[jj];
          const vd_bf16_bits* row = X16p + (long)srow[jj] * ldx;
          w0[u] = *reinterpret_cast<const uint2*>(row + c);
          w1[u] = two ? *reinterpret_cast<const uint2*>(row + c2) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + u >= cnt) break;
          if (t[u] != cur) {
            flush(cur);
            cur = t[u];
          }
          acc0.x += __uint_as_float(w0[u].x << 16); acc0.y += __uint_as_float(w0[u].x & 0xffff0000u);
          acc0.z += __uint_as_float(w0[u].y << 16); acc0.w += __uint_as_float(w0[u].y & 0xffff0000u);
          acc1.x += __uint_as_float(w1[u].x << 16); acc1.y += __uint_as_float(w1[u].x & 0xffff0000u);
          acc1.z += __uint_as_float(w1[u].y << 16); acc1.w += __uint_as_float(w1[u].y & 0xffff0000u);
        }
        j += 4;
      }
      flush(cur);
    }
    return;
  }
  for (int c = threadIdx.x * 4; c < ncol; c += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur = stok[0];
    int j = 0;
    while (j < cnt) {
      // rows are independent loads: keep 4 in flight
      float4 v[4];
      int t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = min(j + u, cnt - 1);
        t[u] = stok[jj];
        v[u] = *reinterpret_cast<const float4*>(X + (long)srow[jj] * ldx + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j + u >= cnt) break;
        if (t[u] != cur) {
          float* o = out + (long)cur * ldo + c;
          unsafeAtomicAdd(o, acc.x);
          unsafeAtomicAdd(o + 1, acc.y);
          unsafeAtomicAdd(o + 2, acc.z);
          unsafeAtomicAdd(o + 3, acc.w);
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          cur = t[u];
        }
        acc.x += v[u].x;
        acc.y += v[u].y;
        acc.z += v[u].z;
        acc.w += v[u].w;
      }
      j += 4;
    }
    float* o = out + (long)cur * ldo + c;
    unsafeAtomicAdd(o, acc.x);
    unsafeAtomicAdd(o + 1, acc.y);
    unsafeAtomicAdd(o + 2, acc.z);
    unsafeAtomicAdd(o + 3, acc.w);
  }
}

// ------------------------------------------------------------------ clamp + Adam
// model.lua:96-99 + model_utils/optim_updates.lua:62-91: g <- clamp(gscale*g, +-clip);
// m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; w <- w - step * m / (sqrt(v) + eps)
// (eps is added to the UNcorrected sqrt(v); step = lr*sqrt(1-b2^t)/(1-b1^t) from the host.)
__global__ void clamp_adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long n, float gscale, float clip, float b1, float b2,
                                  float eps, float step) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float4 W = *reinterpret_cast<float4*>(w + i4), G = *reinterpret_cast<float4*>(g + i4);
    float4 Mv = *reinterpret_cast<float4*>(m + i4), Vv = *reinterpret_cast<float4*>(v + i4);
    float* Wp = &W.x;
    float* Gp = &G.x;
    float* Mp = &Mv.x;
    float* Vp = &Vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = Gp[k] * gscale;
      gg = fminf(fmaxf(gg, -clip), clip);
      Gp[k] = gg;
      Mp[k] = b1 * Mp[k] + (1.f - b1) * gg;
      Vp[k] = b2 * Vp[k] + (1.f - b2) * gg * gg;
      Wp[k] -= step * Mp[k] / (sqrtf(Vp[k]) + eps);
    }
    *reinterpret_cast<float4*>(w + i4) = W;
    *reinterpret_cast<float4*>(g + i4) = G;
    *reinterpret_cast<float4*>(m + i4) = Mv;
    *reinterpret_cast<float4*>(v + i4) = Vv;
  } else {
    for (long i = i4; i < n; ++i) {
      float gg = g[i] * gscale;
      gg = fminf(fmaxf(gg, -clip), clip);
      g[i] = gg;
      m[i] = b1 * m[i] + (1.f - b1) * gg;
      v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
      w[i] -= step * m[i] / (sqrtf(v[i]) + eps);
    }
  }
}

static inline dim3 grid1d(long n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

extern "C" {

int vd_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, float p, void* stream) {
  VD_CHECK_ARG(mask && n >= 0 && p >= 0.f && p < 1.f && ((uintptr_t)mask & 3) == 0, "vd_dropout_mask: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, grid1d((n + 3) / 4, 256), dim3(256), 0, (hipStream_t)stream, mask,
                     (long)n, seed, p);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_dropout_apply(const float* x, const uint8_t* mask, float* y, int64_t n, float scale, void* stream) {
  VD_CHECK_ARG(x && mask && y && n >= 0, "vd_dropout_apply: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(dropout_apply_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, x, mask, y,
                     (long)n, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_tanh_backward(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  VD_CHECK_ARG(dy && y && dx && n >= 0, "vd_tanh_backward: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(tanh_bwd_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long)n);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_axpby(const float* a, const float* b, float* c, int64_t n, float alpha, float beta, void* stream) {
  VD_CHECK_ARG(a && c && n >= 0, "vd_axpby: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(axpby_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, a, b, c, (long)n, alpha,
                     beta);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_mask_time_forward(const float* feat, const int32_t* tok, float* out, int T, int N, int D, void* stream) {
  VD_CHECK_ARG(feat && tok && out && T >= 0 && N >= 0 && D > 0, "vd_mask_time_forward: bad args");
  const long total = (long)T * N * D;
  if (total == 0) return VD_OK;
  hipLaunchKernelGGL(mask_time_fwd_kernel, grid1d(total, 256), dim3(256), 0, (hipStream_t)stream, feat, tok, out,
                     T, N, D);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_mask_time_backward(const float* dout, const int32_t* tok, float* dfeat, int T, int N, int D,
                          void* stream) {
  VD_CHECK_ARG(dout && tok && dfeat && T >= 0 && N >= 0 && D > 0, "vd_mask_time_backward: bad args");
  if ((long)N * D == 0) return VD_OK;
  hipLaunchKernelGGL(mask_time_bwd_kernel, grid1d((long)N * D, 256), dim3(256), 0, (hipStream_t)stream, dout,
                     tok, dfeat, T, N, D);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_zero_inactive_rows(float* buf, int64_t tstride, int64_t ld, int ncols, const int32_t* nact_dev, int T,
                          int N, void* stream) {
  VD_CHECK_ARG(buf && nact_dev && T >= 0 && N >= 0 && ncols % 4 == 0 && ld % 4 == 0, "vd_zero_inactive_rows: bad args");
  if (T == 0 || N == 0) return VD_OK;
  hipLaunchKernelGGL(zero_inactive_rows_kernel, dim3(64, T), dim3(256), 0, (hipStream_t)stream, buf, (long)tstride,
                     (long)ld, ncols, nact_dev, N);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_embed_gather(const float* emb, const int32_t* tok, const uint8_t* mask, float* out, int64_t rows, int E,
                    float scale, void* stream);
}  // extern "C"
// (internal, common.h) dense [T x N x ncols[i]] buffers, rows [nact[t], N) of every time slice zeroed in one launch
int vd_zero_inactive_multi(const VdZeroSet& z, const int32_t* nact_dev, int T, int N, hipStream_t stream) {
  VD_CHECK_ARG(z.n >= 1 && z.n <= 6 && nact_dev && T >= 0 && N >= 0, "vd_zero_inactive_multi: bad args");
  if (T == 0 || N == 0) return VD_OK;
  VdZeroSet zz = z;
  zz.quads_per_row = 0;
  for (int i = 0; i < z.n; ++i) {
    VD_CHECK_ARG(z.buf[i] && z.ncols[i] % 4 == 0, "vd_zero_inactive_multi: buffer %d", i);
    zz.quads_per_row += z.ncols[i] >> 2;
  }
  hipLaunchKernelGGL(zero_inactive_multi_kernel, dim3(128, T), dim3(256), 0, stream, zz, nact_dev, N);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
extern "C" {
int vd_embed_gather(const float* emb, const int32_t* tok, const uint8_t* mask, float* out, int64_t rows, int E,
                    float scale, void* stream) {
  VD_CHECK_ARG(emb && tok && out && rows >= 0 && E > 0 && E % 4 == 0, "vd_embed_gather: bad args (E=%d)", E);
  if (rows == 0) return VD_OK;
  hipLaunchKernelGGL(embed_gather_kernel, grid1d(rows * (E / 4), 256), dim3(256), 0, (hipStream_t)stream, emb,
                     tok, mask, out, (long)rows, E, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_embed_scatter_acc(float* demb, const int32_t* tok, const uint8_t* mask, const float* dx, int64_t rows,
                         int E, float scale, void* stream) {
  VD_CHECK_ARG(demb && tok && dx && rows >= 0 && E > 0, "vd_embed_scatter_acc: bad args");
  if (rows == 0) return VD_OK;
  hipLaunchKernelGGL(embed_scatter_kernel, grid1d(rows * E, 256), dim3(256), 0, (hipStream_t)stream, demb, tok,
                     mask, dx, (long)rows, E, scale);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// counting sort of n token ids in [0, V): perm = row indices ordered by token.
// work = int32[2*V + 1] scratch (count, cursor); offset = int32[V+1].
int vd_token_sort(const int32_t* tok, int64_t n, int V, int32_t* offset, int32_t* work, int32_t* perm,
                  void* stream) {
  VD_CHECK_ARG(tok && offset && work && perm && n >= 0 && V > 0, "vd_token_sort: bad args");
  hipStream_t s = (hipStream_t)stream;
  int* count = work;
  int* cursor = work + V;
  VD_HIP(hipMemsetAsync(count, 0, sizeof(int) * V, s));
  if (n > 0) {
    hipLaunchKernelGGL(tok_count_kernel, grid1d(n, 256), dim3(256), 0, s, tok, (long)n, count);
    VD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(tok_scan_kernel, dim3(1), dim3(1024), 0, s, count, V, offset, cursor);
  VD_LAUNCH_CHECK();
  if (n > 0) {
    hipLaunchKernelGGL(tok_fill_kernel, grid1d(n, 256), dim3(256), 0, s, tok, (long)n, cursor, perm);
    VD_LAUNCH_CHECK();
  }
  return VD_OK;
}

// out[tok[r], 0:ncol] += X[r, 0:ncol] for all n rows (rows visited in perm order).
int vd_segment_rowsum_acc(const float* X, int64_t ldx, const int32_t* tok, const int32_t* perm, int64_t n,
                          int ncol, float* out, int64_t ldo, void* stream) {
  VD_CHECK_ARG(X && tok && perm && out && n >= 0 && ncol > 0 && ncol % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0,
               "vd_segment_rowsum_acc: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(segment_rowsum_kernel<256>, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, X, (long)ldx, tok, perm, (long)n, ncol,
                     out, (long)ldo);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

int vd_clamp_adam(float* w, float* g, float* m, float* v, int64_t n, float gscale, float clip, float beta1,
                  float beta2, float eps, float step, void* stream) {
  VD_CHECK_ARG(w && g && m && v && n >= 0, "vd_clamp_adam: bad args");
  VD_CHECK_ARG((((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "vd_clamp_adam: buffers must be 16-byte aligned");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL(clamp_adam_kernel, grid1d((n + 3) / 4, 256), dim3(256), 0, (hipStream_t)stream, w, g, m, v,
                     (long)n, gscale, clip, beta1, beta2, eps, step);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

}  // extern "C"

// internal (common.h): the same segmented row sum over bf16 rows (the compact da of a bf16 pass)
int vd_segment_rowsum_acc_bf16(const vd_bf16_bits* X16, int64_t ldx, const int32_t* tok, const int32_t* perm, int64_t n, int ncol,
                               float* out, int64_t ldo, hipStream_t stream) {
  VD_CHECK_ARG(X16 && tok && perm && out && n >= 0 && ncol % 4 == 0 && ldx % 4 == 0, "vd_segment_rowsum_acc_bf16: bad args");
  if (n == 0) return VD_OK;
  hipLaunchKernelGGL((segment_rowsum_kernel<256, true>), grid1d(n, 256), dim3(256), 0, stream, reinterpret_cast<const float*>(X16), (long)ldx,
                     tok, perm, (long)n, ncol, out, (long)ldo);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
