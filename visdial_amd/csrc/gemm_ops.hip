// C-ABI entry points for the plain (non-recurrent) fp32 MFMA GEMMs: the Linear layers
// of the encoders (nn.Linear y = x*W^T + b, weight [out x in]), the
// hoisted LSTM input projections, and every weight-gradient contraction.
#include "gemm_core.h"

using CfgBigDB = GemmCfg<4, 1, 4, 32>;        // 128 x 128 tile, double-buffered LDS (long K loops: weight grads)
using CfgBig = GemmCfg<4, 1, 4, 16, 0, 3>;    // 128 x 128 tile, BK = 16, single LDS buffer, 3 workgroups / CU
using CfgSmall = GemmCfg<1, 4, 1, 32, 0, 3>;  // 32 x 32 tile, 4-way intra-block split-K (latency shapes)

static inline bool use_small(int M, int N) {
  return (long)vd_cdiv(M, CfgBig::BM) * vd_cdiv(N, CfgBig::BN) < 48;
}

static int check_align(const void* p, long ld, const char* what) {
  if (((uintptr_t)p & 15) != 0 || (ld & 3) != 0) {
    vd_set_error("%s: pointer must be 16-byte aligned and leading dimension a multiple of 4 (ld=%ld)", what, ld);
    return VD_ERR_ARG;
  }
  return VD_OK;
}

// column sums: out[col] += sum_rows X[row][col]
__global__ void colsum_kernel(const float* __restrict__ X, long ld, int M, int N, int rows_per_block,
                              float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= N) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += X[(long)r * ld + col];
    s1 += X[(long)(r + 1) * ld + col];
    s2 += X[(long)(r + 2) * ld + col];
    s3 += X[(long)(r + 3) * ld + col];
  }
  for (; r < r1; ++r) s0 += X[(long)r * ld + col];
  unsafeAtomicAdd(out + col, (s0 + s1) + (s2 + s3));
}

extern "C" {

// C[M x N] (+)= act(A[M x K] * W[N x K]^T + bias)
int vd_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int M, int N, int K, int act, int accumulate, void* stream) {
  VD_CHECK_ARG(A && W && C && M >= 0 && N >= 0 && K >= 0 && K % 4 == 0, "vd_gemm_nt: bad args M=%d N=%d K=%d", M,
               N, K);
  if (int rc = check_align(A, lda, "vd_gemm_nt A")) return rc;
  if (int rc = check_align(W, ldw, "vd_gemm_nt W")) return rc;
  SrcRow a{A, lda}, b{W, ldw};
  hipStream_t s = (hipStream_t)stream;
  if (use_small(M, N)) {
    EpiStore<1> e{C, ldc, bias, act, accumulate};
    return launch_gemm<CfgSmall>(M, N, K, 1, a, b, e, s);
  }
  EpiStore<4> e{C, ldc, bias, act, accumulate};
  // both operands are k-contiguous rows: throughput shapes take the LDS-DMA pipeline (gemm_core.h)
  static const int nt_glds = getenv("VD_NT_GLDS") ? atoi(getenv("VD_NT_GLDS")) : 1;
  if (nt_glds && M >= 1024 && K >= 64 && K % 16 == 0 && (long)M * lda * 4 < (1L << 32) && (long)N * ldw * 4 < (1L << 32))
    return launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 41984>, false>(M, N, K, 1, A, lda, W, ldw, e, s);
  return launch_gemm<CfgBig>(M, N, K, 1, a, b, e, s);
}

// C[M x N] (+)= A[M x K] * B[K x N] + bias
int vd_gemm_nn(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
               int64_t ldc, int M, int N, int K, int accumulate, void* stream) {
  // K need not be a multiple of 4 when every A row is padded to one (lda >= ceil4(K), pad columns finite):
  // the float4 that straddles K multiplies pad values by zero-filled B rows.
  VD_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0 && (K % 4 == 0 || lda >= (K + 3) / 4 * 4) && N % 4 == 0,
               "vd_gemm_nn: bad args M=%d N=%d K=%d", M, N, K);
  if (int rc = check_align(A, lda, "vd_gemm_nn A")) return rc;
  if (int rc = check_align(B, ldb, "vd_gemm_nn B")) return rc;
  SrcRow a{A, lda};
  SrcK b{B, ldb};
  hipStream_t s = (hipStream_t)stream;
  if (use_small(M, N)) {
    EpiStore<1> e{C, ldc, bias, VD_ACT_NONE, accumulate};
    return launch_gemm<CfgSmall>(M, N, K, 1, a, b, e, s);
  }
  EpiStore<4> e{C, ldc, bias, VD_ACT_NONE, accumulate};
  return launch_gemm<CfgBig>(M, N, K, 1, a, b, e, s);
}

// C[M x N] += A[K x M]^T * B[K x N]   (weight gradients; split-K with float atomics)
int vd_gemm_tn_acc(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                   int N, int K, int flags, void* stream) {
  // M (resp. N) need not be a multiple of 4 when the rows of A (resp. B) are padded to one.
  VD_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0 && (M % 4 == 0 || lda >= (M + 3) / 4 * 4) &&
                   (N % 4 == 0 || ldb >= (N + 3) / 4 * 4),
               "vd_gemm_tn_acc: bad args M=%d N=%d K=%d", M, N, K);
  if (int rc = check_align(A, lda, "vd_gemm_tn_acc A")) return rc;
  if (int rc = check_align(B, ldb, "vd_gemm_tn_acc B")) return rc;
  if (K == 0) return VD_OK;
  SrcK a{A, lda}, b{B, ldb};
  EpiAtomic<4> e{C, ldc};
  const long tiles = (long)vd_cdiv(M, CfgBig::BM) * vd_cdiv(N, CfgBig::BN);
  // LDS-DMA pipeline with k-major tiles: both operands are [k][m] rows, which IS the MFMA fragment layout along
  // the lanes (no transposing LDS writes).  One full round of workgroups (3 per CU x 256 CUs = 768) and no K-tile
  // rotation: 133.9 TFLOP/s on the option dWh shape (M=512, N=2048, K=380 000) vs 119.6 for the register-staged
  // kernel at 1024 blocks (profiles/r02_dwh_sweep.txt; with 1024 blocks the ranking was the opposite in round 1:
  // the 1.33-round tail, not the pipeline, decided).
  const int cfg = vd_tune_get("VD_TN_CFG", 20);
  const bool kmaj = cfg == 20 && !(flags & VD_FLAG_BF16) && M % 128 == 0 && N % 128 == 0 &&
                    K >= vd_tune_get("VD_TN_KMAJ_MINK", 1024);
  if (kmaj && K % 16 != 0) {
    // the k-major pipeline moves whole 16-row K tiles: contract the first floor(K / 16) * 16 rows with it and the last
    // < 16 rows with the register-staged kernel (one short launch; e.g. the encoder's (T-1)*N = 7 800 rows)
    const int K1 = K & ~15;
    if (int rc = vd_gemm_tn_acc(A, lda, B, ldb, C, ldc, M, N, K1, flags, stream)) return rc;
    SrcK a2{A + (long)K1 * lda, lda}, b2{B + (long)K1 * ldb, ldb};
    return launch_gemm<GemmCfg<4, 1, 4, 16, 0, 4, 41984>>(M, N, K - K1, 1, a2, b2, e, (hipStream_t)stream);
  }
  // split-K target: one full round of workgroups for the big k-major shape; the register-staged shapes (encoder weight
  // gradients, K <= 11 323) take fewer, longer slices -- every slice ends in 64 KB of float atomics per tile
  const int target = kmaj ? vd_tune_get("VD_TN_BLOCKS", 768) : vd_tune_get("VD_TN_BLOCKS_SMALL", 768);
  long splits = vd_cdiv(target, tiles);
  const long max_splits = vd_cdiv(K, vd_tune_get("VD_TN_MIN_KCHUNK", 1024));   // (sweep in the full step: 26.03 -> 25.70 ms vs 1024 blocks / 64-row slices)
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (flags & VD_FLAG_BF16)   // opt-in reduced precision: bf16 operands, fp32 accumulation
    return launch_gemm<GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  if (kmaj)
    return launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 41984>, true>(M, N, K, (int)splits, A, lda, B, ldb, e,
                                                                      (hipStream_t)stream, vd_tune_get("VD_TN_ROTATE", 0));
  if (cfg == 20 || cfg == 5)   // (cfg 20 on a shape the k-major pipeline does not take: the register-staged default)
    return launch_gemm<GemmCfg<4, 1, 4, 16, 0, 4, 41984>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  if (cfg == 1) return launch_gemm<CfgBig>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  if (cfg == 2) return launch_gemm<GemmCfg<4, 1, 4, 16, 0, 3>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  if (cfg == 3) return launch_gemm<GemmCfg<4, 1, 4, 16, 2, 3>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  if (cfg == 4) return launch_gemm<GemmCfg<4, 1, 4, 32, 2, 3>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  return launch_gemm<CfgBigDB>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
}

// C[M x N] += sum_k A[a_rows[k], :M]^T * B[b_rows[k], :N]: the weight-gradient contraction over an explicit list of K
// (row of A, row of B) pairs instead of all rows -- the non-pad (timestep, row) pairs of a maskZero recurrence.
int vd_gemm_tn_rows_acc(const float* A, int64_t lda, const int32_t* a_rows, const float* B, int64_t ldb,
                        const int32_t* b_rows, float* C, int64_t ldc, int M, int N, int K, void* stream) {
  VD_CHECK_ARG(A && B && C && a_rows && b_rows && M >= 0 && N >= 0 && K >= 0 && (M % 4 == 0 || lda >= (M + 3) / 4 * 4) &&
                   (N % 4 == 0 || ldb >= (N + 3) / 4 * 4),
               "vd_gemm_tn_rows_acc: bad args M=%d N=%d K=%d", M, N, K);
  if (int rc = check_align(A, lda, "vd_gemm_tn_rows_acc A")) return rc;
  if (int rc = check_align(B, ldb, "vd_gemm_tn_rows_acc B")) return rc;
  if (K == 0) return VD_OK;
  SrcKRows a{A, lda, a_rows}, b{B, ldb, b_rows};
  EpiAtomic<4> e{C, ldc};
  using Cfg = GemmCfg<4, 1, 4, 16, 0, 4, 41984>;
  const long tiles = (long)vd_cdiv(M, Cfg::BM) * vd_cdiv(N, Cfg::BN);
  long splits = vd_cdiv(vd_tune_get("VD_TN_BLOCKS_SMALL", 768), tiles);
  const long max_splits = vd_cdiv(K, vd_tune_get("VD_TN_MIN_KCHUNK", 1024));
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return launch_gemm<Cfg>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
}

// out[N] += column sums of X[M x N]  (bias gradients)
int vd_colsum_acc(const float* X, int64_t ld, int M, int N, float* out, void* stream) {
  VD_CHECK_ARG(X && out && M >= 0 && N >= 0, "vd_colsum_acc: bad args");
  if (M == 0 || N == 0) return VD_OK;
  const int cb = vd_cdiv(N, 256);
  int rb = vd_cdiv(1024, cb);
  int rows_per_block = vd_cdiv(M, rb);
  if (rows_per_block < 16) rows_per_block = 16;
  rb = vd_cdiv(M, rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, dim3(cb, rb), dim3(256), 0, (hipStream_t)stream, X, ld, M, N,
                     rows_per_block, out);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

}  // extern "C"
