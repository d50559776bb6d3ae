// split_core.h -- exact-operand bf16 split of an fp32 GEMM for the option recurrence (opt-in, `lstmPrecision = split9`).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 value is the exact sum of three bf16 values
// (hi = RNE(v), mid = RNE(v - hi), lo = RNE(v - hi - mid): 3 x 8 significand bits + signs cover the 24), the product of two
// bf16 values is exact in fp32, so  A * B = sum_{i,j in {hi,mid,lo}} A_i * B_j  -- nine bf16 MFMAs (v_mfma_f32_32x32x16_bf16,
// fp32 accumulation) per fp32 one, 9/16 of the matrix-pipe time, every product exact.  NPROD selects the products that are
// issued: 9 = all (fp32-grade), 6 = those with i + j <= 2 (drops terms below 2^-24 of the largest), 3 = i + j <= 1, 1 = plain
// bf16.  Only 9 is a substitute for fp32; the others exist for the error table (tests/test_ops_gpu.py).
//
// Operands: A [M x K] fp32 rows exactly as the fp32 kernels read them -- the activations stay fp32 in memory, a wave splits its
// fragments into three bf16 planes IN REGISTERS -- and B as three precomputed bf16 planes Bt_p [N x K] (k contiguous; the recurrent
// weights, converted once per pass).  Tiles go global -> LDS by DMA; one macro step = 16 k = one fp32 A tile [128 x 16] (64-byte rows,
// chunk swizzle of gemm_block_glds) + three bf16 B tiles [128 x 16] (32-byte rows, chunk swizzle (row >> 4) & 1: conflict-free for
// ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}, SQ_LDS_BANK_CONFLICT = 0; round 4's (row >> 2) & 1 was two-way
// conflicted on every B read): 20 KB per stage, two stages = 40 KB, four workgroups per CU -- the epilogue of one (gates, c, h, the
// projection-table gather) runs under the MFMAs of the others.  Per K step a B plane's four fragments are read once and multiplied
// with the three A planes, column tile innermost: consecutive MFMAs never share an accumulator.
//
// What bounds it (profiles/r05_experiments.txt section 1): POWER.  v_mfma_f32_32x32x16_bf16 on non-zero operands sustains 1.85 PFLOP/s
// on this chip, not 2.5 (scripts/probes/mfma_bf16_peak.hip: the clock falls to ~1.75 GHz); 9 x 41.9 GFLOP take 204 us at that rate with
// the pipe 100 % busy.  The K loop alone runs 299 us on random operands and 206 us on zeros (same code, same cycles, higher clock);
// bigger wave tiles (64 x 128 per wave: half the LDS-DMA bytes and fragment reads per MFMA) do not change it.
#pragma once
#include "gemm_core.h"

typedef __bf16 vd_bf16x8 __attribute__((ext_vector_type(8)));
typedef float vd_f32x4 __attribute__((ext_vector_type(4)));

// v (8 consecutive k of one row) -> hi / mid / lo planes
__device__ __forceinline__ void vd_split3(const float4& u, const float4& w, vd_bf16x8& hi, vd_bf16x8& mid, vd_bf16x8& lo) {
  const float v[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    const float r1 = v[i] - (float)h;          // exact: the leading 8 bits cancel
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;            // exact
    hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
  }
}

// fp32 matrix -> three bf16 planes (dst + p * plane), same element order
__global__ void __launch_bounds__(256) f32_to_bf16x3_kernel(const float* __restrict__ src, vd_bf16_bits* __restrict__ dst, long n4, long plane) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(src)[i];
  float4 r1, r2;
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  const bf4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  r1.x = v.x - (float)h[0]; r1.y = v.y - (float)h[1]; r1.z = v.z - (float)h[2]; r1.w = v.w - (float)h[3];
  const bf4 m = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
  r2.x = r1.x - (float)m[0]; r2.y = r1.y - (float)m[1]; r2.z = r1.z - (float)m[2]; r2.w = r1.w - (float)m[3];
  const bf4 l = {(__bf16)r2.x, (__bf16)r2.y, (__bf16)r2.z, (__bf16)r2.w};
  *reinterpret_cast<bf4*>(dst + i * 4) = h;
  *reinterpret_cast<bf4*>(dst + plane + i * 4) = m;
  *reinterpret_cast<bf4*>(dst + 2 * plane + i * 4) = l;
}
static int weights_to_bf16x3(const float* src, vd_bf16_bits* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_bf16x3_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, src, dst, n / 4, n);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

struct SplitCfg {
  static constexpr int NT = 4, WM = 4, THREADS = 256;
  static constexpr int BM = 128, BN = 128;
  static constexpr int ATILE = BM * 64;                           // [BM rows][16 fp32]      = 64-byte rows
  static constexpr int BTILE = BN * 32;                           // [BN rows][16 bf16]      = 32-byte rows, one per plane
  static constexpr int STAGE = ATILE + 3 * BTILE;                 // one macro step = 16 k
#ifndef VD_SPLIT_LDS
#define VD_SPLIT_LDS 41984                                        // as the fp32 step kernels: three workgroups per CU + room for one latency-shape
#endif                                                            // workgroup of the encoder streams (`make variant DEFS=-DVD_SPLIT_LDS=40960`: four)
  static constexpr int LDS_BYTES = VD_SPLIT_LDS;
  static_assert(LDS_BYTES >= 2 * STAGE, "two stages of 20 KB");
};

// workgroup id -> tile.  When the column tiles divide over the 8 XCDs, every XCD owns a COLUMN slice of the weight planes (resident
// in its 4 MB L2: the three planes together are 6 MB) and walks all row tiles (block b runs on XCD b % 8); else XCD-contiguous
// ranges with the column tile fastest.
__device__ __forceinline__ void split_tile_of(int bid, int nwg, int tiles_n, int& tile_m, int& tile_n) {
  if ((tiles_n & 7) == 0) {
    const int xcd = bid & 7, li = bid >> 3, cpx = tiles_n >> 3;
    tile_n = xcd * cpx + li % cpx;
    tile_m = li / cpx;
    return;
  }
  const int wg = xcd_remap(bid, nwg);
  tile_n = wg % tiles_n;
  tile_m = wg / tiles_n;
}

template <int NPROD, class Epi>
__global__ void __launch_bounds__(256, 3)
gemm_split_kernel(int M, int N, int K, int tiles_n, const float* A, long lda, const vd_bf16_bits* B, long ldb, long bplane, Epi epi) {
  using Cfg = SplitCfg;
  constexpr int NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int tile_m, tile_n;
  split_tile_of((int)blockIdx.x, (int)gridDim.x, tiles_n, tile_m, tile_n);
  const int row_base = tile_m * Cfg::BM, col_base = tile_n * Cfg::BN;
  constexpr int NIA = 2;                                         // this wave's own 32 rows = 2 pieces of 16 rows x 64 bytes
  constexpr int NIB = 3;                                         // row group `wave` (32 rows x 32 bytes) of every plane
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nm = K / 16;                                         // macro steps
  const int wrow = row_base + wm * 32;                           // first row of this wave

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  typename EpiPreOf<Epi>::type pre;
  if constexpr (EpiPreOf<Epi>::value) epi.preload(pre, wrow, lane, M);

  // per-thread source byte offsets of this wave's DMA instructions (macro step 0); chunk index swizzled on the source side
  unsigned voffa[NIA], voffb[NIB];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int r = wm * 32 + i * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    voffa[i] = (unsigned)(((long)min(row_base + r, M - 1) * lda + c * 4) * 4);
  }
  {
    const int r = wm * 32 + (lane >> 1);
    const int c = (lane & 1) ^ ((r >> 4) & 1);
    const unsigned v = (unsigned)(((long)min(col_base + r, N - 1) * ldb + c * 8) * 2);
#pragma unroll
    for (int i = 0; i < NIB; ++i) voffb[i] = v;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto issue = [&](int m, int st) {
    const unsigned base = lds0 + st * Cfg::STAGE;
#pragma unroll
    for (int i = 0; i < NIB; ++i)
      glds16(voffb[i], reinterpret_cast<const float*>(B + (long)i * bplane + (long)m * 16), base + Cfg::ATILE + i * Cfg::BTILE + wm * 1024);
    const float* ak = A + (long)m * 16;
#pragma unroll
    for (int i = 0; i < NIA; ++i) glds16(voffa[i], ak, base + (wm * NIA + i) * 1024);
  };
  const int hi = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 2) & 3;
  const int cb = (hi ^ ((l31 >> 4) & 1)) * 16;                   // this lane's 8 bf16 k inside a 32-byte B row

  if (nm > 0) {
    issue(0, 0);
    int st = 0;
    for (int m = 0; m < nm; ++m) {
      if (m + 1 < nm) {
        issue(m + 1, st ^ 1);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");          // NIA + NIB requests of step m + 1 may stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");                    // every wave's share of macro step m has landed
      const char* sbase = reinterpret_cast<const char*>(smem) + st * Cfg::STAGE;
      const float* sa = reinterpret_cast<const float*>(sbase) + (wm * 32 + l31) * 16;
      const float4 u = *reinterpret_cast<const float4*>(sa + (((hi * 2) ^ sw) * 4));
      const float4 w = *reinterpret_cast<const float4*>(sa + (((hi * 2 + 1) ^ sw) * 4));
      vd_bf16x8 ap[3];
      vd_split3(u, w, ap[0], ap[1], ap[2]);
      // plane p of B against planes i of A, smallest terms first; j innermost
#pragma unroll
      for (int p = 2; p >= 0; --p) {
        vd_bf16x8 bp[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const vd_f32x4 t = *reinterpret_cast<const vd_f32x4*>(sbase + Cfg::ATILE + p * Cfg::BTILE + (j * 32 + l31) * 32 + cb);
          bp[j] = __builtin_bit_cast(vd_bf16x8, t);
        }
#pragma unroll
        for (int i = 2; i >= 0; --i) {
          constexpr int LIM = NPROD >= 9 ? 4 : NPROD >= 6 ? 2 : NPROD >= 3 ? 1 : 0;     // products with i + p <= LIM are issued
          if (i + p > LIM) continue;
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i], bp[j], acc[j], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage `st` may be refilled
      st ^= 1;
    }
  }
  if constexpr (EpiPreOf<Epi>::value) epi(acc, wrow, col_base, lane, M, N, smem + wm * 1024, &pre);
  else epi(acc, wrow, col_base, lane, M, N, smem + wm * 1024);
}

// C[M x N] = epi(A[M x K] (fp32 rows) * Bt[N x K]^T), Bt given as three bf16 planes; K % 16 == 0
template <int NPROD, class Epi>
static int launch_gemm_split(int M, int N, int K, const float* A, long lda, const vd_bf16_bits* B, long ldb, long bplane, const Epi& e,
                             hipStream_t stream) {
  if (M <= 0 || N <= 0) return VD_OK;
  using Cfg = SplitCfg;
  VD_CHECK_ARG(K % 16 == 0 && lda % 4 == 0 && ldb % 8 == 0 && (long)M * lda * 4 < (1L << 32) && (long)N * ldb * 2 < (1L << 32),
               "launch_gemm_split: unsupported shape M=%d N=%d K=%d", M, N, K);
  const int tiles_m = vd_cdiv(M, Cfg::BM), tiles_n = vd_cdiv(N, Cfg::BN);
  auto kern = gemm_split_kernel<NPROD, Epi>;
  static bool attr_set = false;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::LDS_BYTES, stream, M, N, K, tiles_n, A, lda, B, ldb, bplane, e);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
