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
#ifdef VD_PROBE_NOSPLIT
  // PROBE build only (`make variant NAME=nosplit DEFS=-DVD_PROBE_NOSPLIT`; results are garbage): the operand bits reinterpreted as bf16 planes,
  // no conversion -- the instruction stream a kernel over PRE-SPLIT planes would run (optimistic: it moves 4 instead of 6 bytes per value)
  // (exponent fields forced to 0x7E: random signs and mantissas with magnitudes in [0.5, 1) -- bit patterns of ordinary operands, so the
  //  matrix pipe draws what it draws on real data instead of saturating to NaN; one v_and_or_b32 per register)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 a = (__builtin_bit_cast(u32x4, u) & 0x807F807Fu) | 0x3F003F00u, b = (__builtin_bit_cast(u32x4, w) & 0x807F807Fu) | 0x3F003F00u;
  hi = __builtin_bit_cast(vd_bf16x8, a);
  mid = __builtin_bit_cast(vd_bf16x8, b);
  lo = __builtin_bit_cast(vd_bf16x8, a ^ 0x00150015u);
  return;
#endif
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
static __global__ void __launch_bounds__(256) f32_to_bf16x3_kernel(const float* __restrict__ src, vd_bf16_bits* __restrict__ dst, long n4, long plane) {
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

// workgroup id -> tile.  Block b runs on XCD b % 8 (round-robin dispatch), so the map decides which operand bytes each XCD's private L2
// sees.  When the column tiles divide over VD_SPLIT_CG column groups, the 8 XCDs form a CG x (8 / CG) grid: an XCD owns the weight planes
// of tiles_n / CG column tiles (resident in its 4 MB L2) and walks 1 / (8 / CG) of the row tiles, column tile fastest (a row tile's A rows
// are fetched once per XCD that needs them and shared by its column tiles through L2).  CG = 8 (round 5): smallest weight slice, but every
// XCD fetches ALL of A -- 8 x 41 MB per forward launch, the 1.6 x over-fetch of profiles/r05_pmc_option_lstm_kernels.txt; CG = 2: A is
// fetched twice, 3 MB of planes per XCD.  Else XCD-contiguous ranges with the column tile fastest (the backward step: 4 column tiles).
// The grid is padded to 8 x rows-per-group x columns-per-group workgroups; a workgroup whose row tile is past the end leaves at once.
#ifndef VD_SPLIT_CG
#define VD_SPLIT_CG 2
#endif
__device__ __forceinline__ bool split_tile_of(int bid, int nwg, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
  constexpr int CG = VD_SPLIT_CG, RG = 8 / CG;
  if ((tiles_n % CG) == 0 && tiles_n >= 8) {
    const int xcd = bid & 7, li = bid >> 3, cpg = tiles_n / CG, rpg = (tiles_m + RG - 1) / RG;
    tile_n = (xcd % CG) * cpg + li % cpg;
    tile_m = (xcd / CG) * rpg + li / cpg;
    return tile_m < tiles_m && li / cpg < rpg;
  }
  const int wg = xcd_remap(bid, nwg);
  tile_n = wg % tiles_n;
  tile_m = wg / tiles_n;
  return true;
}
static int split_grid(int tiles_m, int tiles_n) {
  constexpr int CG = VD_SPLIT_CG, RG = 8 / CG;
  if ((tiles_n % CG) == 0 && tiles_n >= 8) return 8 * ((tiles_m + RG - 1) / RG) * (tiles_n / CG);
  return tiles_m * tiles_n;
}

template <int NPROD, class Epi>
__global__ void __launch_bounds__(256, 3)
gemm_split_kernel(int M, int N, int K, int tiles_n, const float* A, long lda, const vd_bf16_bits* B, long ldb, long bplane, Epi epi) {
  using Cfg = SplitCfg;
  constexpr int NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int tile_m, tile_n;
  if (!split_tile_of((int)blockIdx.x, (int)gridDim.x, (M + Cfg::BM - 1) / Cfg::BM, tiles_n, tile_m, tile_n)) return;
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
      // plane p of B against planes i of A, smallest terms first; j innermost.  (Measured and dropped: producing the A planes one at a time,
      // hi first, with the next plane's VALU instructions interleaved into the MFMAs by sched_group_barrier -- the technique that pays in
      // the dWh kernel below: 337 / 330 vs 342 / 322 us alone, 23.6 vs 23.4 ms in the step, 158 instead of 124 VGPRs.)
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
  hipLaunchKernelGGL(kern, dim3(split_grid(tiles_m, tiles_n)), dim3(256), Cfg::LDS_BYTES, stream, M, N, K, tiles_n, A, lda, B, ldb, bplane, e);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same exact split for the weight-gradient contraction  C[M x N] += A[K x M]^T * B[K x N]  (dWh = h^T * da over all
// (timestep, row) pairs: K = 380 000 at the headline shape).  Both operands are fp32 rows contracted over their ROW index, so
// NEITHER can be converted ahead of time without another pass over 4 GB: tiles go global -> LDS by DMA as [16 k][256 m] and
// [16 k][128 n] fp32 (k-major, dense: a row of the tile is a row segment of the operand), a wave reads its MFMA fragments with
// ds_read_b32 (lane = one m or n, 8 consecutive k: conflict-free, lanes 0-31 walk one k row) and splits BOTH in registers.
// A wave owns 64 m x 128 n (8 accumulator tiles): 48 split values (16 of A, 32 of B) per 72 MFMAs.  Split-K over the rows with
// float atomics at the end; the splits of one K range run on ONE XCD (block b runs on XCD b % 8), so the 16 column tiles that
// read the same h rows and the 2 row tiles that read the same da rows share them through that XCD's L2.
// ---------------------------------------------------------------------------------------------------------------------------
struct SplitTnCfg {
  static constexpr int BM = 256, BN = 128, THREADS = 256;
  static constexpr int ATILE = 16 * BM * 4;                       // [16 k][256 m] fp32
  static constexpr int BTILE = 16 * BN * 4;                       // [16 k][128 n] fp32
  static constexpr int STAGE = ATILE + BTILE;                     // 24 KB
  static constexpr int LDS_BYTES = 2 * STAGE;                     // 48 KB: two workgroups per CU (the registers allow no more)
};

template <int NPROD>
__global__ void __launch_bounds__(256, 2)
gemm_split_tn_kernel(int M, int N, int kchunk, int K, int tiles, int tiles_n, int splits, const float* A, long lda, const float* B, long ldb,
                     float* C, long ldc) {
  using Cfg = SplitTnCfg;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // tile / K-range of this workgroup: XCD x takes the K ranges [x * spx, (x + 1) * spx) (spx = splits / 8) and all tiles of each
  int tile, split;
  if ((splits & 7) == 0) {
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    split = xcd * (splits >> 3) + li / tiles;
    tile = li % tiles;
  } else {
    split = (int)blockIdx.x / tiles;
    tile = (int)blockIdx.x % tiles;
  }
  const int m_base = (tile / tiles_n) * Cfg::BM, n_base = (tile % tiles_n) * Cfg::BN;
  const int ks = split * kchunk, ke = min(K, ks + kchunk);
  const int nm = ke > ks ? (ke - ks) / 16 : 0;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);

  f32x16 acc[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][j][r] = 0.f;

  // DMA: A rows are 1 KB (one instruction per k row: 4 per wave), B rows 512 bytes (two k rows per instruction: 2 per wave)
  unsigned voffa[4], voffb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) voffa[i] = (unsigned)(((long)(wm * 4 + i) * lda + m_base + lane * 4) * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) voffb[i] = (unsigned)(((long)((wm * 2 + i) * 2 + (lane >> 5)) * ldb + n_base + (lane & 31) * 4) * 4);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto issue = [&](int m, int st) {
    const unsigned base = lds0 + st * Cfg::STAGE;
    const float* ak = A + (long)(ks + m * 16) * lda;
    const float* bk = B + (long)(ks + m * 16) * ldb;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(voffb[i], bk, base + Cfg::ATILE + (wm * 2 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(voffa[i], ak, base + (wm * 4 + i) * 1024);
  };
  const int hi = lane >> 5, l31 = lane & 31;

  if (nm > 0) {
    issue(0, 0);
    int st = 0;
    for (int m = 0; m < nm; ++m) {
      if (m + 1 < nm) {
        issue(m + 1, st ^ 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");
      const float* sA = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + st * Cfg::STAGE) + (hi * 8) * Cfg::BM + wm * 64 + l31;
      const float* sB = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + st * Cfg::STAGE + Cfg::ATILE) + (hi * 8) * Cfg::BN + l31;
      // Software pipeline inside the step: the A planes and the B planes of column tile 0 are converted up front; then the 18 MFMAs of column
      // tile j are issued INTERLEAVED with the fragment reads and the split of column tile j + 1 (sched_group_barrier: one MFMA, four VALU
      // instructions, ...), so most of the ~240 VALU instructions a step spends on splitting sit in the shadow of the matrix pipe instead of
      // in front of it (left to itself the compiler hoists every conversion above the first MFMA: 5.91 -> 5.51 ms on the headline dWh).
      // (Measured and dropped: three LDS stages with one barrier per step and the NEXT step's first conversions under the last column
      //  tile's MFMAs -- 5.82 ms; largest products first so that a step's first MFMAs need only the hi planes -- 5.68 vs 5.61 ms.)
      vd_bf16x8 ap[2][3], bp[2][3];
      auto load_b = [&](int j, vd_bf16x8 (&dst)[3]) {
        const float* p = sB + j * 32;
        const float4 u = make_float4(p[0], p[Cfg::BN], p[2 * Cfg::BN], p[3 * Cfg::BN]);
        const float4 w = make_float4(p[4 * Cfg::BN], p[5 * Cfg::BN], p[6 * Cfg::BN], p[7 * Cfg::BN]);
        vd_split3(u, w, dst[0], dst[1], dst[2]);
      };
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float* p = sA + mb * 32;
        const float4 u = make_float4(p[0], p[Cfg::BM], p[2 * Cfg::BM], p[3 * Cfg::BM]);
        const float4 w = make_float4(p[4 * Cfg::BM], p[5 * Cfg::BM], p[6 * Cfg::BM], p[7 * Cfg::BM]);
        vd_split3(u, w, ap[mb][0], ap[mb][1], ap[mb][2]);
      }
      load_b(0, bp[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < 4) load_b(j + 1, bp[(j + 1) & 1]);
#pragma unroll
        for (int pb = 2; pb >= 0; --pb)
#pragma unroll
          for (int pa = 2; pa >= 0; --pa) {
            constexpr int LIM = NPROD >= 9 ? 4 : NPROD >= 6 ? 2 : NPROD >= 3 ? 1 : 0;
            if (pa + pb > LIM) continue;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[mb][pa], bp[j & 1][pb], acc[mb][j], 0, 0, 0);
          }
        if (j + 1 < 4) {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);        // the next tile's fragment reads first (their latency hides under the MFMAs)
#pragma unroll
          for (int i = 0; i < 18; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // ... four VALU instructions of the next tile's split
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      st ^= 1;
    }
  }
  if (nm == 0) return;                            // (a K range past the end: the range count is rounded up to a multiple of 8)
  const EpiAtomic<4> e{C, ldc};
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) e(acc[mb], m_base + wm * 64 + mb * 32, n_base, lane, M, N);
}

// C[M x N] += A[K x M]^T * B[K x N] on the exact split; M % 256 == 0, N % 128 == 0, K % 16 == 0 (the caller contracts a ragged tail elsewhere)
template <int NPROD>
static int launch_gemm_split_tn(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, hipStream_t stream) {
  using Cfg = SplitTnCfg;
  VD_CHECK_ARG(M % Cfg::BM == 0 && N % Cfg::BN == 0 && K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && 16L * lda * 4 < (1L << 31) && 16L * ldb * 4 < (1L << 31),
               "launch_gemm_split_tn: unsupported shape M=%d N=%d K=%d", M, N, K);
  if (K == 0) return VD_OK;
  const int tiles_n = N / Cfg::BN, tiles = (M / Cfg::BM) * tiles_n;
  // one full round of workgroups at two per CU; a multiple of 8 K ranges so that each XCD owns whole ranges
  int splits = vd_cdiv(2L * vd_num_cus(), tiles);
  splits = (splits + 7) & ~7;
  int kchunk = vd_cdiv(vd_cdiv(K, splits), 16) * 16;
  if (kchunk < 64) kchunk = 64;
  splits = vd_cdiv(K, kchunk);
  if (splits >= 8) splits = (splits + 7) & ~7;        // (trailing K ranges past the end are empty workgroups)
  auto kern = gemm_split_tn_kernel<NPROD>;
  static bool attr_set = false;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(256), Cfg::LDS_BYTES, stream, M, N, kchunk, K, tiles, tiles_n, splits, A, lda, B, ldb, C, ldc);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
