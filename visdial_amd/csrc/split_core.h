// split_core.h -- exact-operand bf16 split of an fp32 GEMM for the option recurrence (opt-in, `lstmPrecision = split9`).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 value is the exact sum of three bf16 values
// (hi = RNE(v), mid = RNE(v - hi), lo = RNE(v - hi - mid): 3 x 8 significand bits + signs cover the 24), the product of two
// bf16 values is exact in fp32, so  A * B = sum_{i,j in {hi,mid,lo}} A_i * B_j  -- nine bf16 MFMAs (v_mfma_f32_32x32x16_bf16,
// fp32 accumulation) per fp32 one, 9/16 of the matrix-pipe time, every product exact.  NPROD selects the products that are
// issued: 9 = all (fp32-grade), 6 = those with i + j <= 2 (drops terms below 2^-24 of the largest), 3 = i + j <= 1, 1 = plain
// bf16.  Only 9 is a substitute for fp32; the others exist for the error table (tests/test_split_gpu.py).
//
// Operands: A [M x K] fp32 rows exactly as the fp32 kernels read them -- the activations stay fp32 in memory, a wave splits its
// fragments into three bf16 planes IN REGISTERS (VALU work that hides under the MFMA burst) -- and B as three precomputed bf16
// planes Bt_p [N x K] (k contiguous; the recurrent weights, converted once per pass).  Tiles go global -> LDS by DMA with the
// chunk swizzle of gemm_block_glds; one macro step = 16 k = one fp32 A tile [BM x 16] (64-byte rows) + three bf16 B tiles [BN x 16]
// (32-byte rows): 20 KB per stage at 128 x 128, two stages = 40 KB, so THREE workgroups share a CU like the fp32 kernels -- the
// epilogue of one (HBM-bound: gates, c, h, the projection-table gather) runs under the MFMAs of the others.
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

template <int NT_, int NSTAGE_ = 2>
struct SplitCfg {
  static constexpr int NT = NT_, WM = 4, THREADS = 256, NSTAGE = NSTAGE_;
  static constexpr int BM = 128, BN = NT * 32;
  static constexpr int ATILE = BM * 64;                           // [BM rows][16 fp32]      = 64-byte rows
  static constexpr int BTILE = BN * 32;                           // [BN rows][16 bf16]      = 32-byte rows, one per plane
  static constexpr int STAGE = ATILE + 3 * BTILE;                 // one macro step = 16 k
  static constexpr int LDS_BYTES = (NSTAGE * STAGE > 4 * 4096) ? NSTAGE * STAGE : 4 * 4096;   // (>= the epilogue's 4 KB per wave)
};

template <int NT, int NPROD, int NSTAGE, class Epi>
__device__ __forceinline__ void gemm_block_split(int M, int N, int K, int row_base, int col_base, const float* A, long lda,
                                                 const vd_bf16_bits* B, long ldb, long bplane, const Epi& epi, float* smem) {
  using Cfg = SplitCfg<NT, NSTAGE>;
  constexpr int WM = Cfg::WM, BM = Cfg::BM;
  constexpr int NIA = (BM / 16) / WM;                            // 1 KB DMA instructions per wave: A tile = 8
  // B: one plane tile = BN / 32 instructions (32 rows of 32 bytes each); 3 planes.  NT = 4: 12 instructions, 3 per wave (wave w
  // takes row group w of every plane).  NT = 2: 6 instructions -> 2 per wave, instruction ids {w, w + 4} mod 6 (two are issued
  // twice: identical bytes to the identical place) so that every wave has the SAME count and the vmcnt immediates are uniform
  constexpr int NIB = NT == 4 ? 3 : 2;
  static_assert(NT == 4 || NT == 2, "split pipeline: 128- or 64-column tiles");
  constexpr int PER_STEP = NIA + NIB;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nm = K / 16;                                         // macro steps

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  typename EpiPreOf<Epi>::type pre;
  if constexpr (EpiPreOf<Epi>::value) epi.preload(pre, row_base + wm * 32, lane, M);

  // per-thread source byte offsets of this wave's DMA instructions (macro step 0); chunk index swizzled on the source side
  unsigned voffa[NIA], voffb[NIB], ldsb[NIB];
  long planeb[NIB];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int r = (i * WM + wm) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    voffa[i] = (unsigned)(((long)min(row_base + r, M - 1) * lda + c * 4) * 4);
  }
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    int plane, grp;                                              // which plane, which 32-row group of its tile
    if constexpr (NT == 4) { plane = i; grp = wm; }
    else { const int id = (wm + 4 * i) % 6; plane = id >> 1; grp = id & 1; }
    const int r = grp * 32 + (lane >> 1);
    const int c = (lane & 1) ^ ((r >> 2) & 1);
    voffb[i] = (unsigned)(((long)min(col_base + r, N - 1) * ldb + c * 8) * 2);
    planeb[i] = (long)plane * bplane;
    ldsb[i] = (unsigned)(Cfg::ATILE + plane * Cfg::BTILE + grp * 1024);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto issue = [&](int m, int st) {
    const unsigned base = lds0 + st * Cfg::STAGE;
#pragma unroll
    for (int i = 0; i < NIB; ++i) glds16(voffb[i], reinterpret_cast<const float*>(B + planeb[i] + (long)m * 16), base + ldsb[i]);
    const float* ak = A + (long)m * 16;
#pragma unroll
    for (int i = 0; i < NIA; ++i) glds16(voffa[i], ak, base + (i * WM + wm) * 1024);
  };
  auto wait_vm = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  const int hi = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 2) & 3;

  if (nm > 0) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nm) issue(s, s);
    int st = 0;
    for (int m = 0; m < nm; ++m) {
      const int ahead = m + NSTAGE - 1;
      if (ahead < nm) {
        int sa = st + NSTAGE - 1;
        if (sa >= NSTAGE) sa -= NSTAGE;
        issue(ahead, sa);
        if constexpr (PER_STEP * (NSTAGE - 1) == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (PER_STEP * (NSTAGE - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (PER_STEP * (NSTAGE - 1) == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (PER_STEP * (NSTAGE - 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (PER_STEP * (NSTAGE - 1) == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if constexpr (PER_STEP * (NSTAGE - 1) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else static_assert(PER_STEP * (NSTAGE - 1) <= 15, "steady-state vmcnt immediate");
      } else {
        wait_vm(PER_STEP * (nm - 1 - m));                        // the tail: only the later steps' requests may still fly
      }
      asm volatile("s_barrier" ::: "memory");                    // every wave's share of macro step m has landed
      const char* sbase = reinterpret_cast<const char*>(smem) + st * Cfg::STAGE;
      const float* sa = reinterpret_cast<const float*>(sbase) + (wm * 32 + l31) * 16;
      const float4 u = *reinterpret_cast<const float4*>(sa + (((hi * 2) ^ sw) * 4));
      const float4 w = *reinterpret_cast<const float4*>(sa + (((hi * 2 + 1) ^ sw) * 4));
      vd_bf16x8 ap[3];
      vd_split3(u, w, ap[0], ap[1], ap[2]);
      const int cb = (hi ^ (sw & 1)) * 16;                       // byte offset of this lane's 8 bf16 k inside a 32-byte B row
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        vd_bf16x8 bp[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const vd_f32x4 t = *reinterpret_cast<const vd_f32x4*>(sbase + Cfg::ATILE + p * Cfg::BTILE + (j * 32 + l31) * 32 + cb);
          bp[p] = __builtin_bit_cast(vd_bf16x8, t);
        }
        // smallest products first
#define VD_PROD(i, jj) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i], bp[jj], acc[j], 0, 0, 0);
        if constexpr (NPROD >= 9) { VD_PROD(2, 2) VD_PROD(2, 1) VD_PROD(1, 2) }
        if constexpr (NPROD >= 6) { VD_PROD(2, 0) VD_PROD(1, 1) VD_PROD(0, 2) }
        if constexpr (NPROD >= 3) { VD_PROD(1, 0) VD_PROD(0, 1) }
        VD_PROD(0, 0)
#undef VD_PROD
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage `st` may be refilled
      if (++st == NSTAGE) st = 0;
    }
  }
  if constexpr (EpiPreOf<Epi>::value) epi(acc, row_base + wm * 32, col_base, lane, M, N, smem + wm * 1024, &pre);
  else epi(acc, row_base + wm * 32, col_base, lane, M, N, smem + wm * 1024);
}

template <int NT, int NPROD, int NSTAGE, class Epi>
__global__ void __launch_bounds__(256, 3)
gemm_split_kernel(int M, int N, int K, int tiles_m, int tiles_n, const float* A, long lda, const vd_bf16_bits* B, long ldb, long bplane, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
  gemm_block_split<NT, NPROD, NSTAGE>(M, N, K, tile_m * SplitCfg<NT>::BM, tile_n * SplitCfg<NT>::BN, A, lda, B, ldb, bplane, epi, smem);
}

// C[M x N] = epi(A[M x K] (fp32 rows) * Bt[N x K]^T), Bt given as three bf16 planes; K % 16 == 0
template <int NT, int NPROD, class Epi, int NSTAGE = 2>
static int launch_gemm_split(int M, int N, int K, const float* A, long lda, const vd_bf16_bits* B, long ldb, long bplane, const Epi& e,
                             hipStream_t stream) {
  if (M <= 0 || N <= 0) return VD_OK;
  using Cfg = SplitCfg<NT, NSTAGE>;
  VD_CHECK_ARG(K % 16 == 0 && lda % 4 == 0 && ldb % 8 == 0 && (long)M * lda * 4 < (1L << 32) && (long)N * ldb * 2 < (1L << 32),
               "launch_gemm_split: unsupported shape M=%d N=%d K=%d", M, N, K);
  const int tiles_m = vd_cdiv(M, Cfg::BM), tiles_n = vd_cdiv(N, Cfg::BN);
  auto kern = gemm_split_kernel<NT, NPROD, NSTAGE, Epi>;
  static bool attr_set = false;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::LDS_BYTES, stream, M, N, K, tiles_m, tiles_n, A, lda, B, ldb, bplane, e);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
