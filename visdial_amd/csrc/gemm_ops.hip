// C-ABI entry points for the plain (non-recurrent) fp32 MFMA GEMMs: the Linear layers
// of the encoders (nn.Linear y = x*W^T + b, weight [out x in]), the
// hoisted LSTM input projections, and every weight-gradient contraction.
#include "gemm_core.h"
#include "split_core.h"

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

// the same for N % 4 == 0, 16-byte aligned rows: a thread owns 4 consecutive columns (float4 loads: 4 x the bytes in flight per
// thread), the four row lanes of a 256-thread block (64 column groups x 4) are folded through LDS before the atomics.  The scalar
// kernel above moved 65 MB in 130-800 us inside a step (4 bytes per load, 4 loads in flight per thread); it remains for ragged N.
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ X, long ld, int M, int N, int rows_per_block,
                                                      float* __restrict__ out) {
  __shared__ float4 red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cx) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (col < N) {
    const float* p = X + col;
    int r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (long)r * ld);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (long)(r + 4) * ld);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (long)(r + 8) * ld);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (long)(r + 12) * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; r < r1; r += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (long)r * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  red[ry][cx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                            (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (ry == 0 && col < N) {
    const float4 b0 = red[0][cx], b1 = red[1][cx], b2 = red[2][cx], b3 = red[3][cx];
    unsafeAtomicAdd(out + col, (b0.x + b1.x) + (b2.x + b3.x));
    unsafeAtomicAdd(out + col + 1, (b0.y + b1.y) + (b2.y + b3.y));
    unsafeAtomicAdd(out + col + 2, (b0.z + b1.z) + (b2.z + b3.z));
    unsafeAtomicAdd(out + col + 3, (b0.w + b1.w) + (b2.w + b3.w));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 weight-gradient contraction on bf16 operands (BASELINE.json configs[4]): C[M x N] += A[K x M]^T * B[K x N] with A, B
// the bf16 shadows the producing kernels wrote (common.h).  Both operands are k-major ([k][m] rows), the MFMA wants 8
// consecutive k per lane: tiles go global -> LDS by DMA with NO conversion or transposing write, and the fragments are
// fetched with the gfx950 LDS transpose read (ds_read_b64_tr_b16).
//   * tile = 32 k x 128 m (8 KB) per operand; LDS image = [8 k-groups][8 m-groups] blocks of [4 k][16 m] bf16 (128 B
//     contiguous: the 16 lanes of a transpose-read group fetch one block conflict-free).  One DMA instruction moves one
//     k-group (4 k-rows x 128 m = 1 KB); the block layout is produced on the SOURCE side: LDS position p of an
//     instruction (16 B, lane-linear) takes row k0 + (p & 7) / 2, columns (p >> 3) * 16 + (p & 1) * 8 .. +7.
//   * MFMA 32x32x16 bf16: lane l holds row l % 32 and k = 8 * (l / 32) .. +7: two transpose reads (k-groups 2h, 2h + 1
//     of the step, h = l / 32) of the block (m-group = (l % 32) / 16), lane i = l % 16 of the group receives column i.
//   * three LDS buffers per operand (48 KB per workgroup, 3 workgroups per CU), one barrier per K tile, counted vmcnt.
// Bound: operand delivery (L2 -> LDS), not the matrix pipe: 16 KB per 8 MFMAs of 32 cycles.
// ---------------------------------------------------------------------------------------------------------------
typedef short vd_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 vd_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ vd_bf16x8 tr_read8(const char* lds_base, int off) {
  typedef __attribute__((address_space(3))) vd_s16x4* lds_p;
  const vd_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_base + off));
  const vd_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_base + off + 1024));
  union {
    vd_s16x4 h[2];
    vd_bf16x8 v;
  } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.v;
}

__global__ void __launch_bounds__(256, 3)
gemm_bf16_tn_tr_kernel(const vd_bf16_bits* __restrict__ A, const vd_bf16_bits* __restrict__ B, float* __restrict__ C, long ldc,
                       int M, int N, int K, int kchunk, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TILE = 8192;                            // bytes per operand tile; 3 buffers per operand
  char* const lds = reinterpret_cast<char*>(smem);     // [3][A tile | B tile]
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % tiles_n, tile_m = (wg / tiles_n) % tiles_m, split = wg / (tiles_n * tiles_m);
  const int ks = split * kchunk, ke = min(K, ks + kchunk);
  const int nk = ke > ks ? (ke - ks) / 32 : 0;
  const int lane = threadIdx.x & 63;
  const int wm = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int m0 = tile_m * 128, n0 = tile_n * 128;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // DMA source offsets (bytes from the tile's first k-row) of this wave's two instructions per operand: k-groups wm, wm + 4
  const int key = (lane & 7) >> 1, col = (lane >> 3) * 16 + (lane & 1) * 8;
  unsigned voffa[2], voffb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int krow = (i * 4 + wm) * 4 + key;
    voffa[i] = (unsigned)(((long)krow * M + m0 + col) * 2);
    voffb[i] = (unsigned)(((long)krow * N + n0 + col) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  auto issue = [&](int kt, int buf) {
    const float* ak = reinterpret_cast<const float*>(A + (long)(ks + kt * 32) * M);
    const float* bk = reinterpret_cast<const float*>(B + (long)(ks + kt * 32) * N);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(voffb[i], bk, lds0 + buf * 2 * TILE + TILE + (i * 4 + wm) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(voffa[i], ak, lds0 + buf * 2 * TILE + (i * 4 + wm) * 1024);
  };
  // transpose-read offsets inside a tile: k-group pair by lane half, m-group by lane quarter, 8 bytes per lane
  const int g = lane >> 4, li = lane & 15;
  const int frag = ((g >> 1) * 2) * 1024 + (g & 1) * 128 + li * 8;

  if (nk > 0) {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed when at most the 4 instructions of tile kt + 1 are still in flight
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");   // everybody's share of tile kt is in LDS; tile kt - 1 is fully consumed
      if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);   // (kt + 2) % 3 == (kt - 1) % 3
      const char* ta = lds + buf * 2 * TILE;
      const char* tb = ta + TILE;
#pragma unroll
      for (int st = 0; st < 2; ++st) {            // two 16-k MFMA steps per 32-k tile
        const vd_bf16x8 a8 = tr_read8(ta, frag + st * 4096 + wm * 256);
        vd_bf16x8 b8[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b8[j] = tr_read8(tb, frag + st * 4096 + j * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8[j], acc[j], 0, 0, 0);
      }
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
  // split-K partial sums into the gradient buffer with hardware float atomics
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = n0 + j * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + mfma_row(r, lane);
      unsafeAtomicAdd(C + (long)row * ldc + c, acc[j][r]);
    }
  }
}

// 256 x 256 output tiles, eight waves (wave = 64 x 128: 2 x 4 accumulator tiles): the 128 x 128 kernel above moves 16 KB through LDS-DMA and
// 40 KB through the transpose reads per MFLOP-tile of 8 MFMAs per wave -- LDS-bound 1.75 x over its matrix time.  Here a 32-k tile is
// 32 KB of DMA + 96 KB of fragment reads for 16 MFMAs per wave (LDS and matrix time balanced), and the L2 -> LDS volume of a contraction
// halves.  The LDS image is the plain [32 k][256 m] tile (each DMA instruction copies two whole 512-byte rows -- fully coalesced on the
// memory side, where the block layout above makes every lane pair jump to another row); the transpose reads gather their [4 k][16 m]
// blocks from four rows, kept on different banks by a 64-byte XOR swizzle.  NBUF buffers of 32 KB: one workgroup per CU.
template <int NBUF>
__global__ void __launch_bounds__(512)
gemm_bf16_tn_tr256_kernel(const vd_bf16_bits* __restrict__ A, const vd_bf16_bits* __restrict__ B, float* __restrict__ C, long ldc,
                          int M, int N, int K, int kchunk, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TILE = 16384;                           // bytes per operand tile (32 k x 256 columns); NBUF buffers per operand
  char* const lds = reinterpret_cast<char*>(smem);     // [NBUF][A tile | B tile]
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % tiles_n, tile_m = (wg / tiles_n) % tiles_m, split = wg / (tiles_n * tiles_m);
  const int ks = split * kchunk, ke = min(K, ks + kchunk);
  const int nk = ke > ks ? (ke - ks) / 32 : 0;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = w & 3, wn = w >> 2;
  const int m0 = tile_m * 256, n0 = tile_n * 256;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA: one instruction moves TWO whole k-rows of the tile (2 x 512 B, each contiguous in memory: lanes 0-31 row 2 id, lanes 32-63 row
  // 2 id + 1), 16 instructions per operand and tile; this wave issues ids w and w + 8 of both operands.  LDS image = plain [32 k][256 m]
  // rows of 512 B whose 64-byte chunks are XOR-swizzled with k & 3 inside every 256 bytes (the four k-rows one transpose read touches
  // fall on different banks); the swizzle is applied to the SOURCE column, the LDS position of a lane is fixed by the instruction.
  unsigned voffa[2], voffb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int krow = (i * 8 + w) * 2 + (lane >> 5);
    const int cb = ((lane & 31) * 16) ^ ((krow & 3) << 6);          // source column, bytes
    voffa[i] = (unsigned)(((long)krow * M + m0) * 2 + cb);
    voffb[i] = (unsigned)(((long)krow * N + n0) * 2 + cb);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  auto issue = [&](int kt, int buf) {
    const float* ak = reinterpret_cast<const float*>(A + (long)(ks + kt * 32) * M);
    const float* bk = reinterpret_cast<const float*>(B + (long)(ks + kt * 32) * N);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(voffb[i], bk, lds0 + buf * 2 * TILE + TILE + (i * 8 + w) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(voffa[i], ak, lds0 + buf * 2 * TILE + (i * 8 + w) * 1024);
  };
  // transpose-read addresses: a 16-lane group fetches [4 k][16 m] (lane li: k-row li / 4, 8 bytes = 4 columns at (li % 4) * 4); lane halves
  // take the k-groups {0, 1} / {2, 3} of a 16-k step (2 KB per k-group), lane quarters the two 16-row halves of a 32-row fragment
  const int g = lane >> 4, li = lane & 15;
  const int frag = (g >> 1) * 4096 + (li >> 2) * 512 + (g & 1) * 32 + (li & 3) * 8;
  int xa[2], xb[4];                                                   // swizzled 64-byte chunk of fragment i / j in this lane's k-row
#pragma unroll
  for (int i = 0; i < 2; ++i) xa[i] = frag + (((wm * 2 + i) ^ (li >> 2)) << 6);
#pragma unroll
  for (int j = 0; j < 4; ++j) xb[j] = frag + (((wn * 4 + j) ^ (li >> 2)) << 6);
  auto rd8 = [&](const char* base, int off) {
    typedef __attribute__((address_space(3))) vd_s16x4* lds_p;
    union {
      vd_s16x4 h[2];
      vd_bf16x8 v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off + 2048));
    return u.v;
  };

  // Fragments are read one 16-k step AHEAD of the MFMAs that consume them, across the tile boundary: while the matrix pipe works on step 1
  // of tile kt the transpose reads of step 0 of tile kt + 1 are in flight, so the barrier sits between two batches of ready MFMAs and the
  // two waves of a SIMD never wait for LDS at the same time.  One barrier per tile, placed after step 0: it certifies tile kt + 1 (landed,
  // every wave's share) and that nobody reads tile kt - 1 any more, whose buffer the next DMA overwrites.
  struct Frags {
    vd_bf16x8 a[2], b[4];
  };
  auto read_step = [&](Frags& f, int buf, int st) {
    const char* ta = lds + buf * 2 * TILE;
    const char* tb = ta + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = rd8(ta, xa[i] + st * 8192);
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = rd8(tb, xb[j] + st * 8192);
  };
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
  };
  auto wait_tiles_in_flight = [&](int later) {   // DMA instructions of at most `later` tiles (4 each) may still be pending
    if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
      if (i < nk) issue(i, i);
    wait_tiles_in_flight(min(NBUF - 2, nk - 1));
    asm volatile("s_barrier" ::: "memory");       // tile 0 is in LDS
    Frags f0, f1;
    read_step(f0, 0, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
      read_step(f1, buf, 1);
      mfma_step(f0);
      if (kt + 1 < nk) {
        wait_tiles_in_flight(min(NBUF - 3, nk - 2 - kt));   // tile kt + 1 has landed: only tiles kt + 2 .. kt + NBUF - 2 may be pending
        asm volatile("s_barrier" ::: "memory");
        if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);   // into the buffer of tile kt - 1
        read_step(f0, nbuf, 0);
      }
      mfma_step(f1);
      buf = nbuf;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = n0 + wn * 128 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + mfma_row(r, lane);
        unsafeAtomicAdd(C + (long)row * ldc + c, acc[i][j][r]);
      }
    }
}

static int launch_gemm_bf16_tn_tr256(const vd_bf16_bits* A, const vd_bf16_bits* B, float* C, long ldc, int M, int N, int K,
                                     hipStream_t stream) {
  const int tiles_m = M / 256, tiles_n = N / 256;
  long splits = vd_cdiv(256, (long)tiles_m * tiles_n);       // one workgroup per CU: one full round
  int kchunk = vd_cdiv(vd_cdiv(K, splits), 32) * 32;
  if (kchunk < 32) kchunk = 32;
  splits = vd_cdiv(K, kchunk);
  static bool attr_set = false;
#ifndef VD_TR256_NBUF
#define VD_TR256_NBUF 3
#endif
  constexpr int NBUF = VD_TR256_NBUF;
  const int lds = NBUF * 2 * 16384;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_tr256_kernel<NBUF>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_bf16_tn_tr256_kernel<NBUF>, dim3((unsigned)(tiles_m * tiles_n * splits)), dim3(512), lds, stream, A, B, C,
                     ldc, M, N, K, kchunk, tiles_m, tiles_n);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

static int launch_gemm_bf16_tn_tr(const vd_bf16_bits* A, const vd_bf16_bits* B, float* C, long ldc, int M, int N, int K,
                                  hipStream_t stream) {
  const int tiles_m = M / 128, tiles_n = N / 128;
  long splits = vd_cdiv(768, (long)tiles_m * tiles_n);       // one full round of workgroups (3 per CU x 256 CUs)
  int kchunk = vd_cdiv(vd_cdiv(K, splits), 32) * 32;
  if (kchunk < 32) kchunk = 32;
  splits = vd_cdiv(K, kchunk);
  static bool attr_set = false;
  const int lds = 3 * 2 * 8192;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_tr_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_bf16_tn_tr_kernel, dim3((unsigned)(tiles_m * tiles_n * splits)), dim3(256), lds, stream, A, B, C,
                     ldc, M, N, K, kchunk, tiles_m, tiles_n);
  VD_LAUNCH_CHECK();
  return VD_OK;
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
  if (M >= 1024 && K >= 64 && K % 16 == 0 && (long)M * lda * 4 < (1L << 32) && (long)N * ldw * 4 < (1L << 32)) {
    // atomic accumulation (accumulate == 2, no bias / activation) may split K: a narrow product such as dEmb += dTable * Wx^T
    // (N = 300: 267 tiles, one workgroup per CU, every K tile an exposed round trip) runs three K slices per tile instead
    int splits = 1;
    const long tiles = (long)vd_cdiv(M, 128) * vd_cdiv(N, 128);
    if (accumulate == 2 && !bias && act == VD_ACT_NONE && tiles < 512 && K >= 1024) {
      splits = (int)vd_cdiv(768, tiles);
      if (splits > K / 512) splits = K / 512;
    }
    return launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 41984>, false>(M, N, K, splits, A, lda, W, ldw, e, s, -1, splits > 1);
  }
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
  // exact-operand split (split_core.h): the big contraction of a split9 pass (dWh of the option recurrence); a ragged row tail (< 16
  // rows) goes through the fp32-MFMA kernel below
  if ((flags & VD_FLAG_SPLIT9) && !(flags & VD_FLAG_BF16) && M % SplitTnCfg::BM == 0 && N % SplitTnCfg::BN == 0 && K >= 8192 &&
      16L * lda * 4 < (1L << 31) && 16L * ldb * 4 < (1L << 31)) {
    const int K1 = K & ~15;
    if (int rc = launch_gemm_split_tn<9>(M, N, K1, A, lda, B, ldb, C, ldc, (hipStream_t)stream)) return rc;
    if (K1 == K) return VD_OK;
    SrcK a2{A + (long)K1 * lda, lda}, b2{B + (long)K1 * ldb, ldb};
    return launch_gemm<GemmCfg<4, 1, 4, 16, 0, 4, 41984>>(M, N, K - K1, 1, a2, b2, e, (hipStream_t)stream);
  }
  const bool kmaj = !(flags & VD_FLAG_BF16) && M % 128 == 0 && N % 128 == 0 && K >= 1024;
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
  long splits = vd_cdiv(768, tiles);
  const long max_splits = vd_cdiv(K, 1024);   // (sweep in the full step: 26.03 -> 25.70 ms vs 1024 blocks / 64-row slices)
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (flags & VD_FLAG_BF16) {   // opt-in reduced precision: bf16 operands, fp32 accumulation
    // both operands already exist as bf16 (shadows written by the LSTM step kernels of this pass): multiply them directly
    if (M % 128 == 0 && N % 128 == 0 && K >= 32 && lda == M && ldb == N && (long)32 * N * 2 + 256 < (1L << 31)) {
      const vd_bf16_bits* a16 = vd_bf16_shadow_find(A, (size_t)K * M);
      const vd_bf16_bits* b16 = vd_bf16_shadow_find(B, (size_t)K * N);
      if (a16 && b16) {
        const int K1 = K & ~31;     // whole 32-row K tiles; the last < 32 rows go through the staging kernel below
        if (int rc = launch_gemm_bf16_tn_tr(a16, b16, C, ldc, M, N, K1, (hipStream_t)stream)) return rc;
        if (K1 == K) return VD_OK;
        SrcK a2{A + (long)K1 * lda, lda}, b2{B + (long)K1 * ldb, ldb};
        return launch_gemm<GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>>(M, N, K - K1, 1, a2, b2, e, (hipStream_t)stream);
      }
    }
    // fp32 operands in memory (the encoder's saved state in a bf16 pass of the model-level runtime): LDS-DMA tiles of the fp32 rows, both
    // operands rounded to bf16 in registers, one v_mfma_f32_32x32x16_bf16 per 16 k = the exact-split kernel with its first product
    // only (split_core.h).  The last < 16 rows through the staging kernel.
    if (M % SplitTnCfg::BM == 0 && N % SplitTnCfg::BN == 0 && K >= 1024 && lda % 4 == 0 && ldb % 4 == 0 && 16L * lda * 4 < (1L << 31) &&
        16L * ldb * 4 < (1L << 31)) {
      const int K1 = K & ~15;
      if (int rc = launch_gemm_split_tn<1>(M, N, K1, A, lda, B, ldb, C, ldc, (hipStream_t)stream)) return rc;
      if (K1 == K) return VD_OK;
      SrcK a2{A + (long)K1 * lda, lda}, b2{B + (long)K1 * ldb, ldb};
      return launch_gemm<GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>>(M, N, K - K1, 1, a2, b2, e, (hipStream_t)stream);
    }
    return launch_gemm<GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
  }
  if (kmaj)
    return launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 41984>, true>(M, N, K, (int)splits, A, lda, B, ldb, e, (hipStream_t)stream, 0);
  // a shape the k-major pipeline does not take: the register-staged kernel
  return launch_gemm<GemmCfg<4, 1, 4, 16, 0, 4, 41984>>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
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
  long splits = vd_cdiv(768, tiles);
  const long max_splits = vd_cdiv(K, 1024);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return launch_gemm<Cfg>(M, N, K, (int)splits, a, b, e, (hipStream_t)stream);
}

// out[N] += column sums of X[M x N]  (bias gradients)
int vd_colsum_acc(const float* X, int64_t ld, int M, int N, float* out, void* stream) {
  VD_CHECK_ARG(X && out && M >= 0 && N >= 0, "vd_colsum_acc: bad args");
  if (M == 0 || N == 0) return VD_OK;
  if (N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X & 15) == 0 && M >= 64) {
    const int cb4 = vd_cdiv(N, 256);                       // 64 column groups of 4 per block
    int rb4 = vd_cdiv(2048, cb4);
    int rpb = vd_cdiv(vd_cdiv(M, rb4), 4) * 4;             // a multiple of the 4 row lanes
    if (rpb < 32) rpb = 32;
    rb4 = vd_cdiv(M, rpb);
    hipLaunchKernelGGL(colsum4_kernel, dim3(cb4, rb4), dim3(256), 0, (hipStream_t)stream, X, ld, M, N, rpb, out);
    VD_LAUNCH_CHECK();
    return VD_OK;
  }
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

// internal (common.h): C[M x N] += A16[K x M]^T * B16[K x N], both operands bf16 rows (the compact h / da of a bf16 pass)
__global__ void __launch_bounds__(256) bf16_tn_tail_kernel(const vd_bf16_bits* __restrict__ A, const vd_bf16_bits* __restrict__ B, float* __restrict__ C,
                                                           long ldc, int M, int N, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k)
    acc += __uint_as_float((unsigned)A[(long)k * M + m] << 16) * __uint_as_float((unsigned)B[(long)k * N + n] << 16);
  unsafeAtomicAdd(C + (long)m * ldc + n, acc);
}
int vd_gemm_tn_acc_bf16(const vd_bf16_bits* A16, const vd_bf16_bits* B16, float* C, int64_t ldc, int M, int N, int K, hipStream_t stream) {
  VD_CHECK_ARG(A16 && B16 && C && M % 128 == 0 && N % 128 == 0 && K >= 0, "vd_gemm_tn_acc_bf16: M, N must be multiples of 128");
  const int K1 = K & ~31;
  if (K1 > 0) {
#ifdef VD_NO_TR256   // (A/B build knob: `make variant NAME=notr256 DEFS=-DVD_NO_TR256`)
    const bool big = false;
#else
    const bool big = M % 256 == 0 && N % 256 == 0 && K1 >= 8192;
#endif
    if (int rc = big ? launch_gemm_bf16_tn_tr256(A16, B16, C, ldc, M, N, K1, stream) : launch_gemm_bf16_tn_tr(A16, B16, C, ldc, M, N, K1, stream)) return rc;
  }
  if (K1 < K) {   // the last < 32 rows
    hipLaunchKernelGGL(bf16_tn_tail_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, stream, A16 + (long)K1 * M, B16 + (long)K1 * N,
                       C, (long)ldc, M, N, K - K1);
    VD_LAUNCH_CHECK();
  }
  return VD_OK;
}
