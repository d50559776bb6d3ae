// fp32 MFMA GEMM core for gfx950 (CDNA4), written for 64-wide wavefronts.
//
//   C[M x N] = epilogue( A[M x K] * B[K x N] )
//
// * matrix op: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TF/chip).
// * block = WM x WK waves.  Wave (wm, wk) owns a 32 x (NT*32) output strip and the
//   wk-th KW-wide slice of every K tile (intra-block split-K for latency-bound
//   small-M shapes; WK == 1 for the throughput shapes).
// * both operands are staged global -> registers -> LDS as [row][k] (k contiguous,
//   row stride BK+4 floats => conflict-free ds_read_b128 fragment reads).  LDS is
//   double-buffered: one s_barrier per K tile, next tile's global loads are issued
//   before the MFMA burst of the current one.
// * k order inside each 8-wide chunk is permuted (lanes 0-31 take k0..k0+3, lanes
//   32-63 take k0+4..k0+7) so that one ds_read_b128 feeds four MFMAs.
// * sources are functors so operand fusion (dropout masks, per-image broadcast,
//   LSTM gate-column interleave) happens in the loader instead of extra HBM passes.
// * workgroup id -> tile mapping is XCD-aware (block b runs on XCD b % 8; every XCD
//   gets a contiguous range of tiles so its private L2 sees a compact working set).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

// Diagnostic build only (-DVD_TIMING, `make timing`): wave 0 of every workgroup stamps the shader clock
// (s_memtime) at the phase boundaries of gemm_block and the 100 MHz chip-wide clock (s_memrealtime) at
// its start and end; scripts/mb_c16.py reads the table back (vd_debug_timing).  Never compiled into the product.
#ifdef VD_TIMING
#define VD_TSLOTS 12
#define VD_TBLOCKS 8192
static __device__ unsigned long long vd_tbuf[VD_TBLOCKS * VD_TSLOTS];
#define VD_TSTAMP(i, clk)                                                                \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < VD_TBLOCKS) vd_tbuf[blockIdx.x * VD_TSLOTS + (i)] = (clk); \
  } while (0)
#define VD_T(i) VD_TSTAMP(i, __builtin_amdgcn_s_memtime())
#define VD_TREAL(i) VD_TSTAMP(i, __builtin_amdgcn_s_memrealtime())
#else
#define VD_T(i) do {} while (0)
#define VD_TREAL(i) do {} while (0)
#endif

// LDSMIN: request at least this much LDS per workgroup (occupancy shaping, see lstm.hip: throughput
// shapes are held to 3 workgroups per CU so that a latency-shape workgroup of another stream always fits).
// BF16: operands are rounded to bf16 (RNE) while they are staged into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the opt-in reduced-precision recurrence of BASELINE.json
// configs[4]; never used by the fp32 headline path.
template <int WM_, int WK_, int NT_, int KW_, int DB_ = 1, int MINW_ = 1, int LDSMIN_ = 0, int BF16_ = 0>
struct GemmCfg {
  static constexpr int WM = WM_, WK = WK_, NT = NT_, KW = KW_, DB = DB_, MINW = MINW_;
  static constexpr int BM = WM * 32, BN = NT * 32, BK = WK * KW;
  static constexpr int THREADS = WM * WK * 64;
  static constexpr int BF16 = BF16_;
  static constexpr int ELEM_BYTES = BF16 ? 2 : 4;
  static constexpr int STRIDE = BF16 ? BK + 8 : BK + 4;  // LDS row stride in elements (fp32 or bf16)
  static constexpr int BUF_FLOATS = (BM + BN) * STRIDE;
  static constexpr int STAGE_BYTES = (DB == 1 ? 2 : 1) * BUF_FLOATS * ELEM_BYTES;  // BUF_FLOATS counts elements
  static constexpr int RED_BYTES = (WK - 1) * WM * 16 * 64 * 4;  // one 32x32 tile per parked wave at a time
  static constexpr int EPI_BYTES = WM * 32 * 32 * 4;  // per-wave 32x32 transposition scratch for row-vectorised epilogues
  static constexpr int LDS_BYTES0 = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static constexpr int LDS_BYTES1 = LDS_BYTES0 > EPI_BYTES ? LDS_BYTES0 : EPI_BYTES;
  static constexpr int LDS_BYTES = LDS_BYTES1 > LDSMIN_ ? LDS_BYTES1 : LDSMIN_;
  static constexpr int NA = (BM * BK / 4) / THREADS;
  static constexpr int NB = (BN * BK / 4) / THREADS;
  static_assert(KW % 8 == 0, "KW must be a multiple of 8");
  static_assert(!BF16 || (KW % 16 == 0 && DB == 0), "bf16 tiles: 16 k per MFMA, single-buffer pipeline only");
  static_assert((BM * BK / 4) % THREADS == 0 && NA >= 1, "A tile must split evenly");
  static_assert((BN * BK / 4) % THREADS == 0 && NB >= 1, "B tile must split evenly");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------------------
// Operand sources.  KMAJOR == false: element (r, k) lives at p[r*ld + k] and
// ld4(r, k) returns k..k+3.  KMAJOR == true: element (r, k) lives at p[k*ld + r]
// and ld4(r, k) returns r..r+3.
// ---------------------------------------------------------------------------
struct SrcRow {
  static constexpr bool KMAJOR = false;
  static constexpr bool PLAIN = true;  // element address is affine in k: hoisted out of the K loop
  const float* p;
  long ld;
  __device__ __forceinline__ const float* ptr(int r, int k) const { return p + (long)r * ld + k; }
  __device__ __forceinline__ long kstep() const { return 1; }
  __device__ __forceinline__ float4 ld4(int r, int k) const {
    return *reinterpret_cast<const float4*>(p + (long)r * ld + k);
  }
};
struct SrcK {
  static constexpr bool KMAJOR = true;
  static constexpr bool PLAIN = true;
  const float* p;
  long ld;
  __device__ __forceinline__ const float* ptr(int r, int k) const { return p + (long)k * ld + r; }
  __device__ __forceinline__ long kstep() const { return ld; }
  __device__ __forceinline__ float4 ld4(int r, int k) const {
    return *reinterpret_cast<const float4*>(p + (long)k * ld + r);
  }
};
// k-major rows picked through an index list: element (r, k) = p[rows[k]*ld + r].  Weight gradients of the masked
// (maskZero) encoder recurrences contract only the (timestep, row) pairs that are not padding.
struct SrcKRows {
  static constexpr bool KMAJOR = true;
  static constexpr bool PLAIN = false;
  const float* p;
  long ld;
  const int* rows;
  __device__ __forceinline__ float4 ld4(int r, int k) const {
    return *reinterpret_cast<const float4*>(p + (long)rows[k] * ld + r);
  }
};
// LSTM recurrent weight Wh [H x 4H] (gate order i,f,o,g) seen through "virtual"
// columns vc = jb*128 + gate*32 + jj  ->  real column gate*H + jb*32 + jj, so one
// wave strip holds all four gates of 32 hidden units.
struct SrcKGate4 {
  static constexpr bool KMAJOR = true;
  static constexpr bool PLAIN = true;
  const float* p;
  long ld;
  int H;
  __device__ __forceinline__ const float* ptr(int vc, int k) const {
    const int jb = vc >> 7, g = (vc >> 5) & 3, jj = vc & 31;
    return p + (long)k * ld + g * H + jb * 32 + jj;
  }
  __device__ __forceinline__ long kstep() const { return ld; }
  __device__ __forceinline__ float4 ld4(int vc, int k) const {
    const int jb = vc >> 7, g = (vc >> 5) & 3, jj = vc & 31;
    return *reinterpret_cast<const float4*>(p + (long)k * ld + g * H + jb * 32 + jj);
  }
};

__device__ __forceinline__ int mfma_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Stage one 32x32 accumulator tile through the wave's private LDS scratch (1024 floats) and hand it back
// row-vectorised: out[p] = tile[row = p*8 + (lane>>3)][col = (lane&7)*4 .. +3].  Lets an epilogue issue
// 16-byte global accesses (8 lanes = one 128-byte row segment) instead of 4-byte ones.  Single wave:
// LDS operations of a wave execute in order, no barrier needed.
__device__ __forceinline__ void tile_to_rows(const f32x16& a, float* scr, int lane, float4 (&out)[4]) {
  asm volatile("" ::: "memory");  // compiler-only fences: keep the scratch writes/reads in program order
#pragma unroll
  for (int r = 0; r < 16; ++r) scr[mfma_row(r, lane) * 32 + (lane & 31)] = a[r];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int p = 0; p < 4; ++p)
    out[p] = *reinterpret_cast<const float4*>(scr + (p * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
  asm volatile("" ::: "memory");
}

// Epilogue functors may declare `struct Pre` + `preload(Pre&, row0, lane, M)`: state fetched before the K loop of the
// LDS-DMA pipeline and handed to operator() as a trailing `const Pre*` (nullptr from the other pipelines).
struct EpiNoPre {};
template <class Epi, class = void>
struct EpiPreOf {
  static constexpr bool value = false;
  using type = EpiNoPre;
};
#ifndef VD_NO_EPI_PRE   // (A/B build knob: `make variant NAME=nopre DEFS=-DVD_NO_EPI_PRE`)
template <class Epi>
struct EpiPreOf<Epi, std::void_t<typename Epi::Pre>> {
  static constexpr bool value = true;
  using type = typename Epi::Pre;
};
#endif

// Epilogue functors of the split-K latency shapes may offer a DISTRIBUTED form (HAS_DIST, DOps, dist_ok / dist_load / dist_store): the four
// K-slice waves of a 32-row tile each finish one 8-row group instead of wave 0 finishing all four.
template <class Epi, class = void>
struct EpiDistOf {
  static constexpr bool value = false;
};
template <class Epi>
struct EpiDistOf<Epi, std::void_t<typename Epi::DOps>> {
  static constexpr bool value = Epi::HAS_DIST;
};

// XCD-aware bijective remap of the flat workgroup id (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, li = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + li;
}

// Lane mapping of the k-major staging loads: piece f of a [BK k][D columns] tile -> (group of 4 columns r4, k row kk).  SIXTEEN adjacent
// lanes take sixteen consecutive k of one column group: the transposing LDS stores (ds_write_b32: two groups of 32 lanes, 32 banks) then
// write 16 consecutive floats per row group, and the two row groups of a lane group sit 4 * STRIDE = 16 (mod 32) banks apart --
// conflict-free.  (With runs of eight k the four row groups of a lane group fell on two bank sets: 2-way conflicts, 8 instead of 4 cycles
// on each of the 64 store instructions of a forward-tick iteration.)   `make variant DEFS=-DVD_KMAJ_OLD` restores runs of eight.
template <int D, int BK>
__device__ __forceinline__ void kmaj_piece(int f, int& r4, int& kk) {
#ifndef VD_KMAJ_OLD
  if constexpr (BK % 16 == 0) {
    const int klo = f & 15, g = f >> 4;
    r4 = g % (D / 4);
    kk = (g / (D / 4)) * 16 + klo;
    return;
  }
#endif
  const int klo = f & 7, g = f >> 3;
  r4 = g % (D / 4);
  kk = (g / (D / 4)) * 8 + klo;
}

// One workgroup's tile: C[row_base.., col_base..] over K range [ks, ke).  Shared by the plain and the
// grouped kernels.
template <class Cfg, class ASrc, class BSrc, class Epi>
__device__ __forceinline__ void gemm_block(int M, int N, int ks, int ke, int row_base, int col_base, int rot_seed,
                                           const ASrc& asrc, const BSrc& bsrc, const Epi& epi, float* smem,
                                           int tid_in = -1) {
  constexpr int WM = Cfg::WM, NT = Cfg::NT, KW = Cfg::KW;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
  constexpr int THREADS = Cfg::THREADS, STR = Cfg::STRIDE;
  constexpr int NA = Cfg::NA, NB = Cfg::NB;
  // tid_in: callers that loop over tiles pass a laundered thread id so that per-lane address terms are re-derived per
  // tile instead of being hoisted out of the tile loop (they would stay live across the epilogue)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wk = wave / WM;
  const int nk = ke > ks ? (ke - ks + BK - 1) / BK : 0;

  VD_T(0);
  VD_TREAL(6);
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra[NA], rb[NB];
  float4 ra2[NA], rb2[NB];  // second register stage (DB == 2 only; dead otherwise)

  // loop-invariant part of every staging load: this thread's source address at k = 0, its k offset
  // inside a tile and whether its row exists.  Per K tile only `k0 * kstep` (wave-uniform) is added.
  const float* pa[NA];
  const float* pb[NB];
  int ka[NA], kb[NB];
  if constexpr (ASrc::PLAIN) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * THREADS;
      int gr;
      if constexpr (!ASrc::KMAJOR) {
        gr = row_base + f / (BK / 4);
        ka[i] = (f % (BK / 4)) * 4;
      } else {
        int r4;
        kmaj_piece<BM, BK>(f, r4, ka[i]);
        gr = row_base + r4 * 4;
      }
      pa[i] = gr < M ? asrc.ptr(gr, ka[i]) : nullptr;
    }
  }
  if constexpr (BSrc::PLAIN) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * THREADS;
      int gr;
      if constexpr (!BSrc::KMAJOR) {
        gr = col_base + f / (BK / 4);
        kb[i] = (f % (BK / 4)) * 4;
      } else {
        int r4;
        kmaj_piece<BN, BK>(f, r4, kb[i]);
        gr = col_base + r4 * 4;
      }
      pb[i] = gr < N ? bsrc.ptr(gr, kb[i]) : nullptr;
    }
  }

  auto load_tile_into = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB]) {
    if constexpr (ASrc::PLAIN) {
      const long off = (long)k0 * asrc.kstep();
#pragma unroll
      for (int i = 0; i < NA; ++i)
        ra[i] = (pa[i] != nullptr && k0 + ka[i] < ke) ? *reinterpret_cast<const float4*>(pa[i] + off)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!ASrc::KMAJOR) {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        const int gr = row_base + r, gk = k0 + kq * 4;
        if (gr < M && gk < ke) v = asrc.ld4(gr, gk);
      } else {
        int r4, kk;
        kmaj_piece<BM, BK>(f, r4, kk);
        const int gr = row_base + r4 * 4, gk = k0 + kk;
        if (gr < M && gk < ke) v = asrc.ld4(gr, gk);
      }
      ra[i] = v;
    }
    }
    if constexpr (BSrc::PLAIN) {
      const long off = (long)k0 * bsrc.kstep();
#pragma unroll
      for (int i = 0; i < NB; ++i)
        rb[i] = (pb[i] != nullptr && k0 + kb[i] < ke) ? *reinterpret_cast<const float4*>(pb[i] + off)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!BSrc::KMAJOR) {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        const int gr = col_base + r, gk = k0 + kq * 4;
        if (gr < N && gk < ke) v = bsrc.ld4(gr, gk);
      } else {
        int r4, kk;
        kmaj_piece<BN, BK>(f, r4, kk);
        const int gr = col_base + r4 * 4, gk = k0 + kk;
        if (gr < N && gk < ke) v = bsrc.ld4(gr, gk);
      }
      rb[i] = v;
    }
    }
  };

  auto load_tile = [&](int k0) { load_tile_into(k0, ra, rb); };

  using elem_t = std::conditional_t<Cfg::BF16 != 0, __bf16, float>;
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  elem_t* const smem_e = reinterpret_cast<elem_t*>(smem);
  auto put_row4 = [&](elem_t* d, const float4& v) {  // 4 consecutive k of one row
    if constexpr (Cfg::BF16) {
      bf16x4_t t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<bf16x4_t*>(d) = t;
    } else {
      *reinterpret_cast<float4*>(d) = v;
    }
  };
  auto put_col4 = [&](elem_t* d, const float4& v) {  // one k of 4 consecutive rows (transposing store)
    d[0] = (elem_t)v.x;
    d[STR] = (elem_t)v.y;
    d[2 * STR] = (elem_t)v.z;
    d[3 * STR] = (elem_t)v.w;
  };
  auto store_tile_from = [&](int buf, const float4 (&ra)[NA], const float4 (&rb)[NB]) {
    elem_t* sa = smem_e + buf * Cfg::BUF_FLOATS;
    elem_t* sb = sa + BM * STR;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * THREADS;
      if constexpr (!ASrc::KMAJOR) {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        put_row4(sa + r * STR + kq * 4, ra[i]);
      } else {
        int r4, kk;
        kmaj_piece<BM, BK>(f, r4, kk);
        put_col4(sa + (r4 * 4) * STR + kk, ra[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * THREADS;
      if constexpr (!BSrc::KMAJOR) {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        put_row4(sb + r * STR + kq * 4, rb[i]);
      } else {
        int r4, kk;
        kmaj_piece<BN, BK>(f, r4, kk);
        put_col4(sb + (r4 * 4) * STR + kk, rb[i]);
      }
    }
  };

  // K-tile rotation: co-resident blocks start at different K offsets so that power-of-two
  // row strides do not funnel every block's loads into the same L2/HBM channel at once.
  const int rot = (rot_seed >= 0 && nk > 1) ? rot_seed % nk : 0;
  auto ktile = [&](int kt) {
    int t = kt + rot;
    if (t >= nk) t -= nk;
    return ks + t * BK;
  };
  auto store_tile = [&](int buf) { store_tile_from(buf, ra, rb); };

  const int frag_off = (lane & 31) * STR + wk * KW + (lane >> 5) * (Cfg::BF16 ? 8 : 4);
  auto mfma_tile = [&](int buf) {
    const elem_t* sa = smem_e + buf * Cfg::BUF_FLOATS + (wm * 32) * STR + frag_off;
    const elem_t* sb = smem_e + buf * Cfg::BUF_FLOATS + BM * STR + frag_off;
    if constexpr (Cfg::BF16) {
      // 32x32x16: a lane feeds 8 consecutive k (lanes 0-31: k..k+7, lanes 32-63: k+8..k+15) = one ds_read_b128
#pragma unroll
      for (int kc = 0; kc < KW / 16; ++kc) {
        const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(sa + kc * 16);
        bf16x8_t b8[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b8[j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 32 * STR + kc * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8[j], acc[j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < KW / 8; ++kc) {
        const float4 a4 = *reinterpret_cast<const float4*>(sa + kc * 8);
        float4 b4[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const float4*>(sb + j * 32 * STR + kc * 8);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[j].x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[j].y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[j].z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[j].w, acc[j], 0, 0, 0);
      }
    }
  };

  if constexpr (Cfg::DB == 2) {
    // one LDS buffer, TWO register stages: the loads of tiles k+1 and k+2 are in flight while tile k is
    // multiplied, so a global-load round trip may span two MFMA bursts
    if (nk > 0) load_tile_into(ktile(0), ra, rb);
    if (nk > 1) load_tile_into(ktile(1), ra2, rb2);
    for (int kt = 0; kt < nk; kt += 2) {
      store_tile_from(0, ra, rb);
      __syncthreads();
      if (kt == 0) VD_T(1);
      if (kt + 2 < nk) load_tile_into(ktile(kt + 2), ra, rb);
      mfma_tile(0);
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile_from(0, ra2, rb2);
        __syncthreads();
        if (kt + 3 < nk) load_tile_into(ktile(kt + 3), ra2, rb2);
        mfma_tile(0);
        __syncthreads();
      }
    }
    VD_T(2);
  } else if constexpr (Cfg::DB == 1) {
    // two LDS buffers, one barrier per K tile
    if (nk > 0) {
      load_tile(ktile(0));
      store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tile(ktile(kt + 1));
      mfma_tile(cur);
      if (kt + 1 < nk) store_tile(cur ^ 1);
      __syncthreads();
    }
  } else {
    // one LDS buffer (half the LDS => more co-resident workgroups), two barriers per K tile;
    // the next tile's global loads stay in flight in registers across the MFMA burst
    if (nk > 0) load_tile(ktile(0));
    for (int kt = 0; kt < nk; ++kt) {
      store_tile(0);
      __syncthreads();
      if (kt == 0) VD_T(1);
      if (kt + 1 < nk) load_tile(ktile(kt + 1));
      mfma_tile(0);
      __syncthreads();
    }
    VD_T(2);
  }

  if constexpr (Cfg::WK == 4 && Cfg::WM == 1 && (NT == 4 || NT == 1) && EpiDistOf<Epi>::value) {
    // distributed epilogue (see EpiLstmFwdT::dist_load): every wave requests the operands of ITS row group, then the four partial tiles of
    // each gate meet in LDS and every wave sums the 8 rows it finishes
    if (epi.dist_ok()) {
      typename Epi::DOps q;
      float4 a[NT];
      float* red = smem;                       // 4 waves x 32 x 32 floats
      static_assert(Cfg::LDS_BYTES >= 4 * 1024 * 4, "the distributed split-K epilogue exchanges 4 x 32 x 32 floats (16 KB) through LDS");
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // the operand requests go out once half of the accumulator tiles are dead (register budget: 128 VGPRs beside the throughput
        // kernels); the round trip hides under the remaining exchanges
        if (j == (NT == 4 ? 2 : 0)) epi.dist_load(q, row_base, col_base, lane, wk, M);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wk * 1024 + mfma_row(r, lane) * 32 + (lane & 31)] = acc[j][r];
        __syncthreads();
        const float* src = red + (wk * 8 + (lane >> 3)) * 32 + (lane & 7) * 4;
        float4 t = *reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(src + w * 1024);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        a[j] = t;
      }
      epi.dist_store(a, q, row_base, col_base, lane, wk, M);
      VD_T(4);
      VD_TREAL(7);
      return;
    }
  }
  if constexpr (Cfg::WK > 1) {
    // intra-block split-K: waves wk>0 park their partial sums in LDS one 32x32 tile at a time
    // (keeps the scratch at 4 KB per parked wave), wk==0 adds them.
    float* red = smem;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      __syncthreads();
      if (wk > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wk - 1) * WM + wm) * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int w = 0; w < Cfg::WK - 1; ++w)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[((w * WM + wm) * 16 + r) * 64 + lane];
      }
    }
    if (wk > 0) return;
  }

  epi(acc, row_base + wm * 32, col_base, lane, M, N, smem + wm * 1024);
  VD_T(4);
  VD_TREAL(7);
}

template <class Cfg, class ASrc, class BSrc, class Epi>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW)
gemm_f32_kernel(int M, int N, int K, int kchunk, int tiles_m, int tiles_n, int rotate, ASrc asrc,
                BSrc bsrc, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Latency shapes (intra-block split-K configs) are the short dependent launches of the encoder
  // recurrences; they share CUs with throughput-shape workgroups of other streams.  Raising their wave
  // priority lets them win MFMA/VALU issue arbitration on the SIMD (priority outranks age).
  if constexpr (Cfg::WK > 1) __builtin_amdgcn_s_setprio(3);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % tiles_n;
  const int tile_m = (wg / tiles_n) % tiles_m;
  const int split = wg / (tiles_n * tiles_m);
  const int ks = split * kchunk;
  const int ke = min(K, ks + kchunk);
  gemm_block<Cfg>(M, N, ks, ke, tile_m * Cfg::BM, tile_n * Cfg::BN, rotate ? tile_m * 5 + tile_n * 3 + split : -1,
                  asrc, bsrc, epi, smem);
}

// ---------------------------------------------------------------------------
// LDS-DMA pipeline for the throughput recurrences (both operands k-contiguous: A[M x K] rows, B given as
// its transpose Bt[N x K] rows; K % 16 == 0).  Differences from gemm_block:
//   * operand tiles go global -> LDS directly (global_load_lds_dwordx4): no staging VGPRs, no ds_write;
//   * two LDS buffers of [rows][16 floats] with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3 on
//     the SOURCE side (the DMA writes lane-linear), which keeps every ds_read_b128 fragment read
//     conflict-free without padding;
//   * ONE barrier per K tile, in the middle of the 32-MFMA burst: by then this wave has read its last
//     fragment of tile k (so buffer k&1 may be refilled with tile k+2) and its share of tile k+1 has landed
//     (so tile k+1's first fragments can be fetched under the second half of the burst).
// The waits are explicit (hipcc does not count inline-asm memory operations).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void glds16(unsigned voff, const float* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// KMAJ == false: A[M x K] rows, Bt[N x K] rows (k contiguous), tiles [row][16 k] with swizzled chunks.
// KMAJ == true : A[K x M] rows, B[K x N] rows (k = row index: weight gradients X^T * dY), tiles [16 k][128]
//                dense; fragments are ds_read_b32 (32 consecutive floats per half-wave, conflict-free).
template <class Cfg, bool KMAJ, class Epi>
__device__ __forceinline__ void gemm_block_glds(int M, int N, int ks, int ke, int row_base, int col_base,
                                                int rot_seed, const float* A, long lda, const float* B, long ldb,
                                                const Epi& epi, float* smem, int tid_in = -1) {
  constexpr int WM = Cfg::WM, NT = Cfg::NT;
  constexpr int BM = Cfg::BM, BN = Cfg::BN;
  constexpr int NIA = (BM / 16) / WM, NIB = (BN / 16) / WM;  // DMA instructions per wave per tile
  constexpr int ABUF = BM * 16, BBUF = BN * 16;               // floats per buffer
  // Buffer counts: tile kt+NB is requested in the middle of iteration kt, NB-1 iterations before its
  // first fragment read.  A gets 3 buffers, B 3 if they fit the LDS request, else 2.
  constexpr int NBA = (3 * ABUF + 2 * BBUF) * 4 <= Cfg::LDS_BYTES ? 3 : 2;
  constexpr int NBB = (NBA == 3 && (3 * ABUF + 3 * BBUF) * 4 <= Cfg::LDS_BYTES) ? 3 : 2;
  static_assert(Cfg::WK == 1 && Cfg::KW == 16, "LDS-DMA pipeline: BK = 16, no intra-block split-K");
  static_assert((BM / 16) % WM == 0 && (BN / 16) % WM == 0, "16-row DMA groups must split evenly over the waves");
  static_assert((NBA * ABUF + NBB * BBUF) * 4 <= Cfg::LDS_BYTES, "DMA buffers must fit the LDS request");
  static_assert(!KMAJ || (BM == 128 && BN == 128), "k-major tiles: one DMA instruction = two 128-float k rows");
  static_assert(!KMAJ || Cfg::BF16 == 0, "bf16 operands: row-major (k-contiguous) tiles only");
  // tid_in: callers that loop over tiles pass a laundered thread id so per-lane address terms are re-derived
  // per tile instead of being hoisted out of the tile loop (they would stay live across the epilogue)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = ke > ks ? (ke - ks) / 16 : 0;
  VD_T(0);
  VD_TREAL(6);

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // epilogues with a dependent first load (row -> token id -> projection row) fetch the ids HERE: the round trip hides
  // under the whole K loop instead of standing in front of the epilogue's first row loads
  typename EpiPreOf<Epi>::type pre;
  if constexpr (EpiPreOf<Epi>::value) epi.preload(pre, row_base + wm * 32, lane, M);

  // per-thread source offsets (bytes, first K tile) of this wave's DMA instructions
  unsigned voffa[NIA], voffb[NIB];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    if constexpr (KMAJ) {
      const int krow = (i * WM + wm) * 2 + (lane >> 5);
      voffa[i] = (unsigned)(((long)krow * lda + row_base + (lane & 31) * 4) * 4);
    } else {
      const int r = (i * WM + wm) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);  // chunk index swizzled on the source side
      voffa[i] = (unsigned)(((long)min(row_base + r, M - 1) * lda + c * 4) * 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    if constexpr (KMAJ) {
      const int krow = (i * WM + wm) * 2 + (lane >> 5);
      voffb[i] = (unsigned)(((long)krow * ldb + col_base + (lane & 31) * 4) * 4);
    } else {
      const int r = (i * WM + wm) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      voffb[i] = (unsigned)(((long)min(col_base + r, N - 1) * ldb + c * 4) * 4);
    }
  }
  float* const sA = smem;
  float* const sB = smem + NBA * ABUF;
  const unsigned ldsA = (unsigned)(uintptr_t)sA, ldsB = (unsigned)(uintptr_t)sB;  // LDS byte addresses
  const int rot = (rot_seed >= 0 && nk > 1) ? rot_seed % nk : 0;
  auto koff = [&](int kt) {
    int t = kt + rot;
    if (t >= nk) t -= nk;
    return ks + t * 16;
  };
  auto issue_a = [&](int kt, int buf) {
    const float* ak = A + (KMAJ ? (long)koff(kt) * lda : (long)koff(kt));
#pragma unroll
    for (int i = 0; i < NIA; ++i) glds16(voffa[i], ak, ldsA + buf * (ABUF * 4) + (i * WM + wm) * 1024);
  };
  auto issue_b = [&](int kt, int buf) {
    const float* bk = B + (KMAJ ? (long)koff(kt) * ldb : (long)koff(kt));
#pragma unroll
    for (int i = 0; i < NIB; ++i) glds16(voffb[i], bk, ldsB + buf * (BBUF * 4) + (i * WM + wm) * 1024);
  };

  // fragments of half a tile (8 k values = 4 MFMA k-steps): af[s] / bf[j][s] feed k-step s
  struct Frag {
    float a[4];
    float b[NT][4];
  };
  const int hi = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 2) & 3;
  auto read_frags = [&](int bufa, int bufb, int half, Frag& f) {
    if constexpr (KMAJ) {
      const float* sa = sA + bufa * ABUF + (half * 8 + hi) * BM + wm * 32 + l31;
      const float* sb = sB + bufb * BBUF + (half * 8 + hi) * BN + l31;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f.a[s4] = sa[s4 * 2 * BM];
#pragma unroll
        for (int j = 0; j < NT; ++j) f.b[j][s4] = sb[s4 * 2 * BN + j * 32];
      }
    } else {
      const int ch = ((half * 2 + hi) ^ sw) * 4;
      const float4 a4 = *reinterpret_cast<const float4*>(sA + bufa * ABUF + (wm * 32 + l31) * 16 + ch);
      f.a[0] = a4.x; f.a[1] = a4.y; f.a[2] = a4.z; f.a[3] = a4.w;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float4 b4 = *reinterpret_cast<const float4*>(sB + bufb * BBUF + (j * 32 + l31) * 16 + ch);
        f.b[j][0] = b4.x; f.b[j][1] = b4.y; f.b[j][2] = b4.z; f.b[j][3] = b4.w;
      }
    }
  };
  auto mfma16 = [&](const Frag& f) {
    if constexpr (Cfg::BF16 != 0) {
      // bf16 operands (A, B point at bf16 rows; all K arguments count 4-byte units = bf16 PAIRS): a 16-byte chunk is 8
      // consecutive k = one operand of v_mfma_f32_32x32x16_bf16 (lanes 0-31: k .. k+7, lanes 32-63: k+8 .. k+15), so the
      // same tiles, the same swizzle and the same reads feed ONE instruction per column tile and half tile
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t af = {f.a[0], f.a[1], f.a[2], f.a[3]};
      const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, af);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4_t bf = {f.b[j][0], f.b[j][1], f.b[j][2], f.b[j][3]};
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8_t, bf), acc[j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s4], f.b[j][s4], acc[j], 0, 0, 0);
    }
  };
  // s_waitcnt needs an immediate.  n = DMA instructions of this wave allowed to stay in flight.
  auto wait_vm = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  if (nk > 0) {
    Frag f0, f1;
    // request order inside every group is B first, then A: the operand with fewer buffers is always the
    // older request, so "tile kt+1 of both operands has landed" == "at most the younger A requests are left"
    issue_b(0, 0);
    issue_a(0, 0);
#pragma unroll
    for (int b = 1; b < NBA; ++b) {
      if (b < NBB && b < nk) issue_b(b, b);
      if (b < nk) issue_a(b, b);
    }
    wait_vm(NIA * (min(nk, NBA) - 1) + NIB * (min(nk, NBB) - 1));
    asm volatile("s_barrier" ::: "memory");
    VD_T(1);
    VD_TREAL(8);
    read_frags(0, 0, 0, f0);
    int ca = 0, cb = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int na = (ca + 1 == NBA) ? 0 : ca + 1, nb = (cb + 1 == NBB) ? 0 : cb + 1;
      read_frags(ca, cb, 1, f1);
      mfma16(f0);
      // tile kt fully consumed by this wave, its share of tile kt+1 landed -> meet the other waves.
      // (sched_barrier: MFMAs carry no side effects, the scheduler would otherwise sink the burst below
      // the barrier and expose the fragment-read latency in front of it)
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NBA <= nk) {  // steady state: constant allowance, one immediate
        constexpr int STEADY = NIA * (NBA - 2) + NIB * (NBB - 2);
        if constexpr (STEADY == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (STEADY == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if constexpr (STEADY == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (STEADY == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (STEADY == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (STEADY == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (STEADY == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else static_assert(STEADY <= 6, "steady-state vmcnt immediate");
      } else {
        wait_vm(NIA * max(0, min(kt + NBA - 1, nk - 1) - (kt + 1)) + NIB * max(0, min(kt + NBB - 1, nk - 1) - (kt + 1)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NBB < nk) issue_b(kt + NBB, cb);
      if (kt + NBA < nk) issue_a(kt + NBA, ca);
      if (kt + 1 < nk) read_frags(na, nb, 0, f0);
      mfma16(f1);
      ca = na;
      cb = nb;
    }
    VD_T(2);
    VD_TREAL(9);
  }
  if constexpr (NT == 8) {
    // 256-column tiles (two gate groups of 32 hidden units per wave, 128 accumulator registers): the epilogue functors
    // are written for ONE group of four 32 x 32 tiles and run once per group
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (col_base + g * 128 >= N) break;
      const f32x16 half[4] = {acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]};
      if constexpr (EpiPreOf<Epi>::value) epi(half, row_base + wm * 32, col_base + g * 128, lane, M, N, smem + wm * 1024, &pre);
      else epi(half, row_base + wm * 32, col_base + g * 128, lane, M, N, smem + wm * 1024);
    }
  } else if constexpr (EpiPreOf<Epi>::value) epi(acc, row_base + wm * 32, col_base, lane, M, N, smem + wm * 1024, &pre);
  else epi(acc, row_base + wm * 32, col_base, lane, M, N, smem + wm * 1024);
  VD_T(4);
  VD_TREAL(7);
}

template <class Cfg, bool KMAJ, class Epi>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW)
gemm_f32_glds_kernel(int M, int N, int K, int kchunk, int tiles_m, int tiles_n, int rotate, const float* A, long lda,
                     const float* B, long ldb, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % tiles_n, tile_m = (wg / tiles_n) % tiles_m, split = wg / (tiles_n * tiles_m);
  const int ks = split * kchunk, ke = min(K, ks + kchunk);
  gemm_block_glds<Cfg, KMAJ>(M, N, ks, ke, tile_m * Cfg::BM, tile_n * Cfg::BN,
                             rotate ? tile_m * 5 + tile_n * 3 + split : -1, A, lda, B, ldb, epi, smem);
}

// KMAJ == false: C[M x N] = epi(A[M x K] * Bt[N x K]^T), K % 16 == 0, row byte offsets below 4 GB.
// KMAJ == true : C[M x N] (+)= epi over `splits` K ranges of A[K x M]^T * B[K x N]; M, N % 128 == 0, K % 16 == 0.
// (row-major operands take splits > 1 only with `atomic_epi`: the caller vouches that the epilogue ACCUMULATES atomically)
template <class Cfg, bool KMAJ, class Epi>
static int launch_gemm_glds(int M, int N, int K, int splits, const float* A, long lda, const float* B, long ldb,
                            const Epi& e, hipStream_t stream, int rotate_in = -1, bool atomic_epi = false) {
  if (M <= 0 || N <= 0) return VD_OK;
  VD_CHECK_ARG(K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && splits >= 1,
               "launch_gemm_glds: unsupported shape M=%d N=%d K=%d", M, N, K);
  if (KMAJ)
    VD_CHECK_ARG(M % Cfg::BM == 0 && N % Cfg::BN == 0, "launch_gemm_glds: k-major tiles need M, N %% 128 == 0");
  else
    VD_CHECK_ARG((long)M * lda * 4 < (1L << 32) && (long)N * ldb * 4 < (1L << 32) && (splits == 1 || atomic_epi),
                 "launch_gemm_glds: row offsets beyond 32 bits, or split-K without an accumulating epilogue");
  const int tiles_m = vd_cdiv(M, Cfg::BM), tiles_n = vd_cdiv(N, Cfg::BN);
  int kchunk = vd_cdiv(vd_cdiv(K, splits), 16) * 16;
  if (kchunk < 16) kchunk = 16;
  splits = vd_cdiv(K, kchunk);
  auto kern = gemm_f32_glds_kernel<Cfg, KMAJ, Epi>;
  static bool attr_set = false;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    attr_set = true;
  }
  const int rotate = rotate_in >= 0 ? rotate_in : 1;
  const int grid = tiles_m * tiles_n * splits;
  const int lds = Cfg::LDS_BYTES;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), lds, stream, M, N, K,
                     kchunk, tiles_m, tiles_n, rotate, A, lda, B, ldb, e);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// ---------------------------------------------------------------------------
// Grouped launch: up to MAXP independent problems (same tile config and functor types, different
// pointers / dims) in ONE launch -- used to advance several recurrences by one tick per launch.
// ---------------------------------------------------------------------------
template <class Prob, int MAXP>
struct GroupArgs {
  Prob p[MAXP];
  int tile_start[MAXP + 1];
  int nprob;
};

template <class Cfg, class Prob, int MAXP>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW)
gemm_f32_grouped_kernel(GroupArgs<Prob, MAXP> g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (Cfg::WK > 1) __builtin_amdgcn_s_setprio(3);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i)
    if (i < g.nprob && wg >= g.tile_start[i]) pi = i;
  const Prob& P = g.p[pi];
  const int t = wg - g.tile_start[pi];
  const int tile_n = t % P.tiles_n, tile_m = t / P.tiles_n;
  gemm_block<Cfg>(P.M, P.N, 0, P.K, tile_m * Cfg::BM, tile_n * Cfg::BN, -1, P.a, P.b, P.e, smem);
}

template <class Cfg, class Prob, int MAXP>
static int launch_grouped(GroupArgs<Prob, MAXP>& g, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < g.nprob; ++i) {
    g.p[i].tiles_n = vd_cdiv(g.p[i].N, Cfg::BN);
    g.tile_start[i] = total;
    total += vd_cdiv(g.p[i].M, Cfg::BM) * g.p[i].tiles_n;
  }
  g.tile_start[g.nprob] = total;
  if (total == 0) return VD_OK;
  auto kern = gemm_f32_grouped_kernel<Cfg, Prob, MAXP>;
  static bool attr_set = false;
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Cfg::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, g);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// ---------------------------------------------------------------------------
// Generic epilogues
// ---------------------------------------------------------------------------
enum { VD_ACT_NONE = 0, VD_ACT_TANH = 1 };

// C = act(acc + bias[col]) (or C += ...).  One block owns each C element (no split-K).
template <int NT>
struct EpiStore {
  float* C;
  long ldc;
  const float* bias;  // nullable
  int act;
  int accumulate;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M,
                                             int N, float* /*scr*/ = nullptr) const {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col0 + j * 32 + (lane & 31);
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + mfma_row(r, lane);
        if (row >= M) continue;
        float v = acc[j][r] + bv;
        if (act == VD_ACT_TANH) v = tanhf(v);
        float* d = C + (long)row * ldc + col;
        if (accumulate == 2) {   // C += ... with a hardware float atomic: safe beside other (atomic) writers of C
          atomicAdd(d, v);
          continue;
        }
        if (accumulate) v += *d;
        *d = v;
      }
    }
  }
};

// C += acc with hardware float atomics (split-K partial sums into a gradient buffer).
template <int NT>
struct EpiAtomic {
  float* C;
  long ldc;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M,
                                             int N, float* /*scr*/ = nullptr) const {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col0 + j * 32 + (lane & 31);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + mfma_row(r, lane);
        if (row >= M) continue;
        unsafeAtomicAdd(C + (long)row * ldc + col, acc[j][r]);
      }
    }
  }
};

// host-side launcher
template <class Cfg, class ASrc, class BSrc, class Epi>
static int launch_gemm(int M, int N, int K, int splits, ASrc a, BSrc b, Epi e, hipStream_t stream) {
  if (M <= 0 || N <= 0) return VD_OK;
  const int tiles_m = vd_cdiv(M, Cfg::BM), tiles_n = vd_cdiv(N, Cfg::BN);
  if (splits < 1) splits = 1;
  int kchunk = K > 0 ? vd_cdiv(vd_cdiv(K, splits), Cfg::BK) * Cfg::BK : Cfg::BK;
  if (K > 0) splits = vd_cdiv(K, kchunk);
  else splits = 1;
  auto kern = gemm_f32_kernel<Cfg, ASrc, BSrc, Epi>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Cfg::LDS_BYTES));
    attr_set = true;
  }
  const int grid = tiles_m * tiles_n * splits;
  const int rotate = (1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, M, N, K, kchunk, tiles_m,
                     tiles_n, rotate, a, b, e);
  VD_LAUNCH_CHECK();
  return VD_OK;
}
