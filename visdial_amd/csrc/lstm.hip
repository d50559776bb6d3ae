// Fused SeqLSTM timestep kernels (forward + backward) and the sequence drivers.
//
// Semantics restated from Element-Research rnn `nn.SeqLSTM` as used by the reference
// (encoders/mn-att-ques-im-hist.lua:27-45, decoders/disc.lua:4-15, decoders/gen.lua:17-22;
// SURVEY.md App. A1): weight [(D+H) x 4H] with gate column order i,f,o,g; per step
//   a = b + x_t*Wx + h_{t-1}*Wh ; i,f,o = sigmoid ; g = tanh ; c = f*c_prev + i*g ; h = o*tanh(c)
// maskZero(): rows whose step input is all-zero (<=> token id 0) get h = c = gates = 0.
//
// MI355X mapping: the input projection x_t*Wx + b is hoisted out of the recurrence
// (one batched GEMM, or a gather from the [V+1 x 4H] table Emb*Wx+b for the option
// LSTM), so a timestep is ONE kernel: h_{t-1}[N x H] * Wh[H x 4H] on fp32 MFMA with
// the gate columns interleaved so each wave strip holds i,f,o,g of 32 hidden units,
// and the whole cell update (+ mask) runs in the epilogue on the accumulator registers.
#include "gemm_core.h"

// defined in gemm_ops.hip (declared in include/visdial_hip.h)
extern "C" int vd_gemm_tn_acc(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                              int N, int K, int flags, void* stream);

// ---------------------------------------------------------------------------
// forward epilogue: acc[g] = (h_prev*Wh)[row, g*H + j]
// ---------------------------------------------------------------------------
template <int SEQ>
struct EpiLstmFwdT {
  const float* xproj;  // dense: [N x 4H] rows (ld = xld); table mode: table base [V+1 x 4H]
  long xld;
  const int* tok_gather;  // nullable; if set, x row = xproj + tok_gather[row]*xld
  const int* tok_mask;    // nullable; row is masked (zero state) when tok_mask[row] == 0
  const float* c_prev;    // nullable -> zeros
  float* gates;           // [N x 4H] post-activation i,f,o,g
  float* c_out;           // [N x H]
  float* h_out;           // [N x H]
  int H;
  vd_bf16_bits* h16 = nullptr;   // nullable: bf16 copy of h_out (bf16 pass: operand of the weight-gradient contraction)
  // SEQ = 1: compiler-scheduled epilogue of round 1 (A/B build, knob VD_LSTM_FWD_EPI_SEQ)
  // The four accumulator tiles are i,f,o,g of hidden units [j0, j0+32).  Each is staged through the wave's LDS
  // scratch so that a lane ends up with 4 consecutive hidden units of one row: every global access of the cell
  // update is a 16-byte one (8 lanes = one 128-byte segment of a row).
  //
  // Memory-level parallelism is scheduled by hand: a lane serves 4 rows (p = 0..3) and every row needs a dependent
  // chain token id -> projection row -> math -> stores.  Left to the compiler this became 4 x (id load, wait, 5 row
  // loads, wait, math, stores): 8 serialised round trips, about as long as the whole K loop of the tile (the
  // workgroup's matrix pipe share sits idle meanwhile).  Here the 4 token / mask ids are fetched in ONE batch whose
  // latency hides under the LDS transposes, and the row loads of p+1 are issued as soon as the loaded values of p
  // have been folded into the pre-activations (one buffer of 20 VGPRs, reused), so they fly under the
  // transcendental math and the stores of p.
  // round-1 form: one row at a time, loads / waits / math / stores in the order the compiler picks
  __device__ __forceinline__ void sequential(const f32x16 (&acc)[4], int row0, int vcol0, int lane, int M, float* scr) const {
    // The four accumulator tiles are i,f,o,g of hidden units [j0, j0+32).  Each is staged through the
    // wave's LDS scratch so that a lane ends up with 4 consecutive hidden units of one row: every global
    // access of the cell update is then a 16-byte one (8 lanes = one 128-byte segment of a row).
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;
    float4 ai[4], af[4], ao[4], ag[4];
    tile_to_rows(acc[0], scr, lane, ai);
    tile_to_rows(acc[1], scr, lane, af);
    tile_to_rows(acc[2], scr, lane, ao);
    tile_to_rows(acc[3], scr, lane, ag);
    VD_T(3);
    if (j >= H) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + (lane >> 3);
      if (row >= M) continue;
      float4 gi, gf, go, gg, c, h;
      if (tok_mask && tok_mask[row] == 0) {
        gi = gf = go = gg = c = h = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const float* xr = xproj + (tok_gather ? (long)tok_gather[row] : (long)row) * xld + j;
        const float4 xi = *reinterpret_cast<const float4*>(xr);
        const float4 xf = *reinterpret_cast<const float4*>(xr + H);
        const float4 xo = *reinterpret_cast<const float4*>(xr + 2 * H);
        const float4 xg = *reinterpret_cast<const float4*>(xr + 3 * H);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_prev) cp = *reinterpret_cast<const float4*>(c_prev + (long)row * H + j);
#define VD_CELL(E)                                        \
        gi.E = vd_sigmoid(ai[p].E + xi.E);                \
        gf.E = vd_sigmoid(af[p].E + xf.E);                \
        go.E = vd_sigmoid(ao[p].E + xo.E);                \
        gg.E = vd_tanh(ag[p].E + xg.E);                   \
        c.E = gf.E * cp.E + gi.E * gg.E;                  \
        h.E = go.E * vd_tanh(c.E);
        VD_CELL(x) VD_CELL(y) VD_CELL(z) VD_CELL(w)
#undef VD_CELL
      }
      float* gr = gates + (long)row * 4 * H + j;
      *reinterpret_cast<float4*>(gr) = gi;
      *reinterpret_cast<float4*>(gr + H) = gf;
      *reinterpret_cast<float4*>(gr + 2 * H) = go;
      *reinterpret_cast<float4*>(gr + 3 * H) = gg;
      *reinterpret_cast<float4*>(c_out + (long)row * H + j) = c;
      *reinterpret_cast<float4*>(h_out + (long)row * H + j) = h;
      if (h16) vd_st4_bf16(h16 + (long)row * H + j, h);
    }
  }
  // token / mask ids of the lane's 4 rows, fetched before the K loop by the LDS-DMA pipeline (gemm_block_glds): the
  // dependent chain row -> token id -> projection row loses its first round trip (~3 us under load)
  struct Pre {
    int tk[4];
    int keepbits;
  };
  __device__ __forceinline__ void preload(Pre& q, int row0, int lane, int M) const {
    q.keepbits = 15;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + (lane >> 3);
      q.tk[p] = row < M ? row : M - 1;
    }
    if (tok_gather) {
#pragma unroll
      for (int p = 0; p < 4; ++p) q.tk[p] = tok_gather[q.tk[p]];
    }
    if (tok_mask) {
      int kb = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = row0 + p * 8 + (lane >> 3);
        kb |= (tok_mask[row < M ? row : M - 1] != 0) << p;
      }
      q.keepbits = kb;
    }
  }
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int vcol0, int lane, int M,
                                             int /*Nv*/, float* scr, const Pre* pre = nullptr) const {
    if constexpr (SEQ == 2) {
      // DIAGNOSTIC build (knob VD_LSTM_FWD_EPI_SEQ=2, scripts/microbench.py): no epilogue loads, no gate math, one
      // 16-byte store per row -- isolates the K loop of the step kernel.  Results are garbage by construction.
      float4 a0[4];
      const f32x16 sum = acc[0] + acc[1] + acc[2] + acc[3];   // keeps every MFMA of the K loop live
      tile_to_rows(sum, scr, lane, a0);
      const int j0 = (vcol0 >> 7) * 32 + (lane & 7) * 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = row0 + p * 8 + (lane >> 3);
        if (row < M) *reinterpret_cast<float4*>(h_out + (long)row * H + j0) = a0[p];
      }
      return;
    }
    if constexpr (SEQ == 1) {
      sequential(acc, row0, vcol0, lane, M, scr);
      return;
    }
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;  // < H: H % 32 == 0 and vcol0 < 4H
    const int rl = lane >> 3;
    int rowc[4], tk[4], keep[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + rl;
      rowc[p] = row < M ? row : M - 1;  // rows past the end load from a valid address and are not stored
      tk[p] = rowc[p];
      keep[p] = 1;
    }
    // (one uniform branch per id array with its 4 loads back to back: a per-row `ptr ? ptr[row] : row` select made
    //  the compiler wait for each load before the next)
    if (pre) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        tk[p] = pre->tk[p];
        keep[p] = (pre->keepbits >> p) & 1;
      }
    } else {
      if (tok_gather) {
#pragma unroll
        for (int p = 0; p < 4; ++p) tk[p] = tok_gather[rowc[p]];
      }
      if (tok_mask) {
#pragma unroll
        for (int p = 0; p < 4; ++p) keep[p] = tok_mask[rowc[p]];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 a[4][4];  // [gate][p]
    tile_to_rows(acc[0], scr, lane, a[0]);
    tile_to_rows(acc[1], scr, lane, a[1]);
    tile_to_rows(acc[2], scr, lane, a[2]);
    tile_to_rows(acc[3], scr, lane, a[3]);
    VD_T(3);
    float4 x[4], cp;
#if VD_EPI_BUF & 1
    // descriptors (SGPRs) + 32-bit byte offsets: per row group one multiply-add for the gathered projection row and one for the
    // clamped state row; the gate stride and the row-group stride are SGPR offsets
    const __amdgpu_buffer_rsrc_t rx = vd_rsrc(xproj), rc = vd_rsrc(c_prev), rg = vd_rsrc(gates), rco = vd_rsrc(c_out),
                                 rh = vd_rsrc(h_out), rh16 = vd_rsrc(h16);
    const unsigned uH4 = (unsigned)H * 4u, uj4 = (unsigned)j * 4u, uxld4 = (unsigned)xld * 4u;
    const unsigned vrow = ((unsigned)(row0 + rl)) * uH4 + uj4;          // byte offset of (row0 + rl, j) in an [M x H] tensor
    auto issue = [&](int p) {
      const unsigned xo = (unsigned)tk[p] * uxld4 + uj4;
      x[0] = vd_buf_ld4(rx, xo, 0);
      x[1] = vd_buf_ld4(rx, xo, uH4);
      x[2] = vd_buf_ld4(rx, xo, 2 * uH4);
      x[3] = vd_buf_ld4(rx, xo, 3 * uH4);
      cp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c_prev) cp = vd_buf_ld4(rc, (unsigned)rowc[p] * uH4 + uj4, 0);
    };
#else
    auto issue = [&](int p) {
      const float* xr = xproj + (long)tk[p] * xld + j;
      x[0] = *reinterpret_cast<const float4*>(xr);
      x[1] = *reinterpret_cast<const float4*>(xr + H);
      x[2] = *reinterpret_cast<const float4*>(xr + 2 * H);
      x[3] = *reinterpret_cast<const float4*>(xr + 3 * H);
      cp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c_prev) cp = *reinterpret_cast<const float4*>(c_prev + (long)rowc[p] * H + j);
    };
#endif
    issue(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // fold the loaded values into the pre-activations; x / cp are dead afterwards
      float4 pi = a[0][p], pf = a[1][p], po = a[2][p], pg = a[3][p];
      pi.x += x[0].x; pi.y += x[0].y; pi.z += x[0].z; pi.w += x[0].w;
      pf.x += x[1].x; pf.y += x[1].y; pf.z += x[1].z; pf.w += x[1].w;
      po.x += x[2].x; po.y += x[2].y; po.z += x[2].z; po.w += x[2].w;
      pg.x += x[3].x; pg.y += x[3].y; pg.z += x[3].z; pg.w += x[3].w;
      const float4 cq = cp;
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < 4) issue(p + 1);  // in flight under the math and stores below
      __builtin_amdgcn_sched_barrier(0);
      const float km = keep[p] != 0 ? 1.f : 0.f;  // maskZero(): h = c = gates = 0 for pad rows
      float4 gi, gf, go, gg, c, h;
#define VD_CELL(E)                                  \
      gi.E = km * vd_sigmoid(pi.E);                 \
      gf.E = km * vd_sigmoid(pf.E);                 \
      go.E = km * vd_sigmoid(po.E);                 \
      gg.E = km * vd_tanh(pg.E);                    \
      c.E = gf.E * cq.E + gi.E * gg.E;              \
      h.E = go.E * vd_tanh(c.E);
      VD_CELL(x) VD_CELL(y) VD_CELL(z) VD_CELL(w)
#undef VD_CELL
      const int row = row0 + p * 8 + rl;
      if (row < M) {
#if VD_EPI_BUF & 1
        const unsigned sp = (unsigned)p * 8u * uH4;                  // row-group stride (uniform)
        const unsigned vg = 4u * vrow - 3u * uj4;                    // (row0 + rl, j) in the [M x 4H] gates tensor
        vd_buf_st4(rg, vg, 4u * sp, gi);                              // saved for the backward pass
        vd_buf_st4(rg, vg, 4u * sp + uH4, gf);
        vd_buf_st4(rg, vg, 4u * sp + 2 * uH4, go);
        vd_buf_st4(rg, vg, 4u * sp + 3 * uH4, gg);
        vd_buf_st4(rco, vrow, sp, c);
        vd_buf_st4(rh, vrow, sp, h);
        if (h16) vd_buf_st4_bf16(rh16, vrow >> 1, sp >> 1, h);
#else
        float* gr = gates + (long)row * 4 * H + j;
        vd_st4_stream(gr, gi);           // saved for the backward pass: written once, read ~10 ms later
        vd_st4_stream(gr + H, gf);
        vd_st4_stream(gr + 2 * H, go);
        vd_st4_stream(gr + 3 * H, gg);
        *reinterpret_cast<float4*>(c_out + (long)row * H + j) = c;
        *reinterpret_cast<float4*>(h_out + (long)row * H + j) = h;
        if (h16) vd_st4_bf16(h16 + (long)row * H + j, h);
#endif
      }
    }
  }
};

using EpiLstmFwd = EpiLstmFwdT<0>;

struct EpiLstmFwdSpread;
// DIAGNOSTIC (knob VD_LSTM_FWD_EPI_SEQ=4): as EpiLstmFwdSpread below, but the traffic is FIRE-AND-FORGET -- the gather goes
// global -> LDS by DMA into 1 KB of spare LDS (no register, no consumer, hence no wait), the store writes a constant -- so the
// only thing measured is whether epilogue-sized traffic issued smoothly from inside the K loop slows the K loop.
template <int MODE>   // 1 = the DMA gathers only, 2 = the stores only, 3 = both
struct EpiLstmFwdSpreadNoWaitT {
  const float* xproj;
  long xld;
  const int* tok_gather;
  float* gates;
  float* h_out;
  int H;
  static constexpr int KHOOK_VM_OPS = MODE == 3 ? 2 : 1;
  struct KHook {
    int tk[4];
  };
  __device__ __forceinline__ void khook_init(KHook& s) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.tk[i] = 0;
  }
  __device__ __forceinline__ void khook(KHook& s, int kt, int nk, int row0, int vcol0, int lane, int M, unsigned lds_spare) const {
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;
    const int row = min(row0 + (kt & 3) * 8 + (lane >> 3), M - 1);
    const int g = (kt >> 2) & 3;
    if (kt < 4) s.tk[kt & 3] = tok_gather ? tok_gather[row] : row;      // 4 id loads per tile, like the real epilogue
    const unsigned voff = (unsigned)(((long)s.tk[kt & 3] * xld + g * H + j) * 4);
    if constexpr (MODE & 1) glds16(voff, xproj, lds_spare);
    if constexpr (MODE & 2) *reinterpret_cast<float4*>(gates + (long)row * 4 * H + g * H + j) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void khook_finish(KHook&, int, int, int, int) const {}
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int vcol0, int lane, int M, int /*Nv*/,
                                             float* scr) const {
    float4 a0[4];
    const f32x16 sum = acc[0] + acc[1] + acc[2] + acc[3];
    tile_to_rows(sum, scr, lane, a0);
    const int j0 = (vcol0 >> 7) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + (lane >> 3);
      if (row < M) *reinterpret_cast<float4*>(h_out + (long)row * H + j0) = a0[p];
    }
  }
};

// DIAGNOSTIC (knob VD_LSTM_FWD_EPI_SEQ=3, scripts/mb_r3.py): the K loop of the forward step kernel with memory traffic of the
// real epilogue's volume and kind ISSUED FROM INSIDE the K loop -- per K tile and lane one 16-byte gather from the projection
// table (a pseudo-random row) and one 16-byte store into the gates buffer, the loaded value consumed 4 K tiles later -- and the
// minimal tail of the K-loop-only build.  Answers one question: does epilogue traffic that is spread over the K loop cost the
// matrix pipe what the epilogue after the K loop costs?  Results are garbage by construction.
struct EpiLstmFwdSpread {
  const float* xproj;
  long xld;
  const int* tok_gather;
  float* gates;
  float* h_out;
  int H;
  static constexpr int KHOOK_VM_OPS = 0;   // its loads are consumed (compiler-managed waits): nothing fire-and-forget
  struct KHook {
    float4 ring[4];
    float4 sum;
  };
  __device__ __forceinline__ void khook_init(KHook& s) const {
    s.sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.ring[i] = s.sum;
  }
  __device__ __forceinline__ void khook(KHook& s, int kt, int nk, int row0, int vcol0, int lane, int M, unsigned /*lds_spare*/) const {
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;
    const int row = min(row0 + (kt & 3) * 8 + (lane >> 3), M - 1);
    const int g = (kt >> 2) & 3;
    // consume the load of 4 K tiles ago, then reuse its slot
    const float4 old = s.ring[kt & 3];
    s.sum.x += old.x; s.sum.y += old.y; s.sum.z += old.z; s.sum.w += old.w;
    const long trow = tok_gather ? (long)tok_gather[row] : (long)row;
    s.ring[kt & 3] = *reinterpret_cast<const float4*>(xproj + trow * xld + g * H + j);
    *reinterpret_cast<float4*>(gates + (long)row * 4 * H + g * H + j) = s.sum;
  }
  __device__ __forceinline__ void khook_finish(KHook& s, int row0, int vcol0, int lane, int M) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.sum.x += s.ring[i].x; s.sum.y += s.ring[i].y; s.sum.z += s.ring[i].z; s.sum.w += s.ring[i].w; }
  }
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int vcol0, int lane, int M, int /*Nv*/,
                                             float* scr) const {
    float4 a0[4];
    const f32x16 sum = acc[0] + acc[1] + acc[2] + acc[3];
    tile_to_rows(sum, scr, lane, a0);
    const int j0 = (vcol0 >> 7) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + (lane >> 3);
      if (row < M) *reinterpret_cast<float4*>(h_out + (long)row * H + j0) = a0[p];
    }
  }
};
// DIAGNOSTIC (knob VD_LSTM_FWD_EPI_SEQ=7 / 8, scripts/mb_r3.py): the K loop of the forward step kernel plus the real epilogue's
// ARITHMETIC and nothing else -- per tile and lane 80 v_exp_f32 + 80 v_rcp_f32 (quarter rate) + 640 full-rate FMAs on dummy
// registers, no loads, no stores beyond the minimal tail.  MODE 0: as one lump after the K loop (where the epilogue is: other
// waves' MFMAs are the only thing that can hide it); MODE 1: spread over the K loop from the hook between the two MFMA bursts of
// every K tile (the wave's own MFMAs can hide it too).  Do VALU instructions co-execute with the matrix pipe at all here?
template <int MODE>
struct EpiLstmFwdValuProbeT {
  float* h_out;
  int H;
  static constexpr int KHOOK_VM_OPS = 0;
  struct KHook {
    float v[5];
  };
  __device__ __forceinline__ static void work(float (&v)[5], int reps) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {      // five independent chains, like the five activations of a hidden unit
        float e, q;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(v[i]));
        asm volatile("v_rcp_f32 %0, %1" : "=v"(q) : "v"(e));
        asm volatile("v_fma_f32 %0, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %3"
                     : "=&v"(v[i]) : "v"(q), "v"(e), "v"(v[(i + 1) % 5]));
      }
    }
  }
  __device__ __forceinline__ void khook_init(KHook& s) const {
#pragma unroll
    for (int i = 0; i < 5; ++i) s.v[i] = 0.001f * (threadIdx.x + i);
  }
  __device__ __forceinline__ void khook(KHook& s, int kt, int nk, int, int, int, int, unsigned) const {
    // 32 K tiles per tile at H = 512: 80 / 32 = 2.5 exp per K tile -> alternate 3 and 2 rounds of (1 exp + 1 rcp + 4 fma) x 5 / 5
    if constexpr (MODE == 1) {
      float one[5] = {s.v[0], s.v[1], s.v[2], s.v[3], s.v[4]};
      work(one, 1);                     // 5 exp + 5 rcp + 20 fma per K tile  (x 32 = 160 + 160 + 640: twice the trans of the real epilogue's 80 + 80)
#pragma unroll
      for (int i = 0; i < 5; ++i) s.v[i] = (kt & 1) ? one[i] : s.v[i] * 0.5f + one[i] * 0.5f;
    }
  }
  __device__ __forceinline__ void khook_finish(KHook&, int, int, int, int) const {}
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int vcol0, int lane, int M, int /*Nv*/,
                                             float* scr) const {
    float v[5] = {acc[0][0], acc[1][1], acc[2][2], acc[3][3], acc[0][4]};
    if constexpr (MODE == 0) work(v, 32);      // the same 160 exp + 160 rcp + 640 fma, after the K loop
    float4 a0[4];
    f32x16 sum = acc[0] + acc[1] + acc[2] + acc[3];
    sum[0] += v[0] + v[1] + v[2] + v[3] + v[4];
    tile_to_rows(sum, scr, lane, a0);
    const int j0 = (vcol0 >> 7) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = row0 + p * 8 + (lane >> 3);
      if (row < M) *reinterpret_cast<float4*>(h_out + (long)row * H + j0) = a0[p];
    }
  }
};
#ifndef VD_TICK_EPI_SEQ
#define VD_TICK_EPI_SEQ 1   // epilogue flavour of the encoder tick kernels: 1 = compiler-scheduled (3 spilled VGPRs in the persistent
                           // kernel at the 128-register cap), 0 = the hand-scheduled one of the throughput kernels (12 spills)
#endif

// ---------------------------------------------------------------------------
// backward epilogue: acc = (da_{t+1} * Wh^T)[row, j]  (zero at the last step)
//   dh = acc + dh_a + dh_b ; tc = tanh(c_t)
//   dc = dc_next + dh*o*(1-tc^2)
//   da_i = dc*g*i(1-i) ; da_f = dc*c_prev*f(1-f) ; da_o = dh*tc*o(1-o) ; da_g = dc*i*(1-g^2)
//   dc_next <- dc*f
// da_t overwrites the saved gates of step t in place.
// ---------------------------------------------------------------------------
// BATCH = slots whose loads are issued together before any is consumed (1 or 2; see the comment inside).
// TWO = false: the caller guarantees at most ONE incoming-gradient operand (dh_b == nullptr: every step but the last of a
// recurrence).  The second operand is fetched inside the consume phase, and the s_waitcnt behind that (uniformly skipped) load
// drains the previous slot's stores even when the load is never executed.
template <int NT, int BATCH = (NT == 1 ? 2 : 1), bool TWO = true>
struct EpiLstmBwd {
  const float* dh_a;  // nullable [N x H]
  const float* dh_b;  // nullable [N x H]
  float* gates;       // [N x 4H] in: gates_t, out: da_t
  const float* c_t;   // [N x H]
  const float* c_prev;  // nullable -> zeros
  float* dc;            // [N x H] in: dc_next (ignored when dc_first), out: dc for step t-1
  int dc_first;
  int H;
  vd_bf16_bits* da16 = nullptr;   // nullable: bf16 copy of da_t [N x 4H] (bf16 pass)
  // A lane serves NT x 4 (column tile, row) slots of 4 consecutive hidden units.  Each slot reads 7-9 float4 (saved
  // gates, c_t, c_{t-1}, dc, incoming dh) before ~30 flops of math, so the epilogue is pure memory latency.  Left to
  // the compiler the slots ran back to back, each behind its own s_waitcnt vmcnt(0) (plus one more round trip per
  // optional dh operand): 8-24 serialised round trips per tile.  Here every operand of a slot (optional ones included)
  // is requested in one batch, and with BATCH = 2 the loads of TWO slots are issued before any of them is consumed
  // (half the round trips).  BATCH = 2 needs ~140 VGPRs next to the second accumulator tile of the 128x64
  // throughput shape (it spills at the 128-register cap), so that shape has two builds: <=128 VGPRs with BATCH = 1
  // (a latency-shape workgroup of another stream still fits beside three of these on a SIMD) and <=168 VGPRs with
  // BATCH = 2; the single-tile latency shapes (NT = 1) fit BATCH = 2 inside 128.
  struct Slot {
    float4 g[4], ct, cp, dcv, dhx;
  };
  __device__ __forceinline__ void load_slot(Slot& L, int rc, int jc) const {
    const long o = (long)rc * H + jc;
    const float* gr = gates + (long)rc * 4 * H + jc;
    L.g[0] = vd_ld4_stream(gr);          // saved gates: read exactly once
    L.g[1] = vd_ld4_stream(gr + H);
    L.g[2] = vd_ld4_stream(gr + 2 * H);
    L.g[3] = vd_ld4_stream(gr + 3 * H);
    L.ct = *reinterpret_cast<const float4*>(c_t + o);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    L.cp = c_prev ? *reinterpret_cast<const float4*>(c_prev + o) : z;
    L.dcv = dc_first ? z : *reinterpret_cast<const float4*>(dc + o);
    const float* dh1 = dh_a ? dh_a : dh_b;   // the common case has at most one incoming-gradient operand
    L.dhx = dh1 ? *reinterpret_cast<const float4*>(dh1 + o) : z;
  }
#if VD_EPI_BUF & 6
  // buffer addressing (common.h): `o4` = byte offset of (row, j) in an [M x H] tensor; the gates tensor's is 4 * o4 - 12 * j
  struct Rsrc {
    __amdgpu_buffer_rsrc_t g, ct, cp, dc, dh1, dh2, g16;
  };
  __device__ __forceinline__ void load_slot_buf(Slot& L, const Rsrc& R, unsigned o4, unsigned og4, unsigned uH4) const {
    L.g[0] = vd_buf_ld4(R.g, og4, 0);          // saved gates: read exactly once
    L.g[1] = vd_buf_ld4(R.g, og4, uH4);
    L.g[2] = vd_buf_ld4(R.g, og4, 2 * uH4);
    L.g[3] = vd_buf_ld4(R.g, og4, 3 * uH4);
    L.ct = vd_buf_ld4(R.ct, o4, 0);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    L.cp = c_prev ? vd_buf_ld4(R.cp, o4, 0) : z;
    L.dcv = dc_first ? z : vd_buf_ld4(R.dc, o4, 0);
    L.dhx = (dh_a || dh_b) ? vd_buf_ld4(R.dh1, o4, 0) : z;   // the common case has at most one incoming-gradient operand
  }
#endif
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M,
                                             int N, float* scr) const {
    const int rl = lane >> 3, cl = (lane & 7) * 4;
    const bool two_dh = TWO && dh_a && dh_b;
#if VD_EPI_BUF & 6
    const Rsrc R{vd_rsrc(gates), vd_rsrc(c_t), vd_rsrc(c_prev), vd_rsrc(dc), vd_rsrc(dh_a ? dh_a : dh_b), vd_rsrc(dh_b), vd_rsrc(da16)};
    const unsigned uH4 = (unsigned)H * 4u;
#endif
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      float4 d4[4];
      tile_to_rows(acc[jt], scr, lane, d4);  // row-vectorised: 4 consecutive hidden units per lane
      const int j = col0 + jt * 32 + cl;
      const int jc = j < N ? j : N - 4;      // N = H here; columns past the end load a valid address, never stored
#pragma unroll
      for (int pp = 0; pp < 4; pp += BATCH) {
        Slot L[BATCH];
#if VD_EPI_BUF & 6
        unsigned o4[BATCH], og4[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
          const int row = row0 + (pp + q) * 8 + rl;
          o4[q] = (unsigned)(row < M ? row : M - 1) * uH4 + (unsigned)jc * 4u;
          og4[q] = 4u * o4[q] - 12u * (unsigned)jc;
#if VD_EPI_BUF & 2
          load_slot_buf(L[q], R, o4[q], og4[q], uH4);
#else
          load_slot(L[q], row < M ? row : M - 1, jc);
#endif
        }
#else
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
          const int row = row0 + (pp + q) * 8 + rl;
          load_slot(L[q], row < M ? row : M - 1, jc);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
          const Slot& C = L[q];
          float4 dh = d4[pp + q];
          dh.x += C.dhx.x; dh.y += C.dhx.y; dh.z += C.dhx.z; dh.w += C.dhx.w;
          const int row = row0 + (pp + q) * 8 + rl;
#if VD_EPI_BUF & 6
          if constexpr (TWO) {
            if (two_dh) {
              const float4 t = vd_buf_ld4(R.dh2, o4[q], 0);
              dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
            }
          }
#else
          const long o = (long)(row < M ? row : M - 1) * H + jc;
          if (two_dh) {
            const float4 t = *reinterpret_cast<const float4*>(dh_b + o);
            dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
          }
#endif
          float4 ai, af, ao, ag, dn;
#define VD_CELLB(E)                                                       \
          {                                                               \
            const float tc = vd_tanh(C.ct.E);                             \
            const float d = C.dcv.E + dh.E * C.g[2].E * (1.f - tc * tc);  \
            ai.E = d * C.g[3].E * C.g[0].E * (1.f - C.g[0].E);            \
            af.E = d * C.cp.E * C.g[1].E * (1.f - C.g[1].E);              \
            ao.E = dh.E * tc * C.g[2].E * (1.f - C.g[2].E);               \
            ag.E = d * C.g[0].E * (1.f - C.g[3].E * C.g[3].E);            \
            dn.E = d * C.g[1].E;                                          \
          }
          VD_CELLB(x) VD_CELLB(y) VD_CELLB(z) VD_CELLB(w)
#undef VD_CELLB
          if (row < M && j < N) {
#if VD_EPI_BUF & 4
            // (row < M and j < N here: the clamped offsets of the loads are the true ones)
            vd_buf_st4(R.g, og4[q], 0, ai);
            vd_buf_st4(R.g, og4[q], uH4, af);
            vd_buf_st4(R.g, og4[q], 2 * uH4, ao);
            vd_buf_st4(R.g, og4[q], 3 * uH4, ag);
            vd_buf_st4(R.dc, o4[q], 0, dn);
            if (da16) {
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, 0, ai);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, uH4 >> 1, af);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, uH4, ao);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, 3 * (uH4 >> 1), ag);
            }
#else
#if VD_EPI_BUF & 6
            const long o = (long)row * H + j;
#endif
            float* gr = gates + (long)row * 4 * H + j;
            *reinterpret_cast<float4*>(gr) = ai;
            *reinterpret_cast<float4*>(gr + H) = af;
            *reinterpret_cast<float4*>(gr + 2 * H) = ao;
            *reinterpret_cast<float4*>(gr + 3 * H) = ag;
            *reinterpret_cast<float4*>(dc + o) = dn;
            if (da16) {
              vd_bf16_bits* g16 = da16 + (long)row * 4 * H + j;
              vd_st4_bf16(g16, ai);
              vd_st4_bf16(g16 + H, af);
              vd_st4_bf16(g16 + 2 * H, ao);
              vd_st4_bf16(g16 + 3 * H, ag);
            }
#endif
          }
        }
      }
    }
  }
};

// The backward epilogue of the THROUGHPUT step kernel (fp32, a step with a recurrent product: at most one incoming-gradient
// operand, no bf16 shadow), branch-free.  In EpiLstmBwd every optional operand (`ptr ? load : zero`), the tile-edge predicate around
// the stores and the shadow stores are uniform BRANCHES; behind each join the compiler no longer knows how many memory
// operations are in flight and falls back to s_waitcnt vmcnt(0) -- so the 8 slots of a tile were 8 load -> math -> store -> DRAIN
// round trips (ISA: profiles/r03_experiments.txt section 13d).  Here every access is unconditional and is masked by the buffer
// range check instead: a null operand gets a descriptor with num_records = 0 (loads return 0 = the value the branch supplied,
// stores are dropped), rows past M fall outside num_records by themselves, columns past N get the out-of-range offset
// 0x80000000.  The compiler then counts exactly: loads of two slots, vmcnt(8), math, 5 stores, vmcnt(5), math, 5 stores, next loads.
// Needs every tensor of the step below 2 GB (num_records is a positive int).
template <int NT>
struct EpiLstmBwdLean {
  const float* dh_a;  // nullable [N x H] (the one incoming-gradient operand)
  float* gates;       // [N x 4H] in: gates_t, out: da_t
  const float* c_t;   // [N x H]
  const float* c_prev;  // nullable -> zeros
  float* dc;            // [N x H] in: dc_next (ignored when dc_first), out: dc for step t-1
  int dc_first;
  int H;
  struct Slot {
    float4 g[4], ct, cp, dcv, dhx;
  };
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
  }
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M, int N,
                                             float* scr) const {
    const int rl = lane >> 3, cl = (lane & 7) * 4;
    const unsigned uH4 = (unsigned)H * 4u, bytes = (unsigned)M * uH4;
    const __amdgpu_buffer_rsrc_t Rg = rsrc(gates, 4u * bytes), Rct = rsrc(c_t, bytes), Rcp = rsrc(c_prev, bytes),
                                 Rdc = rsrc(dc, bytes), Rdcl = rsrc(dc_first ? nullptr : dc, bytes), Rdh = rsrc(dh_a, bytes);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      float4 d4[4];
      tile_to_rows(acc[jt], scr, lane, d4);  // row-vectorised: 4 consecutive hidden units per lane
      const int j = col0 + jt * 32 + cl;
      const bool jbad = j >= N;              // N = H here
#pragma unroll
      for (int pp = 0; pp < 4; pp += 2) {
        Slot L[2];
        unsigned o4[2], og4[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned row = (unsigned)(row0 + (pp + q) * 8 + rl);
          o4[q] = jbad ? 0x80000000u : row * uH4 + (unsigned)j * 4u;           // rows >= M are out of range by themselves
          og4[q] = jbad ? 0x80000000u : row * 4u * uH4 + (unsigned)j * 4u;
          L[q].g[0] = vd_buf_ld4(Rg, og4[q], 0);          // saved gates: read exactly once
          L[q].g[1] = vd_buf_ld4(Rg, og4[q], uH4);
          L[q].g[2] = vd_buf_ld4(Rg, og4[q], 2 * uH4);
          L[q].g[3] = vd_buf_ld4(Rg, og4[q], 3 * uH4);
          L[q].ct = vd_buf_ld4(Rct, o4[q], 0);
          L[q].cp = vd_buf_ld4(Rcp, o4[q], 0);
          L[q].dcv = vd_buf_ld4(Rdcl, o4[q], 0);
          L[q].dhx = vd_buf_ld4(Rdh, o4[q], 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const Slot& C = L[q];
          float4 dh = d4[pp + q];
          dh.x += C.dhx.x; dh.y += C.dhx.y; dh.z += C.dhx.z; dh.w += C.dhx.w;
          float4 ai, af, ao, ag, dn;
#define VD_CELLB(E)                                                       \
          {                                                               \
            const float tc = vd_tanh(C.ct.E);                             \
            const float d = C.dcv.E + dh.E * C.g[2].E * (1.f - tc * tc);  \
            ai.E = d * C.g[3].E * C.g[0].E * (1.f - C.g[0].E);            \
            af.E = d * C.cp.E * C.g[1].E * (1.f - C.g[1].E);              \
            ao.E = dh.E * tc * C.g[2].E * (1.f - C.g[2].E);               \
            ag.E = d * C.g[0].E * (1.f - C.g[3].E * C.g[3].E);            \
            dn.E = d * C.g[1].E;                                          \
          }
          VD_CELLB(x) VD_CELLB(y) VD_CELLB(z) VD_CELLB(w)
#undef VD_CELLB
          vd_buf_st4(Rg, og4[q], 0, ai);
          vd_buf_st4(Rg, og4[q], uH4, af);
          vd_buf_st4(Rg, og4[q], 2 * uH4, ao);
          vd_buf_st4(Rg, og4[q], 3 * uH4, ag);
          vd_buf_st4(Rdc, o4[q], 0, dn);
        }
      }
    }
  }
};


// throughput shapes (option LSTM: N = 20 000 rows).  Defaults chosen by on-device sweeps
// (scripts/microbench.py, profiles/r01_config_sweep.txt): forward = 128x128 tile, BK = 16, single LDS
// buffer, 3 workgroups/CU (99 TF); backward = 128x64 tile (1 256 half-size workgroups balance 256 CUs),
// BK = 16, single buffer, 4 workgroups/CU (102 TF).
// VD_LSTM_FWD_CFG / VD_LSTM_BWD_CFG select the alternatives for A/B runs.
using CfgF0 = GemmCfg<4, 1, 4, 32, 1, 1>;  // double-buffered, 2 WG/CU
using CfgF1 = GemmCfg<4, 1, 4, 32, 0, 3>;  // default
using CfgF2 = GemmCfg<8, 1, 4, 32, 0, 4>;  // 256x128 tile, 512 threads, 2 WG/CU
using CfgF3 = GemmCfg<4, 1, 4, 64, 0, 2>;  // BK = 64, 2 WG/CU
using CfgF4 = GemmCfg<4, 1, 4, 16, 0, 3>;  // BK = 16
using CfgF5 = GemmCfg<4, 1, 4, 16, 2, 3>;  // two register stages
using CfgF6 = GemmCfg<4, 1, 4, 8, 2, 3>;
using CfgF7 = GemmCfg<4, 1, 4, 32, 2, 3>;
using CfgF8 = GemmCfg<4, 1, 4, 16, 1, 3>;
using CfgF9 = GemmCfg<4, 1, 4, 16, 0, 4, 41984>;  // <=128 VGPR, LDS padded to 41 KB: exactly 3 WG/CU + room for one latency WG
using CfgB0 = GemmCfg<4, 1, 4, 32, 1, 1>;
using CfgB1 = GemmCfg<4, 1, 4, 32, 0, 3>;
using CfgB2 = GemmCfg<4, 1, 2, 32, 0, 3>;  // default
using CfgB3 = GemmCfg<4, 1, 2, 32, 1, 3>;
using CfgB4 = GemmCfg<4, 1, 2, 32, 0, 4>;  // 4 WG/CU
using CfgB5 = GemmCfg<4, 1, 2, 64, 0, 3>;  // BK = 64
using CfgB6 = GemmCfg<8, 1, 2, 32, 0, 4>;  // 256x64 tile
using CfgB7 = GemmCfg<4, 1, 2, 16, 2, 3>;  // two register stages
using CfgB8 = GemmCfg<4, 1, 2, 16, 0, 3>;
using CfgB10 = GemmCfg<4, 1, 2, 32, 2, 3>;
using CfgB11 = GemmCfg<4, 1, 2, 16, 0, 4, 41984>;
using CfgB12 = GemmCfg<4, 1, 2, 16, 0, 3, 41984>;  // <=168 VGPR build of B11 for the two-slot epilogue (EpiLstmBwd<2, 2>)
// latency shapes (N ~ 200 rows): 4-way intra-block split-K.  Single LDS buffer and <= 152 VGPRs so one
// of these workgroups fits into the footprint a retiring throughput-shape workgroup frees (they run
// concurrently on other streams).
using CfgFbf16 = GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>;  // bf16 operands (opt-in), BK = 32 = two 32x32x16 MFMAs per tile
using CfgBbf16 = GemmCfg<4, 1, 2, 32, 0, 3, 0, 1>;
// bf16 pass with shadows: the LDS-DMA pipeline of the fp32 step kernels over bf16 rows (h / da shadows, bf16 weight copies)
using CfgF9bf16 = GemmCfg<4, 1, 4, 16, 0, 4, 41984, 1>;
using CfgB11bf16 = GemmCfg<4, 1, 2, 16, 0, 4, 41984, 1>;
using CfgFwdSmallA = GemmCfg<1, 4, 4, 8, 0, 3>;   // 32 x (32 j x 4 gates), BK = 32, 23 KB LDS
using CfgFwdSmallB = GemmCfg<1, 4, 4, 16, 0, 3>;  // BK = 64, 43.5 KB LDS, half the dependent K iterations
using CfgBwdSmallA = GemmCfg<1, 4, 1, 32, 0, 3>;  // 32 x 32 tile, BK = 128, 33.8 KB LDS
using CfgBwdSmallB = GemmCfg<1, 4, 1, 64, 0, 3>;  // BK = 256, 66.6 KB LDS
using CfgFwdSmallC = GemmCfg<1, 4, 4, 8, 0, 4>;   // as A, <=128 VGPR: fits beside 3 padded throughput workgroups
using CfgBwdSmallC = GemmCfg<1, 4, 1, 32, 0, 4>;
using CfgFwdSmallD = GemmCfg<1, 4, 4, 8, 2, 4>;   // as C with two register stages (tiles k+1, k+2 in flight)
using CfgFwdSmallE = GemmCfg<2, 2, 4, 8, 0, 4>;   // 64 x 128 tiles, 2-way split-K, BK = 16: 40 % fewer operand bytes per output than 32 x 128
using CfgFwdSmallF = GemmCfg<2, 2, 4, 8, 2, 4>;   // E with two register stages
using CfgFwdSmallG = GemmCfg<2, 2, 4, 16, 0, 4>;  // E with BK = 32
using CfgBwdSmallD = GemmCfg<1, 4, 1, 32, 2, 4>;
using CfgBwdSmallE = GemmCfg<1, 4, 2, 16, 0, 4>;   // 32 x 64 tiles (NT = 2), BK = 64: the A panel is re-read by 8 column tiles, not 16
using CfgBwdSmallF = GemmCfg<2, 2, 2, 32, 0, 4>;   // 64 x 64 tiles, 2-way intra-block split-K, BK = 64: -43 % operand traffic
using CfgBwdSmallG = GemmCfg<1, 4, 2, 16, 2, 4>;   // E with two register stages

static int env_int(const char* name, int dflt) {
  const char* ev = getenv(name);
  return ev ? atoi(ev) : dflt;
}

// WhT[vc][k] = Wh[k][g*H + jb*32 + jj], vc = jb*128 + g*32 + jj: the recurrent weights as k-contiguous rows
// in the gate-interleaved column order of the forward step, so that both operands of the LDS-DMA pipeline
// are plain row-major.  4 MB, rebuilt once per forward pass (the weights change every update).
__global__ void __launch_bounds__(256) wh_gate_transpose_kernel(const float* __restrict__ Wh,
                                                                float* __restrict__ WhT, int H, int plain = 0) {
  __shared__ float tile[32][33];
  const int vt = blockIdx.x, kt = blockIdx.y;
  const int jb = vt >> 2, g = vt & 3;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // plain: WhT[c][k] = Wh[k][c] (column order kept: the layer-2 input projection of the encoder ticks)
  const long src_col = plain ? (long)vt * 32 : (long)g * H + jb * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = kt * 32 + ty + i * 8;
    tile[ty + i * 8][tx] = Wh[(long)k * 4 * H + src_col + tx];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vc = vt * 32 + ty + i * 8;
    WhT[(long)vc * H + kt * 32 + tx] = tile[tx][ty + i * 8];
  }
}

// fp32 -> bf16 copy of a weight matrix (bf16 pass of the option recurrence: the LDS-DMA step kernels multiply bf16 rows
// as they lie in memory; same rounding as the activations' shadows, vd_st4_bf16)
__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float* __restrict__ src, vd_bf16_bits* __restrict__ dst, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) vd_st4_bf16(dst + i * 4, reinterpret_cast<const float4*>(src)[i]);
}
static int weights_to_bf16(const float* src, vd_bf16_bits* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, src, dst, n / 4);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// LDS-DMA pipeline eligibility: throughput shape, K % 16 == 0, 32-bit row byte offsets
static bool use_glds_fwd(int N, int H) {
  static const int cfg = env_int("VD_LSTM_FWD_CFG", 20);
  return cfg == 20 && N >= 2048 && H % 32 == 0 && (long)N * H * 4 < (1L << 32);
}
static bool use_glds_bwd(int N, int H) {
  static const int cfg = env_int("VD_LSTM_BWD_CFG", 20);
  return cfg == 20 && N >= 2048 && H % 32 == 0 && (long)N * 4 * H * 4 < (1L << 32);
}

// ---------------------------------------------------------------------------
// Persistent sequence kernels: the WHOLE T-step recurrence of a throughput shape in one launch.
//
// The per-step launches above start 768 workgroups at the same instant with identical lifetimes: they reach
// their epilogues together (matrix pipe idle while the gate/cell traffic drains), every step ends in a
// partially filled last round (2512 tiles on 768 slots), and the pattern repeats 20 times per direction.
// Here the tiles of ALL steps form one work list.  Resident workgroups pull (step, row tile, column tile)
// items from it and the only ordering is the real data dependency: tile (r, *, s) needs the 128 rows r of
// step s-1 complete, i.e. all column tiles of (r, s-1) -- one arrival counter per (row tile, step).  Items are
// listed step-major, so the rows a tile depends on were issued ~3 rounds earlier and the wait is almost
// never taken; workgroups drift out of phase, epilogues of some overlap K loops of others, and there is one
// tail per direction instead of one per step.
//
// Work lists: 8 queues (one per XCD, selected at run time from HW_REG_XCC_ID) that own contiguous ranges of
// ROW tiles, so a row tile's h / da panels stay in one XCD's L2; a workgroup whose queue is exhausted takes
// from the next queue.  Placement is a speed choice only: every cross-workgroup hand-off uses the
// agent-scope release (producer, after its stores) / relaxed poll + agent-scope acquire (consumer) protocol,
// which is correct for any workgroup->CU/XCD placement and any dispatch order.  Deadlock freedom needs no
// co-residency: a workgroup only ever waits for items that were dequeued (hence are being executed by a
// resident workgroup) earlier in the same queue, and dependencies point to strictly earlier steps.
// Every spin is bounded (VD_SEQ_TIMEOUT_TICKS of the 100 MHz clock); a timeout sets a sticky error word that
// the host can read back with vd_lstm_seq_status().
// ---------------------------------------------------------------------------
#define VD_SEQ_QUEUES 8
#define VD_SEQ_HEAD_STRIDE 32                                   // uints: one 128-byte line per queue head
#define VD_SEQ_ERR_WORD (VD_SEQ_QUEUES * VD_SEQ_HEAD_STRIDE)    // sticky timeout flag
#define VD_SEQ_CU_WORDS (VD_SEQ_ERR_WORD + 32)                  // 8 x 256 per-CU arrival tickets (start stagger)
#define VD_SEQ_CNT0 (VD_SEQ_CU_WORDS + VD_SEQ_QUEUES * 256)     // arrival counters [T x tiles_m]
#define VD_SEQ_TIMEOUT_TICKS 200000000ull                       // 2 s of s_memrealtime

static size_t seq_sync_bytes(int T, int tiles_m) { return ((size_t)VD_SEQ_CNT0 + (size_t)T * tiles_m) * sizeof(unsigned); }

struct SeqSched {
  unsigned* sync;
  int T, tiles_m, tiles_n;
  int stagger_ticks;  // >0: the k-th workgroup to arrive on a CU sleeps k * stagger_ticks (100 MHz ticks) once
};

__device__ __forceinline__ unsigned vd_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x & 7u;
}
__device__ __forceinline__ unsigned vd_cu_key() {  // CU / SH / SE id bits of HW_ID: identifies the CU inside its XCC
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 8)" : "=s"(x));
  return x & 255u;
}

// Wave-level helpers.  All control flow of the scheduler is WAVE-UNIFORM (scalar branch conditions): a divergent
// `if (threadIdx.x == 0) { loops ... }` region next to s_barrier lets the compiler's CFG structurizer rotate the
// tile loop so that lanes 1..63 of wave 0 run ahead through the barrier while lane 0 is parked (observed: an endless
// loop re-reading a stale work item).  The only per-lane regions left are single instructions under `lane == 0`.
__device__ __forceinline__ unsigned wave_fetch_add(unsigned* p, unsigned v, int lane) {
  unsigned r = 0;
  if (lane == 0) r = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}
__device__ __forceinline__ unsigned wave_load(unsigned* p, int lane) {
  unsigned r = 0;
  if (lane == 0) r = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}

// Executed by ONE wave (all lanes, uniformly).  Pull the next item (processing step s, row tile r, column tile j)
// and wait until the rows it depends on are complete.  Returns false when every queue is exhausted.
__device__ __forceinline__ bool seq_next(const SeqSched& sc, int lane, int& q, int& tried, int& s, int& r, int& j) {
  unsigned* const sync = sc.sync;
  for (;;) {
    const int rt0 = (int)((long)q * sc.tiles_m / VD_SEQ_QUEUES), rt1 = (int)((long)(q + 1) * sc.tiles_m / VD_SEQ_QUEUES);
    const int per_step = (rt1 - rt0) * sc.tiles_n;
    if (per_step > 0) {
      const unsigned idx = wave_fetch_add(sync + q * VD_SEQ_HEAD_STRIDE, 1u, lane);
      if (idx < (unsigned)per_step * (unsigned)sc.T) {
        s = (int)(idx / (unsigned)per_step);
        const int rem = (int)(idx - (unsigned)s * (unsigned)per_step);
        r = rt0 + rem / sc.tiles_n;
        j = rem % sc.tiles_n;
        break;
      }
    }
    q = (q + 1) & (VD_SEQ_QUEUES - 1);
    if (++tried >= VD_SEQ_QUEUES) return false;
  }
  if (s > 0) {
    unsigned* const cnt = sync + VD_SEQ_CNT0 + (long)(s - 1) * sc.tiles_m + r;
    unsigned* const err = sync + VD_SEQ_ERR_WORD;
    if (wave_load(cnt, lane) < (unsigned)sc.tiles_n) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        __builtin_amdgcn_s_sleep(4);
        if (wave_load(cnt, lane) >= (unsigned)sc.tiles_n) break;
        if (wave_load(err, lane) != 0u) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > VD_SEQ_TIMEOUT_TICKS) {
          if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale L1 lines; plain loads follow the barrier
  }
  return true;
}

// All threads.  Publish a finished tile: every wave drains its stores, then one wave releases at agent scope
// and bumps the (row tile, step) arrival counter.
__device__ __forceinline__ void seq_publish(const SeqSched& sc, int s, int r, int wave, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the wait behind buffer_wbl2 (guide, G16 pitfall 12)
    if (lane == 0)
      __hip_atomic_fetch_add(sc.sync + VD_SEQ_CNT0 + (long)s * sc.tiles_m + r, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One-time start stagger: the k-th workgroup of this launch to arrive on a CU waits k * stagger_ticks, so the
// co-resident workgroups of a CU begin out of phase instead of drifting apart over the first steps.
__device__ __forceinline__ void seq_stagger(const SeqSched& sc, unsigned xcc, int wave, int lane) {
  if (sc.stagger_ticks <= 0) return;
  if (wave == 0) {
    const unsigned k = wave_fetch_add(sc.sync + VD_SEQ_CU_WORDS + xcc * 256 + vd_cu_key(), 1u, lane);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(k % 3u) * (unsigned)sc.stagger_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
  }
  __syncthreads();
}

// Scheduler state of a workgroup (wave 0 owns it, in SGPRs) + the LDS mailbox the other waves read the item from.
#define VD_SEQ_LOOP_HEAD(ARGS)                                                                          \
  int* const slot = reinterpret_cast<int*>(smem) + Cfg::LDS_BYTES / 4 - 4; /* beyond DMA buffers / scratch */ \
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);                               \
  const int lane0 = (int)threadIdx.x & 63;                                                              \
  const unsigned xcc = vd_xcc_id();                                                                     \
  int q = (int)xcc, tried = 0;                                                                          \
  seq_stagger((ARGS).sc, xcc, wave, lane0);

#define VD_SEQ_NEXT_ITEM(ARGS, S, R, J)                                                                 \
  if (wave == 0) {                                                                                      \
    int s_ = -1, r_ = 0, j_ = 0;                                                                        \
    if (!seq_next((ARGS).sc, lane0, q, tried, s_, r_, j_)) s_ = -1;                                     \
    if (lane0 == 0) {                                                                                   \
      slot[0] = s_;                                                                                     \
      slot[1] = r_;                                                                                     \
      slot[2] = j_;                                                                                     \
    }                                                                                                   \
  }                                                                                                     \
  __syncthreads();                                                                                      \
  const int S = __builtin_amdgcn_readfirstlane(slot[0]);                                                \
  const int R = __builtin_amdgcn_readfirstlane(slot[1]);                                                \
  const int J = __builtin_amdgcn_readfirstlane(slot[2]);

struct LstmSeqFwdArgs {
  const float* xproj;   // dense: [T x N x 4H] (x_tstride per step); table mode: [V+1 x 4H]
  long x_tstride, xld;
  const int* tok_gather;  // [T x N] or null
  const int* tok_mask;    // [T x N] or null
  const float* WhT;       // [4H x H] gate-interleaved transpose of Wh
  const float *h0, *c0;   // [N x H] or null
  float *gates, *h, *c;   // [T x N x 4H], [T x N x H] x 2
  int N, H, rotate;
  int t_base;             // single-step launches (sc.T == 1): the real step index of item step 0
  SeqSched sc;
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW) lstm_seq_fwd_kernel(LstmSeqFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  VD_SEQ_LOOP_HEAD(a)
  const long NH = (long)a.N * a.H;
  for (;;) {
    VD_SEQ_NEXT_ITEM(a, tq, r, j)
    if (tq < 0) break;
    const int t = tq + a.t_base;
    const float* hp = t ? a.h + (t - 1) * NH : a.h0;
    EpiLstmFwd e;
    e.xproj = a.xproj + (long)t * a.x_tstride;
    e.xld = a.xld;
    e.tok_gather = a.tok_gather ? a.tok_gather + (long)t * a.N : nullptr;
    e.tok_mask = a.tok_mask ? a.tok_mask + (long)t * a.N : nullptr;
    e.c_prev = t ? a.c + (t - 1) * NH : a.c0;
    e.gates = a.gates + (long)t * 4 * NH;
    e.c_out = a.c + t * NH;
    e.h_out = a.h + t * NH;
    e.H = a.H;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // per-lane address terms are rebuilt per tile, not kept live across tiles
    gemm_block_glds<Cfg, false>(a.N, 4 * a.H, 0, hp ? a.H : 0, r * Cfg::BM, j * Cfg::BN, a.rotate ? r * 5 + j * 3 : -1, hp,
                                (long)a.H, a.WhT, (long)a.H, e, smem, tid);
    if (a.sc.T > 1) seq_publish(a.sc, tq, r, wave, lane0);
    else __syncthreads();   // single-step launch: nobody waits for this tile; only the LDS image is reused
  }
}

struct LstmSeqBwdArgs {
  const float* Wh;        // [H x 4H]
  float* gates;           // [T x N x 4H] in: gates, out: da
  const float *c, *c0;    // [T x N x H], [N x H] or null
  const float* dh_seq;    // [T x N x H] or null
  const float* dh_last;   // [N x H] or null
  float* dc;              // [N x H] work (holds dc_last on entry when dc_has_last)
  int dc_has_last;
  int T, N, H, rotate;
  int s_base;             // single-step launches (sc.T == 1): the real processing-step index of item step 0
  SeqSched sc;
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW) lstm_seq_bwd_kernel(LstmSeqBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  VD_SEQ_LOOP_HEAD(a)
  const long NH = (long)a.N * a.H;
  for (;;) {
    VD_SEQ_NEXT_ITEM(a, sq, r, j)
    if (sq < 0) break;
    const int s = sq + a.s_base;
    const int t = a.T - 1 - s;
    const bool last = (s == 0);
    const float* da_next = last ? nullptr : a.gates + (long)(t + 1) * 4 * NH;
    EpiLstmBwd<Cfg::NT, (Cfg::MINW <= 3 ? 2 : 1)> e;
    e.dh_a = a.dh_seq ? a.dh_seq + t * NH : nullptr;
    e.dh_b = (last && a.dh_last) ? a.dh_last : nullptr;
    e.gates = a.gates + (long)t * 4 * NH;
    e.c_t = a.c + t * NH;
    e.c_prev = t ? a.c + (t - 1) * NH : a.c0;
    e.dc = a.dc;
    e.dc_first = (last && !a.dc_has_last) ? 1 : 0;
    e.H = a.H;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    gemm_block_glds<Cfg, false>(a.N, a.H, 0, last ? 0 : 4 * a.H, r * Cfg::BM, j * Cfg::BN, a.rotate ? r * 5 + j * 3 : -1,
                                da_next, 4L * a.H, a.Wh, 4L * a.H, e, smem, tid);
    if (a.sc.T > 1) seq_publish(a.sc, sq, r, wave, lane0);
    else __syncthreads();
  }
}

// persistent path eligibility (on top of the LDS-DMA eligibility): more than one step, one row chain
// Opt-in (default off).  Measured on MI355X (profiles/r02_persistent_vs_per_step.txt): alone the persistent launches
// run at the per-step kernels' speed (fwd 7.36 vs 7.02 ms, bwd 6.90 vs 6.90 ms), and inside the training step they
// LOSE 1.9 ms (29.5 vs 27.6 ms/step): a 7 ms launch that owns every workgroup slot leaves the encoder's ~170 small
// launches only the one spare wave slot per SIMD, so they pile up behind it and surface on the critical path.
static bool use_persistent(const char* knob) { return vd_tune_get(knob, 0) != 0; }

template <class Kern, class Args>
static int launch_seq(Kern kern, const Args& a, int lds_bytes, int threads, int total_tiles, hipStream_t s) {
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  // resident capacity: 3 workgroups per CU (41 KB LDS request each); a larger grid would only queue
  int grid = vd_tune_get("VD_LSTM_SEQ_WGS_PER_CU", 3) * vd_num_cus();
  if (grid > total_tiles) grid = total_tiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, s, a);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

static int lstm_step_fwd(const float* h_prev, const float* Wh, int N, int H, int K, const EpiLstmFwd& epi,
                         hipStream_t s) {
  SrcRow a{h_prev, H};
  SrcKGate4 b{Wh, 4L * H, H};
  if (N >= 2048) {
    static const int cfg0 = env_int("VD_LSTM_FWD_CFG", 20);
    const int cfg = cfg0 == 20 ? 9 : cfg0;   // 20 = LDS-DMA pipeline, handled by the drivers
    switch (cfg) {
      case 0: return launch_gemm<CfgF0>(N, 4 * H, K, 1, a, b, epi, s);
      case 2: return launch_gemm<CfgF2>(N, 4 * H, K, 1, a, b, epi, s);
      case 3: return launch_gemm<CfgF3>(N, 4 * H, K, 1, a, b, epi, s);
      case 4: return launch_gemm<CfgF4>(N, 4 * H, K, 1, a, b, epi, s);
      case 5: return launch_gemm<CfgF5>(N, 4 * H, K, 1, a, b, epi, s);
      case 6: return launch_gemm<CfgF6>(N, 4 * H, K, 1, a, b, epi, s);
      case 7: return launch_gemm<CfgF7>(N, 4 * H, K, 1, a, b, epi, s);
      case 8: return launch_gemm<CfgF8>(N, 4 * H, K, 1, a, b, epi, s);
      case 9: return launch_gemm<CfgF9>(N, 4 * H, K, 1, a, b, epi, s);
      default: return launch_gemm<CfgF1>(N, 4 * H, K, 1, a, b, epi, s);
    }
  }
  // (the <=128-VGPR build C spills 20 registers with the hand-scheduled epilogue; this un-bundled path is not on the
  //  headline step, so it takes the 140-VGPR build A)
  static const int scfg = env_int("VD_LSTM_FWD_SMALL", 0);
  if (scfg == 0) return launch_gemm<CfgFwdSmallA>(N, 4 * H, K, 1, a, b, epi, s);
  if (scfg == 2) return launch_gemm<CfgFwdSmallC>(N, 4 * H, K, 1, a, b, epi, s);
  return launch_gemm<CfgFwdSmallB>(N, 4 * H, K, 1, a, b, epi, s);
}

static int lstm_step_bwd(const float* da_next, const float* Wh, int N, int H, int K, const float* dh_a,
                         const float* dh_b, float* gates, const float* c_t, const float* c_prev, float* dc,
                         int dc_first, hipStream_t s, int flags = 0, vd_bf16_bits* da16 = nullptr,
                         const vd_bf16_bits* da16_next = nullptr, const vd_bf16_bits* Wh16 = nullptr) {
  SrcRow a{da_next, 4L * H};
  SrcRow b{Wh, 4L * H};  // B[k][n] = Wh[n][k]
  if ((flags & VD_FLAG_BF16) && N >= 2048 && K > 0) {
    EpiLstmBwd<2> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H, da16};
    // shadows on: da_{t+1} and Wh are read as the bf16 rows their producers wrote (half the operand bytes, no conversion
    // while staging, LDS-DMA pipeline); K counts bf16 pairs
    if (da16_next && Wh16 && K % 32 == 0 && vd_tune_get("VD_BF16_OCC4", 0))   // A/B: 36 KB request = 4 workgroups per CU
      return launch_gemm_glds<GemmCfg<4, 1, 2, 16, 0, 4, 36864, 1>, false>(N, H, K / 2, 1, reinterpret_cast<const float*>(da16_next), 2L * H,
                                                                       reinterpret_cast<const float*>(Wh16), 2L * H, e, s);
    if (da16_next && Wh16 && K % 32 == 0)
      return launch_gemm_glds<CfgB11bf16, false>(N, H, K / 2, 1, reinterpret_cast<const float*>(da16_next), 2L * H,
                                                 reinterpret_cast<const float*>(Wh16), 2L * H, e, s);
    return launch_gemm<CfgBbf16>(N, H, K, 1, a, b, e, s);
  }
  if (da16 && N >= 2048) {   // bf16 pass, step without a recurrent product (the last one): same shadow, generic kernel
    EpiLstmBwd<2> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H, da16};
    return launch_gemm<CfgB11>(N, H, K, 1, a, b, e, s);
  }
  if (N >= 2048) {
    static const int cfg0 = env_int("VD_LSTM_BWD_CFG", 20);
    const int cfg = cfg0 == 20 ? 11 : cfg0;
    if (use_glds_bwd(N, H) && K > 0) {
      // LDS-DMA pipeline: A = da_{t+1} rows, Bt = Wh rows (both contiguous in k = the 4H gate columns)
      if (const int nt4 = vd_tune_get("VD_LSTM_BWD_NT4", 0)) {
        // A/B: 128 x 128 tiles (every da_{t+1} panel is read by 4 column tiles instead of 8: half the L2 -> LDS operand
        // traffic per FLOP; 628 tiles leave a 2.45-round tail when the kernel runs alone)
        EpiLstmBwd<4, 1> e4{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
        if (nt4 == 2) return launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 3, 41984>, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e4, s);
        return launch_gemm_glds<CfgF9, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e4, s);
      }
      // the two-slot epilogue (loads of two slots in flight before either is consumed: half the serialised round trips).  With
      // buffer addressing (-DVD_EPI_BUF & 6) it fits the 128-VGPR build (126, no spills) = 2, the default; with 64-bit global
      // addressing it needs the 168-VGPR build (= 1) or spills 8 registers.  0 = one slot at a time.
      if (const int b2 = vd_tune_get("VD_LSTM_BWD_BATCH2", (VD_EPI_BUF & 6) == 6 ? 2 : 0)) {
        if (b2 == 3 && !(dh_a && dh_b) && (long)N * 4 * H * 4 < (1L << 31)) {   // branch-free epilogue (range-check masking)
          EpiLstmBwdLean<2> e3{dh_a ? dh_a : dh_b, gates, c_t, c_prev, dc, dc_first, H};
          return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e3, s);
        }
        if (b2 >= 2 && !(dh_a && dh_b)) {   // (a step with a recurrent product has at most one incoming-gradient operand)
          EpiLstmBwd<2, 2, false> e2c{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
          return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e2c, s);
        }
        EpiLstmBwd<2, 2> e2b{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
        if (b2 >= 2) return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e2b, s);
        return launch_gemm_glds<CfgB12, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e2b, s);
      }
      EpiLstmBwd<2> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
      return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e, s);
    }
    EpiLstmBwd<4> e4{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
    EpiLstmBwd<2> e2{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
    switch (cfg) {
      case 0: return launch_gemm<CfgB0>(N, H, K, 1, a, b, e4, s);
      case 1: return launch_gemm<CfgB1>(N, H, K, 1, a, b, e4, s);
      case 3: return launch_gemm<CfgB3>(N, H, K, 1, a, b, e2, s);
      case 4: return launch_gemm<CfgB4>(N, H, K, 1, a, b, e2, s);
      case 5: return launch_gemm<CfgB5>(N, H, K, 1, a, b, e2, s);
      case 6: return launch_gemm<CfgB6>(N, H, K, 1, a, b, e2, s);
      case 7: return launch_gemm<CfgB7>(N, H, K, 1, a, b, e2, s);
      case 8: return launch_gemm<CfgB8>(N, H, K, 1, a, b, e2, s);
      case 11: return launch_gemm<CfgB11>(N, H, K, 1, a, b, e2, s);
      case 10: return launch_gemm<CfgB10>(N, H, K, 1, a, b, e2, s);
      default: return launch_gemm<CfgB2>(N, H, K, 1, a, b, e2, s);
    }
  }
  EpiLstmBwd<1> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
  static const int scfg = env_int("VD_LSTM_BWD_SMALL", 2);
  if (scfg == 0) return launch_gemm<CfgBwdSmallA>(N, H, K, 1, a, b, e, s);
  if (scfg == 2) return launch_gemm<CfgBwdSmallC>(N, H, K, 1, a, b, e, s);
  return launch_gemm<CfgBwdSmallB>(N, H, K, 1, a, b, e, s);
}

// ---------------------------------------------------------------------------
// Two-layer stacks advanced as a skewed wavefront, several stacks per launch.
//
// The encoder recurrences (history / question, 2 x SeqLSTM each) are chains of ~30 us dependent
// launches.  Tick tau runs, for every stack, THREE independent sub-problems in one grouped launch:
//   forward :  L1 step tau | X2 = h1[tau-1]*Wx2 + b2 | L2 step tau-2
//   backward:  L2 step t   | dh1[t+1] = da2[t+1]*Wx2^T | L1 step t+2      (t = T-1-tau)
// so a T-step, 2-layer stack needs T+2 launches instead of 2T, all sub-problems keep K = H (4H in
// backward), and history + question stacks share the launches.
// ---------------------------------------------------------------------------
struct SrcKSel {
  static constexpr bool KMAJOR = true;
  static constexpr bool PLAIN = true;
  const float* p;
  long ld;
  int H;
  int gate4;
  __device__ __forceinline__ const float* ptr(int vc, int k) const {
    int col = vc;
    if (gate4) {
      const int jb = vc >> 7, g = (vc >> 5) & 3, jj = vc & 31;
      col = g * H + jb * 32 + jj;
    }
    return p + (long)k * ld + col;
  }
  __device__ __forceinline__ long kstep() const { return ld; }
  __device__ __forceinline__ float4 ld4(int vc, int k) const {
    int col = vc;
    if (gate4) {
      const int jb = vc >> 7, g = (vc >> 5) & 3, jj = vc & 31;
      col = g * H + jb * 32 + jj;
    }
    return *reinterpret_cast<const float4*>(p + (long)k * ld + col);
  }
};
using EpiLstmFwdTick = EpiLstmFwdT<VD_TICK_EPI_SEQ>;
struct EpiTickFwd {
  int kind;  // 0 = LSTM cell update, 1 = plain store (+bias)
  EpiLstmFwdTick f;
  EpiStore<4> s;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int col0, int lane, int M,
                                             int N, float* scr) const {
    if (kind == 0) f(acc, row0, col0, lane, M, N, scr);
    else s(acc, row0, col0, lane, M, N);
  }
};
struct TickFwdProb {
  int M, N, K, tiles_n;
  SrcRow a;
  SrcKSel b;
  EpiTickFwd e;
};
struct TickFwdProbT {      // forward tick sub-problem for gemm_block_glds_wk: B given as its transpose (k-contiguous rows)
  int M, N, K, tiles_n;
  const float* a;
  long lda;
  const float* bt;
  long ldb;
  EpiTickFwd e;
};
template <int NT>   // column tiles per wave: 1 = 32 x 32 tiles, 2 = 32 (or 64) x 64 tiles (less operand traffic per FLOP)
struct EpiTickBwdT {
  int kind;
  EpiLstmBwd<NT> f;      // (default BATCH: 2 slots in flight for NT = 1, 1 for NT >= 2)
  EpiStore<NT> s;
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M,
                                             int N, float* scr) const {
    if (kind == 0) f(acc, row0, col0, lane, M, N, scr);
    else s(acc, row0, col0, lane, M, N);
  }
};
template <int NT>
struct TickBwdProbT {
  int M, N, K, tiles_n;
  SrcRow a;
  SrcRow b;
  EpiTickBwdT<NT> e;
};
using EpiTickBwd = EpiTickBwdT<1>;
using TickBwdProb = TickBwdProbT<1>;

struct vd_lstm2_fwd_t {
  int T, N;
  const int32_t* tok_mask;
  const float *Wh1, *Wx2, *b2, *Wh2;
  float *gates1, *h1, *c1, *gates2, *h2, *c2;
  const int32_t* nact;  // HOST array [T] or NULL: rows [0, nact[t]) are the only non-pad rows at step t
};
struct vd_lstm2_bwd_t {
  int T, N;
  const float *Wh1, *Wx2, *Wh2;
  float* gates1;
  const float* c1;
  float* gates2;
  const float* c2;
  const float* dh_last2;
  float* dh1_seq;
  float *dc1, *dc2;
  const int32_t* nact;  // HOST array [T] or NULL (see vd_lstm2_fwd_t)
};

#define VD_MAX_STACKS 2

// ---------------------------------------------------------------------------
// Persistent two-layer recurrences: ALL T+2 ticks of a direction in ONE launch.
//
// The per-tick grouped launches above are short dependent kernels.  Inside the training step they share one hardware
// queue with the option-LSTM step kernels (2 512 workgroups on 768 slots each): a tick's packet is only dispatched
// once the big kernel in front of it has started its last workgroup, so a tick that takes 34 us alone takes ~150 us
// there (profiles/r03_kernel_stats_bench.txt) and the 42-tick encoder forward (6.3 ms + prologue) ends AFTER the
// 7.5 ms option forward: the main stream waits ~0.6 ms for the encoder output before the criterion.
//
// Here the tiles of every tick form one work list in tick-major order.  <= one workgroup per CU (the slot a throughput
// workgroup cannot use: <= 34 KB LDS, <= 128 VGPRs) pulls items from it; an item only waits for the tiles it really
// depends on -- the same row tile of <= 2 sub-problems of the PREVIOUS tick (forward: L1 <- L1; X2 <- L1; L2 <- X2, L2;
// backward: L2 <- L2; X <- L2; L1 <- L1, X), one arrival counter per (tick, stack, sub-problem, row tile).  The hand-off
// is the release / acquire protocol of the option-LSTM persistent kernels (seq_publish / seq_next above); deadlock
// freedom needs no co-residency: a workgroup only waits for items dequeued earlier from the same list.
// Sub-problem descriptors are rebuilt on the device from the stack descriptors (kernel arguments: no per-call table
// upload); the per-step active-row counts travel in the kernel arguments too (T <= VD_L2_MAXT, else per-tick launches).
// ---------------------------------------------------------------------------
#define VD_L2_MAXT 128
#define VD_L2_KINDS 3

struct L2SeqFwdArgs {
  vd_lstm2_fwd_t st[VD_MAX_STACKS];           // .nact is NOT dereferenced on the device (host pointer)
  unsigned short nact[VD_MAX_STACKS][VD_L2_MAXT];
  int nstacks, H, Tmax, row_tiles;            // row_tiles = counters per (tick, stack, kind)
  unsigned* sync;
};
struct L2SeqBwdArgs {
  vd_lstm2_bwd_t st[VD_MAX_STACKS];
  unsigned short nact[VD_MAX_STACKS][VD_L2_MAXT];
  int nstacks, H, Tmax, row_tiles;
  unsigned* sync;
};

// step index of sub-problem `kind` of a stack at tick tau (or -1): forward kinds 0 = L1 cell, 1 = L2 cell, 2 = X2
// projection; backward kinds 0 = L2 cell, 1 = L1 cell, 2 = dh1 projection -- the order the per-tick host loops use
__device__ __forceinline__ int l2_fwd_step(int kind, int tau, int T) {
  const int t = kind == 0 ? tau : (kind == 1 ? tau - 2 : tau - 1);
  return (t >= 0 && t < T) ? t : -1;
}
__device__ __forceinline__ int l2_bwd_step(int kind, int tau, int T) {
  const int t = kind == 0 ? T - 1 - tau : (kind == 1 ? T + 1 - tau : T - tau);
  return (t >= 0 && t < T) ? t : -1;
}

__device__ __forceinline__ TickFwdProb l2_fwd_prob(const vd_lstm2_fwd_t& S, int kind, int t, int rows, int H) {
  const long NH = (long)S.N * H;
  TickFwdProb P;
  P.M = rows; P.N = 4 * H; P.tiles_n = (4 * H) / 128;
  if (kind < 2) {
    float* gates = kind == 0 ? S.gates1 : S.gates2;
    float* h = kind == 0 ? S.h1 : S.h2;
    float* c = kind == 0 ? S.c1 : S.c2;
    P.K = t ? H : 0;
    P.a = SrcRow{t ? h + (t - 1) * NH : h, H};
    P.b = SrcKSel{kind == 0 ? S.Wh1 : S.Wh2, 4L * H, H, 1};
    P.e.kind = 0;
    P.e.f.xproj = gates + (long)t * 4 * NH; P.e.f.xld = 4L * H;
    P.e.f.tok_gather = nullptr;
    P.e.f.tok_mask = S.tok_mask ? S.tok_mask + (long)t * S.N : nullptr;
    P.e.f.c_prev = t ? c + (t - 1) * NH : nullptr;
    P.e.f.gates = gates + (long)t * 4 * NH; P.e.f.c_out = c + t * NH; P.e.f.h_out = h + t * NH; P.e.f.H = H;
    P.e.s = EpiStore<4>{nullptr, 0, nullptr, 0, 0};
  } else {
    P.K = H;
    P.a = SrcRow{S.h1 + t * NH, H};
    P.b = SrcKSel{S.Wx2, 4L * H, H, 0};
    P.e.kind = 1;
    P.e.s = EpiStore<4>{S.gates2 + (long)t * 4 * NH, 4L * H, S.b2, VD_ACT_NONE, 0};
    P.e.f = EpiLstmFwdTick{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, H};
  }
  return P;
}

__device__ __forceinline__ TickBwdProb l2_bwd_prob(const vd_lstm2_bwd_t& S, int kind, int t, int rows, int H) {
  const long NH = (long)S.N * H;
  TickBwdProb P;
  P.M = rows; P.N = H; P.tiles_n = H / 32;
  if (kind < 2) {
    const bool l2 = kind == 0;
    float* gates = l2 ? S.gates2 : S.gates1;
    const float* c = l2 ? S.c2 : S.c1;
    const bool last = (t == S.T - 1);
    P.K = last ? 0 : 4 * H;
    P.a = SrcRow{last ? gates : gates + (long)(t + 1) * 4 * NH, 4L * H};
    P.b = SrcRow{l2 ? S.Wh2 : S.Wh1, 4L * H};
    P.e.kind = 0;
    P.e.f.dh_a = l2 ? (last ? S.dh_last2 : nullptr) : S.dh1_seq + t * NH;
    P.e.f.dh_b = nullptr;
    P.e.f.gates = gates + (long)t * 4 * NH;
    P.e.f.c_t = c + t * NH;
    P.e.f.c_prev = t ? c + (t - 1) * NH : nullptr;
    P.e.f.dc = l2 ? S.dc2 : S.dc1;
    P.e.f.dc_first = last ? 1 : 0;
    P.e.f.H = H;
    P.e.s = EpiStore<1>{nullptr, 0, nullptr, 0, 0};
  } else {
    P.K = 4 * H;
    P.a = SrcRow{S.gates2 + (long)t * 4 * NH, 4L * H};
    P.b = SrcRow{S.Wx2, 4L * H};
    P.e.kind = 1;
    P.e.s = EpiStore<1>{S.dh1_seq + t * NH, H, nullptr, VD_ACT_NONE, 0};
    P.e.f = EpiLstmBwd<1>{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, H};
  }
  return P;
}

// Setup kernel (one small launch per direction): thread tau builds the <= 6 sub-problem descriptors of its tick into a
// table in device memory, thread 0 then prefix-sums the tile counts -- the persistent kernel reads descriptors with
// scalar loads on demand, like the per-tick grouped kernel reads its kernel arguments.
template <class Cfg, bool FWD, class Args, class Prob>
__global__ void __launch_bounds__(VD_L2_MAXT + 2) lstm2_seq_setup_kernel(Args a, Prob* tab, int* item_start) {
  __shared__ int ntiles[VD_L2_MAXT + 2];
  const int tau = threadIdx.x, nticks = a.Tmax + 2, H = a.H;
  const int tiles_n = FWD ? (4 * H) / Cfg::BN : H / Cfg::BN;
  if (tau < nticks) {
    int n = 0;
    for (int s = 0; s < VD_MAX_STACKS; ++s)
      for (int kind = 0; kind < VD_L2_KINDS; ++kind) {
        Prob P;
        int rows = 0;
        if (s < a.nstacks) {
          const int t = FWD ? l2_fwd_step(kind, tau, a.st[s].T) : l2_bwd_step(kind, tau, a.st[s].T);
          rows = t < 0 ? 0 : (int)a.nact[s][t];
          if constexpr (FWD) P = l2_fwd_prob(a.st[s], kind, t < 0 ? 0 : t, rows, H);
          else P = l2_bwd_prob(a.st[s], kind, t < 0 ? 0 : t, rows, H);
        } else {
          if constexpr (FWD) P = l2_fwd_prob(a.st[0], kind, 0, 0, H);
          else P = l2_bwd_prob(a.st[0], kind, 0, 0, H);
        }
        P.M = rows;
        tab[(tau * VD_MAX_STACKS + s) * VD_L2_KINDS + kind] = P;
        n += ((rows + Cfg::BM - 1) / Cfg::BM) * tiles_n;
      }
    ntiles[tau] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < nticks; ++i) {
      item_start[i] = acc;
      acc += ntiles[i];
    }
    item_start[nticks] = acc;
  }
}

// FWD = forward direction.  One workgroup = Cfg (the latency shape of the per-tick launches).
template <class Cfg, bool FWD, class Prob>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW)
lstm2_seq_kernel(const Prob* __restrict__ tab, const int* __restrict__ item_start, int nticks, int tiles_n, int RT,
                 unsigned* sync) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (Cfg::WK > 1) __builtin_amdgcn_s_setprio(3);
  int* const slot = reinterpret_cast<int*>(smem) + Cfg::LDS_BYTES / 4;   // scheduler mailbox behind the GEMM's LDS image
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane0 = (int)threadIdx.x & 63;
  const int total = item_start[nticks];
  unsigned* const head = sync;
  unsigned* const err = sync + VD_SEQ_ERR_WORD;
  unsigned* const cnt = sync + VD_SEQ_CNT0;
  auto pidx = [&](int tau, int s, int kind) { return (tau * VD_MAX_STACKS + s) * VD_L2_KINDS + kind; };
  for (;;) {
    if (wave == 0) {   // wave-uniform scheduler (see the note on seq_next): dequeue, decode, wait for the producers
      const int idx = (int)wave_fetch_add(head, 1u, lane0);
      int tau = -1, pi = 0, r = 0, j = 0;
      if (idx < total) {
        tau = 0;
        while (item_start[tau + 1] <= idx) ++tau;
        int rem = idx - item_start[tau];
        pi = pidx(tau, 0, 0);
        for (int k = 0; k < VD_MAX_STACKS * VD_L2_KINDS; ++k) {
          const int n = ((tab[pi].M + Cfg::BM - 1) / Cfg::BM) * tiles_n;
          if (rem < n) break;
          rem -= n;
          ++pi;
        }
        r = rem / tiles_n;
        j = rem - r * tiles_n;
        if (tau > 0) {
          // producers at tick tau-1 (same stack, same row tile); kinds: fwd 0 = L1, 1 = L2, 2 = X2; bwd 0 = L2, 1 = L1, 2 = X
          const int sk = pi - pidx(tau, 0, 0), s = sk / VD_L2_KINDS, kind = sk - s * VD_L2_KINDS;
          int dep0, dep1 = -1;
          if (FWD) { dep0 = kind == 1 ? 2 : 0; if (kind == 1) dep1 = 1; }
          else     { dep0 = kind == 1 ? 1 : 0; if (kind == 1) dep1 = 2; }
          bool waited = false;
          for (int d = 0; d < 2; ++d) {
            const int dep = d == 0 ? dep0 : dep1;
            if (dep < 0) continue;
            const int pp = pidx(tau - 1, s, dep);
            if (tab[pp].M <= r * Cfg::BM) continue;   // that row tile did not exist in the producer: state is the pre-filled zero
            unsigned* const c = cnt + (long)pp * RT + r;
            if (wave_load(c, lane0) < (unsigned)tiles_n) {
              const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
              for (;;) {
                __builtin_amdgcn_s_sleep(2);
                if (wave_load(c, lane0) >= (unsigned)tiles_n) break;
                if (wave_load(err, lane0) != 0u) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > VD_SEQ_TIMEOUT_TICKS) {
                  if (lane0 == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  break;
                }
              }
            }
            waited = true;
          }
          (void)waited;
        }
      }
      if (lane0 == 0) {
        slot[0] = tau; slot[1] = pi; slot[2] = r; slot[3] = j;
      }
    }
    __syncthreads();
    const int tau = __builtin_amdgcn_readfirstlane(slot[0]);
    if (tau < 0) break;
    const int pi = __builtin_amdgcn_readfirstlane(slot[1]);
    const int r = __builtin_amdgcn_readfirstlane(slot[2]);
    const int j = __builtin_amdgcn_readfirstlane(slot[3]);
    // the tile reads rows other workgroups wrote in this launch: every wave drops its stale L1 lines first
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const Prob& P = tab[pi];
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // per-lane address terms are rebuilt per tile, not kept live across tiles
    gemm_block<Cfg>(P.M, P.N, 0, P.K, r * Cfg::BM, j * Cfg::BN, -1, P.a, P.b, P.e, smem, tid);
    // publish: every wave drains its stores, then one wave releases at agent scope and bumps the counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane0 == 0) __hip_atomic_fetch_add(cnt + (long)pi * RT + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// host side: shared by both directions.  Returns VD_OK after launching, or 1 when the persistent path does not apply
// (caller falls back to the per-tick launches).
template <class Cfg, bool FWD, class Args, class Prob, class Stack>
static int launch_lstm2_seq(const Stack* st, int nstacks, int H, int Tmax, hipStream_t stream) {
  if (Tmax > VD_L2_MAXT) return 1;
  Args a;
  memset(&a, 0, sizeof(a));
  int Nmax = 0;
  long items = 0;
  for (int s = 0; s < nstacks; ++s) {
    a.st[s] = st[s];
    if (st[s].N > 65535) return 1;
    Nmax = st[s].N > Nmax ? st[s].N : Nmax;
    for (int t = 0; t < st[s].T; ++t) {
      const int rows = st[s].nact ? st[s].nact[t] : st[s].N;
      a.nact[s][t] = (unsigned short)(rows < 0 ? 0 : rows);
      items += 3L * vd_cdiv(rows, Cfg::BM);
    }
    a.st[s].nact = nullptr;
  }
  a.nstacks = nstacks; a.H = H; a.Tmax = Tmax;
  a.row_tiles = vd_cdiv(Nmax, Cfg::BM);
  a.sync = nullptr;
  const int tiles_n = FWD ? (4 * H) / Cfg::BN : H / Cfg::BN;
  items *= tiles_n;
  if (items <= 0) return VD_OK;
  const int nprob = (Tmax + 2) * VD_MAX_STACKS * VD_L2_KINDS;
  const size_t words = (size_t)VD_SEQ_CNT0 + (size_t)nprob * a.row_tiles;
  const size_t tab_bytes = (size_t)nprob * sizeof(Prob) + (VD_L2_MAXT + 4) * sizeof(int);
  VdStreamScratch scr;
  if (int rc = vd_stream_scratch(stream, tab_bytes, words * sizeof(unsigned), &scr)) return rc;
  Prob* tab = reinterpret_cast<Prob*>(scr.wht);
  int* item_start = reinterpret_cast<int*>(reinterpret_cast<char*>(scr.wht) + (size_t)nprob * sizeof(Prob));
  VD_HIP(hipMemsetAsync(scr.sync, 0, words * sizeof(unsigned), stream));
  hipLaunchKernelGGL((lstm2_seq_setup_kernel<Cfg, FWD, Args, Prob>), dim3(1), dim3(VD_L2_MAXT + 2), 0, stream, a, tab, item_start);
  VD_LAUNCH_CHECK();
  auto kern = lstm2_seq_kernel<Cfg, FWD, Prob>;
  const int lds = Cfg::LDS_BYTES + 64;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  // one workgroup per CU: the slot left beside three throughput workgroups; never more workgroups than items
  long grid = (long)vd_tune_get("VD_LSTM2_SEQ_WGS_PER_CU", 1) * vd_num_cus();
  if (const int abs_wgs = vd_tune_get("VD_LSTM2_SEQ_WGS", 0)) grid = abs_wgs;
  if (grid > items) grid = items;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::THREADS), lds, stream, (const Prob*)tab, (const int*)item_start,
                     Tmax + 2, tiles_n, a.row_tiles, scr.sync);
  VD_LAUNCH_CHECK();
  return VD_OK;
}

// the per-tick grouped launches of the backward direction, for one tile configuration (NT = column tiles per wave)
template <class Cfg, int NT>
static int lstm2_backward_ticks(const vd_lstm2_bwd_t* st, int nstacks, int H, int Tmax, hipStream_t stream) {
  for (int tau = 0; tau < Tmax + 2; ++tau) {
    GroupArgs<TickBwdProbT<NT>, 3 * VD_MAX_STACKS> g;
    g.nprob = 0;
    for (int s = 0; s < nstacks; ++s) {
      const vd_lstm2_bwd_t& S = st[s];
      const long NH = (long)S.N * H;
      for (int layer = 2; layer >= 1; --layer) {  // cell backward: L2 at t = T-1-tau, L1 at t = T+1-tau
        const int t = layer == 2 ? S.T - 1 - tau : S.T + 1 - tau;
        if (t < 0 || t >= S.T) continue;
        float* gates = layer == 2 ? S.gates2 : S.gates1;
        const float* c = layer == 2 ? S.c2 : S.c1;
        const bool last = (t == S.T - 1);
        const int rows = S.nact ? S.nact[t] : S.N;
        if (rows <= 0) continue;
        TickBwdProbT<NT>& P = g.p[g.nprob++];
        P.M = rows; P.N = H; P.K = last ? 0 : 4 * H;
        P.a = SrcRow{last ? gates : gates + (long)(t + 1) * 4 * NH, 4L * H};
        P.b = SrcRow{layer == 2 ? S.Wh2 : S.Wh1, 4L * H};
        P.e.kind = 0;
        P.e.f.dh_a = layer == 2 ? (last ? S.dh_last2 : nullptr) : S.dh1_seq + t * NH;
        P.e.f.dh_b = nullptr;
        P.e.f.gates = gates + (long)t * 4 * NH;
        P.e.f.c_t = c + t * NH;
        P.e.f.c_prev = t ? c + (t - 1) * NH : nullptr;
        P.e.f.dc = layer == 2 ? S.dc2 : S.dc1;
        P.e.f.dc_first = last ? 1 : 0;
        P.e.f.H = H;
        P.e.s = EpiStore<NT>{nullptr, 0, nullptr, 0, 0};
      }
      const int t = S.T - tau;  // dh1[t] = da2[t] * Wx2^T
      if (t >= 0 && t < S.T && (S.nact ? S.nact[t] : S.N) > 0) {
        TickBwdProbT<NT>& P = g.p[g.nprob++];
        P.M = S.nact ? S.nact[t] : S.N; P.N = H; P.K = 4 * H;
        P.a = SrcRow{S.gates2 + (long)t * 4 * NH, 4L * H};
        P.b = SrcRow{S.Wx2, 4L * H};
        P.e.kind = 1;
        P.e.s = EpiStore<NT>{S.dh1_seq + t * NH, H, nullptr, VD_ACT_NONE, 0};
        P.e.f = EpiLstmBwd<NT>{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, H};
      }
    }
    if (g.nprob == 0) continue;
    if (int rc = launch_grouped<Cfg>(g, stream)) return rc;
  }
  return VD_OK;
}

extern "C" {

// ---------------------------------------------------------------------------
// Row chains: fork/join of library-owned streams around a per-row-independent recurrence.
// ---------------------------------------------------------------------------
#define VD_MAX_CHAINS 4
struct RowChains {
  int n = 1;
  int row0[VD_MAX_CHAINS + 1];
  hipStream_t stream[VD_MAX_CHAINS];

  struct Pool {
    hipStream_t side[VD_MAX_CHAINS - 1];
    hipEvent_t fork_ev, join_ev[VD_MAX_CHAINS - 1];
    bool ready = false;
  };
  static Pool& pool() {
    static thread_local Pool p;
    return p;
  }

  int fork(int N, hipStream_t s, int want) {
    n = (N >= 4096 && want > 1) ? (want > VD_MAX_CHAINS ? VD_MAX_CHAINS : want) : 1;
    stream[0] = s;
    row0[0] = 0;
    if (n == 1) {
      row0[1] = N;
      return VD_OK;
    }
    Pool& p = pool();
    if (!p.ready) {
      int prio = 0;
      (void)hipStreamGetPriority(s, &prio);
      for (int i = 0; i < VD_MAX_CHAINS - 1; ++i) {
        VD_HIP(hipStreamCreateWithPriority(&p.side[i], hipStreamNonBlocking, prio));
        VD_HIP(hipEventCreateWithFlags(&p.join_ev[i], hipEventDisableTiming));
      }
      VD_HIP(hipEventCreateWithFlags(&p.fork_ev, hipEventDisableTiming));
      p.ready = true;
    }
    // chain boundaries on 128-row tile boundaries
    const int tiles = (N + 127) / 128;
    for (int i = 1; i < n; ++i) row0[i] = (int)((long)tiles * i / n) * 128;
    row0[n] = N;
    VD_HIP(hipEventRecord(p.fork_ev, s));
    for (int i = 1; i < n; ++i) {
      stream[i] = p.side[i - 1];
      VD_HIP(hipStreamWaitEvent(stream[i], p.fork_ev, 0));
    }
    return VD_OK;
  }

  int join(hipStream_t s) {
    if (n == 1) return VD_OK;
    Pool& p = pool();
    for (int i = 1; i < n; ++i) {
      VD_HIP(hipEventRecord(p.join_ev[i - 1], stream[i]));
      VD_HIP(hipStreamWaitEvent(s, p.join_ev[i - 1], 0));
    }
    return VD_OK;
  }
};

// see include/visdial_hip.h
int vd_lstm_forward(const float* xproj, int64_t x_tstride, int64_t x_ld, const int32_t* tok_gather,
                    const int32_t* tok_mask, const float* Wh, const float* h0, const float* c0, float* gates,
                    float* h, float* c, int T, int N, int H, int flags, void* stream) {
  VD_CHECK_ARG(T >= 0 && N >= 0 && H > 0 && H % 32 == 0, "vd_lstm_forward: bad dims T=%d N=%d H=%d", T, N, H);
  VD_CHECK_ARG(xproj && Wh && gates && h && c, "vd_lstm_forward: null pointer");
  VD_CHECK_ARG((h0 == nullptr) == (c0 == nullptr), "vd_lstm_forward: h0 and c0 must both be set or both null");
  VD_CHECK_ARG(x_ld % 4 == 0, "vd_lstm_forward: x_ld must be a multiple of 4");
  // the step epilogue addresses its tensors with 32-bit byte offsets (buffer descriptors, common.h): one step's slice of every
  // tensor must stay below 4 GB (gates: N * 4H * 4 bytes; the dense projection: N * x_ld * 4; a projection TABLE: rows * x_ld * 4)
  VD_CHECK_ARG((long)N * 4 * H * 4 < (1L << 32) && (tok_gather || (long)N * x_ld * 4 < (1L << 32)),
               "vd_lstm_forward: N=%d rows x 4H=%d exceed 4 GB per step: split the batch", N, 4 * H);
  hipStream_t s = (hipStream_t)stream;
  const long NH = (long)N * H;
  // The recurrence is independent per row: throughput shapes run as row chains on separate streams so the
  // tail of one chain's step kernel is filled by the other chain's workgroups (no chip-wide drain per step).
  float* WhT = nullptr;
  const bool bf16 = (flags & VD_FLAG_BF16) && N >= 2048 && H % 32 == 0;
  const bool glds = (use_glds_fwd(N, H) || bf16) && T > 1;   // both paths multiply by the transposed copy
  static const int nchains = env_int("VD_LSTM_CHAINS_FWD", 1);
  const bool persist = glds && !bf16 && nchains <= 1 && use_persistent("VD_LSTM_PERSIST_FWD");
  VdStreamScratch scr;
  if (glds) {
    const int tiles_m = vd_cdiv(N, CfgF9::BM);
    if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * (bf16 ? 6 : 4), seq_sync_bytes(T, tiles_m), &scr)) return rc0;
    WhT = scr.wht;
    hipLaunchKernelGGL(wh_gate_transpose_kernel, dim3(4 * H / 32, H / 32), dim3(256), 0, s, Wh, WhT, H);
    VD_LAUNCH_CHECK();
  }
  if (persist) {
    // one launch for all T steps (see "Persistent sequence kernels" above)
    const int tiles_m = vd_cdiv(N, CfgF9::BM), tiles_n = vd_cdiv(4 * H, CfgF9::BN);
    VD_HIP(hipMemsetAsync(scr.sync, 0, seq_sync_bytes(T, tiles_m), s));
    LstmSeqFwdArgs a;
    a.xproj = xproj; a.x_tstride = x_tstride; a.xld = x_ld;
    a.tok_gather = tok_gather; a.tok_mask = tok_mask;
    a.WhT = WhT; a.h0 = h0; a.c0 = c0; a.gates = gates; a.h = h; a.c = c;
    a.N = N; a.H = H; a.rotate = vd_tune_get("VD_GEMM_ROTATE", 1);
    a.t_base = 0;
    a.sc = SeqSched{scr.sync, T, tiles_m, tiles_n, vd_tune_get("VD_LSTM_STAGGER_US", 0) * 100};
    return launch_seq(lstm_seq_fwd_kernel<CfgF9>, a, CfgF9::LDS_BYTES, CfgF9::THREADS, T * tiles_m * tiles_n, s);
  }
  // VD_LSTM_STEP_QUEUE=1 (A/B): every step is ONE round of resident workgroups that pull tiles from per-XCD queues
  // (the persistent kernel with a one-step work list): the launch dispatches at once, so later packets of the shared
  // hardware queue (the encoder's ticks) are not held back until the step kernel has started its last workgroup
  const bool step_queue = glds && !bf16 && nchains <= 1 && vd_tune_get("VD_LSTM_STEP_QUEUE", 0) != 0;
  size_t sq_words = 0;
  if (step_queue) {
    const int tiles_m = vd_cdiv(N, CfgF9::BM);
    sq_words = ((size_t)VD_SEQ_CNT0 + tiles_m + 31) / 32 * 32;
    if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * sizeof(float), sq_words * T * sizeof(unsigned), &scr)) return rc0;
    VD_HIP(hipMemsetAsync(scr.sync, 0, sq_words * T * sizeof(unsigned), s));
  }
  // bf16 pass: the step kernels also write a bf16 copy of h (the operand of the dWh contraction of the same pass)
  vd_bf16_bits* h16 = nullptr;
  if (bf16 && vd_tune_get("VD_BF16_SHADOW", 1)) {
    if (int rc0 = vd_bf16_shadow_get(0, h, (size_t)T * NH, &h16)) return rc0;
  } else {
    vd_bf16_shadow_invalidate(h, (size_t)T * NH);
  }
  // ... and READ the shadow of h_{t-1} and a bf16 copy of the transposed weights through the LDS-DMA pipeline (VD_BF16_GLDS)
  vd_bf16_bits* WhT16 = nullptr;
  if (h16 && glds && vd_tune_get("VD_BF16_GLDS", 1)) {
    WhT16 = reinterpret_cast<vd_bf16_bits*>(scr.wht + (size_t)4 * H * H);
    if (int rc0 = weights_to_bf16(WhT, WhT16, 4L * H * H, s)) return rc0;
  }
  RowChains rc_;
  int rc = rc_.fork(N, s, nchains);
  if (rc) return rc;
  for (int t = 0; t < T; ++t) {
    if (step_queue && t > 0) {
      const int tiles_m = vd_cdiv(N, CfgF9::BM), tiles_n = vd_cdiv(4 * H, CfgF9::BN);
      LstmSeqFwdArgs a;
      a.xproj = xproj; a.x_tstride = x_tstride; a.xld = x_ld;
      a.tok_gather = tok_gather; a.tok_mask = tok_mask;
      a.WhT = WhT; a.h0 = h0; a.c0 = c0; a.gates = gates; a.h = h; a.c = c;
      a.N = N; a.H = H; a.rotate = vd_tune_get("VD_GEMM_ROTATE", 1);
      a.t_base = t;
      a.sc = SeqSched{scr.sync + sq_words * t, 1, tiles_m, tiles_n, 0};
      rc = launch_seq(lstm_seq_fwd_kernel<CfgF9>, a, CfgF9::LDS_BYTES, CfgF9::THREADS, tiles_m * tiles_n, s);
      if (rc) return rc;
      continue;
    }
    for (int ch = 0; ch < rc_.n; ++ch) {
      const long r0 = rc_.row0[ch];
      const int nr = rc_.row0[ch + 1] - rc_.row0[ch];
      const float* hp = t ? h + (t - 1) * NH + r0 * H : (h0 ? h0 + r0 * H : nullptr);
      const float* cp = t ? c + (t - 1) * NH + r0 * H : (c0 ? c0 + r0 * H : nullptr);
      EpiLstmFwd e;
      e.xproj = xproj + (long)t * x_tstride + (tok_gather ? 0 : r0 * x_ld);
      e.xld = x_ld;
      e.tok_gather = tok_gather ? tok_gather + (long)t * N + r0 : nullptr;
      e.tok_mask = tok_mask ? tok_mask + (long)t * N + r0 : nullptr;
      e.c_prev = cp;
      e.gates = gates + (long)t * 4 * NH + r0 * 4 * H;
      e.c_out = c + t * NH + r0 * H;
      e.h_out = h + t * NH + r0 * H;
      e.H = H;
      e.h16 = h16 ? h16 + t * NH + r0 * H : nullptr;
      if (bf16 && hp && WhT16 && t > 0 && vd_tune_get("VD_BF16_OCC4", 0))
        rc = launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 40960, 1>, false>(nr, 4 * H, H / 2, 1, reinterpret_cast<const float*>(h16 + (t - 1) * NH + r0 * H),
                                                (long)H / 2, reinterpret_cast<const float*>(WhT16), (long)H / 2, e, rc_.stream[ch]);
      else if (bf16 && hp && WhT16 && t > 0)
        rc = launch_gemm_glds<CfgF9bf16, false>(nr, 4 * H, H / 2, 1, reinterpret_cast<const float*>(h16 + (t - 1) * NH + r0 * H),
                                                (long)H / 2, reinterpret_cast<const float*>(WhT16), (long)H / 2, e, rc_.stream[ch]);
      else if (bf16 && hp)
        rc = launch_gemm<CfgFbf16>(nr, 4 * H, H, 1, SrcRow{hp, H}, SrcRow{WhT, H}, e, rc_.stream[ch]);
      else if (glds && hp) {
        const int epi = vd_tune_get("VD_LSTM_FWD_EPI_SEQ", 0);
        if (vd_tune_get("VD_LSTM_FWD_DEEP", 0)) {
          // A/B: three LDS buffers for BOTH operands (48 KB request, still 3 workgroups per CU but no room for a latency
          // workgroup beside them): is the epilogue's cost the extra latency its traffic puts on the K loops' DMA?
          rc = launch_gemm_glds<GemmCfg<4, 1, 4, 16, 0, 4, 49152>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
        } else if (const int wide = vd_tune_get("VD_LSTM_FWD_NT8", 0)) {
          // A/B: 128 x 256 workgroup tiles, FOUR waves of 32 x 256 (two gate groups per wave, 128 accumulator registers,
          // 2 workgroups per CU = 2 waves per SIMD): the 25 % operand-traffic cut of the 256 x 128 variant below without
          // its 4 waves per SIMD, and twice the MFMA burst per barrier.  wide = 2: two A buffers (56 KB request)
          if (epi == 2) {
            EpiLstmFwdT<2> e2{e.xproj, e.xld, e.tok_gather, e.tok_mask, e.c_prev, e.gates, e.c_out, e.h_out, e.H};
            rc = launch_gemm_glds<GemmCfg<4, 1, 8, 16, 0, 2, 73728>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e2, rc_.stream[ch]);
          } else if (wide == 2) rc = launch_gemm_glds<GemmCfg<4, 1, 8, 16, 0, 2, 57344>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
          else rc = launch_gemm_glds<GemmCfg<4, 1, 8, 16, 0, 2, 73728>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
        } else if (const int big = vd_tune_get("VD_LSTM_FWD_BM256", 0)) {
          // A/B: 256 x 128 workgroup tiles (8 waves): 25 % less operand traffic per FLOP than 128 x 128 -- the step kernels
          // pay ~6.5 % per extra 25 % of memory traffic through the CU (profiles/r03_experiments.txt section 13)
          if (big == 2) rc = launch_gemm_glds<GemmCfg<8, 1, 4, 16, 0, 2, 49152>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
          else rc = launch_gemm_glds<GemmCfg<8, 1, 4, 16, 0, 2, 65536>, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
        } else if (epi == 1) {
          EpiLstmFwdT<1> e1{e.xproj, e.xld, e.tok_gather, e.tok_mask, e.c_prev, e.gates, e.c_out, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e1, rc_.stream[ch]);
        } else if (epi == 4) {
          EpiLstmFwdSpreadNoWaitT<3> e4{e.xproj, e.xld, e.tok_gather, e.gates, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e4, rc_.stream[ch]);
        } else if (epi == 5) {
          EpiLstmFwdSpreadNoWaitT<1> e5{e.xproj, e.xld, e.tok_gather, e.gates, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e5, rc_.stream[ch]);
        } else if (epi == 6) {
          EpiLstmFwdSpreadNoWaitT<2> e6{e.xproj, e.xld, e.tok_gather, e.gates, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e6, rc_.stream[ch]);
        } else if (epi == 7) {
          EpiLstmFwdValuProbeT<0> e7{e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e7, rc_.stream[ch]);
        } else if (epi == 8) {
          EpiLstmFwdValuProbeT<1> e8{e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e8, rc_.stream[ch]);
        } else if (epi == 3) {
          EpiLstmFwdSpread e3{e.xproj, e.xld, e.tok_gather, e.gates, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e3, rc_.stream[ch]);
        } else if (epi == 2) {
          EpiLstmFwdT<2> e2{e.xproj, e.xld, e.tok_gather, e.tok_mask, e.c_prev, e.gates, e.c_out, e.h_out, e.H};
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e2, rc_.stream[ch]);
        } else {
          rc = launch_gemm_glds<CfgF9, false>(nr, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, rc_.stream[ch]);
        }
      }
      else
        rc = lstm_step_fwd(hp, Wh, nr, H, hp ? H : 0, e, rc_.stream[ch]);
      if (rc) return rc;
    }
  }
  return rc_.join(s);
}

// Library-owned side stream + event ring for work that trails the backward recurrence step by step.
#define VD_TRAIL_EVENTS 64
struct TrailStream {
  hipStream_t side = nullptr;
  hipEvent_t ev[VD_TRAIL_EVENTS];
  hipEvent_t done;
  bool ready = false;
  int init(hipStream_t s) {
    if (ready) return VD_OK;
    int prio = 0;
    (void)hipStreamGetPriority(s, &prio);
    VD_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio));
    for (int i = 0; i < VD_TRAIL_EVENTS; ++i) VD_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    VD_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    ready = true;
    return VD_OK;
  }
};
static TrailStream& trail_stream() {
  static thread_local TrailStream t;
  return t;
}

int vd_lstm_backward(const float* Wh, float* gates, const float* c, const float* c0, const float* dh_seq,
                     const float* dh_last, const float* dc_last, float* dc_work, float* dh0, const float* h_seq,
                     float* dWh_acc, int T, int N, int H, int flags, void* stream) {
  VD_CHECK_ARG(T >= 1 && N >= 0 && H > 0 && H % 32 == 0, "vd_lstm_backward: bad dims T=%d N=%d H=%d", T, N, H);
  VD_CHECK_ARG(Wh && gates && c && dc_work, "vd_lstm_backward: null pointer");
  VD_CHECK_ARG((h_seq == nullptr) == (dWh_acc == nullptr), "vd_lstm_backward: h_seq and dWh_acc go together");
  VD_CHECK_ARG((long)N * 4 * H * 4 < (1L << 32), "vd_lstm_backward: N=%d rows x 4H=%d exceed 4 GB per step: split the batch", N, 4 * H);
  hipStream_t s = (hipStream_t)stream;
  const long NH = (long)N * H;
  if (dc_last && dc_last != dc_work) VD_HIP(hipMemcpyAsync(dc_work, dc_last, NH * sizeof(float), hipMemcpyDeviceToDevice, s));
  // bf16 pass: the step kernels also write a bf16 copy of da (the other operand of the dWh contraction)
  vd_bf16_bits* da16 = nullptr;
  vd_bf16_bits* Wh16 = nullptr;
  if ((flags & VD_FLAG_BF16) && N >= 2048 && H % 32 == 0 && vd_tune_get("VD_BF16_SHADOW", 1)) {
    if (int rc0 = vd_bf16_shadow_get(1, gates, (size_t)T * 4 * NH, &da16)) return rc0;
    if (T > 1 && vd_tune_get("VD_BF16_GLDS", 1)) {
      VdStreamScratch wscr;
      if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * 6, 0, &wscr)) return rc0;
      Wh16 = reinterpret_cast<vd_bf16_bits*>(wscr.wht + (size_t)4 * H * H);
      if (int rc0 = weights_to_bf16(Wh, Wh16, 4L * H * H, s)) return rc0;
    }
  } else {
    vd_bf16_shadow_invalidate(gates, (size_t)T * 4 * NH);
  }
  // Recurrent weight gradient dWh += sum_{t>=1} h_{t-1}^T da_t.  Default: one contraction over all (T-1)*N
  // rows after the recurrence.  VD_LSTM_WGRAD_OVERLAP=1 issues it as one chunk per step on a side stream as
  // soon as step t has produced da_t, so the chunks fill the tails / epilogue phases of the step kernels:
  // 13.9 -> 12.6 ms for both together on the option LSTM alone, but -1.8 % on the whole training step (the
  // encoder streams lose the idle phases they live on), hence opt-in.
  static const int overlap = env_int("VD_LSTM_WGRAD_OVERLAP", 0);
  const bool trail = dWh_acc && N >= 2048 && T > 1 && T <= VD_TRAIL_EVENTS && overlap;
  TrailStream& ts = trail_stream();
  if (trail)
    if (int rc0 = ts.init(s)) return rc0;
  static const int nchains = env_int("VD_LSTM_CHAINS_BWD", 1);
  const bool persist = use_glds_bwd(N, H) && !(flags & VD_FLAG_BF16) && T > 1 && nchains <= 1 && !trail &&
                       use_persistent("VD_LSTM_PERSIST_BWD");
  if (persist) {
    const int tiles_m = vd_cdiv(N, CfgB11::BM), tiles_n = vd_cdiv(H, CfgB11::BN);
    VdStreamScratch scr;
    if (int rc0 = vd_stream_scratch(s, 0, seq_sync_bytes(T, tiles_m), &scr)) return rc0;
    VD_HIP(hipMemsetAsync(scr.sync, 0, seq_sync_bytes(T, tiles_m), s));
    LstmSeqBwdArgs a;
    a.Wh = Wh; a.gates = gates; a.c = c; a.c0 = c0; a.dh_seq = dh_seq; a.dh_last = dh_last; a.dc = dc_work;
    a.dc_has_last = dc_last ? 1 : 0;
    a.T = T; a.N = N; a.H = H; a.rotate = vd_tune_get("VD_GEMM_ROTATE", 1);
    a.s_base = 0;
    a.sc = SeqSched{scr.sync, T, tiles_m, tiles_n, vd_tune_get("VD_LSTM_STAGGER_US", 0) * 100};
    const int rc0 = vd_tune_get("VD_LSTM_BWD_BATCH2", 0)
                        ? launch_seq(lstm_seq_bwd_kernel<CfgB12>, a, CfgB12::LDS_BYTES, CfgB12::THREADS, T * tiles_m * tiles_n, s)
                        : launch_seq(lstm_seq_bwd_kernel<CfgB11>, a, CfgB11::LDS_BYTES, CfgB11::THREADS, T * tiles_m * tiles_n, s);
    if (rc0) return rc0;
  }
  const bool step_queue = use_glds_bwd(N, H) && !(flags & VD_FLAG_BF16) && T > 1 && nchains <= 1 && !trail && !persist &&
                          vd_tune_get("VD_LSTM_STEP_QUEUE", 0) != 0;
  VdStreamScratch sq_scr;
  size_t sq_words = 0;
  if (step_queue) {
    const int tiles_m = vd_cdiv(N, CfgB11::BM);
    sq_words = ((size_t)VD_SEQ_CNT0 + tiles_m + 31) / 32 * 32;
    if (int rc0 = vd_stream_scratch(s, 0, sq_words * T * sizeof(unsigned), &sq_scr)) return rc0;
    VD_HIP(hipMemsetAsync(sq_scr.sync, 0, sq_words * T * sizeof(unsigned), s));
  }
  RowChains rc_;
  int rc = rc_.fork(N, s, persist ? 1 : nchains);
  if (rc) return rc;
  for (int t = T - 1; t >= 0 && !persist; --t) {
    const bool last = (t == T - 1);
    if (step_queue && !last) {
      const int tiles_m = vd_cdiv(N, CfgB11::BM), tiles_n = vd_cdiv(H, CfgB11::BN);
      LstmSeqBwdArgs a;
      a.Wh = Wh; a.gates = gates; a.c = c; a.c0 = c0; a.dh_seq = dh_seq; a.dh_last = dh_last; a.dc = dc_work;
      a.dc_has_last = dc_last ? 1 : 0;
      a.T = T; a.N = N; a.H = H; a.rotate = vd_tune_get("VD_GEMM_ROTATE", 1);
      a.s_base = T - 1 - t;
      a.sc = SeqSched{sq_scr.sync + sq_words * t, 1, tiles_m, tiles_n, 0};
      rc = launch_seq(lstm_seq_bwd_kernel<CfgB11>, a, CfgB11::LDS_BYTES, CfgB11::THREADS, tiles_m * tiles_n, s);
      if (rc) return rc;
      continue;
    }
    for (int ch = 0; ch < rc_.n; ++ch) {
      const long r0 = rc_.row0[ch];
      const int nr = rc_.row0[ch + 1] - rc_.row0[ch];
      const float* da_next = last ? nullptr : gates + (long)(t + 1) * 4 * NH + r0 * 4 * H;
      const float* c0r = c0 ? c0 + r0 * H : nullptr;
      rc = lstm_step_bwd(da_next, Wh, nr, H, last ? 0 : 4 * H, dh_seq ? dh_seq + t * NH + r0 * H : nullptr,
                         (last && dh_last) ? dh_last + r0 * H : nullptr, gates + (long)t * 4 * NH + r0 * 4 * H,
                         c + t * NH + r0 * H, t ? c + (t - 1) * NH + r0 * H : c0r, dc_work + r0 * H,
                         (last && !dc_last) ? 1 : 0, rc_.stream[ch], flags,
                         da16 ? da16 + (long)t * 4 * NH + r0 * 4 * H : nullptr,
                         (da16 && !last) ? da16 + (long)(t + 1) * 4 * NH + r0 * 4 * H : nullptr, Wh16);
      if (rc) return rc;
    }
    if (trail && t >= 1 && rc_.n == 1) {
      VD_HIP(hipEventRecord(ts.ev[t], s));
      VD_HIP(hipStreamWaitEvent(ts.side, ts.ev[t], 0));
      rc = vd_gemm_tn_acc(h_seq + (long)(t - 1) * NH, H, gates + (long)t * 4 * NH, 4L * H, dWh_acc, 4L * H, H, 4 * H, N,
                          flags, ts.side);
      if (rc) return rc;
    }
  }
  rc = rc_.join(s);
  if (rc) return rc;
  if (dWh_acc && T > 1) {
    if (trail && rc_.n == 1) {
      VD_HIP(hipEventRecord(ts.done, ts.side));
      VD_HIP(hipStreamWaitEvent(s, ts.done, 0));
    } else {   // one contraction over all (T-1)*N rows
      rc = vd_gemm_tn_acc(h_seq, H, gates + 4 * NH, 4L * H, dWh_acc, 4L * H, H, 4 * H, (T - 1) * N, flags, s);
      if (rc) return rc;
    }
  }
  if (dh0) {
    // gradient w.r.t. the initial hidden state: da_0 * Wh^T (dc_work already holds dL/dc0)
    SrcRow a{gates, 4L * H};
    SrcRow b{Wh, 4L * H};
    EpiStore<4> e{dh0, H, nullptr, VD_ACT_NONE, 0};
    rc = launch_gemm<CfgB1>(N, H, 4 * H, 1, a, b, e, s);
    if (rc) return rc;
  }
  return VD_OK;
}

// see include/visdial_hip.h
int vd_lstm_seq_status(void* stream, int* timed_out) {
  VD_CHECK_ARG(timed_out, "vd_lstm_seq_status: null");
  *timed_out = 0;
  hipStream_t s = (hipStream_t)stream;
  VdStreamScratch scr;
  if (int rc = vd_stream_scratch(s, 0, 0, &scr)) return rc;
  VD_HIP(hipStreamSynchronize(s));
  if (!scr.sync || scr.sync_bytes < (VD_SEQ_ERR_WORD + 1) * sizeof(unsigned)) return VD_OK;  // no persistent launch yet
  unsigned v = 0;
  VD_HIP(hipMemcpy(&v, scr.sync + VD_SEQ_ERR_WORD, sizeof(v), hipMemcpyDeviceToHost));
  *timed_out = (int)v;
  return VD_OK;
}

int vd_lstm2_forward(const vd_lstm2_fwd_t* st, int nstacks, int H, void* stream) {
  VD_CHECK_ARG(st && nstacks >= 1 && nstacks <= VD_MAX_STACKS && H > 0 && H % 32 == 0,
               "vd_lstm2_forward: bad args (nstacks=%d, max %d)", nstacks, VD_MAX_STACKS);
  int Tmax = 0;
  for (int s = 0; s < nstacks; ++s) {
    VD_CHECK_ARG(st[s].T >= 1 && st[s].N >= 1 && st[s].Wh1 && st[s].Wx2 && st[s].b2 && st[s].Wh2 && st[s].gates1 &&
                     st[s].h1 && st[s].c1 && st[s].gates2 && st[s].h2 && st[s].c2,
                 "vd_lstm2_forward: stack %d has null/empty fields", s);
    Tmax = st[s].T > Tmax ? st[s].T : Tmax;
  }
  if (vd_tune_get("VD_LSTM2_PERSIST", 0)) {   // one persistent launch for all T + 2 ticks (see lstm2_seq_kernel); opt-in
    const int rc = launch_lstm2_seq<CfgFwdSmallC, true, L2SeqFwdArgs, TickFwdProb>(st, nstacks, H, Tmax, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  static const int scfg_fwd = env_int("VD_LSTM_FWD_SMALL", 2);
  if (scfg_fwd == 7 && H % 32 == 0) {
    // LDS-DMA ticks (gemm_block_glds_wk): the three weight matrices of every stack as k-contiguous rows -- Wh1 / Wh2
    // gate-interleaved (the cell-update epilogue wants i,f,o,g of a hidden unit in one wave), Wx2 plain -- rebuilt per call
    const size_t mat = (size_t)4 * H * H;
    VdStreamScratch scr;
    if (int rc = vd_stream_scratch((hipStream_t)stream, 3 * VD_MAX_STACKS * mat * sizeof(float), 0, &scr)) return rc;
    for (int s = 0; s < nstacks; ++s) {
      const float* src[3] = {st[s].Wh1, st[s].Wh2, st[s].Wx2};
      for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(wh_gate_transpose_kernel, dim3(4 * H / 32, H / 32), dim3(256), 0, (hipStream_t)stream, src[i],
                           scr.wht + (3 * s + i) * mat, H, i == 2 ? 1 : 0);
        VD_LAUNCH_CHECK();
      }
    }
    for (int tau = 0; tau < Tmax + 2; ++tau) {
      GroupArgs<TickFwdProbT, 3 * VD_MAX_STACKS> g;
      g.nprob = 0;
      for (int s = 0; s < nstacks; ++s) {
        const vd_lstm2_fwd_t& S = st[s];
        const long NH = (long)S.N * H;
        for (int layer = 1; layer <= 2; ++layer) {
          const int t = layer == 1 ? tau : tau - 2;
          if (t < 0 || t >= S.T) continue;
          float* gates = layer == 1 ? S.gates1 : S.gates2;
          float* h = layer == 1 ? S.h1 : S.h2;
          float* c = layer == 1 ? S.c1 : S.c2;
          const int rows = S.nact ? S.nact[t] : S.N;
          if (rows <= 0) continue;
          TickFwdProbT& P = g.p[g.nprob++];
          P.M = rows; P.N = 4 * H; P.K = t ? H : 0;
          P.a = t ? h + (t - 1) * NH : h; P.lda = H;
          P.bt = scr.wht + (3 * s + (layer == 1 ? 0 : 1)) * mat; P.ldb = H;
          P.e.kind = 0;
          P.e.f.xproj = gates + (long)t * 4 * NH; P.e.f.xld = 4L * H;
          P.e.f.tok_gather = nullptr;
          P.e.f.tok_mask = S.tok_mask ? S.tok_mask + (long)t * S.N : nullptr;
          P.e.f.c_prev = t ? c + (t - 1) * NH : nullptr;
          P.e.f.gates = gates + (long)t * 4 * NH; P.e.f.c_out = c + t * NH; P.e.f.h_out = h + t * NH; P.e.f.H = H;
          P.e.s = EpiStore<4>{nullptr, 0, nullptr, 0, 0};
        }
        const int t = tau - 1;
        if (t >= 0 && t < S.T && (S.nact ? S.nact[t] : S.N) > 0) {
          TickFwdProbT& P = g.p[g.nprob++];
          P.M = S.nact ? S.nact[t] : S.N; P.N = 4 * H; P.K = H;
          P.a = S.h1 + t * NH; P.lda = H;
          P.bt = scr.wht + (3 * s + 2) * mat; P.ldb = H;
          P.e.kind = 1;
          P.e.s = EpiStore<4>{S.gates2 + (long)t * 4 * NH, 4L * H, S.b2, VD_ACT_NONE, 0};
          P.e.f = EpiLstmFwdTick{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, H};
        }
      }
      if (g.nprob == 0) continue;
      if (int rc = launch_grouped_glds_wk<4>(g, (hipStream_t)stream)) return rc;
    }
    return VD_OK;
  }
  for (int tau = 0; tau < Tmax + 2; ++tau) {
    GroupArgs<TickFwdProb, 3 * VD_MAX_STACKS> g;
    g.nprob = 0;
    for (int s = 0; s < nstacks; ++s) {
      const vd_lstm2_fwd_t& S = st[s];
      const long NH = (long)S.N * H;
      for (int layer = 1; layer <= 2; ++layer) {  // cell updates: L1 at t = tau, L2 at t = tau - 2
        const int t = layer == 1 ? tau : tau - 2;
        if (t < 0 || t >= S.T) continue;
        float* gates = layer == 1 ? S.gates1 : S.gates2;
        float* h = layer == 1 ? S.h1 : S.h2;
        float* c = layer == 1 ? S.c1 : S.c2;
        const int rows = S.nact ? S.nact[t] : S.N;
        if (rows <= 0) continue;
        TickFwdProb& P = g.p[g.nprob++];
        P.M = rows; P.N = 4 * H; P.K = t ? H : 0;
        P.a = SrcRow{t ? h + (t - 1) * NH : h, H};
        P.b = SrcKSel{layer == 1 ? S.Wh1 : S.Wh2, 4L * H, H, 1};
        P.e.kind = 0;
        P.e.f.xproj = gates + (long)t * 4 * NH; P.e.f.xld = 4L * H;
        P.e.f.tok_gather = nullptr;
        P.e.f.tok_mask = S.tok_mask ? S.tok_mask + (long)t * S.N : nullptr;
        P.e.f.c_prev = t ? c + (t - 1) * NH : nullptr;
        P.e.f.gates = gates + (long)t * 4 * NH; P.e.f.c_out = c + t * NH; P.e.f.h_out = h + t * NH; P.e.f.H = H;
        P.e.s = EpiStore<4>{nullptr, 0, nullptr, 0, 0};
      }
      const int t = tau - 1;  // layer-2 input projection of step t
      if (t >= 0 && t < S.T && (S.nact ? S.nact[t] : S.N) > 0) {
        TickFwdProb& P = g.p[g.nprob++];
        P.M = S.nact ? S.nact[t] : S.N; P.N = 4 * H; P.K = H;
        P.a = SrcRow{S.h1 + t * NH, H};
        P.b = SrcKSel{S.Wx2, 4L * H, H, 0};
        P.e.kind = 1;
        P.e.s = EpiStore<4>{S.gates2 + (long)t * 4 * NH, 4L * H, S.b2, VD_ACT_NONE, 0};
        P.e.f = EpiLstmFwdTick{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, H};
      }
    }
    if (g.nprob == 0) continue;
    static const int scfg = env_int("VD_LSTM_FWD_SMALL", 2);
    if (int rc = scfg == 2 ? launch_grouped<CfgFwdSmallC>(g, (hipStream_t)stream)
                 : scfg == 3 ? launch_grouped<CfgFwdSmallD>(g, (hipStream_t)stream)
                 : scfg == 4 ? launch_grouped<CfgFwdSmallE>(g, (hipStream_t)stream)
                 : scfg == 5 ? launch_grouped<CfgFwdSmallF>(g, (hipStream_t)stream)
                 : scfg == 6 ? launch_grouped<CfgFwdSmallG>(g, (hipStream_t)stream)
                             : launch_grouped<CfgFwdSmallA>(g, (hipStream_t)stream))
      return rc;
  }
  return VD_OK;
}

int vd_lstm2_backward(const vd_lstm2_bwd_t* st, int nstacks, int H, void* stream) {
  VD_CHECK_ARG(st && nstacks >= 1 && nstacks <= VD_MAX_STACKS && H > 0 && H % 32 == 0,
               "vd_lstm2_backward: bad args (nstacks=%d, max %d)", nstacks, VD_MAX_STACKS);
  int Tmax = 0;
  for (int s = 0; s < nstacks; ++s) {
    VD_CHECK_ARG(st[s].T >= 1 && st[s].N >= 1 && st[s].Wh1 && st[s].Wx2 && st[s].Wh2 && st[s].gates1 && st[s].c1 &&
                     st[s].gates2 && st[s].c2 && st[s].dh_last2 && st[s].dh1_seq && st[s].dc1 && st[s].dc2,
                 "vd_lstm2_backward: stack %d has null/empty fields", s);
    Tmax = st[s].T > Tmax ? st[s].T : Tmax;
  }
  if (vd_tune_get("VD_LSTM2_PERSIST", 0)) {
    const int rc = launch_lstm2_seq<CfgBwdSmallC, false, L2SeqBwdArgs, TickBwdProb>(st, nstacks, H, Tmax, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  static const int scfg = env_int("VD_LSTM_BWD_SMALL", 3);   // 3 = two register stages: -0.2 ms per headline step vs 2
  switch (scfg) {
    case 2: return lstm2_backward_ticks<CfgBwdSmallC, 1>(st, nstacks, H, Tmax, (hipStream_t)stream);
    case 3: return lstm2_backward_ticks<CfgBwdSmallD, 1>(st, nstacks, H, Tmax, (hipStream_t)stream);
    case 4: return lstm2_backward_ticks<CfgBwdSmallE, 2>(st, nstacks, H, Tmax, (hipStream_t)stream);   // 32 x 64 tiles
    case 5: return lstm2_backward_ticks<CfgBwdSmallF, 2>(st, nstacks, H, Tmax, (hipStream_t)stream);   // 64 x 64 tiles
    case 6: return lstm2_backward_ticks<CfgBwdSmallG, 2>(st, nstacks, H, Tmax, (hipStream_t)stream);   // 32 x 64, two register stages
    default: return lstm2_backward_ticks<CfgBwdSmallA, 1>(st, nstacks, H, Tmax, (hipStream_t)stream);
  }
}

}  // extern "C"

#ifdef VD_TIMING
// diagnostic build only: copy the per-workgroup phase stamps of the last launches to the host
extern "C" int vd_debug_timing(unsigned long long* out, int n) {
  VD_HIP(hipDeviceSynchronize());
  VD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(vd_tbuf), (size_t)n * sizeof(unsigned long long)));
  return VD_OK;
}
#endif
