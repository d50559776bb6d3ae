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
#include "split_core.h"

// defined in gemm_ops.hip (declared in include/visdial_hip.h)
extern "C" int vd_gemm_tn_acc(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                              int N, int K, int flags, void* stream);

// ---------------------------------------------------------------------------
// forward epilogue: acc[g] = (h_prev*Wh)[row, g*H + j]
// ---------------------------------------------------------------------------
// C16 (bf16 pass of the model-level runtime, configs[4]): COMPACT state -- `xproj` is a bf16 projection table (xld in bf16
// elements), `gates` a bf16 buffer [N x 4H], `h_out` may be null (only the last step's fp32 h has a reader; the recurrence and the
// weight gradient read the bf16 copy h16): 291 instead of 496 MB of epilogue traffic per launch at the headline shape
template <int SEQ, bool C16 = false>
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
  // SEQ = 1: compiler-scheduled epilogue (kept for the tick kernels' build switch VD_TICK_EPI_SEQ); 2: diagnostic, K loop only
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
  // Split-K latency shapes (one 32-row tile, four waves each holding a K slice of it): the cell update DISTRIBUTED over the four waves.
  // Wave g takes row group g (8 rows x 32 hidden units, one row x 4 units per lane): it requests its operands (mask id, the four projection
  // pieces, c_{t-1}) right after the K loop -- the round trip hides under the cross-wave reduction through LDS (gemm_block) -- and does
  // a quarter of the math and of the stores.  With wave 0 alone serving the four row groups one after the other the epilogue of a tick
  // workgroup was 5.8 of its 22 us (four dependent round trips; scripts/mb_ticks.py MB_PHASES=1).
  static constexpr bool HAS_DIST = !C16;
  struct DOps {
    float4 x[4], cp;
    int keep;
  };
  __device__ __forceinline__ bool dist_ok() const { return true; }
  __device__ __forceinline__ void dist_load(DOps& q, int row0, int vcol0, int lane, int grp, int M) const {
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;
    const int row = row0 + grp * 8 + (lane >> 3);
    const int rc = row < M ? row : M - 1;
    q.keep = tok_mask ? (tok_mask[rc] != 0) : 1;
    const float* xr = xproj + (tok_gather ? (long)tok_gather[rc] : (long)rc) * xld + j;
    q.x[0] = *reinterpret_cast<const float4*>(xr);
    q.x[1] = *reinterpret_cast<const float4*>(xr + H);
    q.x[2] = *reinterpret_cast<const float4*>(xr + 2 * H);
    q.x[3] = *reinterpret_cast<const float4*>(xr + 3 * H);
    q.cp = c_prev ? *reinterpret_cast<const float4*>(c_prev + (long)rc * H + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // a[g] = the reduced pre-activation sums of gate g for this lane's (row, 4 hidden units)
  __device__ __forceinline__ void dist_store(const float4 (&a)[4], const DOps& q, int row0, int vcol0, int lane, int grp, int M) const {
    const int j = (vcol0 >> 7) * 32 + (lane & 7) * 4;
    const int row = row0 + grp * 8 + (lane >> 3);
    if (row >= M || j >= H) return;
    const float km = q.keep ? 1.f : 0.f;  // maskZero(): h = c = gates = 0 for pad rows
    float4 gi, gf, go, gg, c, h;
#define VD_CELLD(E)                                   \
    gi.E = km * vd_sigmoid(a[0].E + q.x[0].E);        \
    gf.E = km * vd_sigmoid(a[1].E + q.x[1].E);        \
    go.E = km * vd_sigmoid(a[2].E + q.x[2].E);        \
    gg.E = km * vd_tanh(a[3].E + q.x[3].E);           \
    c.E = gf.E * q.cp.E + gi.E * gg.E;                \
    h.E = go.E * vd_tanh(c.E);
    VD_CELLD(x) VD_CELLD(y) VD_CELLD(z) VD_CELLD(w)
#undef VD_CELLD
    float* gr = gates + (long)row * 4 * H + j;
    *reinterpret_cast<float4*>(gr) = gi;
    *reinterpret_cast<float4*>(gr + H) = gf;
    *reinterpret_cast<float4*>(gr + 2 * H) = go;
    *reinterpret_cast<float4*>(gr + 3 * H) = gg;
    *reinterpret_cast<float4*>(c_out + (long)row * H + j) = c;
    *reinterpret_cast<float4*>(h_out + (long)row * H + j) = h;
    if (h16) vd_st4_bf16(h16 + (long)row * H + j, h);
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
      // DIAGNOSTIC build (SEQ = 2): no epilogue loads, no gate math, one
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
    float4 x[4], cp;
    // descriptors (SGPRs) + 32-bit byte offsets: per row group one multiply-add for the gathered projection row and one for the
    // clamped state row; the gate stride and the row-group stride are SGPR offsets
    const __amdgpu_buffer_rsrc_t rx = vd_rsrc(xproj), rc = vd_rsrc(c_prev), rg = vd_rsrc(gates), rco = vd_rsrc(c_out),
                                 rh = vd_rsrc(h_out), rh16 = vd_rsrc(h16);
    const unsigned uH4 = (unsigned)H * 4u, uj4 = (unsigned)j * 4u, uxld4 = (unsigned)xld * 4u;
    const unsigned vrow = ((unsigned)(row0 + rl)) * uH4 + uj4;          // byte offset of (row0 + rl, j) in an [M x H] tensor
    // C16: the projection rows are bf16 -- 12 VGPRs per row group while in flight (4 x 8 bytes packed + c) -- so TWO row groups are
    // kept in flight: groups 0 and 1 are requested before the LDS transposes, group p + 2 when group p has been folded
    vd_u32x2 xs[2][4];
    float4 cs[2];
    auto issue16 = [&](int p, int st) {
      const unsigned xo = ((unsigned)tk[p] * uxld4 + uj4) >> 1;
      xs[st][0] = __builtin_amdgcn_raw_buffer_load_b64(rx, xo, 0, 0);
      xs[st][1] = __builtin_amdgcn_raw_buffer_load_b64(rx, xo, uH4 >> 1, 0);
      xs[st][2] = __builtin_amdgcn_raw_buffer_load_b64(rx, xo, uH4, 0);
      xs[st][3] = __builtin_amdgcn_raw_buffer_load_b64(rx, xo, 3 * (uH4 >> 1), 0);
      cs[st] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c_prev) cs[st] = vd_buf_ld4(rc, (unsigned)rowc[p] * uH4 + uj4, 0);
    };
    if constexpr (C16) {
      issue16(0, 0);
      issue16(1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    tile_to_rows(acc[0], scr, lane, a[0]);
    tile_to_rows(acc[1], scr, lane, a[1]);
    tile_to_rows(acc[2], scr, lane, a[2]);
    tile_to_rows(acc[3], scr, lane, a[3]);
    VD_T(3);
    auto issue = [&](int p) {
      if constexpr (C16) {
        issue16(p, p & 1);
      } else {
        const unsigned xo = (unsigned)tk[p] * uxld4 + uj4;
        x[0] = vd_buf_ld4(rx, xo, 0);
        x[1] = vd_buf_ld4(rx, xo, uH4);
        x[2] = vd_buf_ld4(rx, xo, 2 * uH4);
        x[3] = vd_buf_ld4(rx, xo, 3 * uH4);
        cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_prev) cp = vd_buf_ld4(rc, (unsigned)rowc[p] * uH4 + uj4, 0);
      }
    };
    if constexpr (!C16) issue(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // fold the loaded values into the pre-activations; x / cp are dead afterwards
      float4 pi = a[0][p], pf = a[1][p], po = a[2][p], pg = a[3][p];
      if constexpr (C16) {
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) x[gk] = vd_bf16x4_unpack(xs[p & 1][gk]);
        cp = cs[p & 1];
      }
      pi.x += x[0].x; pi.y += x[0].y; pi.z += x[0].z; pi.w += x[0].w;
      pf.x += x[1].x; pf.y += x[1].y; pf.z += x[1].z; pf.w += x[1].w;
      po.x += x[2].x; po.y += x[2].y; po.z += x[2].z; po.w += x[2].w;
      pg.x += x[3].x; pg.y += x[3].y; pg.z += x[3].z; pg.w += x[3].w;
      const float4 cq = cp;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (C16) {
        if (p + 2 < 4) issue(p + 2);  // two groups ahead: in flight under the math and stores of p AND p + 1
      } else {
        if (p + 1 < 4) issue(p + 1);  // in flight under the math and stores below
      }
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
        const unsigned sp = (unsigned)p * 8u * uH4;                  // row-group stride (uniform)
        const unsigned vg = 4u * vrow - 3u * uj4;                    // (row0 + rl, j) in the [M x 4H] gates tensor
        if constexpr (C16) {                                           // compact state: bf16 gates, fp32 c, bf16 h (+ fp32 h at the last step)
          // a sigmoid gate s > 1/2 is stored as s - 1 = -(1 - s): the backward pass needs s AND 1 - s, and near saturation bf16
          // keeps only one of them (spacing 2^-8 below 1); the sign says which one was stored, both come back to 2^-9 relative
          // (a gate that saturates to exactly 1.0f -- pre-activation above ~16.6 -- must not encode as +0 = "s = 0": the stored value of the
          //  s - 1 case is kept strictly negative; -1e-30 is a bf16 normal and decodes to s = 1, s (1 - s) = 1e-30)
#define VD_ENC1(x) x = x > 0.5f ? fminf(x - 1.f, -1e-30f) : x;
#define VD_ENC(v) VD_ENC1(v.x) VD_ENC1(v.y) VD_ENC1(v.z) VD_ENC1(v.w)
          VD_ENC(gi) VD_ENC(gf) VD_ENC(go)
#undef VD_ENC
#undef VD_ENC1
          vd_buf_st4_bf16(rg, vg >> 1, 2u * sp, gi);
          vd_buf_st4_bf16(rg, vg >> 1, 2u * sp + (uH4 >> 1), gf);
          vd_buf_st4_bf16(rg, vg >> 1, 2u * sp + uH4, go);
          vd_buf_st4_bf16(rg, vg >> 1, 2u * sp + 3 * (uH4 >> 1), gg);
          vd_buf_st4(rco, vrow, sp, c);
          if (h_out) vd_buf_st4(rh, vrow, sp, h);
          vd_buf_st4_bf16(rh16, vrow >> 1, sp >> 1, h);
        } else {
        vd_buf_st4(rg, vg, 4u * sp, gi);                              // saved for the backward pass
        vd_buf_st4(rg, vg, 4u * sp + uH4, gf);
        vd_buf_st4(rg, vg, 4u * sp + 2 * uH4, go);
        vd_buf_st4(rg, vg, 4u * sp + 3 * uH4, gg);
        vd_buf_st4(rco, vrow, sp, c);
        vd_buf_st4(rh, vrow, sp, h);
        if (h16) vd_buf_st4_bf16(rh16, vrow >> 1, sp >> 1, h);
        }
      }
    }
  }
};

using EpiLstmFwd = EpiLstmFwdT<0>;

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
// C16: compact bf16 state (see EpiLstmFwdT): `gates` is a bf16 buffer [N x 4H]; da_t overwrites it in place AS bf16 -- the only copy
// (operand of the next step's product, of the weight gradient and of the table gradient)
template <int NT, int BATCH = (NT == 1 ? 2 : 1), bool TWO = true, bool C16 = false>
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
  // split-K latency shape (NT = 1: a 32 x 32 tile, four K-slice waves): the gate gradients DISTRIBUTED over the four waves like the forward
  // cell update (EpiLstmFwdT::dist_load): wave g finishes rows g*8 .. g*8+7, one (row, 4 hidden units) slot per lane, ONE round trip
  static constexpr bool HAS_DIST = NT == 1 && !C16;
  struct DOps {
    float4 g[4], ct, cp, dcv, dha, dhb;
  };
  __device__ __forceinline__ bool dist_ok() const { return true; }
  __device__ __forceinline__ void dist_load(DOps& q, int row0, int col0, int lane, int grp, int M) const {
    const int j = col0 + (lane & 7) * 4;
    const int row = row0 + grp * 8 + (lane >> 3);
    const int rc = row < M ? row : M - 1;
    const long o = (long)rc * H + j;
    const float* gr = gates + (long)rc * 4 * H + j;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    q.g[0] = *reinterpret_cast<const float4*>(gr);
    q.g[1] = *reinterpret_cast<const float4*>(gr + H);
    q.g[2] = *reinterpret_cast<const float4*>(gr + 2 * H);
    q.g[3] = *reinterpret_cast<const float4*>(gr + 3 * H);
    q.ct = *reinterpret_cast<const float4*>(c_t + o);
    q.cp = c_prev ? *reinterpret_cast<const float4*>(c_prev + o) : z;
    q.dcv = dc_first ? z : *reinterpret_cast<const float4*>(dc + o);
    q.dha = dh_a ? *reinterpret_cast<const float4*>(dh_a + o) : z;
    q.dhb = dh_b ? *reinterpret_cast<const float4*>(dh_b + o) : z;
  }
  __device__ __forceinline__ void dist_store(const float4 (&a)[1], const DOps& q, int row0, int col0, int lane, int grp, int M) const {
    const int j = col0 + (lane & 7) * 4;
    const int row = row0 + grp * 8 + (lane >> 3);
    if (row >= M || j >= H) return;
    float4 dh = a[0], ai, af, ao, ag, dn;
    dh.x += q.dha.x + q.dhb.x; dh.y += q.dha.y + q.dhb.y; dh.z += q.dha.z + q.dhb.z; dh.w += q.dha.w + q.dhb.w;
#define VD_CELLBD(E)                                                      \
    {                                                                     \
      const float gi_ = q.g[0].E, gf_ = q.g[1].E, go_ = q.g[2].E;         \
      const float tc = vd_tanh(q.ct.E);                                   \
      const float d = q.dcv.E + dh.E * go_ * (1.f - tc * tc);             \
      ai.E = d * q.g[3].E * gi_ * (1.f - gi_);                            \
      af.E = d * q.cp.E * gf_ * (1.f - gf_);                              \
      ao.E = dh.E * tc * go_ * (1.f - go_);                               \
      ag.E = d * gi_ * (1.f - q.g[3].E * q.g[3].E);                       \
      dn.E = d * gf_;                                                     \
    }
    VD_CELLBD(x) VD_CELLBD(y) VD_CELLBD(z) VD_CELLBD(w)
#undef VD_CELLBD
    float* gr = gates + (long)row * 4 * H + j;
    *reinterpret_cast<float4*>(gr) = ai;
    *reinterpret_cast<float4*>(gr + H) = af;
    *reinterpret_cast<float4*>(gr + 2 * H) = ao;
    *reinterpret_cast<float4*>(gr + 3 * H) = ag;
    *reinterpret_cast<float4*>(dc + (long)row * H + j) = dn;
    if (da16) {
      vd_bf16_bits* g16 = da16 + (long)row * 4 * H + j;
      vd_st4_bf16(g16, ai);
      vd_st4_bf16(g16 + H, af);
      vd_st4_bf16(g16 + 2 * H, ao);
      vd_st4_bf16(g16 + 3 * H, ag);
    }
  }
  struct Slot {
    float4 g[4], ct, cp, dcv, dhx;
    vd_u32x2 g16[4];   // C16: the saved gates as loaded (packed bf16), unpacked when the slot is consumed -- NOT in the load phase, where the
                       // unpacking would sit in front of the scheduling barrier and wait for every load of the slot
  };
  // buffer addressing (common.h): `o4` = byte offset of (row, j) in an [M x H] tensor; the gates tensor's is 4 * o4 - 12 * j
  struct Rsrc {
    __amdgpu_buffer_rsrc_t g, ct, cp, dc, dh1, dh2, g16;
  };
  __device__ __forceinline__ void load_slot_buf(Slot& L, const Rsrc& R, unsigned o4, unsigned og4, unsigned uH4) const {
    if constexpr (C16) {
      L.g16[0] = __builtin_amdgcn_raw_buffer_load_b64(R.g, og4 >> 1, 0, 0);
      L.g16[1] = __builtin_amdgcn_raw_buffer_load_b64(R.g, og4 >> 1, uH4 >> 1, 0);
      L.g16[2] = __builtin_amdgcn_raw_buffer_load_b64(R.g, og4 >> 1, uH4, 0);
      L.g16[3] = __builtin_amdgcn_raw_buffer_load_b64(R.g, og4 >> 1, 3 * (uH4 >> 1), 0);
    } else {
    L.g[0] = vd_buf_ld4(R.g, og4, 0);          // saved gates: read exactly once
    L.g[1] = vd_buf_ld4(R.g, og4, uH4);
    L.g[2] = vd_buf_ld4(R.g, og4, 2 * uH4);
    L.g[3] = vd_buf_ld4(R.g, og4, 3 * uH4);
    }
    L.ct = vd_buf_ld4(R.ct, o4, 0);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    L.cp = c_prev ? vd_buf_ld4(R.cp, o4, 0) : z;
    L.dcv = dc_first ? z : vd_buf_ld4(R.dc, o4, 0);
    L.dhx = (dh_a || dh_b) ? vd_buf_ld4(R.dh1, o4, 0) : z;   // the common case has at most one incoming-gradient operand
  }
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[NT], int row0, int col0, int lane, int M,
                                             int N, float* scr) const {
    const int rl = lane >> 3, cl = (lane & 7) * 4;
    const bool two_dh = TWO && dh_a && dh_b;
    const Rsrc R{vd_rsrc(gates), vd_rsrc(c_t), vd_rsrc(c_prev), vd_rsrc(dc), vd_rsrc(dh_a ? dh_a : dh_b), vd_rsrc(dh_b), vd_rsrc(da16)};
    const unsigned uH4 = (unsigned)H * 4u;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      float4 d4[4];
      tile_to_rows(acc[jt], scr, lane, d4);  // row-vectorised: 4 consecutive hidden units per lane
      if (jt == 0) VD_T(3);
      const int j = col0 + jt * 32 + cl;
      const int jc = j < N ? j : N - 4;      // N = H here; columns past the end load a valid address, never stored
#pragma unroll
      for (int pp = 0; pp < 4; pp += BATCH) {
        Slot L[BATCH];
        unsigned o4[BATCH], og4[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
          const int row = row0 + (pp + q) * 8 + rl;
          o4[q] = (unsigned)(row < M ? row : M - 1) * uH4 + (unsigned)jc * 4u;
          og4[q] = 4u * o4[q] - 12u * (unsigned)jc;
          load_slot_buf(L[q], R, o4[q], og4[q], uH4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
          Slot& C = L[q];
          if constexpr (C16) {
#pragma unroll
            for (int gk = 0; gk < 4; ++gk) C.g[gk] = vd_bf16x4_unpack(C.g16[gk]);
          }
          float4 dh = d4[pp + q];
          dh.x += C.dhx.x; dh.y += C.dhx.y; dh.z += C.dhx.z; dh.w += C.dhx.w;
          const int row = row0 + (pp + q) * 8 + rl;
          if constexpr (TWO) {
            if (two_dh) {
              const float4 t = vd_buf_ld4(R.dh2, o4[q], 0);
              dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
            }
          }
          float4 ai, af, ao, ag, dn;
#define VD_CELLB(E)                                                       \
          {                                                               \
            float gi_ = C.g[0].E, gf_ = C.g[1].E, go_ = C.g[2].E;         \
            float pi_ = gi_ * (1.f - gi_), pf_ = gf_ * (1.f - gf_), po_ = go_ * (1.f - go_);   \
            if constexpr (C16) {   /* stored s <= 1/2 as s, s > 1/2 as s - 1 (see the forward epilogue) */   \
              pi_ = gi_ < 0.f ? (1.f + gi_) * -gi_ : pi_;  gi_ = gi_ < 0.f ? 1.f + gi_ : gi_;   \
              pf_ = gf_ < 0.f ? (1.f + gf_) * -gf_ : pf_;  gf_ = gf_ < 0.f ? 1.f + gf_ : gf_;   \
              po_ = go_ < 0.f ? (1.f + go_) * -go_ : po_;  go_ = go_ < 0.f ? 1.f + go_ : go_;   \
            }                                                             \
            const float tc = vd_tanh(C.ct.E);                             \
            const float d = C.dcv.E + dh.E * go_ * (1.f - tc * tc);       \
            ai.E = d * C.g[3].E * pi_;                                    \
            af.E = d * C.cp.E * pf_;                                      \
            ao.E = dh.E * tc * po_;                                       \
            ag.E = d * gi_ * (1.f - C.g[3].E * C.g[3].E);                 \
            dn.E = d * gf_;                                               \
          }
          VD_CELLB(x) VD_CELLB(y) VD_CELLB(z) VD_CELLB(w)
#undef VD_CELLB
          if (row < M && j < N) {
            // (row < M and j < N here: the clamped offsets of the loads are the true ones)
            if constexpr (C16) {
              vd_buf_st4_bf16(R.g, og4[q] >> 1, 0, ai);
              vd_buf_st4_bf16(R.g, og4[q] >> 1, uH4 >> 1, af);
              vd_buf_st4_bf16(R.g, og4[q] >> 1, uH4, ao);
              vd_buf_st4_bf16(R.g, og4[q] >> 1, 3 * (uH4 >> 1), ag);
            } else {
            vd_buf_st4(R.g, og4[q], 0, ai);
            vd_buf_st4(R.g, og4[q], uH4, af);
            vd_buf_st4(R.g, og4[q], 2 * uH4, ao);
            vd_buf_st4(R.g, og4[q], 3 * uH4, ag);
            }
            vd_buf_st4(R.dc, o4[q], 0, dn);
            if (!C16 && da16) {
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, 0, ai);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, uH4 >> 1, af);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, uH4, ao);
              vd_buf_st4_bf16(R.g16, og4[q] >> 1, 3 * (uH4 >> 1), ag);
            }
          }
        }
      }
    }
  }
};

// throughput shapes (option LSTM: N = 20 000 rows).  Defaults chosen by on-device sweeps
// (scripts/microbench.py, profiles/r01_config_sweep.txt): forward = 128x128 tile, BK = 16, single LDS
// buffer, 3 workgroups/CU (99 TF); backward = 128x64 tile (1 256 half-size workgroups balance 256 CUs),
// BK = 16, single buffer, 4 workgroups/CU (102 TF).
using CfgF9 = GemmCfg<4, 1, 4, 16, 0, 4, 41984>;  // <=128 VGPR, LDS padded to 41 KB: exactly 3 WG/CU + room for one latency WG
using CfgB1 = GemmCfg<4, 1, 4, 32, 0, 3>;
using CfgB11 = GemmCfg<4, 1, 2, 16, 0, 4, 41984>;
// latency shapes (N ~ 200 rows): 4-way intra-block split-K.  Single LDS buffer and <= 152 VGPRs so one
// of these workgroups fits into the footprint a retiring throughput-shape workgroup frees (they run
// concurrently on other streams).
using CfgFbf16 = GemmCfg<4, 1, 4, 32, 0, 3, 0, 1>;  // bf16 operands (opt-in), BK = 32 = two 32x32x16 MFMAs per tile
using CfgBbf16 = GemmCfg<4, 1, 2, 32, 0, 3, 0, 1>;
// bf16 pass with shadows: the LDS-DMA pipeline of the fp32 step kernels over bf16 rows (h / da shadows, bf16 weight copies)
#ifndef VD_C16_LDS
#define VD_C16_LDS 41984     // (`make variant NAME=c16lds DEFS=-DVD_C16_LDS=32768`: four workgroups per CU for the bf16 pass's step kernels)
#endif
using CfgF9bf16 = GemmCfg<4, 1, 4, 16, 0, 4, VD_C16_LDS, 1>;
using CfgB11bf16 = GemmCfg<4, 1, 2, 16, 0, 4, VD_C16_LDS, 1>;
using CfgFwdSmallA = GemmCfg<1, 4, 4, 8, 0, 3>;   // 32 x (32 j x 4 gates), BK = 32, 23 KB LDS
using CfgFwdSmallC = GemmCfg<1, 4, 4, 8, 0, 4>;   // as A, <=128 VGPR: fits beside 3 padded throughput workgroups
using CfgBwdSmallD = GemmCfg<1, 4, 1, 32, 2, 4>;
// bf16 pass of the model-level runtime (configs[4]: "bf16 LSTM step"): the encoder's ticks round both operands to bf16 while staging them
// into LDS and multiply on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 state): an eighth of the matrix time and half of the LDS
// traffic of a workgroup whose K loop is 45 % MFMA time.  BK = 64 (16 k per wave and MFMA): 156 VGPRs forward, hence three waves per SIMD.
using CfgTickFwdBf16 = GemmCfg<1, 4, 4, 16, 0, 3, 0, 1>;
using CfgTickBwdBf16 = GemmCfg<1, 4, 1, 32, 0, 4, 0, 1>;

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
  static const int cfg = (20);
  return cfg == 20 && N >= 2048 && H % 32 == 0 && (long)N * H * 4 < (1L << 32);
}
static bool use_glds_bwd(int N, int H) {
  static const int cfg = (20);
  return cfg == 20 && N >= 2048 && H % 32 == 0 && (long)N * 4 * H * 4 < (1L << 32);
}

static int lstm_step_fwd(const float* h_prev, const float* Wh, int N, int H, int K, const EpiLstmFwd& epi,
                         hipStream_t s) {
  SrcRow a{h_prev, H};
  SrcKGate4 b{Wh, 4L * H, H};
  if (N >= 2048) return launch_gemm<CfgF9>(N, 4 * H, K, 1, a, b, epi, s);   // (throughput shapes normally take the LDS-DMA drivers)
  return launch_gemm<CfgFwdSmallA>(N, 4 * H, K, 1, a, b, epi, s);
}

static int lstm_step_bwd(const float* da_next, const float* Wh, int N, int H, int K, const float* dh_a,
                         const float* dh_b, float* gates, const float* c_t, const float* c_prev, float* dc,
                         int dc_first, hipStream_t s, int flags = 0, vd_bf16_bits* da16 = nullptr,
                         const vd_bf16_bits* da16_next = nullptr, const vd_bf16_bits* Wh16 = nullptr,
                         const vd_bf16_bits* W3 = nullptr) {
  SrcRow a{da_next, 4L * H};
  SrcRow b{Wh, 4L * H};  // B[k][n] = Wh[n][k]
  if (W3 && K > 0 && !(dh_a && dh_b)) {   // exact-operand split: da_{t+1} stays fp32 in memory, Wh as three bf16 planes
    EpiLstmBwd<4, 2, false> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};   // 128 x 128 tiles: half the A re-reads of the fp32 kernel's 128 x 64
    if (flags & VD_FLAG_SPLIT9) return launch_gemm_split<9>(N, H, K, da_next, 4L * H, W3, 4L * H, 4L * H * H, e, s);
    if (flags & VD_FLAG_SPLIT6) return launch_gemm_split<6>(N, H, K, da_next, 4L * H, W3, 4L * H, 4L * H * H, e, s);
    return launch_gemm_split<3>(N, H, K, da_next, 4L * H, W3, 4L * H, 4L * H * H, e, s);
  }
  if ((flags & VD_FLAG_BF16) && N >= 2048 && K > 0) {
    EpiLstmBwd<2> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H, da16};
    // shadows on: da_{t+1} and Wh are read as the bf16 rows their producers wrote (half the operand bytes, no conversion
    // while staging, LDS-DMA pipeline); K counts bf16 pairs
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
    if (use_glds_bwd(N, H) && K > 0) {
      // LDS-DMA pipeline: A = da_{t+1} rows, Bt = Wh rows (both contiguous in k = the 4H gate columns).  The two-slot epilogue
      // (loads of two slots in flight before either is consumed: half the serialised round trips) fits the 128-VGPR build with
      // buffer addressing; a step with a recurrent product has at most one incoming-gradient operand
      if (!(dh_a && dh_b)) {
        EpiLstmBwd<2, 2, false> e2c{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
        return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e2c, s);
      }
      EpiLstmBwd<2> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
      return launch_gemm_glds<CfgB11, false>(N, H, K, 1, da_next, 4L * H, Wh, 4L * H, e, s);
    }
    EpiLstmBwd<2> e2{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
    return launch_gemm<CfgB11>(N, H, K, 1, a, b, e2, s);
  }
  EpiLstmBwd<1> e{dh_a, dh_b, gates, c_t, c_prev, dc, dc_first, H};
  return launch_gemm<CfgBwdSmallD>(N, H, K, 1, a, b, e, s);   // two register stages, like the backward ticks (27 vs 31 us per launch alone)
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
  static constexpr bool HAS_DIST = true;      // both kinds finish distributed over the four K-slice waves (gemm_block): no one-wave path is compiled
  using DOps = EpiLstmFwdTick::DOps;
  __device__ __forceinline__ bool dist_ok() const { return true; }
  __device__ __forceinline__ void dist_load(DOps& q, int row0, int vcol0, int lane, int grp, int M) const {
    if (kind == 0) f.dist_load(q, row0, vcol0, lane, grp, M);
  }
  __device__ __forceinline__ void dist_store(const float4 (&a)[4], const DOps& q, int row0, int vcol0, int lane, int grp, int M) const {
    if (kind == 0) {
      f.dist_store(a, q, row0, vcol0, lane, grp, M);
      return;
    }
    // plain store (+ bias) of the layer-2 input projection: plain column order, 4 consecutive columns per lane and column tile
    const int row = row0 + grp * 8 + (lane >> 3);
    if (row >= M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = vcol0 + j * 32 + (lane & 7) * 4;
      float4 v = a[j];
      if (s.bias) {
        const float4 b = *reinterpret_cast<const float4*>(s.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      *reinterpret_cast<float4*>(s.C + (long)row * s.ldc + col) = v;
    }
  }
  __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], int row0, int col0, int lane, int M,
                                             int N, float* scr) const {
    // never reached: the split-K forward tick shape always takes the distributed form above (gemm_block); compiling the one-wave
    // epilogues next to it cost > 100 spilled VGPRs at the 128-register cap
    __builtin_trap();
  }
};
struct TickFwdProb {
  int M, N, K, tiles_n;
  SrcRow a;
  SrcKSel b;
  EpiTickFwd e;
};
template <int NT>   // column tiles per wave: 1 = 32 x 32 tiles, 2 = 32 (or 64) x 64 tiles (less operand traffic per FLOP)
struct EpiTickBwdT {
  int kind;
  EpiLstmBwd<NT> f;      // (default BATCH: 2 slots in flight for NT = 1, 1 for NT >= 2)
  EpiStore<NT> s;
  static constexpr bool HAS_DIST = NT == 1;   // 32 x 32 split-K tiles: both kinds finish distributed over the four waves (gemm_block)
  using DOps = typename EpiLstmBwd<NT>::DOps;
  __device__ __forceinline__ bool dist_ok() const { return true; }
  __device__ __forceinline__ void dist_load(DOps& q, int row0, int col0, int lane, int grp, int M) const {
    if (kind == 0) f.dist_load(q, row0, col0, lane, grp, M);
  }
  __device__ __forceinline__ void dist_store(const float4 (&a)[1], const DOps& q, int row0, int col0, int lane, int grp, int M) const {
    if (kind == 0) {
      f.dist_store(a, q, row0, col0, lane, grp, M);
      return;
    }
    const int row = row0 + grp * 8 + (lane >> 3), col = col0 + (lane & 7) * 4;      // plain store: dh1[t] = da2[t] * Wx2^T
    if (row < M) *reinterpret_cast<float4*>(s.C + (long)row * s.ldc + col) = a[0];
  }
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
  float* WhT = nullptr;
  const bool bf16 = (flags & VD_FLAG_BF16) && N >= 2048 && H % 32 == 0;
  // exact-operand split (split_core.h): the throughput shapes of the fp32 LDS-DMA pipeline only
  const int split = (!bf16 && (flags & VD_FLAG_SPLIT) && use_glds_fwd(N, H) && T > 1)
                        ? ((flags & VD_FLAG_SPLIT9) ? 9 : (flags & VD_FLAG_SPLIT6) ? 6 : 3) : 0;
  const bool glds = (use_glds_fwd(N, H) || bf16) && T > 1;   // both paths multiply by the transposed copy
  VdStreamScratch scr;
  vd_bf16_bits* W3 = nullptr;
  if (glds) {
    if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * (split ? 10 : bf16 ? 6 : 4), 0, &scr)) return rc0;
    WhT = scr.wht;
    hipLaunchKernelGGL(wh_gate_transpose_kernel, dim3(4 * H / 32, H / 32), dim3(256), 0, s, Wh, WhT, H);
    VD_LAUNCH_CHECK();
    if (split) {   // the transposed weights as three bf16 planes (hi / mid / lo), once per pass
      W3 = reinterpret_cast<vd_bf16_bits*>(scr.wht + (size_t)4 * H * H);
      if (int rc0 = weights_to_bf16x3(WhT, W3, 4L * H * H, s)) return rc0;
    }
  }
  // bf16 pass: the step kernels also write a bf16 copy of h (the operand of the dWh contraction of the same pass) ...
  vd_bf16_bits* h16 = nullptr;
  if (bf16) {
    if (int rc0 = vd_bf16_shadow_get(0, h, (size_t)T * NH, &h16)) return rc0;
  } else {
    vd_bf16_shadow_invalidate(h, (size_t)T * NH);
  }
  vd_bf16_shadow_invalidate(gates, (size_t)T * 4 * NH);     // this pass overwrites `gates`: a da shadow registered over it (last backward) is stale
  // ... and READ the shadow of h_{t-1} and a bf16 copy of the transposed weights through the LDS-DMA pipeline
  vd_bf16_bits* WhT16 = nullptr;
  if (h16 && glds) {
    WhT16 = reinterpret_cast<vd_bf16_bits*>(scr.wht + (size_t)4 * H * H);
    if (int rc0 = weights_to_bf16(WhT, WhT16, 4L * H * H, s)) return rc0;
  }
  for (int t = 0; t < T; ++t) {
    const float* hp = t ? h + (t - 1) * NH : h0;
    const float* cp = t ? c + (t - 1) * NH : c0;
    EpiLstmFwd e;
    e.xproj = xproj + (long)t * x_tstride;
    e.xld = x_ld;
    e.tok_gather = tok_gather ? tok_gather + (long)t * N : nullptr;
    e.tok_mask = tok_mask ? tok_mask + (long)t * N : nullptr;
    e.c_prev = cp;
    e.gates = gates + (long)t * 4 * NH;
    e.c_out = c + t * NH;
    e.h_out = h + t * NH;
    e.H = H;
    e.h16 = h16 ? h16 + t * NH : nullptr;
    int rc;
    if (bf16 && hp && WhT16 && t > 0)
      rc = launch_gemm_glds<CfgF9bf16, false>(N, 4 * H, H / 2, 1, reinterpret_cast<const float*>(h16 + (t - 1) * NH), (long)H / 2,
                                              reinterpret_cast<const float*>(WhT16), (long)H / 2, e, s);
    else if (bf16 && hp)
      rc = launch_gemm<CfgFbf16>(N, 4 * H, H, 1, SrcRow{hp, H}, SrcRow{WhT, H}, e, s);
    else if (split == 9 && hp)
      rc = launch_gemm_split<9>(N, 4 * H, H, hp, (long)H, W3, (long)H, 4L * H * H, e, s);
    else if (split == 6 && hp)
      rc = launch_gemm_split<6>(N, 4 * H, H, hp, (long)H, W3, (long)H, 4L * H * H, e, s);
    else if (split == 3 && hp)
      rc = launch_gemm_split<3>(N, 4 * H, H, hp, (long)H, W3, (long)H, 4L * H * H, e, s);
    else if (glds && hp)
      rc = launch_gemm_glds<CfgF9, false>(N, 4 * H, H, 1, hp, (long)H, WhT, (long)H, e, s);
    else
      rc = lstm_step_fwd(hp, Wh, N, H, hp ? H : 0, e, s);
    if (rc) return rc;
  }
  return VD_OK;
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
  if ((flags & VD_FLAG_BF16) && N >= 2048 && H % 32 == 0) {
    if (int rc0 = vd_bf16_shadow_get(1, gates, (size_t)T * 4 * NH, &da16)) return rc0;
    if (T > 1) {
      VdStreamScratch wscr;
      if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * 6, 0, &wscr)) return rc0;
      Wh16 = reinterpret_cast<vd_bf16_bits*>(wscr.wht + (size_t)4 * H * H);
      if (int rc0 = weights_to_bf16(Wh, Wh16, 4L * H * H, s)) return rc0;
    }
  } else {
    vd_bf16_shadow_invalidate(gates, (size_t)T * 4 * NH);
  }
  const vd_bf16_bits* W3 = nullptr;
  if (!(flags & VD_FLAG_BF16) && (flags & VD_FLAG_SPLIT) && use_glds_bwd(N, H) && T > 1 && (4 * H) % 32 == 0) {
    VdStreamScratch wscr;
    if (int rc0 = vd_stream_scratch(s, (size_t)4 * H * H * 10, 0, &wscr)) return rc0;
    vd_bf16_bits* w3 = reinterpret_cast<vd_bf16_bits*>(wscr.wht + (size_t)4 * H * H);    // (behind the forward pass's transposed copy)
    if (int rc0 = weights_to_bf16x3(Wh, w3, 4L * H * H, s)) return rc0;
    W3 = w3;
  }
  for (int t = T - 1; t >= 0; --t) {
    const bool last = (t == T - 1);
    const float* da_next = last ? nullptr : gates + (long)(t + 1) * 4 * NH;
    const int rc = lstm_step_bwd(da_next, Wh, N, H, last ? 0 : 4 * H, dh_seq ? dh_seq + t * NH : nullptr, (last && dh_last) ? dh_last : nullptr,
                                 gates + (long)t * 4 * NH, c + t * NH, t ? c + (t - 1) * NH : c0, dc_work, (last && !dc_last) ? 1 : 0, s, flags,
                                 da16 ? da16 + (long)t * 4 * NH : nullptr, (da16 && !last) ? da16 + (long)(t + 1) * 4 * NH : nullptr, Wh16, W3);
    if (rc) return rc;
  }
  int rc = VD_OK;
  if (dWh_acc && T > 1) {   // recurrent weight gradient dWh += sum_{t>=1} h_{t-1}^T da_t: one contraction over all (T-1)*N rows
    rc = vd_gemm_tn_acc(h_seq, H, gates + 4 * NH, 4L * H, dWh_acc, 4L * H, H, 4 * H, (T - 1) * N, flags & (VD_FLAG_BF16 | VD_FLAG_SPLIT9), s);
    if (rc) return rc;
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

}  // extern "C"

// the per-tick grouped launches of the forward direction, for one tile configuration
template <class Cfg>
static int lstm2_forward_ticks(const vd_lstm2_fwd_t* st, int nstacks, int H, int Tmax, hipStream_t stream) {
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
    static_assert(Cfg::WK == 4 && Cfg::WM == 1 && Cfg::NT == 4,
                  "EpiTickFwd finishes only through the distributed epilogue of the 32 x 128 four-wave split-K shape (gemm_block)");
    if (int rc = launch_grouped<Cfg>(g, stream)) return rc;
  }
  return VD_OK;
}

int vd_lstm2_forward_p(const vd_lstm2_fwd_t* st, int nstacks, int H, int flags, hipStream_t stream) {
  VD_CHECK_ARG(st && nstacks >= 1 && nstacks <= VD_MAX_STACKS && H > 0 && H % 32 == 0,
               "vd_lstm2_forward: bad args (nstacks=%d, max %d)", nstacks, VD_MAX_STACKS);
  int Tmax = 0;
  for (int s = 0; s < nstacks; ++s) {
    VD_CHECK_ARG(st[s].T >= 1 && st[s].N >= 1 && st[s].Wh1 && st[s].Wx2 && st[s].b2 && st[s].Wh2 && st[s].gates1 &&
                     st[s].h1 && st[s].c1 && st[s].gates2 && st[s].h2 && st[s].c2,
                 "vd_lstm2_forward: stack %d has null/empty fields", s);
    Tmax = st[s].T > Tmax ? st[s].T : Tmax;
  }
  if ((flags & VD_FLAG_BF16) && H % 64 == 0) return lstm2_forward_ticks<CfgTickFwdBf16>(st, nstacks, H, Tmax, stream);
  return lstm2_forward_ticks<CfgFwdSmallC>(st, nstacks, H, Tmax, stream);
}

int vd_lstm2_backward_p(const vd_lstm2_bwd_t* st, int nstacks, int H, int flags, hipStream_t stream) {
  VD_CHECK_ARG(st && nstacks >= 1 && nstacks <= VD_MAX_STACKS && H > 0 && H % 32 == 0,
               "vd_lstm2_backward: bad args (nstacks=%d, max %d)", nstacks, VD_MAX_STACKS);
  int Tmax = 0;
  for (int s = 0; s < nstacks; ++s) {
    VD_CHECK_ARG(st[s].T >= 1 && st[s].N >= 1 && st[s].Wh1 && st[s].Wx2 && st[s].Wh2 && st[s].gates1 && st[s].c1 &&
                     st[s].gates2 && st[s].c2 && st[s].dh_last2 && st[s].dh1_seq && st[s].dc1 && st[s].dc2,
                 "vd_lstm2_backward: stack %d has null/empty fields", s);
    Tmax = st[s].T > Tmax ? st[s].T : Tmax;
  }
  if (flags & VD_FLAG_BF16) return lstm2_backward_ticks<CfgTickBwdBf16, 1>(st, nstacks, H, Tmax, stream);   // (BK = 128 divides 4H)
  return lstm2_backward_ticks<CfgBwdSmallD, 1>(st, nstacks, H, Tmax, stream);   // two register stages
}

extern "C" {
int vd_lstm2_forward(const vd_lstm2_fwd_t* st, int nstacks, int H, void* stream) { return vd_lstm2_forward_p(st, nstacks, H, 0, (hipStream_t)stream); }
int vd_lstm2_backward(const vd_lstm2_bwd_t* st, int nstacks, int H, void* stream) { return vd_lstm2_backward_p(st, nstacks, H, 0, (hipStream_t)stream); }
int vd_lstm2_forward_flags(const vd_lstm2_fwd_t* st, int nstacks, int H, int flags, void* stream) { return vd_lstm2_forward_p(st, nstacks, H, flags, (hipStream_t)stream); }
int vd_lstm2_backward_flags(const vd_lstm2_bwd_t* st, int nstacks, int H, int flags, void* stream) { return vd_lstm2_backward_p(st, nstacks, H, flags, (hipStream_t)stream); }
}  // extern "C"

// ---- compact bf16 state (common.h): the option recurrence of a bf16 pass, model-level runtime --------------------------------------
int vd_f32_to_bf16(const float* src, vd_bf16_bits* dst, int64_t n, hipStream_t stream) { return weights_to_bf16(src, dst, n, stream); }

int vd_lstm_forward_c16(const vd_bf16_bits* table16, int64_t tab_ld, const int32_t* tok_gather, const float* Wh, vd_bf16_bits* gates16,
                        vd_bf16_bits* h16, float* h_last, float* c, int T, int N, int H, hipStream_t s) {
  VD_CHECK_ARG(table16 && tok_gather && Wh && gates16 && h16 && h_last && c && T >= 1 && N >= 2048 && H % 128 == 0 && tab_ld % 4 == 0,
               "vd_lstm_forward_c16: throughput shapes only (N >= 2048, H %% 128 == 0)");
  VD_CHECK_ARG((long)N * 4 * H * 4 < (1L << 32), "vd_lstm_forward_c16: N x 4H exceeds 4 GB per step");
  const long NH = (long)N * H;
  VdStreamScratch scr;
  if (int rc = vd_stream_scratch(s, (size_t)4 * H * H * 6, 0, &scr)) return rc;
  float* WhT = scr.wht;
  vd_bf16_bits* WhT16 = reinterpret_cast<vd_bf16_bits*>(scr.wht + (size_t)4 * H * H);
  hipLaunchKernelGGL(wh_gate_transpose_kernel, dim3(4 * H / 32, H / 32), dim3(256), 0, s, Wh, WhT, H);
  VD_LAUNCH_CHECK();
  if (int rc = weights_to_bf16(WhT, WhT16, 4L * H * H, s)) return rc;
  for (int t = 0; t < T; ++t) {
    EpiLstmFwdT<0, true> e;
    e.xproj = reinterpret_cast<const float*>(table16);
    e.xld = tab_ld;
    e.tok_gather = tok_gather + (long)t * N;
    e.tok_mask = nullptr;
    e.c_prev = t ? c + (t - 1) * NH : nullptr;
    e.gates = reinterpret_cast<float*>(gates16 + (long)t * 4 * NH);
    e.c_out = c + t * NH;
    e.h_out = t == T - 1 ? h_last : nullptr;
    e.H = H;
    e.h16 = h16 + t * NH;
    int rc;
    if (t == 0) rc = launch_gemm<CfgF9>(N, 4 * H, 0, 1, SrcRow{nullptr, H}, SrcKGate4{Wh, 4L * H, H}, e, s);   // no recurrent product yet
    else rc = launch_gemm_glds<CfgF9bf16, false>(N, 4 * H, H / 2, 1, reinterpret_cast<const float*>(h16 + (t - 1) * NH), (long)H / 2,
                                                 reinterpret_cast<const float*>(WhT16), (long)H / 2, e, s);
    if (rc) return rc;
  }
  return VD_OK;
}

int vd_lstm_backward_c16(const float* Wh, vd_bf16_bits* gates16, const float* c, const float* dh_last, float* dc_work, int T, int N, int H,
                         hipStream_t s) {
  VD_CHECK_ARG(Wh && gates16 && c && dh_last && dc_work && T >= 1 && N >= 2048 && H % 128 == 0, "vd_lstm_backward_c16: throughput shapes only");
  const long NH = (long)N * H;
  VdStreamScratch wscr;
  if (int rc = vd_stream_scratch(s, (size_t)4 * H * H * 6, 0, &wscr)) return rc;
  vd_bf16_bits* Wh16 = reinterpret_cast<vd_bf16_bits*>(wscr.wht + (size_t)4 * H * H);
  if (int rc = weights_to_bf16(Wh, Wh16, 4L * H * H, s)) return rc;
  for (int t = T - 1; t >= 0; --t) {
    const bool last = t == T - 1;
    EpiLstmBwd<2, 2, false, true> e{nullptr, last ? dh_last : nullptr, reinterpret_cast<float*>(gates16 + (long)t * 4 * NH), c + t * NH,
                                    t ? c + (t - 1) * NH : nullptr, dc_work, last ? 1 : 0, H};
    int rc;
    if (last) rc = launch_gemm<CfgB11>(N, H, 0, 1, SrcRow{nullptr, 4L * H}, SrcRow{Wh, 4L * H}, e, s);
    else rc = launch_gemm_glds<CfgB11bf16, false>(N, H, 4 * H / 2, 1, reinterpret_cast<const float*>(gates16 + (long)(t + 1) * 4 * NH), 2L * H,
                                                  reinterpret_cast<const float*>(Wh16), 2L * H, e, s);
    if (rc) return rc;
  }
  return VD_OK;
}


#ifdef VD_TIMING
// diagnostic build only: copy the per-workgroup phase stamps of the last launches to the host
extern "C" int vd_debug_timing(unsigned long long* out, int n) {
  VD_HIP(hipDeviceSynchronize());
  VD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(vd_tbuf), (size_t)n * sizeof(unsigned long long)));
  return VD_OK;
}
#endif