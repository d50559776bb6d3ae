// Shared helpers for the visdial HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define VD_OK 0
#define VD_ERR_ARG -1
#define VD_ERR_HIP -2
#define VD_ERR_STATE -3
#define VD_FLAG_BF16 1  // see include/visdial_hip.h
#define VD_FLAG_SPLIT9 2
#define VD_FLAG_SPLIT6 4
#define VD_FLAG_SPLIT3 8
#define VD_FLAG_SPLIT (VD_FLAG_SPLIT9 | VD_FLAG_SPLIT6 | VD_FLAG_SPLIT3)

// thread-local message returned by vd_last_error()
void vd_set_error(const char* fmt, ...);

#define VD_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      vd_set_error(__VA_ARGS__);                                  \
      return VD_ERR_ARG;                                          \
    }                                                             \
  } while (0)

#define VD_HIP(expr)                                                          \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                  \
      vd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                   hipGetErrorString(e__));                                   \
      return VD_ERR_HIP;                                                      \
    }                                                                         \
  } while (0)

#define VD_LAUNCH_CHECK() VD_HIP(hipGetLastError())

static inline int vd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ROCTx ranges around the phases of a step (SURVEY.md section 5 "tracing"): host-side ranges on the enqueuing thread, visible in
// `rocprofv3 --marker-trace --kernel-trace` next to the kernels dispatched inside them.  librocprofiler-sdk-roctx.so / libroctx64.so is
// dlopen'ed on first use and ONLY when the environment says VD_ROCTX=1 (scripts/measure_round.sh sets it for the trace pass); otherwise,
// or when neither library loads, the calls are no-ops.  (api.hip)
void vd_range_push(const char* name);
void vd_range_pop();
struct VdRange {
  explicit VdRange(const char* name) { vd_range_push(name); }
  ~VdRange() { vd_range_pop(); }
  VdRange(const VdRange&) = delete;
  VdRange& operator=(const VdRange&) = delete;
};

// Library-owned per-(device, stream) scratch (api.hip): work buffers whose lifetime is one stream-ordered call
// (the gate-interleaved Wh^T copy and the tile queues / arrival counters of the persistent recurrence kernels).
// Calls on different streams get different buffers, so they may overlap freely.
struct VdStreamScratch {
  float* wht = nullptr;      // [4H x H]
  size_t wht_bytes = 0;
  unsigned* sync = nullptr;  // persistent-kernel queue heads + arrival counters
  size_t sync_bytes = 0;
};
int vd_stream_scratch(hipStream_t stream, size_t wht_bytes, size_t sync_bytes, VdStreamScratch* out);
int vd_num_cus();  // compute units of the current device (cached)

// zero rows [nact[t], N) of every time slice of up to 6 dense [T x N x ncols[i]] buffers in one launch (elementwise.hip)
struct VdZeroSet {
  float* buf[6];
  int ncols[6];
  int n;
  long quads_per_row;   // filled in by the launcher
};
int vd_zero_inactive_multi(const VdZeroSet& z, const int32_t* nact_dev, int T, int N, hipStream_t stream);

typedef unsigned short vd_bf16_bits;
// ---- compact bf16 state of the option recurrence (model-level runtime, lstmBf16 = 1: BASELINE configs[4]) --------------------
// The saved gates / da live ONLY as bf16 [T x N x 4H], the projection table as bf16 rows, h as bf16 (+ the last step's fp32 h):
// 291 / 414 instead of 496 / 660 MB of epilogue traffic per forward / backward launch at the headline shape (lstm.hip).
int vd_lstm_forward_c16(const vd_bf16_bits* table16, int64_t tab_ld, const int32_t* tok_gather, const float* Wh, vd_bf16_bits* gates16,
                        vd_bf16_bits* h16, float* h_last, float* c, int T, int N, int H, hipStream_t stream);
int vd_lstm_backward_c16(const float* Wh, vd_bf16_bits* gates16, const float* c, const float* dh_last, float* dc_work, int T, int N, int H,
                         hipStream_t stream);
int vd_f32_to_bf16(const float* src, vd_bf16_bits* dst, int64_t n, hipStream_t stream);
int vd_gemm_tn_acc_bf16(const vd_bf16_bits* A16, const vd_bf16_bits* B16, float* C, int64_t ldc, int M, int N, int K, hipStream_t stream);
int vd_segment_rowsum_acc_bf16(const vd_bf16_bits* X16, int64_t ldx, const int32_t* tok, const int32_t* perm, int64_t n, int ncol,
                               float* out, int64_t ldo, hipStream_t stream);

// bf16 shadows of fp32 activations (opt-in bf16 option recurrence, VD_FLAG_BF16; api.hip).  The producing kernels of a
// bf16 pass (LSTM forward: h; LSTM backward: da) also write a bf16 copy of what they store, into a library-owned buffer
// registered against the fp32 tensor's address range; the weight-gradient contraction of the same pass finds the two
// shadows by address and multiplies them directly (LDS-DMA + transpose reads, no fp32 -> bf16 conversion while staging).
// One shadow per registered tensor (keyed by device + base address: two models in a process never share one; the `slot` argument of the
// older two-slot registry is ignored).  A shadow is valid until the next producer call on the same tensor, or
// until an fp32 producer overwrites its range (vd_bf16_shadow_invalidate: the fp32 recurrences, a forward pass over the gates buffer,
// vd_memset).  The registry is process-global and keyed by address only: VD_FLAG_BF16 on vd_gemm_tn_acc is meant for the contraction that
// directly follows the bf16 recurrences of the same pass -- a host that writes those ranges by other means (its own kernels, a tensor
// library) must not pass the flag afterwards.
int vd_bf16_shadow_get(int slot, const float* base, size_t floats, vd_bf16_bits** out);
const vd_bf16_bits* vd_bf16_shadow_find(const float* p, size_t floats);
void vd_bf16_shadow_invalidate(const float* p, size_t floats);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate nonlinearities on the hardware exp and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error ~1e-7,
// two orders below the 1e-5 per-op / 1e-4 end-to-end parity budget.  The reciprocal is the bare instruction on
// purpose: `__fdividef` / `1.0f / x` expand to the IEEE division sequence (2 v_div_scale, v_rcp, 4 v_fma, v_div_fmas,
// v_div_fixup = ~10 VALU instructions), which was HALF of the LSTM step epilogue's 1 550 VALU instructions -- and VALU
// issue in the epilogue comes straight out of the co-resident waves' matrix-pipe time.
__device__ __forceinline__ float vd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float vd_sigmoid(float x) { return vd_rcp(1.0f + __expf(-x)); }
#ifndef VD_TANH_FMA
__device__ __forceinline__ float vd_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow
  const float t = (1.0f - e) * vd_rcp(1.0f + e);
  return copysignf(t, x);
}
#else
// A/B build (-DVD_TANH_FMA): tanh(x) = 2 / (1 + e^(-2x)) - 1 -- three VALU + two transcendental instructions instead of five + two,
// the same ~6e-8 absolute error near 0, no NaN (e^(-2x) = inf -> rcp = 0 -> -1).  43 / 53 instructions fewer in the forward / backward
// step epilogues, all 226 GPU tests green, no measurable change of the step (profiles/r03_experiments.txt 13d): not the default.
__device__ __forceinline__ float vd_tanh(float x) { return fmaf(2.0f, vd_rcp(1.0f + __expf(-2.0f * x)), -1.0f); }
#endif

// Streaming (non-temporal) 16-byte accesses for data that is written once and read much later (saved gates, da):
// A/B build knob -DVD_EPI_NT=1 (`make variant NAME=nt DEFS=-DVD_EPI_NT=1`), else plain accesses.
#ifndef VD_EPI_NT
#define VD_EPI_NT 0
#endif
__device__ __forceinline__ void vd_st4_stream(float* p, const float4& v) {
#if VD_EPI_NT
  f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ float4 vd_ld4_stream(const float* p) {
#if VD_EPI_NT
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}

// Buffer addressing for the step kernels' epilogues: an SGPR descriptor per tensor + a 32-bit VGPR byte offset + an
// SGPR byte offset for the uniform part (gate / row-group strides).  buffer_load / buffer_store compute the address in the
// memory pipeline, so the ~130 64-bit VALU address instructions a tile's epilogue spent on `ptr + (long)row * ld + j` go away --
// and VALU issue is what the epilogue costs (profiles/r03_experiments.txt section 13c).  The compiler tracks vmcnt for these
// builtins (unlike inline asm).  No reliance on the hardware range check: the SGPR offset is not part of it on gfx9.
typedef unsigned vd_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned vd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vd_rsrc(const void* p) {
  // num_records = 2^32 - 1: byte offsets are 32-bit, every tensor addressed this way is < 4 GB per launch (checked by the callers)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ float4 vd_buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const vd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void vd_buf_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float4& v) {
  const vd_u32x4 t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 0);
}
__device__ __forceinline__ void vd_buf_st4_bf16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float4& v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  const bf4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(vd_u32x2, t), r, voff, soff, 0);
}

// 4 packed bf16 -> float4
__device__ __forceinline__ float4 vd_bf16x4_unpack(const vd_u32x2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
// 4 consecutive bf16 (one 8-byte buffer load) -> float4
__device__ __forceinline__ float4 vd_buf_ld4_bf16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return vd_bf16x4_unpack(__builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// 4 consecutive fp32 -> 4 bf16 (RNE, v_cvt_pk_bf16_f32), one 8-byte store
__device__ __forceinline__ void vd_st4_bf16(vd_bf16_bits* p, const float4& v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf4*>(p) = t;
}

// wave64 all-reduce helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
