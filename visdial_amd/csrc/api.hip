// Library-level entry points: error string, device memory helpers for hosts that have no
// tensor library of their own (the LuaJIT-FFI host), device queries.
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "common.h"
#include "../../include/visdial_hip.h"

static thread_local char g_err[1024] = "";

void vd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- per-(device, stream) scratch -----------------------------------------------------------------------
static std::mutex g_scr_mu;
static std::map<std::pair<int, hipStream_t>, VdStreamScratch>& scratch_table() {
  static std::map<std::pair<int, hipStream_t>, VdStreamScratch> t;
  return t;
}

int vd_stream_scratch(hipStream_t stream, size_t wht_bytes, size_t sync_bytes, VdStreamScratch* out) {
  int dev = 0;
  VD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_scr_mu);
  VdStreamScratch& s = scratch_table()[std::make_pair(dev, stream)];
  // Growing frees the old buffer: hipFree synchronises the device, so earlier stream-ordered users are done.
  if (s.wht_bytes < wht_bytes) {
    if (s.wht) VD_HIP(hipFree(s.wht));
    s.wht = nullptr;
    s.wht_bytes = 0;
    VD_HIP(hipMalloc((void**)&s.wht, wht_bytes));
    s.wht_bytes = wht_bytes;
  }
  if (s.sync_bytes < sync_bytes) {
    if (s.sync) VD_HIP(hipFree(s.sync));
    s.sync = nullptr;
    s.sync_bytes = 0;
    VD_HIP(hipMalloc((void**)&s.sync, sync_bytes));
    VD_HIP(hipMemset(s.sync, 0, sync_bytes));
    s.sync_bytes = sync_bytes;
  }
  *out = s;
  return VD_OK;
}

// ---- bf16 shadows (common.h) -----------------------------------------------------------------------------
namespace {
struct Bf16Shadow {
  const float* base = nullptr;   // registered fp32 range [base, base + floats)
  size_t floats = 0;
  vd_bf16_bits* buf = nullptr;
  size_t cap = 0;                // elements allocated
  int dev = -1;
};
Bf16Shadow g_shadow[2];
std::mutex g_shadow_mu;
}  // namespace

int vd_bf16_shadow_get(int slot, const float* base, size_t floats, vd_bf16_bits** out) {
  std::lock_guard<std::mutex> lk(g_shadow_mu);
  Bf16Shadow& s = g_shadow[slot & 1];
  int dev = 0;
  VD_HIP(hipGetDevice(&dev));
  if (s.cap < floats || s.dev != dev) {
    if (s.buf) VD_HIP(hipFree(s.buf));     // synchronises the device: earlier users are done
    s.buf = nullptr;
    s.cap = 0;
    VD_HIP(hipMalloc((void**)&s.buf, floats * sizeof(vd_bf16_bits)));
    s.cap = floats;
    s.dev = dev;
  }
  s.base = base;
  s.floats = floats;
  *out = s.buf;
  return VD_OK;
}

const vd_bf16_bits* vd_bf16_shadow_find(const float* p, size_t floats) {
  std::lock_guard<std::mutex> lk(g_shadow_mu);
  for (const Bf16Shadow& s : g_shadow)
    if (s.base && p >= s.base && p + floats <= s.base + s.floats) return s.buf + (p - s.base);
  return nullptr;
}

void vd_bf16_shadow_invalidate(const float* p, size_t floats) {
  std::lock_guard<std::mutex> lk(g_shadow_mu);
  for (Bf16Shadow& s : g_shadow)
    if (s.base && p < s.base + s.floats && s.base < p + floats) s.base = nullptr;
}

int vd_num_cus() {
  static thread_local int cached_dev = -1, cached = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return cached;
  if (dev != cached_dev) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached = n;
    cached_dev = dev;
  }
  return cached;
}

extern "C" {

const char* vd_last_error(void) { return g_err; }

int vd_abi_version(void) { return VD_ABI_VERSION; }

int vd_device_count(int* count) {
  VD_CHECK_ARG(count, "vd_device_count: null");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    vd_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return VD_ERR_HIP;
  }
  return VD_OK;
}

int vd_set_device(int device) {
  VD_HIP(hipSetDevice(device));
  return VD_OK;
}

// name must hold >= 256 bytes; arch e.g. "gfx950..."
int vd_device_info(int device, char* name, char* arch, int* num_cus, int64_t* hbm_bytes) {
  hipDeviceProp_t p;
  VD_HIP(hipGetDeviceProperties(&p, device));
  if (name) strncpy(name, p.name, 255);
  if (arch) strncpy(arch, p.gcnArchName, 255);
  if (num_cus) *num_cus = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return VD_OK;
}

int vd_malloc(void** ptr, int64_t bytes) {
  VD_CHECK_ARG(ptr && bytes >= 0, "vd_malloc: bad args");
  VD_HIP(hipMalloc(ptr, (size_t)bytes));
  return VD_OK;
}

int vd_free(void* ptr) {
  VD_HIP(hipFree(ptr));
  return VD_OK;
}

int vd_memset(void* ptr, int value, int64_t bytes, void* stream) {
  vd_bf16_shadow_invalidate(static_cast<const float*>(ptr), (size_t)(bytes / 4));     // (a writer of an fp32 range: its bf16 shadow is stale)
  VD_HIP(hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream));
  return VD_OK;
}

int vd_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream) {
  VD_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return VD_OK;
}

int vd_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream) {
  VD_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return VD_OK;
}

int vd_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
  VD_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return VD_OK;
}

// strided 2-D device copy: `rows` rows of `cols` floats (nn.JoinTable / nn.Narrow on column blocks)
int vd_copy_2d(float* dst, int64_t dst_ld, const float* src, int64_t src_ld, int64_t rows, int64_t cols,
               void* stream) {
  VD_CHECK_ARG(dst && src && rows >= 0 && cols >= 0 && dst_ld >= cols && src_ld >= cols, "vd_copy_2d: bad args");
  if (rows == 0 || cols == 0) return VD_OK;
  VD_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * 4, src, (size_t)src_ld * 4, (size_t)cols * 4, (size_t)rows,
                          hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return VD_OK;
}

int vd_stream_synchronize(void* stream) {
  VD_HIP(hipStreamSynchronize((hipStream_t)stream));
  return VD_OK;
}

}  // extern "C"
