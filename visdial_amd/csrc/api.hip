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

#include <dlfcn.h>

static thread_local char g_err[1024] = "";

// ---- ROCTx (common.h) ------------------------------------------------------------------------------------------------------------
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve() {
  const char* on = getenv("VD_ROCTX");
  if (!on || on[0] != '1') return;
  for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void* lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!lib) continue;
    g_roctx_push = (roctx_push_fn)dlsym(lib, "roctxRangePushA");
    g_roctx_pop = (roctx_pop_fn)dlsym(lib, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) return;
    g_roctx_push = nullptr;
    g_roctx_pop = nullptr;
  }
}
}  // namespace
void vd_range_push(const char* name) {
  std::call_once(g_roctx_once, roctx_resolve);
  if (g_roctx_push) g_roctx_push(name);
}
void vd_range_pop() {
  if (g_roctx_pop) g_roctx_pop();
}

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
// One entry per registered fp32 range (keyed by device + base address), so that two models -- or two recurrences of one model -- in a
// process never share a shadow buffer.  At most VD_MAX_SHADOWS live entries; an invalidated entry keeps its buffer and is the first to be
// re-used, then the least recently registered one is evicted.  (`slot` of the older two-slot registry is accepted and ignored.)
namespace {
struct Bf16Shadow {
  const float* base = nullptr;   // registered fp32 range [base, base + floats); nullptr = free (buffer kept for re-use)
  size_t floats = 0;
  vd_bf16_bits* buf = nullptr;
  size_t cap = 0;                // elements allocated
  int dev = -1;
  unsigned long stamp = 0;       // registration order (eviction)
};
constexpr int VD_MAX_SHADOWS = 8;
Bf16Shadow g_shadow[VD_MAX_SHADOWS];
unsigned long g_shadow_clock = 0;
std::mutex g_shadow_mu;
}  // namespace

int vd_bf16_shadow_get(int /*slot*/, const float* base, size_t floats, vd_bf16_bits** out) {
  std::lock_guard<std::mutex> lk(g_shadow_mu);
  int dev = 0;
  VD_HIP(hipGetDevice(&dev));
  Bf16Shadow* pick = nullptr;
  for (Bf16Shadow& s : g_shadow)                                   // the same tensor again (every pass of a training loop)
    if (s.base == base && s.dev == dev) pick = &s;
  if (!pick)
    for (Bf16Shadow& s : g_shadow)                                 // a free entry whose buffer already fits
      if (!s.base && s.dev == dev && s.cap >= floats && (!pick || s.cap < pick->cap)) pick = &s;
  if (!pick)
    for (Bf16Shadow& s : g_shadow)                                 // a never-used entry
      if (!s.base && !s.buf) { pick = &s; break; }
  if (!pick)
    for (Bf16Shadow& s : g_shadow)                                 // any free entry, else the least recently registered one
      if (!pick || (!s.base && pick->base) || ((!s.base) == (!pick->base) && s.stamp < pick->stamp)) pick = &s;
  Bf16Shadow& s = *pick;
  // every OTHER live entry that overlaps the new range is stale from now on: the caller is about to (re)write [base, base + floats) and its shadow;
  // an older entry registered over a larger or shifted range (the allocator handed out a sub-block of a freed tensor) would otherwise still
  // match vd_bf16_shadow_find() -- in array order, possibly before this one -- and serve bytes nobody maintains
  for (Bf16Shadow& o : g_shadow)
    if (&o != &s && o.base && o.dev == dev && base < o.base + o.floats && o.base < base + floats) o.base = nullptr;
  // evicting a LIVE entry registered for another tensor hands its buffer to a new owner while contractions enqueued on other streams may
  // still read it: wait for the device once (rare: more than VD_MAX_SHADOWS live tensors)
  if (s.base && s.base != base) VD_HIP(hipDeviceSynchronize());
  if (s.cap < floats || s.dev != dev) {
    if (s.buf) VD_HIP(hipFree(s.buf));     // synchronises the device: earlier users are done
    s.buf = nullptr;
    s.cap = 0;
    s.base = nullptr;
    VD_HIP(hipMalloc((void**)&s.buf, floats * sizeof(vd_bf16_bits)));
    s.cap = floats;
    s.dev = dev;
  }
  s.base = base;
  s.floats = floats;
  s.stamp = ++g_shadow_clock;
  *out = s.buf;
  return VD_OK;
}

const vd_bf16_bits* vd_bf16_shadow_find(const float* p, size_t floats) {
  std::lock_guard<std::mutex> lk(g_shadow_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (const Bf16Shadow& s : g_shadow)
    if (s.base && s.dev == dev && p >= s.base && p + floats <= s.base + s.floats) return s.buf + (p - s.base);
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
