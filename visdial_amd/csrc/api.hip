// Library-level entry points: error string, device memory helpers for hosts that have no
// tensor library of their own (the LuaJIT-FFI host), device queries.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void vd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* vd_last_error(void) { return g_err; }

int vd_abi_version(void) { return 1; }

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
