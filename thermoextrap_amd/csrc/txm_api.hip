// txm_api.hip -- runtime part of the C ABI: device selection, memory helpers,
// error reporting.  Plumbing only; the kernels live in the other txm_*.hip files.
#include <stdarg.h>
#include <string.h>

#include "txm_common.h"

namespace txm {

static thread_local char g_err[512] = "";
static int g_num_cus = 256;  // MI355X; refreshed by txm_init

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return e == hipErrorNoDevice ? TXM_ERR_NO_DEVICE : TXM_ERR_HIP;
}

int num_cus() { return g_num_cus; }

}  // namespace txm

using namespace txm;

// The hash of the kernel sources this library was built from (thermoextrap_amd/_build.py passes it when it compiles this
// file; "unknown" for a hand-made build).  The marker string is what _build.needs_build() looks for in the binary -- a
// library whose hash is not the tree's is rebuilt whatever the file times say -- and txm_csrc_sha() what bench.py reports.
#ifndef TXM_CSRC_SHA
#define TXM_CSRC_SHA "unknown"
#endif
extern "C" const char txm_csrc_sha_marker[] = "TXM_CSRC_SHA=" TXM_CSRC_SHA;
extern "C" const char *txm_csrc_sha(void) { return txm_csrc_sha_marker + 13; }

extern "C" int txm_abi_version(void) { return TXM_ABI_VERSION; }
extern "C" int txm_sampler_stream_version(void) { return TXM_SAMPLER_STREAM_VERSION; }

extern "C" const char *txm_last_error(void) { return g_err; }

extern "C" int txm_device_count(int *count_host) {
  TXM_REQUIRE(count_host, "device_count: null pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count_host = 0;
    return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  *count_host = n;
  return TXM_OK;
}

extern "C" int txm_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    set_error("txm_init: no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return TXM_ERR_NO_DEVICE;
  }
  TXM_REQUIRE(device >= 0 && device < n, "txm_init: device %d outside [0, %d)", device, n);
  TXM_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TXM_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("txm_init: device %d is %s; libtxmom is built for gfx950 (MI355X) only", device,
              prop.gcnArchName);
    return TXM_ERR_NO_DEVICE;
  }
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return TXM_OK;
}

extern "C" int txm_malloc(void **ptr_host, size_t bytes) {
  TXM_REQUIRE(ptr_host, "malloc: null pointer");
  TXM_HIP(hipMalloc(ptr_host, bytes ? bytes : 1));
  return TXM_OK;
}

extern "C" int txm_free(void *ptr) {
  if (ptr) TXM_HIP(hipFree(ptr));
  return TXM_OK;
}

extern "C" int txm_memcpy_h2d(void *dst, const void *src_host, size_t bytes, txm_stream stream) {
  TXM_REQUIRE(dst && src_host, "memcpy_h2d: null pointer");
  TXM_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return TXM_OK;
}

extern "C" int txm_memcpy_d2h(void *dst_host, const void *src, size_t bytes, txm_stream stream) {
  TXM_REQUIRE(dst_host && src, "memcpy_d2h: null pointer");
  TXM_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return TXM_OK;
}

extern "C" int txm_memset(void *dst, int value, size_t bytes, txm_stream stream) {
  TXM_REQUIRE(dst, "memset: null pointer");
  TXM_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return TXM_OK;
}

extern "C" int txm_stream_sync(txm_stream stream) {
  TXM_HIP(hipStreamSynchronize((hipStream_t)stream));
  return TXM_OK;
}
