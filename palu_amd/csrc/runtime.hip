// Error reporting, device queries and host-side helpers of the C ABI (include/palu_hip.h).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "palu_common.h"

static thread_local char g_err[512] = "";

void palu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int palu_num_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device): the attribute belongs to the function on ONE
// device, and the library serves every GPU of the process (palu_amd._lib.on_device), possibly from several threads.
int palu_func_max_lds(const void* func, int bytes) {
  static std::mutex mu;
  static struct { const void* f; int dev; int bytes; } done[1024];
  static int n = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  for (int i = 0; i < n; ++i)
    if (done[i].f == func && done[i].dev == dev && done[i].bytes >= bytes) return PALU_OK;
  hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    palu_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", bytes, hipGetErrorString(e));
    return PALU_ERR_LAUNCH;
  }
  if (n < 1024) {
    done[n].f = func; done[n].dev = dev; done[n].bytes = bytes;
    ++n;
  }
  return PALU_OK;
}

extern "C" const char* palu_last_error(void) { return g_err; }
extern "C" int palu_version(void) { return 100; }

// kernel/pytorch_reference.py:4 -- fp32: 1 / theta^(2i/D).  (Callers that need bit-identity with
// a framework's own pow should pass their own table; see palu_hip.h.)
extern "C" int palu_rope_inv_freq_host(float theta, int head_dim, float* out_host) {
  PALU_REQUIRE(out_host && head_dim > 0 && head_dim % 2 == 0, PALU_ERR_ARG, "rope_inv_freq: bad arguments");
  for (int i = 0; i < head_dim / 2; ++i) {
    float e = (float)(2 * i) / (float)head_dim;
    out_host[i] = 1.0f / powf(theta, e);
  }
  return PALU_OK;
}
