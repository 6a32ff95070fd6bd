// Error string, version and device check of the C ABI (include/dfdclip.h).
#include "common.hpp"
#include <string.h>

static thread_local char g_err[512] = "";

void dfd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dfd_last_error(void) { return g_err; }
extern "C" int dfd_abi_version(void) { return DFD_ABI_VERSION; }

extern "C" int dfd_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    dfd_set_error("dfd_device_check: no HIP device");
    return DFD_ERR_NO_DEVICE;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    dfd_set_error("dfd_device_check: hipGetDeviceProperties failed");
    return DFD_ERR_NO_DEVICE;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    dfd_set_error("dfd_device_check: device %d is %s, this library is built for gfx950 only", dev, p.gcnArchName);
    return DFD_ERR_NO_DEVICE;
  }
  return DFD_OK;
}
