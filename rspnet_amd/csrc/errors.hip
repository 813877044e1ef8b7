// Error reporting for the C ABI (thread-local last-error text).
#include "common.h"
#include <cstdarg>

static thread_local char g_err[512] = "";

void rsp_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof g_err - 1);
  g_err[sizeof g_err - 1] = 0;
}

// Name of the FIRST matrix kernel the current conv call dispatched (thread-local): written by the launchers themselves, so
// rsp_conv3d_kernel_name's prediction can be cross-checked against what actually ran (tests/test_kernels_gpu.py).
static thread_local char g_kernel[96] = "";
static thread_local bool g_kernel_armed = false;

void rsp_note_reset(void) {
  g_kernel[0] = 0;
  g_kernel_armed = true;
}

void rsp_note_kernel(const char* fmt, ...) {
  if (!g_kernel_armed) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof g_kernel, fmt, ap);
  va_end(ap);
  g_kernel_armed = false;
}

extern "C" {
const char* rsp_last_error(void) { return g_err; }
const char* rsp_last_conv_kernel(void) { return g_kernel; }
const char* rsp_strerror(int code) {
  switch (code) {
    case RSP_OK: return "ok";
    case RSP_EINVAL: return "invalid argument";
    case RSP_EWORKSPACE: return "workspace too small";
    case RSP_ELAUNCH: return "kernel launch failed";
    case RSP_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
int rsp_version(void) { return 120; }
}
