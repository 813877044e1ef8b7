// Error reporting for the C ABI (thread-local last-error text).
#include "common.h"

static thread_local char g_err[512] = "";

void rsp_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof g_err - 1);
  g_err[sizeof g_err - 1] = 0;
}

extern "C" {
const char* rsp_last_error(void) { return g_err; }
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
int rsp_version(void) { return 110; }
}
