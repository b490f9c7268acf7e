// extern "C" entry points of libgparml_hip.so (see include/gparml_hip.h).
#include "gp_common.h"

namespace gp {
thread_local std::string g_create_error;

int fail(gp_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_error = buf;
  return code;
}
}  // namespace gp

extern "C" const char* gp_last_error(const gp_ctx* ctx) { return ctx ? ctx->err.c_str() : gp::g_create_error.c_str(); }
extern "C" const char* gp_version(void) { return "gparml_hip 0.1 (gfx950)"; }
