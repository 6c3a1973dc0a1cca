// abi.hip -- version / error-string entry points of the C-ABI (include/mi_detectron_ops.h).
#include "common.h"

namespace mi {
namespace {
thread_local char g_error[512] = {0};
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
void clear_error() { g_error[0] = 0; }
}  // namespace mi

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return mi::g_error; }
