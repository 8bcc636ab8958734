#include "common.h"
#include <stdarg.h>
namespace kdip {
thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
}  // namespace kdip
