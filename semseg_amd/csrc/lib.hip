// Library-wide state: error string, version.
#include "common.h"
thread_local char ss_err_buf[512] = {0};
int ss_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(ss_err_buf, sizeof(ss_err_buf), fmt, ap); va_end(ap);
    return code;
}
extern "C" const char* ss_last_error(void) { return ss_err_buf; }
extern "C" int ss_version(void) { return 1; }
