// Error plumbing and version of the C ABI.
#include "common.h"

namespace ytvln {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ytvln

extern "C" int ytvln_version(void) { return YTVLN_ABI_VERSION; }
extern "C" const char* ytvln_last_error(void) { return ytvln::err_buf(); }
