// Error reporting + version of libpfhip.so.
#include <cstring>

#include "pf_common.h"

namespace pf {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace pf

extern "C" int pf_version(void) { return 1000; }
extern "C" const char *pf_last_error(void) { return pf::error_buffer(); }
