// Error channel + ABI version of libhippomm_hip.so (see include/hippomm_hip.h).
#include "hmm_common.h"

namespace hmm {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace hmm

extern "C" int hmm_abi_version(void) { return 3; }
extern "C" const char* hmm_last_error(void) { return hmm::g_err; }
