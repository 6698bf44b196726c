// Error channel + ABI version of libhippomm_hip.so (see include/hippomm_hip.h).
#include "hmm_common.h"
#include <string.h>

namespace hmm {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace hmm

extern "C" int hmm_abi_version(void) { return 7; }      // 6: hmm_host_rgbx_to_rgb, hmm_host_arrow_rgbx_to_rgb; 7: hmm_rank_segment_hits
extern "C" const char* hmm_last_error(void) { return hmm::g_err; }

// The kernels are written for one target: launch geometries assume 256 CUs in 8 XCDs (hmm_common.h) and the code objects are
// gfx950 only.  HMM_OK when the CURRENT device is that; HMM_E_STATE with a message otherwise.
extern "C" int hmm_device_supported(void) {
    int dev = 0;
    HMM_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HMM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const bool arch_ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    HMM_REQUIRE(arch_ok && prop.multiProcessorCount == hmm::kNumCU, HMM_E_STATE,
                "device %d is %s with %d CUs; libhippomm_hip.so is built for gfx950 (MI355X) with %d CUs", dev,
                prop.gcnArchName, prop.multiProcessorCount, hmm::kNumCU);
    return HMM_OK;
}
