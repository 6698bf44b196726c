// Probe (tools/libhippomm_probe.so only): what read bandwidth does HBM deliver to a kernel that does nothing but stream?
// The ceiling the scan kernels are measured against in profiles/LABNOTES.md 4.1.  Grid-stride over `n_bytes` in float4 pieces,
// `unroll` independent 16-B loads per lane in flight (1 KiB per wave and load), non-temporal or default cache policy.
#include "hmm_common.h"

namespace hmm {

template <bool NT>
__device__ __forceinline__ float4 ld16(const float4* p) {
    if constexpr (NT) {
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *p;
    }
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void hbm_read_kernel(const float4* __restrict__ src, int64_t n_vec, float* __restrict__ sink) {
    const int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
    float acc = 0.f;
    for (int64_t base = ((int64_t)blockIdx.x * 256 * UNROLL) + threadIdx.x; base < n_vec; base += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = base + (int64_t)u * 256;
            v[u] = i < n_vec ? ld16<NT>(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

}  // namespace hmm

extern "C" int hmm_probe_hbm_read(const void* src_dev, int64_t n_bytes, int blocks, int unroll, int nt, float* sink_dev,
                                  hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* s = static_cast<const float4*>(src_dev);
    const int64_t n = n_bytes / 16;
    HMM_REQUIRE(src_dev && sink_dev && blocks > 0 && (unroll == 4 || unroll == 8 || unroll == 16), HMM_E_INVALID, "hbm_read: bad arguments");
#define HMM_RD(U) do { if (nt) hbm_read_kernel<U, true><<<blocks, 256, 0, st>>>(s, n, sink_dev); \
                       else    hbm_read_kernel<U, false><<<blocks, 256, 0, st>>>(s, n, sink_dev); } while (0)
    if (unroll == 4) HMM_RD(4); else if (unroll == 8) HMM_RD(8); else HMM_RD(16);
#undef HMM_RD
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
