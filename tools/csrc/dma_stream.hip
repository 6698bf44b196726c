// Probe (tools/libhippomm_probe.so only): what does the global -> LDS path cost in power when nothing else runs?
// 256 workgroups x 8 waves (one per CU, as the ping-pong GEMM), every wave keeps 16 global_load_lds_dwordx4 pieces (1 KiB each)
// in flight into a 128-KiB LDS ring and does nothing with the data -- the staging stream of the GEMM without its MFMAs, LDS
// reads and stores.  The workgroups of one XCD (blockIdx % 8) cycle through a region of `region_bytes` of their own:
// 2 MiB stays in that XCD's L2, 16 MiB is served by the Infinity Cache, 256 MiB comes from HBM.  Stamps as mfma_power.hip.
#include "hmm_common.h"

namespace hmm {

__global__ __launch_bounds__(512) void dma_stream_kernel(const char* __restrict__ src, unsigned region_bytes, int iters,
                                                         unsigned long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* region = src + (size_t)(blockIdx.x & 7) * region_bytes;
    // a workgroup reads 64 KiB per iteration; the 32 workgroups of an XCD start 64 KiB apart and step by 2 MiB
    unsigned off = ((blockIdx.x >> 3) * 65536u + wave * 8192u + lane * 16u) % region_bytes;
    const unsigned step = (32u * 65536u) % region_bytes;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        char* dst = smem + (it & 1) * 65536 + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(region + ((off + i * 1024u) % region_bytes)),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   // the previous iteration's 8 pieces have landed
        off += step;
        if (off >= region_bytes) off -= region_bytes;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace hmm

extern "C" int hmm_probe_dma_stream(const void* src_dev, unsigned region_bytes, int iters, unsigned long long* ticks_dev,
                                    hmm_stream_t stream) {
    using namespace hmm;
    HMM_REQUIRE(src_dev && ticks_dev && iters > 0 && region_bytes >= (1u << 21) && region_bytes % 65536u == 0, HMM_E_INVALID,
                "dma_stream: bad arguments");
    constexpr int lds = 128 * 1024;
    HMM_ENSURE_DYN_LDS(dma_stream_kernel, lds);
    dma_stream_kernel<<<256, 512, lds, static_cast<hipStream_t>(stream)>>>(static_cast<const char*>(src_dev), region_bytes, iters, ticks_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
