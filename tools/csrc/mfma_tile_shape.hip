// Probe (tools/libhippomm_probe.so only): would FOUR waves with 128 x 128 accumulator tiles (one wave per SIMD, a third fewer LDS
// fragment bytes per MFMA) beat the shipped EIGHT waves with 128 x 64 tiles (two per SIMD), once each wave also has to issue its
// share of the global -> LDS staging stream?  Same CU-level work per "step" (a 256 x 256 x 32 slice: 256 MFMAs, 32 KiB of LDS
// fill), no barriers, operands re-read from LDS every step, fragments double-buffered in registers:
//   waves 8: per wave and step 32 MFMAs, 12 ds_read_b128, 4 global_load_lds_dwordx4 (with DMA)
//   waves 4: per wave and step 64 MFMAs, 16 ds_read_b128, 8 global_load_lds_dwordx4 (with DMA)
// The DMA reads an L2-resident region and lands in a ring the fragment reads do not touch (the data is not used: this prices
// issue slots and power, not correctness).  Stamps as mfma_power.hip.
#include "hmm_common.h"

namespace hmm {

template <int WAVES, bool DMA, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void mfma_tile_shape_kernel(int steps, const char* __restrict__ src, float* __restrict__ sink,
                                                                      unsigned long long* __restrict__ ticks) {
    constexpr int NI = WAVES == 8 ? 4 : 8;                 // 16-column blocks per wave tile (rows: always 8 blocks = 128)
    constexpr int NF = 8 + NI;                             // fragments per K = 32 step
    constexpr int NDMA = 32 / WAVES;                       // 1-KiB pieces per wave and step: 32 KiB per CU and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // fragment region: 2 sets x NF fragments x 1 KiB per wave (lane-linear, conflict-free b128); ring behind it
    unsigned* w32 = reinterpret_cast<unsigned*>(smem) + wave * 2 * NF * 256;
    unsigned s = 0x9e3779b9u * (blockIdx.x * 512 + threadIdx.x + 1);
    for (int i = 0; i < 2 * NF * 4; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned lo = 0x3f000000u | ((s >> 9) & 0x007f0000u) | ((s & 1u) << 31);
        const unsigned hi = 0x3f000000u | ((s >> 1) & 0x007f0000u) | ((s & 2u) << 30);
        w32[(i >> 2) * 256 + lane * 4 + (i & 3)] = (lo >> 16) | (hi & 0xffff0000u);
    }
    __syncthreads();
    char* ring = smem + WAVES * 2 * NF * 1024 + wave * (2 * NDMA * 1024);
    const char* region = src + (size_t)(blockIdx.x & 7) * (2u << 20);
    unsigned off = ((blockIdx.x >> 3) * 32768u + wave * (NDMA * 1024u) + lane * 16u) & ((2u << 20) - 1);
    const bf16x8* frag = reinterpret_cast<const bf16x8*>(w32) + lane;
    bf16x8 a[2][8], b[2][NI];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[0][i] = frag[i * 64];
#pragma unroll
    for (int i = 0; i < NI; ++i) b[0][i] = frag[(8 + i) * 64];
    f32x4 acc[8 * NI];
#pragma unroll
    for (auto& v : acc) v = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int st = 0; st < steps; st += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                      // two steps per trip: fragment set h is consumed, set h ^ 1 loaded
            const bf16x8* f = frag + (h ^ 1) * NF * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[h ^ 1][i] = *reinterpret_cast<const volatile bf16x8*>(f + i * 64);
#pragma unroll
            for (int i = 0; i < NI; ++i) b[h ^ 1][i] = *reinterpret_cast<const volatile bf16x8*>(f + (8 + i) * 64);
            if constexpr (DMA) {
#pragma unroll
                for (int i = 0; i < NDMA; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(region + ((off + i * 1024u) & ((2u << 20) - 1))),
                                                     (__attribute__((address_space(3))) void*)(ring + (h * NDMA + i) * 1024), 16, 0, 0);
                off = (off + 32u * 32768u) & ((2u << 20) - 1);
                // DEPTH steps of pieces stay in flight (the data is never read, so a slot may be overwritten while it lands)
                if constexpr (NDMA * DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (NDMA * DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (NDMA * DEPTH == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (NDMA * DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h][mi], b[h][ni], acc[mi * NI + ni], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
#pragma unroll
    for (auto& v : acc) t += v[0] + v[1] + v[2] + v[3];
    if (t == 123.456f) sink[threadIdx.x] = t;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace hmm

extern "C" int hmm_probe_mfma_tile_shape(int waves, int dma /* 0 = no stream, d = d steps of pieces in flight (1..3) */, int steps, const void* src_dev, float* sink_dev,
                                         unsigned long long* ticks_dev, hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HMM_REQUIRE((waves == 4 || waves == 8) && steps > 0 && steps % 2 == 0 && src_dev && sink_dev && ticks_dev, HMM_E_INVALID,
                "mfma_tile_shape: bad arguments");
    constexpr int lds = 150 * 1024;                        // one workgroup per CU
    const char* src = static_cast<const char*>(src_dev);
#define HMM_GO(W, D, P) do { HMM_ENSURE_DYN_LDS((mfma_tile_shape_kernel<W, D, P>), lds); \
                             mfma_tile_shape_kernel<W, D, P><<<256, W * 64, lds, st>>>(steps, src, sink_dev, ticks_dev); } while (0)
    HMM_REQUIRE(dma >= 0 && dma <= 3, HMM_E_INVALID, "mfma_tile_shape: dma depth %d", dma);
    if (waves == 8) { if (dma == 0) HMM_GO(8, false, 1); else if (dma == 1) HMM_GO(8, true, 1); else if (dma == 2) HMM_GO(8, true, 2); else HMM_GO(8, true, 3); }
    else            { if (dma == 0) HMM_GO(4, false, 1); else if (dma == 1) HMM_GO(4, true, 1); else if (dma == 2) HMM_GO(4, true, 2); else HMM_GO(4, true, 3); }
#undef HMM_GO
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
