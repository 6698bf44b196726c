// Probe (tools/libhippomm_probe.so only): what does the chip sustain on bf16 MFMA alone, under its power cap, and does the
// instruction shape or the LDS operand traffic move that number?  256 workgroups x 8 waves (one per CU, 2 waves per SIMD,
// like gemm_bf16_pp_kernel), every wave owns a 128 x 64 fp32 accumulator tile and per "step" issues the MFMAs of one
// K = 32 slice of it:
//   mode 0  32 x v_mfma_f32_16x16x32_bf16, operands held in registers (loop invariant)
//   mode 1  16 x v_mfma_f32_32x32x16_bf16, operands held in registers
//   mode 2  as 0, operands re-read from LDS every step (12 ds_read_b128 per step: the real kernel's ratio)
//   mode 3  as 1, operands re-read from LDS every step (12 ds_read_b128 per step)
//   mode 4  as 0, but the operands CHANGE every step (each fragment register is XOR-ed with a step-dependent mask of its low
//           mantissa / sign bits: 12 VALU per 32 MFMAs) -- modes 0-3 keep the multiplier inputs static, which is not what a
//           GEMM does to them
//   mode 6  as 2, reading a different 12-fragment set every step (8 sets = 96 KiB of LDS per workgroup... per wave 12 KiB x 8)
// Operand values are pseudo-random bf16 in (-1, 1) (zeros would draw far less power).  Per workgroup the kernel stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop: clock = d(memtime) / d(realtime) * 100 MHz.
#include "hmm_common.h"

namespace hmm {

template <int MODE>
__global__ __launch_bounds__(512) void mfma_power_kernel(int steps, float* __restrict__ sink,
                                                         unsigned long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 24 KiB of pseudo-random bf16 per wave: 12 fragments x 64 lanes x 16 B, lane-linear (conflict-free b128)
    unsigned* w32 = reinterpret_cast<unsigned*>(smem) + wave * 12 * 64 * 4;
    unsigned s = 0x9e3779b9u * (blockIdx.x * 512 + threadIdx.x + 1);
    for (int i = 0; i < 12 * 4; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned lo = 0x3f000000u | ((s >> 9) & 0x007f0000u) | ((s & 1u) << 31);        // bf16 in [0.5, 1) with sign
        const unsigned hi = 0x3f000000u | ((s >> 1) & 0x007f0000u) | ((s & 2u) << 30);
        w32[(i >> 2) * 256 + lane * 4 + (i & 3)] = (lo >> 16) | (hi & 0xffff0000u);
    }
    __syncthreads();
    const bf16x8* frag = reinterpret_cast<const bf16x8*>(w32) + lane;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = frag[i * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = frag[(8 + i) * 64];

    f32x4 acc16[(MODE & 1) ? 1 : 32];
    f32x16 acc32[(MODE & 1) ? 8 : 1];
#pragma unroll
    for (auto& v : acc16) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (auto& v : acc32)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = 0.f;

    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const bf16x8* frag0 = reinterpret_cast<const bf16x8*>(smem) + lane;
    for (int st = 0; st < steps; ++st) {
        if constexpr (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const volatile bf16x8*>(frag + i * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const volatile bf16x8*>(frag + (8 + i) * 64);
        }
        if constexpr (MODE == 6) {                        // another wave's fragment set every step: the data changes, the traffic does not
            const bf16x8* f = frag0 + ((wave + st) & 7) * 12 * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const volatile bf16x8*>(f + i * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const volatile bf16x8*>(f + (8 + i) * 64);
        }
        if constexpr (MODE == 4) {                        // flip sign and low mantissa bits of every operand register, a new mask per step
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned m = 0x80078007u * ((unsigned)st * 2654435761u >> 31) ^ (((unsigned)st * 40503u) & 0x00070007u);
#pragma unroll
            for (int i = 0; i < 8; ++i) { u32x4 t = __builtin_bit_cast(u32x4, a[i]); t[i & 3] ^= m; a[i] = __builtin_bit_cast(bf16x8, t); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { u32x4 t = __builtin_bit_cast(u32x4, b[i]); t[i & 3] ^= m; b[i] = __builtin_bit_cast(bf16x8, t); }
        }
        if constexpr ((MODE & 1) == 0) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc16[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc16[mi * 4 + ni], 0, 0, 0);
        } else {                                          // 4 x 2 blocks of 32 x 32, two K = 16 halves
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc32[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kh * 4 + mi], b[kh * 2 + ni], acc32[mi * 2 + ni], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
#pragma unroll
    for (auto& v : acc16) t += v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (auto& v : acc32)
#pragma unroll
        for (int j = 0; j < 16; ++j) t += v[j];
    if (t == 123.456f) sink[threadIdx.x] = t;             // keeps the accumulators alive
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace hmm

extern "C" int hmm_probe_mfma_power(int mode, int steps, float* sink_dev, unsigned long long* ticks_dev, hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr int lds = 100 * 1024;                       // one workgroup per CU
    HMM_REQUIRE(mode >= 0 && mode <= 6 && mode != 5 && steps > 0 && sink_dev && ticks_dev, HMM_E_INVALID, "mfma_power: bad arguments");
    switch (mode) {
        case 0: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<0>, lds); mfma_power_kernel<0><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
        case 1: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<1>, lds); mfma_power_kernel<1><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
        case 2: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<2>, lds); mfma_power_kernel<2><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
        case 3: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<3>, lds); mfma_power_kernel<3><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
        case 4: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<4>, lds); mfma_power_kernel<4><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
        default: { HMM_ENSURE_DYN_LDS(mfma_power_kernel<6>, lds); mfma_power_kernel<6><<<256, 512, lds, st>>>(steps, sink_dev, ticks_dev); break; }
    }
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
