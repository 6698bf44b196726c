// Internal helpers shared by the gfx950 kernels of libhippomm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/hippomm_hip.h"

namespace hmm {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

void set_error(const char* fmt, ...);

#define HMM_REQUIRE(cond, code, ...)                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            ::hmm::set_error(__VA_ARGS__);                             \
            return (code);                                             \
        }                                                              \
    } while (0)

#define HMM_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::hmm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                             __FILE__, __LINE__);                                        \
            return HMM_E_HIP;                                                            \
        }                                                                                \
    } while (0)

#define HMM_LAUNCH_CHECK() HMM_HIP_CHECK(hipGetLastError())

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernels that need more than 64 KiB of dynamic LDS must raise the limit once per (kernel, DEVICE): the attribute is
// per device, so a second GPU in the same process needs its own call.  One static bit mask per call site (= per kernel
// instantiation); the launch functions are otherwise stateless.
#define HMM_ENSURE_DYN_LDS(kern, bytes)                                                                  \
    do {                                                                                                 \
        static std::atomic<uint64_t> _hmm_attr_done{0};                                                  \
        int _hmm_dev = 0;                                                                                \
        HMM_HIP_CHECK(hipGetDevice(&_hmm_dev));                                                          \
        const uint64_t _hmm_bit = 1ull << (_hmm_dev & 63);                                               \
        if (!(_hmm_attr_done.load(std::memory_order_acquire) & _hmm_bit)) {                              \
            HMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)));     \
            _hmm_attr_done.fetch_or(_hmm_bit, std::memory_order_release);                                \
        }                                                                                                \
    } while (0)

// Tuning knobs exist only in the probe build (tools/, -DHMM_PROBE: libhippomm_probe.so); in the product library they
// are compile-time constants and no setter is exported.
#ifdef HMM_PROBE
#define HMM_TUNABLE(type, name, value) \
    type name = value;                 \
    extern "C" void hmm_probe_set_##name(type v) { name = v; }
#else
#define HMM_TUNABLE(type, name, value) static constexpr type name = value;
#endif

// ---- bf16 ------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- lane exchanges and wave reductions (64 lanes) ---------------------------------------------
// The 32-bit value of lane (lane ^ J) without the LDS crossbar: XOR 1 / 2 are quad permutations and XOR 8 a rotation of the 16-lane
// row (data-parallel-primitive modifiers on a v_mov), XOR 4 two bank-masked row shifts, XOR 16 / 32 gfx950's v_permlane16_swap /
// v_permlane32_swap.  ~10 cycles where ds_bpermute answers in ~120 -- and every step of a bitonic network waits for the previous one
// (topk_tournament.h).
__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int J>
__device__ __forceinline__ int lane_xor_b32(int v) {
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, true);                      // row_shl:4 -> banks 0, 2 (lane + 4)
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);                             // row_shr:4 -> banks 1, 3 (lane - 4)
    } else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);    // row_ror:8
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);     // odd rows of [0] <-> even rows of [1]
        return (int)((lane_id() & 16u) ? r[0] : r[1]);
    } else {
        static_assert(J == 32, "lane_xor_b32: J must be a power of two below 64");
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);     // upper half of [0] <-> lower half of [1]
        return (int)((lane_id() & 32u) ? r[0] : r[1]);
    }
}

// Butterfly reductions stay on __shfl_xor (ds_bpermute): in the streaming kernels they run beside a saturated VALU, and the LDS
// crossbar is the idle unit there -- with the VALU exchanges above the bf16 prefilter pass slowed from 304 to 314 us
// (profiles/r6_scan_trace_summary_valu_wave_sum.json); the same arithmetic and bits either way.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- total order used by the scan (see include/hippomm_hip.h) ------------------------------
// monotone float -> uint32 map; NaN -> 0xFFFFFFFF (ranks first), -0.0 -> +0.0.
__device__ __forceinline__ uint32_t order_bits(float s) {
    if (s != s) return 0xFFFFFFFFu;
    s += 0.0f;
    uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float order_bits_inverse(uint32_t m) {
    if (m == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    uint32_t b = (m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m;
    return __uint_as_float(b);
}

}  // namespace hmm
