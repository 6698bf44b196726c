// Probe (tools/libhippomm_probe.so only): what does a grid-wide barrier cost on MI355X, against the ~4.5 us a dependent
// kernel launch costs in a stream?  `blocks` persistent workgroups (one per CU) run `rounds` barriers; between two barriers
// every workgroup touches `work_bytes` of a buffer (0 = nothing) so that the release / acquire has something to order.
//   mode 0: one device-scope counter, every workgroup adds and spins on it
//   mode 1: one counter per XCD (blockIdx % 8) + a device counter the last arriver of each XCD adds to; everybody spins on
//           the device counter
//   mode 2: as mode 0, spinning with s_sleep between polls
//   mode 3 (round 5): FLAG ARRAY -- no read-modify-write at all: workgroup b stores the round number to flags[b] (release,
//           agent scope), wave 0 polls all flags (lane l: flags[4 l .. 4 l + 3], relaxed agent-scope loads) and fences once
//   mode 4: as mode 3 without any agent-scope fence: the data between barriers is written and read with agent-scope
//           (L2-bypassing, write-through) accesses instead, so no L2 write-back / invalidate is needed at the barrier
//   mode 5: as mode 4 with a master: workgroup 0 polls the flags and publishes one "go" word that the others poll
// Every spin is bounded (2^20 polls): a bug ends in an error flag (ctr[15]), not in a hung GPU.
#include "hmm_common.h"

namespace hmm {

__device__ __forceinline__ unsigned ld_acquire(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(512) void grid_barrier_kernel(unsigned* ctr /* [512] zeroed */, int rounds, float* buf, int work_floats,
                                                           unsigned long long* stamps) {
    const int nb = gridDim.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (work_floats) {
            for (int i = threadIdx.x; i < work_floats; i += blockDim.x) {
                const int j = (blockIdx.x * work_floats + i + r * 64) % (nb * work_floats);
                if constexpr (MODE >= 4) {
                    acc += __hip_atomic_load(buf + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(buf + blockIdx.x * work_floats + i, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    acc += buf[j];
                    buf[blockIdx.x * work_floats + i] = acc;
                }
            }
        }
        __syncthreads();
        if constexpr (MODE >= 3) {
            const unsigned target = (unsigned)(r + 1);
            unsigned* flags = ctr + 64;                                  // [256] flags, then [1] go word at ctr[32]
            if (threadIdx.x < 64) {
                if (threadIdx.x == 0) {
                    if constexpr (MODE == 3) __hip_atomic_store(flags + blockIdx.x, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(flags + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                int spins = 0;
                if (MODE != 5 || blockIdx.x == 0) {
                    bool ok;
                    do {
                        ok = true;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int b = threadIdx.x * 4 + j;
                            if (b < nb) ok = ok && __hip_atomic_load(flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
                        }
                        ok = __all(ok);
                    } while (!ok && ++spins < (1 << 20));
                    if (MODE == 5 && threadIdx.x == 0) __hip_atomic_store(ctr + 32, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(ctr + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 20)) {}
                }
                if (spins >= (1 << 20)) ctr[15] = 1;
                if constexpr (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(r + 1);
            if constexpr (MODE == 1) {
                const int x = blockIdx.x & 7, per = (nb + 7 - x) / 8;                  // workgroups with this blockIdx % 8
                const unsigned old = __hip_atomic_fetch_add(ctr + 1 + x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == target * (unsigned)per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_acquire(ctr) < target * 8u) {}
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_acquire(ctr) < target * (unsigned)nb) {
                    if constexpr (MODE == 2) __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (acc == 123.456f) buf[0] = acc;
}

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 999) p[0] = 1.f; }

}  // namespace hmm

extern "C" int hmm_probe_grid_barrier(unsigned* ctr_dev, int blocks, int threads, int rounds, int mode, float* buf_dev,
                                      int work_floats, unsigned long long* stamps_dev, hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HMM_REQUIRE(ctr_dev && stamps_dev && blocks >= 1 && blocks <= 256 && threads <= 512, HMM_E_INVALID, "grid_barrier: bad arguments");
    HMM_HIP_CHECK(hipMemsetAsync(ctr_dev, 0, 512 * sizeof(unsigned), st));
    if (mode == 0) grid_barrier_kernel<0><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    else if (mode == 1) grid_barrier_kernel<1><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    else if (mode == 2) grid_barrier_kernel<2><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    else if (mode == 3) grid_barrier_kernel<3><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    else if (mode == 4) grid_barrier_kernel<4><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    else grid_barrier_kernel<5><<<blocks, threads, 0, st>>>(ctr_dev, rounds, buf_dev, work_floats, stamps_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// `n` dependent launches alternating two dynamic-LDS sizes (bytes), kernels that do nothing but touch their LDS: does a kernel that
// asks for more than 64 KiB of LDS cost more at the boundary?
__global__ void empty_lds_kernel(float* p) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) sm[0] = 1.f;
    __syncthreads();
    if (p && threadIdx.x == 999) p[0] = sm[0];
}
extern "C" int hmm_probe_empty_launches_lds(int n, int blocks, int threads, int lds_a, int lds_b, hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < n; ++i) empty_lds_kernel<<<blocks, threads, (i & 1) ? lds_b : lds_a, st>>>(nullptr);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// `n` dependent launches of a kernel that does nothing, in one stream: the per-launch floor
extern "C" int hmm_probe_empty_launches(int n, int blocks, int threads, hmm_stream_t stream) {
    using namespace hmm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) empty_kernel<<<blocks, threads, 0, st>>>(nullptr);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
