// Probe (tools/libhippomm_probe.so only): how fast can the CUs of one XCD / of the chip retire the output burst of a
// GEMM epilogue?  One 512-thread workgroup per CU (100 KiB of dynamic LDS keeps a second one out); the workgroup in
// XCD group g = blockIdx % 8, slot s = blockIdx / 8 is active iff g < n_xcd && s < per_xcd.  An active workgroup writes
// `reps` output tiles exactly as gemm_bf16_pp_kernel's epilogue does:
//   mode 0  bf16 tile 256 x 256: every wave-instruction stores 8 rows x 128 B (16 B per lane)
//   mode 1  fp32 tile 256 x 256, read-modify-write: 4 rows x 256 B per wave-instruction, load + add + store
//   mode 2  fp32 tile, store only
// into a [rows][n_cols] matrix, tile t -> (t / tiles_n, t % tiles_n).  Timing is taken by the host (events) and, per
// workgroup, with s_memrealtime (100 MHz) around the burst loop: out_ticks[2*b] = first, [2*b+1] = last.
#include "hmm_common.h"

namespace hmm {

__global__ __launch_bounds__(512) void store_burst_kernel(char* __restrict__ out, int n_cols, int tiles_n, int n_xcd,
                                                          int per_xcd, int reps, int mode,
                                                          unsigned long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x & 7, s = blockIdx.x >> 3;
    if (g >= n_xcd || s >= per_xcd) return;
    const int active = n_xcd * per_xcd;
    const int me = s * n_xcd + g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    if (threadIdx.x == 0) smem[0] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        const int tile = r * active + me;
        const int m0 = (tile / tiles_n) * 256 + wm * 128, n0 = (tile % tiles_n) * 256 + wn * 64;
        if (mode == 0) {
            unsigned short* C = reinterpret_cast<unsigned short*>(out);
            const int rsub = lane >> 3, chunk = lane & 7;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + rsub;
                const uint4 v = make_uint4(tile, row, lane, it);
                *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * n_cols + n0 + chunk * 8) = v;
            }
        } else {
            float* C = reinterpret_cast<float*>(out);
            const int rsub = lane >> 4, chunk = lane & 15;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 xin[16];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int row = half * 64 + it * 4 + rsub;
                    xin[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (mode == 1) xin[it] = *reinterpret_cast<const float4*>(C + (size_t)(m0 + row) * n_cols + n0 + chunk * 4);
                }
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int row = half * 64 + it * 4 + rsub;
                    float4 v = xin[it];
                    v.x += 1.f; v.y += 2.f; v.z += 3.f; v.w += 4.f;
                    *reinterpret_cast<float4*>(C + (size_t)(m0 + row) * n_cols + n0 + chunk * 4) = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t0; ticks[2 * blockIdx.x + 1] = t1; }
}

}  // namespace hmm

extern "C" int hmm_probe_store_burst(void* out_dev, int n_cols, int tiles_n, int n_xcd, int per_xcd, int reps, int mode,
                                     unsigned long long* ticks_dev, hmm_stream_t stream) {
    using namespace hmm;
    HMM_ENSURE_DYN_LDS(store_burst_kernel, 100 * 1024);
    store_burst_kernel<<<256, 512, 100 * 1024, static_cast<hipStream_t>(stream)>>>(
        static_cast<char*>(out_dev), n_cols, tiles_n, n_xcd, per_xcd, reps, mode, ticks_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
