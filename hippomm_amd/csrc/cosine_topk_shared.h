// Shared by the scan kernels (cosine_topk.hip) and the bf16-prefilter path (cosine_topk_prefilter.hip): launch geometry, the
// per-row arithmetic that DEFINES a similarity's bits, and the conditional exact scan.
#pragma once
#include "hmm_common.h"

namespace hmm {

constexpr int kChunk = 4096;                 // keys sorted per block
constexpr int kScanBlocks = kNumCU * 8;      // workgroups of a streaming pass
constexpr int kFusedCap = 1024;              // block-local candidate list of the streaming kernels

template <bool NT>
__device__ __forceinline__ float4 ld16(const float4* p) {
    if constexpr (NT) {
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *p;
    }
}

__device__ __forceinline__ void fma4(float& dot, float& ss, const float4& x, const float4& q) {
    dot = fmaf(x.x, q.x, dot); ss = fmaf(x.x, x.x, ss);
    dot = fmaf(x.y, q.y, dot); ss = fmaf(x.y, x.y, ss);
    dot = fmaf(x.z, q.z, dot); ss = fmaf(x.z, x.z, ss);
    dot = fmaf(x.w, q.w, dot); ss = fmaf(x.w, x.w, ss);
}

__device__ __forceinline__ int pow2_at_least(int n, int lo) {
    int p = lo;
    while (p < n) p <<= 1;
    return p;
}

// The similarity of one fp32 row as scan_topk_kernel / scan_sims_kernel compute it: lane l holds the float4 at columns
// 4 (64 j + l), j = 0..3 (row_lane = row + lane), q[j] the query's, q_len = sqrtf(wave_sum(sum q^2)) in the same lane order.
// Same loads, same fma order, same wave reduction, same division: the same bits (tests/test_gpu_scan.py compares them).
__device__ __forceinline__ float exact_row_sim(const float4* row_lane, const float4 (&q)[4], float q_len) {
    float4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ld16<true>(row_lane + j * 64);
    float d = 0.f, s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) fma4(d, s, a[j], q[j]);
    d = wave_sum(d);
    s = wave_sum(s);
    return d / (sqrtf(s) * q_len);
}

// Up to four rows at once: all sixteen loads first (one memory latency for the group), then each row's arithmetic exactly as
// exact_row_sim has it -- the same bits.  rows[i] for i >= n are not touched; sim[i] is valid for i < n.
__device__ __forceinline__ void exact_row_sim4(const float4* const (&row_lane)[4], int n, const float4 (&q)[4], float q_len,
                                               float (&sim)[4]) {
    float4 a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[i][j] = ld16<true>(row_lane[i] + j * 64);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sim[i] = 0.f;
        if (i < n) {
            float d = 0.f, s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) fma4(d, s, a[i][j], q[j]);
            d = wave_sum(d);
            s = wave_sum(s);
            sim[i] = d / (sqrtf(s) * q_len);
        }
    }
}

// The per-event selection kernels come in two shapes (see segment_topk_kernel): true when the small one serves this call.
constexpr int kSmallSegChunk = 1024;
bool segments_are_small(int64_t n_rows, int n_segments, int k);

int cosine_topk_if(const int* run_if, unsigned* ticket, const float* store, int64_t n, const float* query, int k, int64_t* idx_out,
                   float* sim_out, int32_t* n_out, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace hmm
