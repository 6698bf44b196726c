// Best-64 selection over order keys in LDS without a full sort: a tournament of 64-key chunks, each held one key per
// lane, sorted and merged with cross-lane exchanges (no LDS traffic, no workgroup barrier inside a chunk).
//   1. every chunk of 64 keys is sorted descending by one wave (21 compare-exchange steps on registers);
//   2. per level, two sorted chunks A, B -> max(A[i], B[63 - i]) holds the best 64 of the 128 and is bitonic: six more
//      steps sort it; the winners move to the front half of the chunk list.  log2(chunks) levels, two workgroup barriers
//      each (readers of a chunk / its overwriter).
// Used where only the best k <= 64 of n keys are wanted: the per-block candidate lists of the scan kernels and the final
// selection kernels (a full bitonic sort of 2048 + 1024 keys cost 121 barrier-separated LDS stages = 36 us per query,
// 5.5 % of a 1M-row scan).  Keys are unique (similarity bits << 32 | row) or zero padding, so the result is the same
// set in the same order as a full descending sort would leave in s[0..63].
#pragma once
#include "hmm_common.h"

namespace hmm {

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

// The key of lane (lane ^ J), through the crossbar-free exchanges of hmm_common.h (lane_xor_b32).  A compare-exchange step is a chain
// link: the next one needs its result, and a selection kernel is nothing but such chains (a best-64 tournament over 2048 keys = 72
// dependent steps); with ds_bpermute (~120 cycles a step) prefilter_final_kernel's two rankings took 7.3 us each, 4.7 with the row
// exchanges alone (profiles/r6_prefilter_final_stamps*.json).
template <int J>
__device__ __forceinline__ uint64_t lane_xor(uint64_t v, int) {
    const uint32_t lo = (uint32_t)lane_xor_b32<J>((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)lane_xor_b32<J>((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// One compare-exchange of the bitonic network: stage K (runs of K lanes, descending in even runs), distance J.
template <int K, int J>
__device__ __forceinline__ void bitonic_steps(uint64_t& v, int lane) {
    const uint64_t o = lane_xor<J>(v, lane);
    const bool desc = K >= 64 || (lane & K) == 0;
    const bool keep_max = ((lane & J) == 0) == desc;
    v = ((v > o) == keep_max) ? v : o;
    if constexpr (J > 1) bitonic_steps<K, J / 2>(v, lane);
}
template <int K>
__device__ __forceinline__ void bitonic_stages(uint64_t& v, int lane) {
    bitonic_steps<K, K / 2>(v, lane);
    if constexpr (K < 64) bitonic_stages<K * 2>(v, lane);
}

// v: one key per lane, bitonic across the wave -> sorted descending by lane
__device__ __forceinline__ uint64_t wave_bitonic_merge_desc(uint64_t v, int lane) {
    bitonic_steps<64, 32>(v, lane);
    return v;
}

// v: one key per lane, any order -> sorted descending by lane
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t v, int lane) {
    bitonic_stages<2>(v, lane);
    return v;
}

// s[0 .. n2): keys, n2 = 64 << m.  On return (after a workgroup barrier) s[0..63] holds the 64 largest, descending; the
// rest of s is clobbered.  Called by every thread of the workgroup (blockDim.x a multiple of 64).
template <bool PAIRS = true>
__device__ __forceinline__ void top64_desc(uint64_t* s, int n2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    int chunks = n2 >> 6;
    if constexpr (PAIRS) {
        for (int c = wave; c < chunks; c += 2 * n_waves) {     // two chunks per wave at a time: two independent chains interleave
            const int c1 = c + n_waves;
            uint64_t v0 = s[c * 64 + lane], v1 = c1 < chunks ? s[c1 * 64 + lane] : 0ull;
            bitonic_stages<2>(v0, lane);
            bitonic_stages<2>(v1, lane);
            s[c * 64 + lane] = v0;
            if (c1 < chunks) s[c1 * 64 + lane] = v1;
        }
    } else {                                                   // inside a streaming kernel: registers are the occupancy
        for (int c = wave; c < chunks; c += n_waves) s[c * 64 + lane] = wave_sort_desc(s[c * 64 + lane], lane);
    }
    while (chunks > 1) {
        const int pairs = chunks >> 1;
        __syncthreads();                                       // the chunks of this level are complete
        for (int p0 = 0; p0 < pairs; p0 += n_waves) {          // same trip count for every wave
            const int p = p0 + wave;
            uint64_t v = 0ull;
            if (p < pairs) {
                const uint64_t a = s[(2 * p) * 64 + lane], b = s[(2 * p + 1) * 64 + 63 - lane];
                v = wave_bitonic_merge_desc(a > b ? a : b, lane);
            }
            // chunk p may still be read as chunk 2p' or 2p'+1 by a slower wave of this round: every wave finishes the
            // reads of a round before any wave writes its result (a barrier per round; the rounds after it only read
            // chunks >= 2 (p0 + n_waves), which this round does not write).  Holds for any n2, not only pairs <= 2 n_waves.
            __syncthreads();
            if (p < pairs) s[p * 64 + lane] = v;
        }
        chunks = pairs;
    }
    __syncthreads();
}

}  // namespace hmm
