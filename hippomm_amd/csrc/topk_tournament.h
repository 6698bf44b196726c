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

// v: one key per lane, bitonic across the wave -> sorted descending by lane
__device__ __forceinline__ uint64_t wave_bitonic_merge_desc(uint64_t v, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint64_t o = shfl_xor_u64(v, j);
        const bool keep_max = (lane & j) == 0;
        v = ((v > o) == keep_max) ? v : o;
    }
    return v;
}

// v: one key per lane, any order -> sorted descending by lane
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t o = shfl_xor_u64(v, j);
            const bool desc = (lane & k) == 0;                 // k = 64: every lane
            const bool keep_max = ((lane & j) == 0) == desc;
            v = ((v > o) == keep_max) ? v : o;
        }
    return v;
}

// s[0 .. n2): keys, n2 = 64 << m.  On return (after a workgroup barrier) s[0..63] holds the 64 largest, descending; the
// rest of s is clobbered.  Called by every thread of the workgroup (blockDim.x a multiple of 64).
__device__ __forceinline__ void top64_desc(uint64_t* s, int n2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    int chunks = n2 >> 6;
    for (int c = wave; c < chunks; c += n_waves) s[c * 64 + lane] = wave_sort_desc(s[c * 64 + lane], lane);
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
