// feature_search scan for gfx950: one streaming pass over the (N,1024) fp32 store computing
// dot(row,q) and sum(row^2) per row, then an exact top-k under a documented total order.
// Replaces top_k_cosine_similarity (reference hippomm/utils/vector_ops.py:151-188).
//
// Kernels
//   scan_topk_kernel   HBM-bound (k <= 128, the usual query).  One wave per row: lane l reads the four float4 at
//                      columns 4*(64*j + l), j=0..3, so every load instruction of a wave covers
//                      1 KiB contiguous.  Two rows (8 x 16 B loads per lane) are in flight per
//                      wave; block-local top-k lists, topk_final_kernel finishes.  Algorithmic bytes: 4096 B per row, read once.
//   scan_sims_deferred_kernel  the same pass writing every similarity (per-event scan, k > 128): results held in registers,
//                      one burst of stores per 512 iterations.
//   topk_chunk_kernel  4096-key bitonic sort in LDS per block, keeps the best k of each chunk;
//                      applied repeatedly until one chunk is left (N=1M, k=32: 245 -> 2 -> 1).
//   bitonic_global_*   rare path (k > 1024 and N > 4096): full sort of all keys.
//   decode_kernel      keys -> (int64 row, fp32 sim).
// Keys are 64-bit: (order_bits(sim) << 32) | row, so "larger key" == "better, higher row first
// on ties, NaN first"; 0 is the padding key (no valid row maps to it).
#include "hmm_common.h"
#include "topk_tournament.h"
#include "cosine_topk_shared.h"

namespace hmm {

constexpr int kFastK = 1024;      // chunk-tournament path handles k <= kFastK
// Workgroups of the streaming kernels.  A pass over the store is bound by HBM, and HBM is saturated by bandwidth x latency of bytes in
// flight (~7 TB/s x 1.5 us = ~11 MB): scan_topk_kernel keeps 8 KB per wave in flight, so ~1536 waves = 384 workgroups are enough, and
// MORE of them stream slower -- 2048 workgroups 0.613 ms, 512: 0.598, 384: 0.579 (= the 0.885 of 8 TB/s a pure reader gets), 336:
// 0.597, 256: 0.721 on 1M rows in one session, a smooth bowl around 384 on two boxes (profiles/r6_scan_blocks.json).  Until round 6
// the grid was "eight workgroups per CU" by habit.  The similarity pass that ships (scan_sims_deferred_kernel) deals rows the same way
// and runs on the same grid; its predecessor scan_sims_kernel (32 rows per wave and visit, a store per visit) has no such optimum
// (0.601-0.619 ms from 256 to 2048 workgroups) and keeps 2048 where the probe build still runs it.
HMM_TUNABLE(int, g_scan_blocks, kNumCU * 3 / 2)   // scan_topk_kernel
#ifdef HMM_PROBE
HMM_TUNABLE(int, g_sims_blocks, kScanBlocks)      // scan_sims_kernel (probe build only)
HMM_TUNABLE(int, g_sims_deferred, 1)               // 0: scan_sims_kernel instead of scan_sims_deferred_kernel (probe build only)
#endif
HMM_TUNABLE(int, g_small_segment_rows, 1024)      // mean rows per event up to which the small per-event selection kernels run

bool segments_are_small(int64_t n_rows, int n_segments, int k) {
    return k <= 64 && n_segments >= 1 && n_rows <= (int64_t)n_segments * g_small_segment_rows;
}

#ifdef HMM_PROBE   // the similarity pass until round 6, kept in the probe build as what scan_sims_deferred_kernel is measured against
// sims[r] = dot(store[r], q) / (||store[r]|| * ||q||)     (vector_ops.py:178-182)
// (at most six waves per SIMD, all eight loads of a row pair in flight: see scan_topk_kernel)
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 6))) void scan_sims_kernel(const float4* __restrict__ store,
                                                        int64_t n_rows,
                                                        const float4* __restrict__ query,
                                                        float* __restrict__ sims) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = query[j * 64 + lane];
        qs = fmaf(q[j].x, q[j].x, qs); qs = fmaf(q[j].y, q[j].y, qs);
        qs = fmaf(q[j].z, q[j].z, qs); qs = fmaf(q[j].w, q[j].w, qs);
    }
    const float q_len = sqrtf(wave_sum(qs));

    // A wave takes 32 consecutive rows at a time (16 pairs) and writes their 32 similarities as ONE 128-byte line: with
    // rows dealt pair by pair, every line of `sims` was assembled from 4-byte stores of 16 different waves (partial-line
    // writes; the kernel ran 7 % behind the list-keeping scan kernel that stores nothing in its loop).
    for (int64_t base = wave * 32; base < n_rows; base += n_waves * 32) {
        float mine = 0.f;                                   // lane l < 32 ends up with the similarity of row base + l
        const int64_t left = n_rows - base;
        const int pairs = left >= 32 ? 16 : (int)((left + 1) >> 1);      // wave-uniform
        for (int jj = 0; jj < pairs; ++jj) {
            // Waves start at different pairs of their 128-KB groups: walked in step, every wave of the chip is at the same offset of
            // a 128-KB-strided group at the same time (0.6109 -> 0.6046 ms per pass, same bits; profiles/r6_sims_stagger.json)
            const int j = pairs == 16 ? ((jj + (int)(wave & 15)) & 15) : jj;
            const int64_t r = base + 2 * j;
            const bool two = (r + 1) < n_rows;
            const float4* p0 = store + r * 256 + lane;
            const float4* p1 = p0 + (two ? 256 : 0);
            float4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ld16<NT>(p0 + i * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = ld16<NT>(p1 + i * 64);
            __builtin_amdgcn_sched_barrier(0);              // all eight loads in flight before the first use (see scan_topk_body)
            float d0 = 0.f, s0 = 0.f, d1 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(d0, s0, a[i], q[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(d1, s1, b[i], q[i]);
            d0 = wave_sum(d0); s0 = wave_sum(s0);
            d1 = wave_sum(d1); s1 = wave_sum(s1);
            const float v0 = d0 / (sqrtf(s0) * q_len), v1 = d1 / (sqrtf(s1) * q_len);
            mine = lane == 2 * j ? v0 : (lane == 2 * j + 1 ? v1 : mine);
        }
        if (lane < 32 && base + lane < n_rows) sims[base + lane] = mine;
    }
}
#endif  // HMM_PROBE

// sims[r] = dot(store[r], q) / (||store[r]|| * ||q||)     (vector_ops.py:178-182)
// scan_topk_kernel's row dealing (pair p belongs to wave p % n_waves: at any moment the waves of the chip read one contiguous window
// of the store; at most six waves per SIMD, all eight loads of a row pair in flight) and NO store inside the loop: lane l of two registers keeps the results of iteration
// 64 j + l, and the wave writes everything it computed when it has read its last row.  Stores mixed into the read stream are what
// scan_sims_kernel pays for: groups of 32 rows with one 128-B line of results each 0.606 ms per pass, the same without its stores
// 0.589, this dealing with an 8-byte store per pair in the loop 0.621, without any 0.573 (scan_topk_kernel) -- and deferred: see
// profiles/r6_sims_deferred.json.  Same arithmetic per row, same bits.
constexpr int kSimsHeld = 8;
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 6))) void scan_sims_deferred_kernel(
        const float4* __restrict__ store, int64_t n_rows, const float4* __restrict__ query, float* __restrict__ sims) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = query[j * 64 + lane];
        qs = fmaf(q[j].x, q[j].x, qs); qs = fmaf(q[j].y, q[j].y, qs);
        qs = fmaf(q[j].z, q[j].z, qs); qs = fmaf(q[j].w, q[j].w, qs);
    }
    const float q_len = sqrtf(wave_sum(qs));
    // an epoch = 64 kSimsHeld iterations (1M rows on the scan's grid: one epoch of 326); larger stores write once per epoch
    for (int64_t it0 = 0; (wave + it0 * n_waves) * 2 < n_rows; it0 += 64 * kSimsHeld) {
    float h0[kSimsHeld], h1[kSimsHeld];
#pragma unroll
    for (int j = 0; j < kSimsHeld; ++j) { h0[j] = 0.f; h1[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < kSimsHeld; ++j) {
        for (int l = 0; l < 64; ++l) {
            const int64_t r = (wave + (it0 + j * 64 + l) * n_waves) * 2;
            if (r >= n_rows) break;                                          // wave-uniform; every later iteration is past the end too
            const bool two = (r + 1) < n_rows;
            const float4* p0 = store + r * 256 + lane;
            const float4* p1 = p0 + (two ? 256 : 0);
            float4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ld16<NT>(p0 + i * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = ld16<NT>(p1 + i * 64);
            __builtin_amdgcn_sched_barrier(0);                               // all eight loads in flight before the first use
            float d0 = 0.f, s0 = 0.f, d1 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(d0, s0, a[i], q[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(d1, s1, b[i], q[i]);
            d0 = wave_sum(d0); s0 = wave_sum(s0);
            d1 = wave_sum(d1); s1 = wave_sum(s1);
            const float v0 = d0 / (sqrtf(s0) * q_len), v1 = d1 / (sqrtf(s1) * q_len);
            h0[j] = lane == l ? v0 : h0[j];
            h1[j] = lane == l ? v1 : h1[j];
        }
    }
#pragma unroll
    for (int j = 0; j < kSimsHeld; ++j) {
        const int64_t r = (wave + (it0 + j * 64 + lane) * n_waves) * 2;
        if (r + 1 < n_rows)  *reinterpret_cast<float2*>(sims + r) = make_float2(h0[j], h1[j]);
        else if (r < n_rows) sims[r] = h0[j];
    }
    }
}

// All similarities of a store into sims[].
static void launch_scan_sims(const float* store, int64_t n, const float* query, float* sims, hipStream_t st) {
    int64_t waves_needed = (n + 1) / 2;
    int blocks = (int)((waves_needed + 3) / 4);
#ifdef HMM_PROBE
    if (!g_sims_deferred) {
        if (blocks > g_sims_blocks) blocks = g_sims_blocks;
        scan_sims_kernel<true><<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(store), n, reinterpret_cast<const float4*>(query), sims);
        return;
    }
#endif
    if (blocks > g_scan_blocks) blocks = g_scan_blocks;
    scan_sims_deferred_kernel<true><<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(store), n,
                                                            reinterpret_cast<const float4*>(query), sims);
}

// Descending bitonic sort of N keys in LDS by NT threads.
template <int N, int NT>
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* s) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += NT) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool desc = (i & k) == 0;
                const uint64_t a = s[i], b = s[l];
                if ((a < b) == desc) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
}

// Descending bitonic sort of the first n2 (power of two) keys in LDS by the whole block.
__device__ __forceinline__ void bitonic_sort_desc_rt(uint64_t* s, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool desc = (i & k) == 0;
                const uint64_t a = s[i], b = s[l];
                if ((a < b) == desc) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
}

// Each block sorts one chunk and writes its best `k` keys (0-padded) to out[blockIdx.x*k ..].
template <bool FROM_SIMS>
__global__ __launch_bounds__(1024) void topk_chunk_kernel(const void* __restrict__ in, int64_t n_in,
                                                          int k, uint64_t* __restrict__ out) {
    __shared__ uint64_t s[kChunk];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int64_t left = n_in - base;
    const int n2 = pow2_at_least((int)(left < kChunk ? left : kChunk), 64);
    for (int t = threadIdx.x; t < n2; t += 1024) {
        const int64_t g = base + t;
        uint64_t key = 0;
        if (g < n_in) {
            if constexpr (FROM_SIMS) {
                const float v = static_cast<const float*>(in)[g];
                key = ((uint64_t)order_bits(v) << 32) | (uint64_t)(uint32_t)g;
            } else {
                key = static_cast<const uint64_t*>(in)[g];
            }
        }
        s[t] = key;
    }
    __syncthreads();
    if (k <= 64) top64_desc(s, n2);            // the best 64, sorted, is all that leaves (a store of 3600 rows: 44.7 -> 17.7 us per query;
    else         bitonic_sort_desc_rt(s, n2);  // the full sort was 66 barrier-separated stages over 4096 keys)
    for (int t = threadIdx.x; t < k; t += 1024) out[(int64_t)blockIdx.x * k + t] = t < n2 ? s[t] : 0ull;
}

// Fused path for k <= kFusedK: the streaming kernel keeps its own candidates.  Every wave appends the
// order keys of its rows to a block-local LDS list; whenever the list could overflow, and once at the
// end, the block sorts it and keeps its best k, which go to out[blockIdx.x*k ..] (0-padded).  Nothing
// but k keys per block (8*k B) is written, and the follow-up passes see blocks*k keys instead of N.
constexpr int kFusedK = 128;

// The streaming body: every wave appends the order keys of its rows to the block's LDS list `cand` (kFusedCap entries), which is cut
// back to its best k whenever it could overflow and once at the end; the block's best k keys, sorted, go to out[blockIdx.x * k ..].
template <bool NT>
__device__ __forceinline__ void scan_topk_body(const float4* __restrict__ store, int64_t n_rows, const float4* __restrict__ query, int k,
                                               uint64_t* __restrict__ out, uint64_t* cand, int* count_p) {
    int& count = *count_p;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    if (threadIdx.x == 0) count = 0;

    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = query[j * 64 + lane];
        qs = fmaf(q[j].x, q[j].x, qs); qs = fmaf(q[j].y, q[j].y, qs);
        qs = fmaf(q[j].z, q[j].z, qs); qs = fmaf(q[j].w, q[j].w, qs);
    }
    const float q_len = sqrtf(wave_sum(qs));
    __syncthreads();

    const int64_t iters = (n_rows + n_waves * 2 - 1) / (n_waves * 2);     // same trip count for every wave
    const int compact_every = (kFusedCap - k) / 8;                        // 8 rows per block per iteration
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t r = wave * 2 + it * n_waves * 2;
        const bool one = r < n_rows, two = (r + 1) < n_rows;              // wave-uniform
        if (one) {
            const float4* p0 = store + r * 256 + lane;
            const float4* p1 = p0 + (two ? 256 : 0);
            float4 a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ld16<NT>(p0 + j * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = ld16<NT>(p1 + j * 64);
            // All eight loads leave before the first fma: the kernel lives on bytes in flight per wave.  Left to itself the
            // scheduler waits for the first load alone before it issues the other seven, and under register pressure it folds the
            // second half of the loads into the arithmetic (four in flight instead of eight).
            __builtin_amdgcn_sched_barrier(0);
            float d0 = 0.f, s0 = 0.f, d1 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) fma4(d0, s0, a[j], q[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fma4(d1, s1, b[j], q[j]);
            d0 = wave_sum(d0); s0 = wave_sum(s0);
            d1 = wave_sum(d1); s1 = wave_sum(s1);
            if (lane == 0) {
                const int pos = atomicAdd(&count, two ? 2 : 1);
                cand[pos] = ((uint64_t)order_bits(d0 / (sqrtf(s0) * q_len)) << 32) | (uint64_t)(uint32_t)r;
                if (two)
                    cand[pos + 1] = ((uint64_t)order_bits(d1 / (sqrtf(s1) * q_len)) << 32) | (uint64_t)(uint32_t)(r + 1);
            }
        }
        if ((it + 1) % compact_every == 0 || it + 1 == iters) {          // block-uniform
            __syncthreads();
            const int n = count;
            const int n2 = pow2_at_least(n, 64);
            for (int t = n + threadIdx.x; t < n2; t += 256) cand[t] = 0ull;
            __syncthreads();
            if (k <= 64) top64_desc<false>(cand, n2);                   // block-uniform; leaves cand[0..63] sorted
            else         bitonic_sort_desc_rt(cand, n2);
            if (threadIdx.x == 0) count = n < k ? n : k;
            __syncthreads();
        }
    }
    const int n = count;
    for (int t = threadIdx.x; t < k; t += 256) out[(int64_t)blockIdx.x * k + t] = t < n ? cand[t] : 0ull;
}

// At most SIX waves per SIMD.  The kernel is bound by HBM and wants many bytes in flight per wave (eight 16-byte loads per lane),
// not many waves: at seven or eight waves per SIMD -- what 72 or 64 registers allow -- the same code streams 4 % slower (0.599 ->
// 0.625 ms on 1M rows in one session; 4, 5 and 6 waves are equal; profiles/r6_scan_ab_variants.json).  Round 5's build sat at six by
// the accident of needing 74 registers; round 6's leaner tournament dropped it to 66 and lost the 4 % until this limit was written down.
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 6))) void scan_topk_kernel(const float4* __restrict__ store, int64_t n_rows,
                                                        const float4* __restrict__ query, int k,
                                                        uint64_t* __restrict__ out) {
    __shared__ uint64_t cand[kFusedCap];
    __shared__ int count;
    scan_topk_body<NT>(store, n_rows, query, k, out, cand, &count);
}

// Best 64 of n_keys keys produced by key_at(i), through a 1024-key LDS window (the best 64 so far stay at w[0..63], 960 new keys per
// round): for the exact scan's conditional fallback below, which must not cost the streaming kernel its occupancy.
template <class F>
__device__ __forceinline__ void top64_of_stream(uint64_t* w, int n_keys, F key_at) {
    int base = 0, keep = 0;
    do {
        const int take = kFusedCap - keep;
        for (int t = threadIdx.x; t < take; t += blockDim.x) w[keep + t] = base + t < n_keys ? key_at(base + t) : 0ull;
        base += take;
        keep = 64;
        __syncthreads();
        top64_desc(w, kFusedCap);                                            // ends on a barrier: w[0..63] = the best so far
    } while (base < n_keys);
}

// The exact scan as the CONDITIONAL FALLBACK of the bf16-prefilter path (cosine_topk_prefilter.hip), one launch with its own symbol:
// returns at once unless *run_if != 0 (the usual case: the scan's 384 workgroups exit, ~4.5 us); otherwise scan_topk_kernel's streaming body,
// and the workgroup that takes the last ticket (device-scope counter, zeroed by prefilter_final_kernel) finishes as topk_final_kernel
// does -- the k lists with the largest maxima hold the answer -- in the 8 KB of LDS the streaming body already has.  k <= 64.
__global__ __launch_bounds__(256) void exact_scan_fallback_kernel(const float4* __restrict__ store, int64_t n_rows,
                                                                  const float4* __restrict__ query, int k,
                                                                  uint64_t* __restrict__ lists, const int* __restrict__ run_if,
                                                                  unsigned* __restrict__ ticket, int64_t* __restrict__ idx_out,
                                                                  float* __restrict__ sim_out, int32_t* __restrict__ n_out) {
    __shared__ uint64_t cand[kFusedCap];
    __shared__ uint64_t winners[64];
    __shared__ int count;
    __shared__ int last;
    if (*run_if == 0) return;                                               // workgroup-uniform
    scan_topk_body<true>(store, n_rows, query, k, lists, cand, &count);
    __threadfence();                                                        // this block's list is visible device-wide ...
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;    // ... before its ticket is
    __syncthreads();
    if (!last) return;
    __threadfence();                                                        // every other block's list is visible here
    const int n_blocks = (int)gridDim.x;
    const int64_t n_waves = (int64_t)n_blocks * 4;
    const volatile uint64_t* all = lists;
    top64_of_stream(cand, n_blocks, [&](int b) { return all[(int64_t)b * k]; });          // the block maxima
    if (threadIdx.x < 64) winners[threadIdx.x] = cand[threadIdx.x];
    __syncthreads();
    const int n_win = n_blocks < k ? n_blocks : k;
    top64_of_stream(cand, n_win * k, [&](int t) {
        const uint64_t top = winners[t / k];
        if (top == 0ull) return (uint64_t)0;
        const int blk = (int)((((int64_t)(top & 0xFFFFFFFFull) >> 1) % n_waves) >> 2);     // row r belongs to wave (r / 2) % n_waves
        return (uint64_t)all[(int64_t)blk * k + (t % k)];
    });
    if (threadIdx.x == 0) { *n_out = k; *ticket = 0u; }
    for (int t = threadIdx.x; t < k; t += 256) {
        idx_out[t] = (int64_t)(cand[t] & 0xFFFFFFFFull);
        sim_out[t] = order_bits_inverse((uint32_t)(cand[t] >> 32));
    }
}

// Final selection for the fused path when k*k <= kChunk.  Every scan block left its best k keys, sorted,
// at cand[b*k ..].  The global top-k can only contain keys of the k blocks with the largest maxima: a key
// below the k-th largest block maximum has k better keys (those maxima) ahead of it.  One workgroup sorts
// the block maxima, gathers the k*k keys of the winning blocks, sorts them and decodes the best k.
// Replaces two chunk-sort passes and the decode launch.  Both selections are best-64 tournaments (topk_tournament.h):
// full bitonic sorts here were 66 + 55 barrier-separated stages of 1024 threads, 36 us = 5.5 % of a 1M-row query.
__global__ __launch_bounds__(1024) void topk_final_kernel(const uint64_t* __restrict__ cand, int n_blocks, int k,
                                                          int64_t n_waves, uint64_t* __restrict__ keys_out, int k_pad,
                                                          int64_t* __restrict__ idx_out, float* __restrict__ sim_out,
                                                          int32_t* __restrict__ n_out) {
    __shared__ uint64_t mx[kScanBlocks];       // block maxima
    __shared__ uint64_t s[kChunk];             // keys of the winning blocks
    const int tid = threadIdx.x;
    const int n2 = pow2_at_least(n_blocks, 64);
    for (int t = tid; t < n2; t += 1024) mx[t] = t < n_blocks ? cand[(int64_t)t * k] : 0ull;
    __syncthreads();
    top64_desc(mx, n2);                        // k <= 64 here (k*k <= kChunk)
    // mx[0..k) = the k largest maxima; recover their blocks from the row index (row r belongs to wave
    // (r/2) % n_waves, 4 waves per block) and gather those blocks' lists
    const int n_win = n_blocks < k ? n_blocks : k;
    const int m2 = pow2_at_least(n_win * k, 64);
    for (int t = tid; t < m2; t += 1024) {
        uint64_t key = 0ull;
        if (t < n_win * k) {
            const uint64_t top = mx[t / k];
            if (top != 0ull) {
                const int64_t row = (int64_t)(top & 0xFFFFFFFFull);
                const int blk = (int)(((row >> 1) % n_waves) >> 2);
                key = cand[(int64_t)blk * k + (t % k)];
            }
        }
        s[t] = key;
    }
    __syncthreads();
    top64_desc(s, m2);
    if (keys_out != nullptr) {
        for (int t = tid; t < k_pad; t += 1024) keys_out[t] = t < k ? s[t] : 0ull;
    } else {
        if (tid == 0 && n_out) *n_out = k;
        for (int t = tid; t < k; t += 1024) {
            idx_out[t] = (int64_t)(s[t] & 0xFFFFFFFFull);
            sim_out[t] = order_bits_inverse((uint32_t)(s[t] >> 32));
        }
    }
}

__global__ void keys_from_sims_kernel(const float* __restrict__ sims, int64_t n, int64_t n_pad,
                                      uint64_t* __restrict__ keys) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pad) return;
    keys[g] = g < n ? (((uint64_t)order_bits(sims[g]) << 32) | (uint64_t)(uint32_t)g) : 0ull;
}

__global__ void bitonic_global_step_kernel(uint64_t* __restrict__ keys, int64_t half, int64_t j, int64_t k) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= half) return;
    const int64_t i = 2 * t - (t & (j - 1));
    const int64_t l = i + j;
    const bool desc = (i & k) == 0;
    const uint64_t a = keys[i], b = keys[l];
    if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
}

__global__ void decode_kernel(const uint64_t* __restrict__ keys, int k_out,
                              int64_t* __restrict__ idx_out, float* __restrict__ sim_out,
                              int32_t* __restrict__ n_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && n_out) *n_out = k_out;
    if (t >= k_out) return;
    const uint64_t key = keys[t];
    idx_out[t] = (int64_t)(key & 0xFFFFFFFFull);
    sim_out[t] = order_bits_inverse((uint32_t)(key >> 32));
}

__global__ void copy_keys_kernel(const uint64_t* __restrict__ src, int n_src, uint64_t* __restrict__ dst, int k) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k) dst[t] = t < n_src ? src[t] : 0ull;
}

// Merge n_shards x k shard-local keys (0-padded) into the global top-k.
__global__ __launch_bounds__(256) void merge_keys_kernel(const uint64_t* __restrict__ keys, int n_shards, int k,
                                                         const int64_t* __restrict__ row_offset,
                                                         int64_t* __restrict__ idx_out,
                                                         float* __restrict__ sim_out,
                                                         int32_t* __restrict__ n_out) {
    __shared__ uint64_t s[kChunk];
    __shared__ int n_valid;
    if (threadIdx.x == 0) n_valid = 0;
    __syncthreads();
    const int total = n_shards * k;
    int mine = 0;
    for (int t = threadIdx.x; t < kChunk; t += 256) {
        uint64_t key = 0;
        if (t < total) {
            key = keys[t];
            if (key != 0) {
                const uint64_t row = (key & 0xFFFFFFFFull) + (uint64_t)row_offset[t / k];
                key = (key & 0xFFFFFFFF00000000ull) | (row & 0xFFFFFFFFull);
                ++mine;
            }
        }
        s[t] = key;
    }
    atomicAdd(&n_valid, mine);
    __syncthreads();
    bitonic_sort_desc<kChunk, 256>(s);
    const int k_out = n_valid < k ? n_valid : k;
    if (threadIdx.x == 0) *n_out = k_out;
    for (int t = threadIdx.x; t < k_out; t += 256) {
        idx_out[t] = (int64_t)(s[t] & 0xFFFFFFFFull);
        sim_out[t] = order_bits_inverse((uint32_t)(s[t] >> 32));
    }
}

// ------------------------------------------------------------------------------------------------------
// Per-event feature_search.  The reference loops over events in Python and scans each event's
// matrix with k=5 (hippocampal_memory.py:3143-3153, :3294-3304).  Here all events live concatenated in one
// resident store with a row-offset table: one streaming pass computes every similarity, then one workgroup
// per event selects that event's top-k (indices are rows WITHIN the event) under the same total order.
// Events larger than a chunk are folded in pieces, carrying the running best k.  Two shapes of the same code: 4096 keys / 1024
// threads, and -- when the events average at most g_small_segment_rows rows and k <= 64 -- 1024 keys / 256 threads: a selection is a
// chain of dependent steps whatever the thread count, and eight small workgroups fit a CU where two large ones do (2000 events of
// 500 rows: one round of workgroups instead of four).  Same result: keys are unique and totally ordered.
// ------------------------------------------------------------------------------------------------------
template <int CHUNK, int THREADS>
__global__ __launch_bounds__(THREADS) void segment_topk_kernel(const float* __restrict__ sims,
                                                            const int64_t* __restrict__ seg_off, int k,
                                                            int64_t* __restrict__ idx_out, float* __restrict__ sim_out,
                                                            int32_t* __restrict__ n_out) {
    __shared__ uint64_t s[CHUNK];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = seg_off[e], hi = seg_off[e + 1];
    const int64_t n = hi - lo;
    const int k_out = (int)(n < k ? (n > 0 ? n : 0) : k);
    int have = 0;                                              // best keys carried from earlier pieces, at s[0..have)
    int64_t base = 0;
    do {
        const int64_t left = n - base;
        const int take = (int)(left < (int64_t)(CHUNK - have) ? left : (int64_t)(CHUNK - have));
        const int total = have + take;
        const int n2 = pow2_at_least(total, 64);
        for (int t = have + tid; t < n2; t += THREADS) {
            uint64_t key = 0ull;
            if (t < total) {
                const int64_t r = base + (t - have);
                key = ((uint64_t)order_bits(sims[lo + r]) << 32) | (uint64_t)(uint32_t)r;
            }
            s[t] = key;
        }
        __syncthreads();
        if (k <= 64) top64_desc(s, n2);                        // best-64 tournament: s[0..63] sorted, the rest clobbered
        else         bitonic_sort_desc_rt(s, n2);
        have = total < k ? total : k;
        base += take;
    } while (base < n);
    if (tid == 0) n_out[e] = k_out;
    for (int t = tid; t < k; t += THREADS) {
        const bool ok = t < k_out;
        idx_out[(int64_t)e * k + t] = ok ? (int64_t)(s[t] & 0xFFFFFFFFull) : -1;
        sim_out[(int64_t)e * k + t] = ok ? order_bits_inverse((uint32_t)(s[t] >> 32)) : 0.0f;
    }
}

struct ScanPlan {
    int64_t n; int k_eff; bool full_sort; int64_t n_pad;
    size_t off_sims, off_a, off_b, total;
};

// The global ranking behind the per-event scan (hippocampal_memory.py:3275-3277: the hits of all events in one list, sorted by
// similarity, descending -- Python's sort is stable, so equal similarities stay in event order -- and the best `keep` kept).
// One workgroup: a best-64 tournament over the (E, k) keys through a 4096-key window.  Key = (monotone(sim) << 32) | ~position, so
// equal similarities rank by position (event, then rank inside the event) and every key is unique; slots beyond an event's count
// are 0 and never chosen.  NaN similarities rank first, as they do inside an event.
__global__ __launch_bounds__(1024) void rank_segment_hits_kernel(const int64_t* __restrict__ idx, const float* __restrict__ sims,
                                                                 const int32_t* __restrict__ counts, int n_segments, int k, int keep,
                                                                 int64_t* __restrict__ event_out, int64_t* __restrict__ row_out,
                                                                 float* __restrict__ sim_out, int32_t* __restrict__ n_out) {
    __shared__ uint64_t w[kChunk];
    __shared__ int n_cand;
    const int64_t total = (int64_t)n_segments * k;
    auto key_at = [&](int64_t pos) -> uint64_t {
        if (pos < total && (int)(pos % k) < counts[pos / k])
            return ((uint64_t)order_bits(sims[pos]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)pos);
        return 0ull;
    };
    if (total <= (int64_t)1024 * (kChunk / 64)) {
        // Quick route (2000 events x 5 hits: 33 -> ~10 us): every thread's largest key -> the keep-th largest of those 1024 maxima, M,
        // is a lower bound of the keep-th largest key, and since keys are unique exactly `keep` threads hold keys >= M: the at most
        // keep x ceil(total / 1024) <= 4096 keys at or above M contain the answer and are ranked alone.
        uint64_t best = 0ull;
        for (int64_t pos = threadIdx.x; pos < total; pos += 1024) {
            const uint64_t key = key_at(pos);
            best = key > best ? key : best;
        }
        w[threadIdx.x] = best;
        if (threadIdx.x == 0) n_cand = 0;
        __syncthreads();
        top64_desc(w, 1024);
        const uint64_t m = w[keep - 1];                                      // 0: fewer than `keep` threads hold a hit at all
        __syncthreads();
        for (int64_t pos = threadIdx.x; pos < total; pos += 1024) {
            const uint64_t key = key_at(pos);
            if (key != 0ull && key >= m) w[atomicAdd(&n_cand, 1)] = key;     // m == 0: fewer than keep x ceil(total / 1024) hits exist
        }
        __syncthreads();
        const int n2 = pow2_at_least(n_cand, 64);
        for (int t = n_cand + threadIdx.x; t < n2; t += 1024) w[t] = 0ull;
        __syncthreads();
        top64_desc(w, n2);
    } else {
        int64_t base = 0;
        int held = 0;
        do {
            const int take = kChunk - held;
            for (int t = threadIdx.x; t < take; t += 1024) w[held + t] = key_at(base + t);
            base += take;
            held = 64;
            __syncthreads();
            top64_desc(w, kChunk);
        } while (base < total);
    }
    int n = 0;
    for (int t = 0; t < keep; ++t) n += w[t] != 0ull;                        // every thread the same count (keys are sorted, zeros last)
    if (threadIdx.x == 0) *n_out = n;
    for (int t = threadIdx.x; t < keep; t += 1024) {
        const bool valid = t < n;
        const int64_t pos = valid ? (int64_t)(0xFFFFFFFFu - (uint32_t)(w[t] & 0xFFFFFFFFull)) : 0;
        event_out[t] = valid ? pos / k : -1;
        row_out[t] = valid ? idx[pos] : -1;
        sim_out[t] = valid ? sims[pos] : 0.0f;
    }
}

static int64_t next_pow2(int64_t x) { int64_t p = 1; while (p < x) p <<= 1; return p; }

static ScanPlan make_plan(int64_t n, int k) {
    ScanPlan p{};
    p.n = n;
    p.k_eff = (int)((int64_t)k < n ? k : n);
    p.full_sort = (n > kChunk) && (p.k_eff > kFastK);
    p.n_pad = p.full_sort ? next_pow2(n) : 0;
    size_t sims = align_up((size_t)(n > 0 ? n : 1) * sizeof(float), 256);
    size_t a, b;
    if (p.full_sort) {
        a = (size_t)p.n_pad * 8; b = 0;
    } else {
        int64_t blocks1 = (n + kChunk - 1) / kChunk;
        if (p.k_eff <= kFusedK && blocks1 < kScanBlocks) blocks1 = kScanBlocks;   // fused path: k keys per scan block
        int64_t kk = p.k_eff > 0 ? p.k_eff : 1;
        a = (size_t)blocks1 * kk * 8;
        int64_t blocks2 = (blocks1 * kk + kChunk - 1) / kChunk;
        b = (size_t)blocks2 * kk * 8;
    }
    p.off_sims = 0;
    p.off_a = sims;
    p.off_b = p.off_a + align_up(a, 256);
    p.total = p.off_b + align_up(b, 256) + 256;
    return p;
}

struct ScanOut {                 // where the result of run_scan lives
    const uint64_t* best;         // sorted best keys (>= k_eff entries), or nullptr when `fused` is set
    int k_eff;
    bool fused;                   // candidates of the fused scan still need topk_final_kernel
    const uint64_t* cand; int n_blocks; int64_t n_waves;
};

// Runs scan + selection up to (not including) the final decode / key copy.
static int run_scan(const float* store, int64_t n, int dim, const float* query, int k,
                    void* ws, size_t ws_bytes, hipStream_t st, ScanOut* out) {
    const uint64_t** best = &out->best;
    int* k_eff = &out->k_eff;
    out->fused = false;
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "cosine_topk: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(n >= 1 && n < (int64_t)0xFFFFFFFFll, HMM_E_INVALID, "cosine_topk: n_rows=%lld out of range", (long long)n);
    HMM_REQUIRE(k >= 1, HMM_E_INVALID, "cosine_topk: k must be >= 1, got %d", k);
    HMM_REQUIRE(store && query && ws, HMM_E_INVALID, "cosine_topk: null pointer");
    HMM_REQUIRE(((uintptr_t)store & 15) == 0 && ((uintptr_t)query & 15) == 0 && ((uintptr_t)ws & 15) == 0, HMM_E_INVALID,
                "cosine_topk: store, query and workspace must be 16-byte aligned");
    const ScanPlan p = make_plan(n, k);
    HMM_REQUIRE(ws_bytes >= p.total, HMM_E_WORKSPACE, "cosine_topk: workspace %zu < required %zu", ws_bytes, p.total);
    char* base = static_cast<char*>(ws);
    float* sims = reinterpret_cast<float*>(base + p.off_sims);
    uint64_t* buf_a = reinterpret_cast<uint64_t*>(base + p.off_a);
    uint64_t* buf_b = reinterpret_cast<uint64_t*>(base + p.off_b);

    int64_t waves_needed = (n + 1) / 2;
    int blocks = (int)((waves_needed + 3) / 4);
    if (blocks > g_scan_blocks) blocks = g_scan_blocks;
    *k_eff = p.k_eff;

    if (!p.full_sort && n > kChunk && p.k_eff <= kFusedK) {
        // fused: the streaming kernel emits k candidates per block; reduce blocks*k keys by chunk sorts
        scan_topk_kernel<true><<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(store), n,
                                                       reinterpret_cast<const float4*>(query), p.k_eff, buf_a);
        HMM_LAUNCH_CHECK();
        if ((int64_t)p.k_eff * p.k_eff <= kChunk && blocks <= kScanBlocks) {     // one-kernel finish
            out->fused = true; out->cand = buf_a; out->n_blocks = blocks; out->n_waves = (int64_t)blocks * 4;
            out->best = nullptr;
            return HMM_OK;
        }
        int64_t count = (int64_t)blocks * p.k_eff;
        uint64_t* cur = buf_a;
        uint64_t* nxt = buf_b;
        do {
            const int64_t nblk = (count + kChunk - 1) / kChunk;
            topk_chunk_kernel<false><<<(unsigned)nblk, 1024, 0, st>>>(cur, count, p.k_eff, nxt);
            HMM_LAUNCH_CHECK();
            count = nblk * p.k_eff;
            uint64_t* t = cur; cur = nxt; nxt = t;
            if (nblk == 1) break;
        } while (true);
        *best = cur;
        return HMM_OK;
    }

    launch_scan_sims(store, n, query, sims, st);
    HMM_LAUNCH_CHECK();

    if (p.full_sort) {
        const int64_t np = p.n_pad;
        keys_from_sims_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(sims, n, np, buf_a);
        HMM_LAUNCH_CHECK();
        const int64_t half = np / 2;
        for (int64_t kk = 2; kk <= np; kk <<= 1)
            for (int64_t j = kk >> 1; j > 0; j >>= 1)
                bitonic_global_step_kernel<<<(unsigned)((half + 255) / 256), 256, 0, st>>>(buf_a, half, j, kk);
        HMM_LAUNCH_CHECK();
        *best = buf_a;
        return HMM_OK;
    }
    int64_t count = n;
    int64_t nblk = (count + kChunk - 1) / kChunk;
    topk_chunk_kernel<true><<<(unsigned)nblk, 1024, 0, st>>>(sims, count, p.k_eff, buf_a);
    HMM_LAUNCH_CHECK();
    uint64_t* cur = buf_a;
    uint64_t* nxt = buf_b;
    while (nblk > 1) {
        count = nblk * p.k_eff;
        nblk = (count + kChunk - 1) / kChunk;
        topk_chunk_kernel<false><<<(unsigned)nblk, 1024, 0, st>>>(cur, count, p.k_eff, nxt);
        HMM_LAUNCH_CHECK();
        uint64_t* t = cur; cur = nxt; nxt = t;
    }
    *best = cur;
    return HMM_OK;
}

// hmm_cosine_topk executed only when *run_if != 0 on the device, as ONE conditional launch (exact_scan_fallback_kernel); n > 4096 rows
// and k <= 64.  `ticket` must read 0 when the kernel starts (prefilter_final_kernel zeroes it).  ws: hmm_cosine_topk's workspace.
int cosine_topk_if(const int* run_if, unsigned* ticket, const float* store, int64_t n, const float* query, int k, int64_t* idx_out,
                   float* sim_out, int32_t* n_out, void* ws, size_t ws_bytes, hipStream_t st) {
    HMM_REQUIRE(run_if && ticket && n > kChunk && n < (int64_t)0xFFFFFFFFll && k >= 1 && k <= 64 && k <= n, HMM_E_INVALID,
                "cosine_topk_if: needs more than %d rows and k <= 64", kChunk);
    const ScanPlan p = make_plan(n, k);
    HMM_REQUIRE(ws_bytes >= p.total, HMM_E_WORKSPACE, "cosine_topk_if: workspace %zu < required %zu", ws_bytes, p.total);
    int64_t waves_needed = (n + 1) / 2;
    int blocks = (int)((waves_needed + 3) / 4);
    if (blocks > g_scan_blocks) blocks = g_scan_blocks;
    exact_scan_fallback_kernel<<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(store), n, reinterpret_cast<const float4*>(query),
                                                       k, reinterpret_cast<uint64_t*>(static_cast<char*>(ws) + p.off_a), run_if, ticket,
                                                       idx_out, sim_out, n_out);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_cosine_topk_workspace_bytes(int64_t n_rows, int k) {
    if (n_rows < 1 || k < 1) return 0;
    return make_plan(n_rows, k).total;
}

extern "C" int hmm_cosine_topk(const float* store_dev, int64_t n_rows, int dim, const float* query_dev, int k,
                               int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                               void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(idx_out_dev && sim_out_dev, HMM_E_INVALID, "cosine_topk: null output pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ScanOut r{};
    int rc = run_scan(store_dev, n_rows, dim, query_dev, k, workspace_dev, workspace_bytes, st, &r);
    if (rc != HMM_OK) return rc;
    if (r.fused) {
        topk_final_kernel<<<1, 1024, 0, st>>>(r.cand, r.n_blocks, r.k_eff, r.n_waves, nullptr, 0, idx_out_dev, sim_out_dev,
                                              n_out_dev);
    } else {
        decode_kernel<<<(r.k_eff + 255) / 256, 256, 0, st>>>(r.best, r.k_eff, idx_out_dev, sim_out_dev, n_out_dev);
    }
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

extern "C" int hmm_cosine_topk_keys(const float* store_dev, int64_t n_rows, int dim, const float* query_dev, int k,
                                    uint64_t* keys_out_dev, void* workspace_dev, size_t workspace_bytes,
                                    hmm_stream_t stream) {
    HMM_REQUIRE(keys_out_dev, HMM_E_INVALID, "cosine_topk_keys: null output pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ScanOut r{};
    int rc = run_scan(store_dev, n_rows, dim, query_dev, k, workspace_dev, workspace_bytes, st, &r);
    if (rc != HMM_OK) return rc;
    if (r.fused) {
        topk_final_kernel<<<1, 1024, 0, st>>>(r.cand, r.n_blocks, r.k_eff, r.n_waves, keys_out_dev, k, nullptr, nullptr,
                                              nullptr);
    } else {
        copy_keys_kernel<<<(k + 255) / 256, 256, 0, st>>>(r.best, r.k_eff, keys_out_dev, k);
    }
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

extern "C" int hmm_topk_merge_keys(const uint64_t* keys_dev, int n_shards, int k,
                                   const int64_t* shard_row_offset_dev,
                                   int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                   hmm_stream_t stream) {
    HMM_REQUIRE(keys_dev && shard_row_offset_dev && idx_out_dev && sim_out_dev && n_out_dev, HMM_E_INVALID,
                "topk_merge_keys: null pointer");
    HMM_REQUIRE(n_shards >= 1 && k >= 1 && (int64_t)n_shards * k <= kChunk, HMM_E_INVALID,
                "topk_merge_keys: n_shards*k = %lld exceeds %d", (long long)n_shards * k, kChunk);
    merge_keys_kernel<<<1, 256, 0, static_cast<hipStream_t>(stream)>>>(keys_dev, n_shards, k, shard_row_offset_dev,
                                                                       idx_out_dev, sim_out_dev, n_out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// Timing / test hook (include/hippomm_hip.h): the plain similarity pass alone.
extern "C" int hmm_op_scan_sims(const float* store_dev, int64_t n_rows, const float* query_dev, float* sims_dev,
                                 hmm_stream_t stream) {
    HMM_REQUIRE(((uintptr_t)sims_dev & 15) == 0, HMM_E_INVALID, "scan_sims: sims_dev must be 16-byte aligned");
    launch_scan_sims(store_dev, n_rows, query_dev, sims_dev, static_cast<hipStream_t>(stream));
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// Timing hook: the fused streaming + block-top-k kernel alone (the dominant kernel of a query).
extern "C" int hmm_op_scan_topk_only(const float* store_dev, int64_t n_rows, const float* query_dev, int k,
                                      uint64_t* cand_dev /* [2048*k] */, hmm_stream_t stream) {
    HMM_REQUIRE(k >= 1 && k <= kFusedK, HMM_E_INVALID, "scan_topk_only: k out of range");
    int64_t waves_needed = (n_rows + 1) / 2;
    int blocks = (int)((waves_needed + 3) / 4);
    if (blocks > g_scan_blocks) blocks = g_scan_blocks;
    scan_topk_kernel<true><<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float4*>(store_dev), n_rows, reinterpret_cast<const float4*>(query_dev), k, cand_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

extern "C" size_t hmm_cosine_topk_segmented_workspace_bytes(int64_t n_rows, int n_segments, int k) {
    if (n_rows < 0 || n_segments < 1 || k < 1) return 0;
    return align_up((size_t)(n_rows > 0 ? n_rows : 1) * sizeof(float), 256) + 256;
}

extern "C" int hmm_cosine_topk_segmented(const float* store_dev, int64_t n_rows, int dim, const float* query_dev,
                                         const int64_t* seg_offsets_dev, int n_segments, int k,
                                         int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                         void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "cosine_topk_segmented: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0xFFFFFFFFll, HMM_E_INVALID, "cosine_topk_segmented: n_rows out of range");
    HMM_REQUIRE(n_segments >= 1 && k >= 1 && k <= kFastK, HMM_E_INVALID,
                "cosine_topk_segmented: need n_segments >= 1 and 1 <= k <= %d", kFastK);
    HMM_REQUIRE(query_dev && seg_offsets_dev && idx_out_dev && sim_out_dev && n_out_dev && workspace_dev, HMM_E_INVALID,
                "cosine_topk_segmented: null pointer");
    HMM_REQUIRE(n_rows == 0 || store_dev, HMM_E_INVALID, "cosine_topk_segmented: null store");
    HMM_REQUIRE(workspace_bytes >= hmm_cosine_topk_segmented_workspace_bytes(n_rows, n_segments, k), HMM_E_WORKSPACE,
                "cosine_topk_segmented: workspace too small");
    HMM_REQUIRE(((uintptr_t)workspace_dev & 15) == 0, HMM_E_INVALID, "cosine_topk_segmented: workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* sims = static_cast<float*>(workspace_dev);
    if (n_rows > 0) {
        launch_scan_sims(store_dev, n_rows, query_dev, sims, st);
        HMM_LAUNCH_CHECK();
    }
    if (segments_are_small(n_rows, n_segments, k))
        segment_topk_kernel<kSmallSegChunk, 256><<<n_segments, 256, 0, st>>>(sims, seg_offsets_dev, k, idx_out_dev, sim_out_dev, n_out_dev);
    else
        segment_topk_kernel<kChunk, 1024><<<n_segments, 1024, 0, st>>>(sims, seg_offsets_dev, k, idx_out_dev, sim_out_dev, n_out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

extern "C" int hmm_rank_segment_hits(const int64_t* idx_dev, const float* sims_dev, const int32_t* counts_dev, int n_segments, int k,
                                     int keep, int64_t* event_out_dev, int64_t* row_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                     hmm_stream_t stream) {
    HMM_REQUIRE(idx_dev && sims_dev && counts_dev && event_out_dev && row_out_dev && sim_out_dev && n_out_dev, HMM_E_INVALID,
                "rank_segment_hits: null pointer");
    HMM_REQUIRE(n_segments >= 1 && k >= 1 && keep >= 1 && keep <= 64 && (int64_t)n_segments * k < (int64_t)0xFFFFFFFFll, HMM_E_INVALID,
                "rank_segment_hits: need n_segments >= 1, k >= 1, 1 <= keep <= 64 and fewer than 2^32 hits (got %d x %d, keep %d)",
                n_segments, k, keep);
    rank_segment_hits_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(idx_dev, sims_dev, counts_dev, n_segments, k, keep,
                                                                              event_out_dev, row_out_dev, sim_out_dev, n_out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
