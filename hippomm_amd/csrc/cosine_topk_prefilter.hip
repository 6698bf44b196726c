// feature_search scan through a bf16 SHADOW of the store, exact by construction ("if a bf16 shadow store is also
// offered, report it separately against 2 048 000 000 B").  Same result as hmm_cosine_topk -- the same rows, the same fp32
// similarities bit for bit (reference hippomm/utils/vector_ops.py:178-186) -- for half the bytes streamed:
//
//   shadow   row r = bf16(x_r / ||x_r||)  (round to nearest even; a zero-norm or non-finite row becomes NaN), 2048 B per row,
//            built once per store by hmm_shadow_store_build;
//   pass 1   prefilter_topk_kernel streams the shadow: s~_r = dot(shadow_r, q) / ||q|| in fp32, block-local top-k of the s~ keys
//            exactly as scan_topk_kernel keeps them (cosine_topk.hip);
//   pass 2   prefilter_final_kernel (one workgroup): t = the k-th largest s~ over all blocks (or, when that already yields few
//            candidates, its lower bound the k-th largest block maximum); every row whose exact similarity
//            can be among the k largest has s~_r >= t - 2 eps (below); those rows are re-scored on the fp32 store with the
//            arithmetic of scan_topk_kernel (exact_row_sim: same loads, same fma order, same wave reduction, same division) and
//            the k best of them are the answer.
//   Error bound.  bf16 has an 8-bit significand: x~ = x (1 + d), |d| <= 2^-8, so |dot(x~, q) - dot(x, q)| <= 2^-8 sum |x_i q_i|
//            <= 2^-8 ||x|| ||q||, i.e. |s~ - s| <= 2^-8 (1 + 2^-8) + fp32 summation noise (~3e-6 for both kernels) < eps = 0.0040.
//            If row r is in the exact top-k then s_r >= the k-th largest s >= t - eps (k rows have s~ >= t, so s >= t - eps), hence
//            s~_r >= t - 2 eps: r is a candidate.  A NaN s~ (zero-norm row, NaN query) ranks first, as NaN similarities do.
//   Fallback.  The candidate set is complete only if no block's list is saturated above the threshold (a list keeps max(2k, 32) <= 64
//            entries; its LAST entry >= t - 2 eps means the block may have dropped candidates) and fits the re-scoring buffer; otherwise pass 2 raises a flag and the
//            exact scan (scan_topk_kernel + topk_final_kernel, conditional on that flag) produces the answer.  Either way the
//            outputs are those of hmm_cosine_topk.  Stores of many near-ties (thousands of rows within 0.8 % of the k-th best)
//            take the fallback: 0.3 ms wasted; random or video-like stores do not.
#include "hmm_common.h"
#include "topk_tournament.h"
#include "cosine_topk_shared.h"

namespace hmm {

// Workgroups of the bf16 passes (see g_scan_blocks in cosine_topk.hip): the shadow pass does more arithmetic per byte and wants more
// waves than the fp32 scan, 768 workgroups: 0.3327 -> 0.3199 ms per query (640: 0.3202, 512: 0.328, 384: 0.390; profiles/r6_scan_blocks.json)
HMM_TUNABLE(int, g_prefilter_blocks, kNumCU * 3)       // prefilter_topk_kernel
#ifdef HMM_PROBE
HMM_TUNABLE(int, g_prefilter_sims_blocks, kScanBlocks) // prefilter_sims_kernel (probe build only)
HMM_TUNABLE(int, g_prefilter_sims_deferred, 1)         // 0: prefilter_sims_kernel instead of prefilter_sims_deferred_kernel (probe build only)
#endif
constexpr float kPrefilterEps = 0.0040f;
constexpr int kPrefilterCap = 1024;           // candidate rows pass 2 re-scores itself (16 waves)
constexpr int kPrefilterMaxK = 64;

// ---- shadow build: one wave per row ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadow_build_kernel(const float4* __restrict__ store, int64_t n_rows,
                                                           uint4* __restrict__ shadow) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        // lane l owns elements 8 l .. 8 l + 7 and 512 + 8 l .. + 7 (two 32-B pieces in, two 16-B pieces out)
        const float4* p = store + r * 256;
        float4 v[4] = {p[2 * lane], p[2 * lane + 1], p[128 + 2 * lane], p[128 + 2 * lane + 1]};
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ss = fmaf(v[i].x, v[i].x, ss); ss = fmaf(v[i].y, v[i].y, ss);
            ss = fmaf(v[i].z, v[i].z, ss); ss = fmaf(v[i].w, v[i].w, ss);
        }
        ss = wave_sum(ss);
        float inv = 1.0f / sqrtf(ss);
        if (!(ss > 0.f) || !(ss < INFINITY)) inv = __uint_as_float(0x7FC00000u);      // zero / non-finite norm: the row is NaN
        uint4 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = v[2 * h], b = v[2 * h + 1];
            bf16x8 t = {(bf16_t)(a.x * inv), (bf16_t)(a.y * inv), (bf16_t)(a.z * inv), (bf16_t)(a.w * inv),
                        (bf16_t)(b.x * inv), (bf16_t)(b.y * inv), (bf16_t)(b.z * inv), (bf16_t)(b.w * inv)};
            o[h] = __builtin_bit_cast(uint4, t);
        }
        uint4* q = shadow + r * 128;
        q[lane] = o[0];
        q[64 + lane] = o[1];
    }
}

__device__ __forceinline__ float dot8_bf16(const uint4& x, const float4& qa, const float4& qb, float acc) {
    // bf16 -> fp32 is a 16-bit shift: low half << 16, high half masked
    acc = fmaf(__uint_as_float(x.x << 16), qa.x, acc); acc = fmaf(__uint_as_float(x.x & 0xFFFF0000u), qa.y, acc);
    acc = fmaf(__uint_as_float(x.y << 16), qa.z, acc); acc = fmaf(__uint_as_float(x.y & 0xFFFF0000u), qa.w, acc);
    acc = fmaf(__uint_as_float(x.z << 16), qb.x, acc); acc = fmaf(__uint_as_float(x.z & 0xFFFF0000u), qb.y, acc);
    acc = fmaf(__uint_as_float(x.w << 16), qb.z, acc); acc = fmaf(__uint_as_float(x.w & 0xFFFF0000u), qb.w, acc);
    return acc;
}

// ---- pass 1: stream the shadow, block-local top-k of the approximate keys ------------------------------------------------------
// Rows are dealt four at a time, wave after wave (row r belongs to wave (r / 4) % n_waves): a block reads 32 KiB contiguous per
// iteration and takes 16 consecutive rows of every band of 32 768, so a run of neighbouring rows -- the near-identical frames of
// one scene of a video -- puts at most 16 candidates into one block's list, which keeps at least 32 entries (prefilter_list_len):
// contiguous scenes of any length do not saturate a list.  (Spreading consecutive groups over different blocks, or dealing one
// row per wave, streams 12 % slower -- four distant pieces per block and iteration instead of one run -- for no fewer fallbacks.)
__global__ __launch_bounds__(256) void prefilter_topk_kernel(const uint4* __restrict__ shadow, int64_t n_rows,
                                                             const float4* __restrict__ query, int k,
                                                             uint64_t* __restrict__ out, uint64_t* __restrict__ maxima) {
    __shared__ uint64_t cand[kFusedCap];
    __shared__ int count;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    if (threadIdx.x == 0) count = 0;
    // query elements of this lane, in the shadow's piece order, and 1 / ||q|| exactly as the exact kernels take ||q||
    float4 q[4];
    float qs = 0.f;
    {
        float4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = query[j * 64 + lane];
            qs = fmaf(t[j].x, t[j].x, qs); qs = fmaf(t[j].y, t[j].y, qs);
            qs = fmaf(t[j].z, t[j].z, qs); qs = fmaf(t[j].w, t[j].w, qs);
        }
        q[0] = query[2 * lane]; q[1] = query[2 * lane + 1]; q[2] = query[128 + 2 * lane]; q[3] = query[128 + 2 * lane + 1];
    }
    const float inv_qlen = 1.0f / sqrtf(wave_sum(qs));
    __syncthreads();

    const int64_t iters = (n_rows + n_waves * 4 - 1) / (n_waves * 4);
    const int compact_every = (kFusedCap - k) / 16;                          // 16 rows per block per iteration
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t r0 = wave * 4 + it * n_waves * 4;                      // this wave's group of four rows
        const int have = r0 >= n_rows ? 0 : (n_rows - r0 >= 4 ? 4 : (int)(n_rows - r0));             // wave-uniform
        if (have) {
            uint4 x[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4* p = shadow + (r0 + (i < have ? i : 0)) * 128 + lane;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 64));
                x[i][0] = make_uint4(a[0], a[1], a[2], a[3]);
                x[i][1] = make_uint4(b[0], b[1], b[2], b[3]);
            }
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = dot8_bf16(x[i][1], q[2], q[3], dot8_bf16(x[i][0], q[0], q[1], 0.f));
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = wave_sum(d[i]);
            if (lane == 0) {
                const int pos = atomicAdd(&count, have);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < have) cand[pos + i] = ((uint64_t)order_bits(d[i] * inv_qlen) << 32) | (uint64_t)(uint32_t)(r0 + i);
            }
        }
        if ((it + 1) % compact_every == 0 || it + 1 == iters) {              // block-uniform
            __syncthreads();
            const int n = count;
            const int n2 = pow2_at_least(n, 64);
            for (int t = n + threadIdx.x; t < n2; t += 256) cand[t] = 0ull;
            __syncthreads();
            top64_desc<false>(cand, n2);                                     // k <= 64 on this path
            if (threadIdx.x == 0) count = n < k ? n : k;
            __syncthreads();
        }
    }
    const int n = count;
    for (int t = threadIdx.x; t < k; t += 256) out[(int64_t)blockIdx.x * k + t] = t < n ? cand[t] : 0ull;
    if (threadIdx.x == 0) maxima[blockIdx.x] = n > 0 ? cand[0] : 0ull;       // the block maxima once more, contiguous: pass 2's first read
}

// Probe build only: thread 0 of pass 2 stamps the 100-MHz wall clock after each phase (tools/prefilter_final_stamps_probe.py).
#ifdef HMM_PROBE
static long long* g_pf_stamps = nullptr;
extern "C" void hmm_probe_set_prefilter_stamps(long long* stamps_dev) { g_pf_stamps = stamps_dev; }
#define HMM_PF_STAMP_PARAM , long long* __restrict__ stamps
#define HMM_PF_STAMP_ARG , g_pf_stamps
#define HMM_PF_STAMP(i) do { if (stamps != nullptr && threadIdx.x == 0) stamps[i] = wall_clock64(); } while (0)
#else
#define HMM_PF_STAMP_PARAM
#define HMM_PF_STAMP_ARG
#define HMM_PF_STAMP(i)
#endif

// ---- pass 2: threshold, candidates, exact re-score, answer (or the fallback flag) -------------------------------------------
// lists: n_blocks x kk keys (kk >= k entries per block, sorted descending, 0-padded).
__global__ __launch_bounds__(1024) void prefilter_final_kernel(const uint64_t* __restrict__ lists, const uint64_t* __restrict__ maxima,
                                                               unsigned* __restrict__ fallback_ticket, int n_blocks, int k, int kk,
                                                               int64_t n_waves, const float4* __restrict__ store,
                                                               const float4* __restrict__ query,
                                                               int64_t* __restrict__ idx_out, float* __restrict__ sim_out,
                                                               int32_t* __restrict__ n_out, int* __restrict__ fallback,
                                                               int* __restrict__ stats /* [2]: candidates, saturated lists */
                                                               HMM_PF_STAMP_PARAM) {
    __shared__ uint64_t mx[kScanBlocks];
    __shared__ uint64_t s[kChunk];
    __shared__ uint32_t cand_row[kPrefilterCap];
    __shared__ uint16_t hot[kScanBlocks];
    __shared__ int n_cand, n_sat, n_hot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    HMM_PF_STAMP(0);
    if (tid == 0) { n_cand = 0; n_sat = 0; n_hot = 0; *fallback_ticket = 0u; }          // the ticket of the conditional exact scan behind this kernel
    const int n2 = pow2_at_least(n_blocks, 64);
    uint64_t own[kScanBlocks / 1024];                                         // this thread's block maxima, kept for the hot-list pass
#pragma unroll
    for (int i = 0; i < kScanBlocks / 1024; ++i) {
        const int t = tid + i * 1024;
        own[i] = t < n_blocks ? maxima[t] : 0ull;
        if (t < n2) mx[t] = own[i];
    }
    __syncthreads();
    HMM_PF_STAMP(1);                                                          // maxima loaded
    top64_desc(mx, n2);
    HMM_PF_STAMP(2);                                                          // ... ranked
    HMM_PF_STAMP(3);                                                          // (3, 4: overwritten on the exact route)
    HMM_PF_STAMP(4);
    // Candidates under a threshold: the lists whose maximum reaches it, one wave per such list, one entry per lane; every entry at
    // or above it is a candidate, a list whose LAST entry passes may have dropped some (saturated).  A wave loads its lists of a
    // round (up to 8) before it looks at any of them: one memory latency per 128 hot lists instead of one per 16.
    auto collect = [&](uint32_t thr) {
#pragma unroll
        for (int i = 0; i < kScanBlocks / 1024; ++i)
            if (own[i] != 0ull && (uint32_t)(own[i] >> 32) >= thr) hot[atomicAdd(&n_hot, 1)] = (uint16_t)(tid + i * 1024);
        __syncthreads();
        constexpr int kHotUnroll = 8;
        for (int h0 = wave; h0 < n_hot; h0 += 16 * kHotUnroll) {
            uint64_t keys[kHotUnroll];
#pragma unroll
            for (int u = 0; u < kHotUnroll; ++u) {
                const int h = h0 + 16 * u;
                keys[u] = (h < n_hot && lane < kk) ? lists[(int64_t)hot[h] * kk + lane] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kHotUnroll; ++u) {
                if (h0 + 16 * u >= n_hot) break;                              // wave-uniform
                const uint64_t key = keys[u];
                const bool pass = key != 0ull && (uint32_t)(key >> 32) >= thr;
                const unsigned long long mask = __ballot(pass);
                int base = 0;
                if (lane == 0) base = atomicAdd(&n_cand, __popcll(mask));
                base = __shfl(base, 0, 64);
                if (pass) {
                    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < kPrefilterCap) cand_row[pos] = (uint32_t)(key & 0xFFFFFFFFull);
                    if (lane == kk - 1) atomicAdd(&n_sat, 1);
                }
            }
        }
        __syncthreads();
    };
    auto threshold_below = [](uint64_t key) {                                 // NaN key -> 0xFFFFFFFF: only NaN rows pass
        return order_bits(order_bits_inverse((uint32_t)(key >> 32)) - 2.0f * kPrefilterEps);
    };
    // QUICK route: the k-th largest block MAXIMUM is a lower bound of the k-th largest key (k lists hold a key that large), and it
    // IS that key whenever the k best rows sit in k different lists -- the rule on a store without long runs of near-identical
    // rows (1M random rows, k = 32: always).  Its candidate set contains the exact one; when it is small and no list is saturated
    // the second ranking (gather the winners' lists, rank 2048 keys: 6 us) is skipped.
    constexpr int kQuickCap = 128;
    uint64_t kth = mx[k - 1];                                                 // k <= 64: inside the ranked front of mx
    bool quick = false;
    if (kth != 0ull) {                                                        // block-uniform (mx is shared)
        collect(threshold_below(kth));
        quick = n_sat == 0 && n_cand <= kQuickCap;
    }
    if (!quick) {
        // EXACT route: the k lists with the largest maxima hold the k largest keys (every list keeps kk >= k entries): rank them.
        // (When the k best rows sit in fewer than k blocks, as the frames of one scene do, the block-maximum bound falls to the
        // background level and thousands of rows pass it: a third of the queries of a scene-structured store would take the
        // exact-scan fallback on that bound alone.)
        __syncthreads();
        if (tid == 0) { n_cand = 0; n_sat = 0; n_hot = 0; }
        const int n_win = n_blocks < k ? n_blocks : k;
        const int m2 = pow2_at_least(n_win * kk, 64);
        for (int t = tid; t < m2; t += 1024) {
            uint64_t key = 0ull;
            if (t < n_win * kk) {
                const uint64_t top = mx[t / kk];
                if (top != 0ull) {
                    const int64_t row = (int64_t)(top & 0xFFFFFFFFull);
                    const int blk = (int)(((row >> 2) % n_waves) >> 2);       // rows are dealt four per wave
                    key = lists[(int64_t)blk * kk + (t % kk)];
                }
            }
            s[t] = key;
        }
        __syncthreads();
        HMM_PF_STAMP(3);                                                      // winners' lists gathered
        top64_desc(s, m2);
        HMM_PF_STAMP(4);                                                      // ... ranked: the k-th largest approximate key
        kth = s[k - 1];                                                       // 0 = fewer than k rows in all (launcher excludes it)
        __syncthreads();
        collect(threshold_below(kth));
    }
    HMM_PF_STAMP(5);                                                          // hot lists read, candidates listed
    const int m = n_cand;
    const bool fall = kth == 0ull || n_sat > 0 || m > kPrefilterCap;
    if (tid == 0) {
        *fallback = fall ? 1 : 0;
        if (stats) { stats[0] = m; stats[1] = n_sat; }
    }
    if (fall) return;                                                         // block-uniform: the exact scan answers
    // exact re-score: one wave per candidate row, the arithmetic of scan_topk_kernel
    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = query[j * 64 + lane];
        qs = fmaf(q[j].x, q[j].x, qs); qs = fmaf(q[j].y, q[j].y, qs);
        qs = fmaf(q[j].z, q[j].z, qs); qs = fmaf(q[j].w, q[j].w, qs);
    }
    const float q_len = sqrtf(wave_sum(qs));
    const int m_pad = pow2_at_least(m, 64);
    for (int t = m + tid; t < m_pad; t += 1024) s[t] = 0ull;
    for (int c0 = wave * 4; c0 < m; c0 += 64) {                               // four candidates per wave and round: one latency per round
        const int nr = m - c0 < 4 ? m - c0 : 4;                               // wave-uniform
        const float4* rp[4];
        uint32_t row[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row[i] = cand_row[c0 + (i < nr ? i : 0)];
            rp[i] = store + (int64_t)row[i] * 256 + lane;
        }
        float sim[4];
        exact_row_sim4(rp, nr, q, q_len, sim);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nr) s[c0 + i] = ((uint64_t)order_bits(sim[i]) << 32) | (uint64_t)row[i];
        }
    }
    __syncthreads();
    HMM_PF_STAMP(6);                                                          // candidates re-scored
    top64_desc(s, m_pad);
    HMM_PF_STAMP(7);
    if (tid == 0 && n_out) *n_out = k;
    for (int t = tid; t < k; t += 1024) {
        idx_out[t] = (int64_t)(s[t] & 0xFFFFFFFFull);
        sim_out[t] = order_bits_inverse((uint32_t)(s[t] >> 32));
    }
}

// Entries a block keeps: at least twice k and at least 32, so that a handful of near-ties inside one block (k = 1: ANY second
// candidate) does not saturate its list; 64 at most (the tournament's width).
static int prefilter_list_len(int k) {
    int kk = 2 * k < 32 ? 32 : 2 * k;
    return kk > kPrefilterMaxK ? kPrefilterMaxK : kk;
}

// ---- per-event variant (hmm_cosine_topk_segmented through the shadow) -------------------------------------------------------
#ifdef HMM_PROBE   // the pass until round 6, kept in the probe build as what prefilter_sims_deferred_kernel is measured against
// pass 1: s~ of every row, written as whole 128-B lines (a wave takes 32 consecutive rows, four at a time)
__global__ __launch_bounds__(256) void prefilter_sims_kernel(const uint4* __restrict__ shadow, int64_t n_rows,
                                                             const float4* __restrict__ query, float* __restrict__ sims) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = query[j * 64 + lane];
        qs = fmaf(t.x, t.x, qs); qs = fmaf(t.y, t.y, qs); qs = fmaf(t.z, t.z, qs); qs = fmaf(t.w, t.w, qs);
    }
    q[0] = query[2 * lane]; q[1] = query[2 * lane + 1]; q[2] = query[128 + 2 * lane]; q[3] = query[128 + 2 * lane + 1];
    const float inv_qlen = 1.0f / sqrtf(wave_sum(qs));
    for (int64_t base = wave * 32; base < n_rows; base += n_waves * 32) {
        float mine = 0.f;                                         // lane l < 32: s~ of row base + l
        const int64_t left = n_rows - base;
        const int steps = left >= 32 ? 8 : (int)((left + 3) >> 2);           // wave-uniform
        for (int jj = 0; jj < steps; ++jj) {
            const int j = steps == 8 ? ((jj + (int)(wave & 7)) & 7) : jj;    // not in step with every other wave: see scan_sims_kernel
            const int64_t r = base + 4 * j;
            uint4 x[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4* p = shadow + (r + i < n_rows ? r + i : r) * 128 + lane;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 64));
                x[i][0] = make_uint4(a[0], a[1], a[2], a[3]);
                x[i][1] = make_uint4(b[0], b[1], b[2], b[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = wave_sum(dot8_bf16(x[i][1], q[2], q[3], dot8_bf16(x[i][0], q[0], q[1], 0.f))) * inv_qlen;
                mine = lane == 4 * j + i ? d : mine;
            }
        }
        if (lane < 32 && base + lane < n_rows) sims[base + lane] = mine;
    }
}
#endif  // HMM_PROBE

// pass 1: s~ of every row, with prefilter_topk_kernel's row dealing and no store in its loop (scan_sims_deferred_kernel's reason:
// stores mixed into the read stream cost more than they weigh): lane l of four registers keeps the four results of iteration 64 j + l, one burst
// of 16-byte stores per 256 iterations.  Same arithmetic per row, same bits.
constexpr int kPrefilterSimsHeld = 4;
__global__ __launch_bounds__(256) void prefilter_sims_deferred_kernel(const uint4* __restrict__ shadow, int64_t n_rows,
                                                                      const float4* __restrict__ query, float* __restrict__ sims) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = query[j * 64 + lane];
        qs = fmaf(t.x, t.x, qs); qs = fmaf(t.y, t.y, qs); qs = fmaf(t.z, t.z, qs); qs = fmaf(t.w, t.w, qs);
    }
    q[0] = query[2 * lane]; q[1] = query[2 * lane + 1]; q[2] = query[128 + 2 * lane]; q[3] = query[128 + 2 * lane + 1];
    const float inv_qlen = 1.0f / sqrtf(wave_sum(qs));
    for (int64_t it0 = 0; (wave + it0 * n_waves) * 4 < n_rows; it0 += 64 * kPrefilterSimsHeld) {
        float h[kPrefilterSimsHeld][4];
#pragma unroll
        for (int j = 0; j < kPrefilterSimsHeld; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) h[j][i] = 0.f;
#pragma unroll
        for (int j = 0; j < kPrefilterSimsHeld; ++j) {
            for (int l = 0; l < 64; ++l) {
                const int64_t r0 = (wave + (it0 + j * 64 + l) * n_waves) * 4;
                if (r0 >= n_rows) break;                                     // wave-uniform; every later iteration is past the end too
                uint4 x[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4* p = shadow + (r0 + i < n_rows ? r0 + i : r0) * 128 + lane;
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                    const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 64));
                    x[i][0] = make_uint4(a[0], a[1], a[2], a[3]);
                    x[i][1] = make_uint4(b[0], b[1], b[2], b[3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = wave_sum(dot8_bf16(x[i][1], q[2], q[3], dot8_bf16(x[i][0], q[0], q[1], 0.f))) * inv_qlen;
                    h[j][i] = lane == l ? d : h[j][i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kPrefilterSimsHeld; ++j) {
            const int64_t r0 = (wave + (it0 + j * 64 + lane) * n_waves) * 4;
            if (r0 + 3 < n_rows) {
                *reinterpret_cast<float4*>(sims + r0) = make_float4(h[j][0], h[j][1], h[j][2], h[j][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + i < n_rows) sims[r0 + i] = h[j][i];
            }
        }
    }
}

// pass 2: one workgroup per event.  The event's k-th largest s~ gives the threshold, the rows at or above it are re-scored on the
// fp32 store (exact_row_sim) and the k best of those are the event's answer -- what segment_topk_kernel returns on the exact
// similarities.  More candidates than the buffer holds (an event of near-identical rows): every row of the event is re-scored.
constexpr int kSegCandCap = 1024;
template <int CHUNK, int THREADS>
__global__ __launch_bounds__(THREADS) void segment_prefilter_kernel(const float* __restrict__ sims, const int64_t* __restrict__ seg_off,
                                                                 int k, const float4* __restrict__ store,
                                                                 const float4* __restrict__ query, int64_t* __restrict__ idx_out,
                                                                 float* __restrict__ sim_out, int32_t* __restrict__ n_out) {
    __shared__ uint64_t s[CHUNK];
    __shared__ uint32_t cand[kSegCandCap];
    __shared__ int n_cand;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = seg_off[e], hi = seg_off[e + 1];
    const int64_t n = hi - lo;
    const int k_out = (int)(n < k ? (n > 0 ? n : 0) : k);
    if (tid == 0) n_cand = 0;
    if (n <= 0) {                                                  // block-uniform
        if (tid == 0) n_out[e] = 0;
        for (int t = tid; t < k; t += THREADS) { idx_out[(int64_t)e * k + t] = -1; sim_out[(int64_t)e * k + t] = 0.0f; }
        return;
    }
    // ---- the k-th largest approximate key of the event (pieces of a chunk, carrying the running best k) ---------------------
    uint32_t thr = 0u;                                             // n <= k: every row is a candidate
    if (n > k) {
        int have = 0;
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
            top64_desc(s, n2);
            have = total < k ? total : k;
            base += take;
        } while (base < n);
        const float t_lo = order_bits_inverse((uint32_t)(s[k - 1] >> 32)) - 2.0f * kPrefilterEps;
        thr = order_bits(t_lo);
    }
    __syncthreads();
    // ---- candidates ---------------------------------------------------------------------------------------------------------
    for (int64_t r = tid; r < n; r += THREADS) {
        if (order_bits(sims[lo + r]) >= thr) {
            const int pos = atomicAdd(&n_cand, 1);
            if (pos < kSegCandCap) cand[pos] = (uint32_t)r;
        }
    }
    __syncthreads();
    const bool all_rows = n_cand > kSegCandCap;                    // block-uniform
    const int64_t m = all_rows ? n : (int64_t)n_cand;
    // ---- exact re-score, k best (pieces of a chunk with carry, as above) ------------------------------------------------------
    float4 q[4];
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = query[j * 64 + lane];
        qs = fmaf(q[j].x, q[j].x, qs); qs = fmaf(q[j].y, q[j].y, qs);
        qs = fmaf(q[j].z, q[j].z, qs); qs = fmaf(q[j].w, q[j].w, qs);
    }
    const float q_len = sqrtf(wave_sum(qs));
    int have = 0;
    int64_t base = 0;
    do {
        const int64_t left = m - base;
        const int take = (int)(left < (int64_t)(CHUNK - have) ? left : (int64_t)(CHUNK - have));
        const int total = have + take;
        const int n2 = pow2_at_least(total, 64);
        for (int t = total + tid; t < n2; t += THREADS) s[t] = 0ull;
        for (int c = wave; c < take; c += THREADS / 64) {                    // wave-uniform trip count per wave
            const int64_t r = all_rows ? base + c : (int64_t)cand[base + c];
            const float sim = exact_row_sim(store + (lo + r) * 256 + lane, q, q_len);
            if (lane == 0) s[have + c] = ((uint64_t)order_bits(sim) << 32) | (uint64_t)(uint32_t)r;
        }
        __syncthreads();
        top64_desc(s, n2);
        have = total < k ? total : k;
        base += take;
    } while (base < m);
    if (tid == 0) n_out[e] = k_out;
    for (int t = tid; t < k; t += THREADS) {
        const bool ok = t < k_out;
        idx_out[(int64_t)e * k + t] = ok ? (int64_t)(s[t] & 0xFFFFFFFFull) : -1;
        sim_out[(int64_t)e * k + t] = ok ? order_bits_inverse((uint32_t)(s[t] >> 32)) : 0.0f;
    }
}

struct PrefilterPlan { size_t off_exact, off_lists, off_maxima, off_flag, total; int blocks; };

static PrefilterPlan prefilter_plan(int64_t n, int k) {
    PrefilterPlan p{};
    p.off_exact = 0;
    const size_t exact = hmm_cosine_topk_workspace_bytes(n, k);
    p.off_lists = align_up(exact, 256);
    int64_t waves_needed = (n + 3) / 4;
    int blocks = (int)((waves_needed + 3) / 4);
    p.blocks = blocks > g_prefilter_blocks ? g_prefilter_blocks : blocks;
    p.off_maxima = p.off_lists + align_up((size_t)kScanBlocks * (size_t)kPrefilterMaxK * 8, 256);
    p.off_flag = p.off_maxima + align_up((size_t)kScanBlocks * 8, 256);          // int flag, then the fallback's unsigned ticket
    p.total = p.off_flag + 256;
    return p;
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_shadow_store_bytes(int64_t n_rows) {
    return n_rows < 1 ? 0 : (size_t)n_rows * 2048;
}

extern "C" int hmm_shadow_store_build(const float* store_dev, int64_t n_rows, int dim, void* shadow_dev, size_t shadow_bytes,
                                      hmm_stream_t stream) {
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "shadow_store_build: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(store_dev && shadow_dev && n_rows >= 1, HMM_E_INVALID, "shadow_store_build: bad arguments");
    HMM_REQUIRE(((uintptr_t)store_dev & 15) == 0 && ((uintptr_t)shadow_dev & 15) == 0, HMM_E_INVALID,
                "shadow_store_build: store / shadow must be 16-byte aligned");
    HMM_REQUIRE(shadow_bytes >= hmm_shadow_store_bytes(n_rows), HMM_E_WORKSPACE, "shadow_store_build: shadow buffer %zu < %zu",
                shadow_bytes, hmm_shadow_store_bytes(n_rows));
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > kScanBlocks * 4) blocks = kScanBlocks * 4;
    shadow_build_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float4*>(store_dev), n_rows, static_cast<uint4*>(shadow_dev));
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

extern "C" size_t hmm_cosine_topk_prefilter_workspace_bytes(int64_t n_rows, int k) {
    if (n_rows < 1 || k < 1) return 0;
    return prefilter_plan(n_rows, k).total;
}

extern "C" int hmm_cosine_topk_prefilter(const float* store_dev, const void* shadow_dev, int64_t n_rows, int dim,
                                         const float* query_dev, int k, int64_t* idx_out_dev, float* sim_out_dev,
                                         int32_t* n_out_dev, int32_t* stats_out_dev, void* workspace_dev, size_t workspace_bytes,
                                         hmm_stream_t stream) {
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "cosine_topk_prefilter: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(store_dev && shadow_dev && query_dev && idx_out_dev && sim_out_dev && workspace_dev, HMM_E_INVALID,
                "cosine_topk_prefilter: null pointer");
    HMM_REQUIRE(k >= 1, HMM_E_INVALID, "cosine_topk_prefilter: k must be >= 1, got %d", k);
    HMM_REQUIRE(n_rows >= 1 && n_rows < (int64_t)0xFFFFFFFFll, HMM_E_INVALID, "cosine_topk_prefilter: n_rows=%lld out of range",
                (long long)n_rows);
    HMM_REQUIRE(((uintptr_t)store_dev & 15) == 0 && ((uintptr_t)shadow_dev & 15) == 0 && ((uintptr_t)query_dev & 15) == 0,
                HMM_E_INVALID, "cosine_topk_prefilter: store / shadow / query must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PrefilterPlan p = prefilter_plan(n_rows, k);
    HMM_REQUIRE(workspace_bytes >= p.total, HMM_E_WORKSPACE, "cosine_topk_prefilter: workspace %zu < required %zu", workspace_bytes,
                p.total);
    char* ws = static_cast<char*>(workspace_dev);
    // Small stores, large k: the prefilter has nothing to win (or no list machinery): the exact scan is the whole call.
    if (k > kPrefilterMaxK || n_rows < (int64_t)kChunk * 4 || n_rows <= k) {
        if (stats_out_dev) HMM_HIP_CHECK(hipMemsetAsync(stats_out_dev, 0xFF, 2 * sizeof(int32_t), st));      // -1, -1: not used
        return hmm_cosine_topk(store_dev, n_rows, dim, query_dev, k, idx_out_dev, sim_out_dev, n_out_dev, ws + p.off_exact,
                               p.off_lists, stream);
    }
    uint64_t* lists = reinterpret_cast<uint64_t*>(ws + p.off_lists);
    int* flag = reinterpret_cast<int*>(ws + p.off_flag);
    const int kk = prefilter_list_len(k);
    uint64_t* maxima = reinterpret_cast<uint64_t*>(ws + p.off_maxima);
    unsigned* ticket = reinterpret_cast<unsigned*>(flag + 1);
    prefilter_topk_kernel<<<p.blocks, 256, 0, st>>>(static_cast<const uint4*>(shadow_dev), n_rows,
                                                    reinterpret_cast<const float4*>(query_dev), kk, lists, maxima);
    HMM_LAUNCH_CHECK();
    prefilter_final_kernel<<<1, 1024, 0, st>>>(lists, maxima, ticket, p.blocks, k, kk, (int64_t)p.blocks * 4,
                                               reinterpret_cast<const float4*>(store_dev), reinterpret_cast<const float4*>(query_dev),
                                               idx_out_dev, sim_out_dev, n_out_dev, flag, stats_out_dev HMM_PF_STAMP_ARG);
    HMM_LAUNCH_CHECK();
    // the exact scan, ONE conditional launch executed only when the flag is up (its workgroups return at once otherwise)
    return cosine_topk_if(flag, ticket, store_dev, n_rows, query_dev, k, idx_out_dev, sim_out_dev, n_out_dev, ws + p.off_exact, p.off_lists, st);
}

extern "C" size_t hmm_cosine_topk_segmented_prefilter_workspace_bytes(int64_t n_rows, int n_segments, int k) {
    return hmm_cosine_topk_segmented_workspace_bytes(n_rows, n_segments, k);
}

extern "C" int hmm_cosine_topk_segmented_prefilter(const float* store_dev, const void* shadow_dev, int64_t n_rows, int dim,
                                                   const float* query_dev, const int64_t* seg_offsets_dev, int n_segments, int k,
                                                   int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                                   void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    // no tournament beyond 64 keys; and events of a few dozen rows (one 1024-thread workgroup per event: 20 000 events of 50 rows
    // take 0.75 ms this way, 0.79 ms exactly -- nothing to win): the exact path
    // (n_segments < 1 goes to the exact entry point too: it reports the argument error -- never a division by zero here)
    if (k > kPrefilterMaxK || n_rows == 0 || n_segments < 1 || n_rows / n_segments < 128)
        return hmm_cosine_topk_segmented(store_dev, n_rows, dim, query_dev, seg_offsets_dev, n_segments, k, idx_out_dev, sim_out_dev,
                                         n_out_dev, workspace_dev, workspace_bytes, stream);
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "cosine_topk_segmented_prefilter: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(n_rows > 0 && n_rows < (int64_t)0xFFFFFFFFll, HMM_E_INVALID, "cosine_topk_segmented_prefilter: n_rows out of range");
    HMM_REQUIRE(n_segments >= 1 && k >= 1, HMM_E_INVALID, "cosine_topk_segmented_prefilter: need n_segments >= 1 and k >= 1");
    HMM_REQUIRE(store_dev && shadow_dev && query_dev && seg_offsets_dev && idx_out_dev && sim_out_dev && n_out_dev && workspace_dev,
                HMM_E_INVALID, "cosine_topk_segmented_prefilter: null pointer");
    HMM_REQUIRE(workspace_bytes >= hmm_cosine_topk_segmented_prefilter_workspace_bytes(n_rows, n_segments, k), HMM_E_WORKSPACE,
                "cosine_topk_segmented_prefilter: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HMM_REQUIRE(((uintptr_t)workspace_dev & 15) == 0, HMM_E_INVALID, "cosine_topk_segmented_prefilter: workspace must be 16-byte aligned");
    float* sims = static_cast<float*>(workspace_dev);
    bool deferred = true;
#ifdef HMM_PROBE
    deferred = g_prefilter_sims_deferred != 0;
    if (!deferred) {
        int64_t blocks = (n_rows + 127) / 128;                              // 4 waves x 32 rows
        if (blocks > g_prefilter_sims_blocks) blocks = g_prefilter_sims_blocks;
        prefilter_sims_kernel<<<(unsigned)blocks, 256, 0, st>>>(static_cast<const uint4*>(shadow_dev), n_rows,
                                                                reinterpret_cast<const float4*>(query_dev), sims);
    }
#endif
    if (deferred) {
        int64_t blocks = (n_rows + 15) / 16;                                // 4 waves x 4 rows
        if (blocks > g_prefilter_blocks) blocks = g_prefilter_blocks;
        prefilter_sims_deferred_kernel<<<(unsigned)blocks, 256, 0, st>>>(static_cast<const uint4*>(shadow_dev), n_rows,
                                                                         reinterpret_cast<const float4*>(query_dev), sims);
    }
    HMM_LAUNCH_CHECK();
    if (segments_are_small(n_rows, n_segments, k))          // the two shapes of segment_topk_kernel, for the same reason
        segment_prefilter_kernel<kSmallSegChunk, 256><<<n_segments, 256, 0, st>>>(
            sims, seg_offsets_dev, k, reinterpret_cast<const float4*>(store_dev), reinterpret_cast<const float4*>(query_dev), idx_out_dev,
            sim_out_dev, n_out_dev);
    else
        segment_prefilter_kernel<kChunk, 1024><<<n_segments, 1024, 0, st>>>(
            sims, seg_offsets_dev, k, reinterpret_cast<const float4*>(store_dev), reinterpret_cast<const float4*>(query_dev), idx_out_dev,
            sim_out_dev, n_out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
