// Batched feature_search: top-k cosine rows for up to 16 queries in ONE pass over the (N,1024)
// fp32 store.  The reference scans once per question and per modality (top_k_cosine_similarity,
// hippomm/utils/vector_ops.py:151-188, called at hippocampal_memory.py:3153 / :3304); with several questions queued
// the store would be read once per question.  Here a row is read once for all of them: HBM-bound like the single-query
// scan (4096 B per row), i.e. up to 16 x the per-query throughput.
//
// The Q x rows similarity block is a GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: store rows are the
// A operand, queries the B operand, IEEE fp32 products and sums -- no precision is given up): per row 2 x 1024 x 16
// FLOP, ~50 TFLOP/s at the HBM-bound row rate against a ~157 TFLOP/s fp32 matrix peak, so the kernel stays on the
// memory roofline.  Workgroup = 4 waves; a wave owns 64 rows (four 16-row tiles) per iteration and reads them with
// 2 x 16 B per lane such that the four lanes of a row cover one whole 128-B line per step, three steps in flight.
// Queries live in LDS (row stride padded by 16 B against bank conflicts) and are re-read per step (b128).
//
// Selection is fused: every query has a candidate list of order keys in LDS and a threshold = its current k-th best;
// only keys above the threshold are appended (LDS atomic), lists are sorted down to k when they could overflow and at
// the end, and each workgroup leaves its best k keys per query.  A second kernel (one workgroup per query) finishes
// exactly as topk_final_kernel does: the global top-k lies in the lists of the k workgroups with the largest maxima.
// Keys, total order (NaN first, higher row first on ties) and outputs are those of hmm_cosine_topk.
#include "hmm_common.h"

namespace hmm {

constexpr int kMQ = 16;                 // queries per pass
constexpr int kMWaves = 4;              // one wave per SIMD: 512 registers each (3 steps x 4 tiles of loads in flight)
constexpr int kMTiles = 4;              // 16-row tiles per wave
constexpr int kMRows = kMWaves * kMTiles * 16;    // 256 rows per workgroup iteration
constexpr int kMCap = 512;              // candidate keys per query (>= k + kMRows)
constexpr int kMMaxK = 64;              // k*k <= 4096 for the one-kernel finish
constexpr int kMQStride = 1024 + 4;     // floats per query row in LDS
constexpr int kMMaxBlocks = 2048;

struct MultiLds {
    float q[kMQ * kMQStride];           // 65792 B
    uint64_t keys[kMQ][kMCap];          // 65536 B
    uint64_t tau[kMQ];
    float qlen[kMQ];
    int cnt[kMQ];
};

// Descending bitonic sort, by ONE wave, of the first n2 (power of two, <= kMCap) keys of NL lists at once: the lists'
// compare-exchange steps are independent, so walking them in lockstep overlaps their LDS round trips (a single
// list is latency-bound: ~26 us for 512 keys).
template <int NL>
__device__ __forceinline__ void wave_bitonic_desc(uint64_t* (&s)[NL], int n2, int lane) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool desc = (i & k) == 0;
                uint64_t a[NL], b[NL];
#pragma unroll
                for (int q = 0; q < NL; ++q) { a[q] = s[q][i]; b[q] = s[q][l]; }
#pragma unroll
                for (int q = 0; q < NL; ++q)
                    if ((a[q] < b[q]) == desc) { s[q][i] = b[q]; s[q][l] = a[q]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(kMWaves * 64) void scan_multi_kernel(const float* __restrict__ store, int64_t n_rows,
                                                                  const float* __restrict__ queries, int n_q, int k,
                                                                  uint64_t* __restrict__ out, int g_multi_rot) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    MultiLds& L = *reinterpret_cast<MultiLds*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;

    // queries -> LDS (zero rows past n_q), their lengths, empty lists
    for (int i = tid; i < kMQ * 256; i += kMWaves * 64) {
        const int qi = i >> 8, c = i & 255;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < n_q) v = reinterpret_cast<const float4*>(queries)[(size_t)qi * 256 + c];
        *reinterpret_cast<float4*>(&L.q[qi * kMQStride + 4 * c]) = v;
    }
    if (tid < kMQ) { L.cnt[tid] = 0; L.tau[tid] = 0ull; }
    __syncthreads();
    for (int qi = wave; qi < kMQ; qi += kMWaves) {
        float s = 0.f;
        for (int c = lane; c < 1024; c += 64) { const float v = L.q[qi * kMQStride + c]; s = fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) L.qlen[qi] = sqrtf(s);
    }
    __syncthreads();
    const float my_qlen = L.qlen[r16];
    const float* qb = &L.q[r16 * kMQStride + 8 * g];            // + 32*i (+4): this lane's B fragments

    const int64_t n_chunks = (n_rows + kMRows - 1) / kMRows;
    constexpr int STAGES = 4;                                     // register ring: 3 steps of loads in flight (8 measured slower)
    f32x4 a[STAGES][kMTiles][2];
    const float* a_ptr[kMTiles];
    int rot = 0;
    // A operand: lane (r16, g) reads row r16 of each tile, columns 32*i + 8*g .. +7.  Every wave walks K from a different
    // starting step: the 16 rows of a tile are 4 KiB apart, so at any moment a wave asks for the same 128-B column of 16
    // rows; rotating the walk spreads concurrent waves over the columns.
    auto begin_chunk = [&](int64_t chunk) {                       // row pointers + the first STAGES-1 steps of loads
        const int64_t row0 = chunk * kMRows + wave * (kMTiles * 16);
#pragma unroll
        for (int t = 0; t < kMTiles; ++t) {
            int64_t r = row0 + 16 * t + r16;
            r = r < n_rows ? r : n_rows - 1;
            a_ptr[t] = store + r * 1024 + 8 * g;
        }
        rot = (g_multi_rot & 1) ? (int)((chunk * kMWaves + wave) * 5) & 31 : 0;
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
#pragma unroll
            for (int t = 0; t < kMTiles; ++t) {
                a[s][t][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_ptr[t] + 32 * ((s + rot) & 31)));
                a[s][t][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_ptr[t] + 32 * ((s + rot) & 31) + 4));
            }
    };
    if ((int64_t)blockIdx.x < n_chunks) begin_chunk(blockIdx.x);
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t row0 = chunk * kMRows + wave * (kMTiles * 16);     // this wave's 64 rows
        // the query fragments are invariant across chunks: without this opaque zero the compiler hoists all 64 LDS
        // reads (256 registers) out of the chunk loop and spills
        int zv;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
        const float* qbi = qb + zv;
        const bool abl_nomfma = g_multi_rot & 2, abl_noselect = g_multi_rot & 4;      // timing ablations (wrong results)
        f32x4 acc[kMTiles];
        float ss[kMTiles];
#pragma unroll
        for (int t = 0; t < kMTiles; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ss[t] = 0.f; }
        const int rot_cur = rot;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {                            // unrolled by STAGES: ring slots are compile-time
            const int cur = i % STAGES, nxt = (i + STAGES - 1) % STAGES;
            if (i + STAGES - 1 < 32) {
#pragma unroll
                for (int t = 0; t < kMTiles; ++t) {
                    a[nxt][t][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_ptr[t] + 32 * ((i + STAGES - 1 + rot_cur) & 31)));
                    a[nxt][t][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_ptr[t] + 32 * ((i + STAGES - 1 + rot_cur) & 31) + 4));
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // keep the prefetch depth at STAGES - 1 steps
            const float4 b0 = *reinterpret_cast<const float4*>(qbi + 32 * ((i + rot_cur) & 31));
            const float4 b1 = *reinterpret_cast<const float4*>(qbi + 32 * ((i + rot_cur) & 31) + 4);
#pragma unroll
            for (int t = 0; t < kMTiles; ++t) {
                const f32x4 x0v = a[cur][t][0], x1v = a[cur][t][1];
                const float4 x0 = make_float4(x0v[0], x0v[1], x0v[2], x0v[3]), x1 = make_float4(x1v[0], x1v[1], x1v[2], x1v[3]);
                if (abl_nomfma) { ss[t] += (x0.x + x0.y) + (x1.z + x1.w); continue; }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, b0.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, b0.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, b0.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, b0.w, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, b1.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, b1.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, b1.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, b1.w, acc[t], 0, 0, 0);
                ss[t] = fmaf(x0.x, x0.x, ss[t]); ss[t] = fmaf(x0.y, x0.y, ss[t]);
                ss[t] = fmaf(x0.z, x0.z, ss[t]); ss[t] = fmaf(x0.w, x0.w, ss[t]);
                ss[t] = fmaf(x1.x, x1.x, ss[t]); ss[t] = fmaf(x1.y, x1.y, ss[t]);
                ss[t] = fmaf(x1.z, x1.z, ss[t]); ss[t] = fmaf(x1.w, x1.w, ss[t]);
            }
        }
        // the next chunk's first loads fly during the selection below (the ring is free again)
        if (chunk + gridDim.x < n_chunks) begin_chunk(chunk + gridDim.x);
        // row norms: the four lanes (r16, g = 0..3) of a row hold its partial sums
#pragma unroll
        for (int t = 0; t < kMTiles; ++t) {
            float s = ss[t];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float norm = sqrtf(s);                          // lanes 0..15 (and copies): row 16*t + lane%16
            // D layout: this lane holds query r16, rows 4*g + j of the tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rn = __shfl(norm, 4 * g + j, 64);
                const int64_t row = row0 + 16 * t + 4 * g + j;
                const float sim = acc[t][j] / (rn * my_qlen);
                const uint64_t key = ((uint64_t)order_bits(sim) << 32) | (uint64_t)(uint32_t)row;
                if (r16 < n_q && row < n_rows && key > L.tau[r16] && !abl_noselect) {
                    const int pos = atomicAdd(&L.cnt[r16], 1);
                    L.keys[r16][pos] = key;
                }
            }
        }
        __syncthreads();
        // Sort the lists down to k: at the end, whenever one could overflow in the next iteration, and once after the very
        // first iteration -- that sets the thresholds early, so that from the second iteration on only rows that beat the
        // current k-th best are appended at all.
        bool need = chunk + gridDim.x >= n_chunks || chunk == (int64_t)blockIdx.x;
#pragma unroll
        for (int qi = 0; qi < kMQ; ++qi) need |= L.cnt[qi] > kMCap - kMRows;
        if (need) {                                               // workgroup-uniform
            constexpr int NL = kMQ / kMWaves;                     // lists per wave: wave w owns queries w, w + 4, ...
            uint64_t* lists[NL];
            int n[NL], nmax = 0;
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                lists[q] = L.keys[wave + q * kMWaves];
                n[q] = L.cnt[wave + q * kMWaves];
                nmax = n[q] > nmax ? n[q] : nmax;
            }
            int n2 = 64;
            while (n2 < nmax) n2 <<= 1;
#pragma unroll
            for (int q = 0; q < NL; ++q)
                for (int t = n[q] + lane; t < n2; t += 64) lists[q][t] = 0ull;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            wave_bitonic_desc<NL>(lists, n2, lane);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    const int qi = wave + q * kMWaves;
                    L.cnt[qi] = n[q] < k ? n[q] : k;
                    L.tau[qi] = n[q] >= k ? lists[q][k - 1] : 0ull;
                }
            }
            __syncthreads();
        }
    }
    // this workgroup's best k per query (sorted, 0-padded); a workgroup with no chunk leaves zeros
    for (int i = tid; i < n_q * k; i += kMWaves * 64) {
        const int qi = i / k, t = i - qi * k;
        out[((size_t)qi * gridDim.x + blockIdx.x) * k + t] = t < L.cnt[qi] ? L.keys[qi][t] : 0ull;
    }
}

// One workgroup per query: see topk_final_kernel.  Row r was scanned by workgroup (r / kMRows) % n_blocks.
__global__ __launch_bounds__(1024) void topk_final_multi_kernel(const uint64_t* __restrict__ cand, int n_blocks, int k,
                                                                int k_eff, int64_t* __restrict__ idx_out,
                                                                float* __restrict__ sim_out, int32_t* __restrict__ n_out,
                                                                int k_stride) {
    __shared__ uint64_t mx[kMMaxBlocks];
    __shared__ uint64_t s[4096];
    const int tid = threadIdx.x, qi = blockIdx.x;
    const uint64_t* c = cand + (size_t)qi * n_blocks * k;
    int n2 = 64;
    while (n2 < n_blocks) n2 <<= 1;
    for (int t = tid; t < n2; t += 1024) mx[t] = t < n_blocks ? c[(size_t)t * k] : 0ull;
    __syncthreads();
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += 1024) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const bool desc = (i & kk) == 0;
                const uint64_t a = mx[i], b = mx[l];
                if ((a < b) == desc) { mx[i] = b; mx[l] = a; }
            }
            __syncthreads();
        }
    const int n_win = n_blocks < k ? n_blocks : k;
    int m2 = 64;
    while (m2 < n_win * k) m2 <<= 1;
    for (int t = tid; t < m2; t += 1024) {
        uint64_t key = 0ull;
        if (t < n_win * k) {
            const uint64_t top = mx[t / k];
            if (top != 0ull) {
                const int64_t row = (int64_t)(top & 0xFFFFFFFFull);
                const int blk = (int)((row / kMRows) % n_blocks);
                key = c[(size_t)blk * k + (t % k)];
            }
        }
        s[t] = key;
    }
    __syncthreads();
    for (int kk = 2; kk <= m2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m2 >> 1); t += 1024) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const bool desc = (i & kk) == 0;
                const uint64_t a = s[i], b = s[l];
                if ((a < b) == desc) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    if (tid == 0 && n_out) n_out[qi] = k_eff;
    for (int t = tid; t < k_eff; t += 1024) {
        idx_out[(size_t)qi * k_stride + t] = (int64_t)(s[t] & 0xFFFFFFFFull);
        sim_out[(size_t)qi * k_stride + t] = order_bits_inverse((uint32_t)(s[t] >> 32));
    }
}

static int multi_grid(int64_t n_rows) {
    const int64_t chunks = (n_rows + kMRows - 1) / kMRows;
    return (int)(chunks < 2 * kNumCU ? chunks : 2 * kNumCU);
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_cosine_topk_workspace_bytes(int64_t n_rows, int k);
extern "C" int hmm_cosine_topk(const float* store_dev, int64_t n_rows, int dim, const float* query_dev, int k,
                               int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                               void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

extern "C" size_t hmm_cosine_topk_multi_workspace_bytes(int64_t n_rows, int n_queries, int k) {
    if (n_rows < 1 || n_queries < 1 || k < 1) return 0;
    const int64_t k_eff = k < n_rows ? k : n_rows;
    if (k_eff > kMMaxK) return hmm_cosine_topk_workspace_bytes(n_rows, k);           // per-query fallback
    return align_up((size_t)kMQ * multi_grid(n_rows) * (size_t)k_eff * 8, 256) + 256;
}

extern "C" int hmm_cosine_topk_multi(const float* store_dev, int64_t n_rows, int dim, const float* queries_dev,
                                     int n_queries, int k, int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                     void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "cosine_topk_multi: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(n_rows >= 1 && n_rows < (int64_t)0xFFFFFFFFll, HMM_E_INVALID, "cosine_topk_multi: n_rows=%lld out of range",
                (long long)n_rows);
    HMM_REQUIRE(n_queries >= 1 && k >= 1, HMM_E_INVALID, "cosine_topk_multi: n_queries=%d k=%d", n_queries, k);
    HMM_REQUIRE(store_dev && queries_dev && idx_out_dev && sim_out_dev && workspace_dev, HMM_E_INVALID,
                "cosine_topk_multi: null pointer");
    HMM_REQUIRE(((uintptr_t)store_dev & 15) == 0 && ((uintptr_t)queries_dev & 15) == 0, HMM_E_INVALID,
                "cosine_topk_multi: store/queries must be 16-byte aligned");
    const size_t need = hmm_cosine_topk_multi_workspace_bytes(n_rows, n_queries, k);
    HMM_REQUIRE(workspace_bytes >= need, HMM_E_WORKSPACE, "cosine_topk_multi: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int k_eff = (int)(k < n_rows ? k : n_rows);
    if (k_eff > kMMaxK) {                                          // large k: one ordinary scan per query
        for (int qi = 0; qi < n_queries; ++qi) {
            const int rc = hmm_cosine_topk(store_dev, n_rows, dim, queries_dev + (size_t)qi * dim, k,
                                           idx_out_dev + (size_t)qi * k, sim_out_dev + (size_t)qi * k,
                                           n_out_dev ? n_out_dev + qi : nullptr, workspace_dev, workspace_bytes, stream);
            if (rc != HMM_OK) return rc;
        }
        return HMM_OK;
    }
    HMM_ENSURE_DYN_LDS(scan_multi_kernel, (int)sizeof(MultiLds));
    const int grid = multi_grid(n_rows);
    uint64_t* cand = static_cast<uint64_t*>(workspace_dev);
    for (int q0 = 0; q0 < n_queries; q0 += kMQ) {                  // 16 queries per pass over the store
        const int nq = n_queries - q0 < kMQ ? n_queries - q0 : kMQ;
        scan_multi_kernel<<<grid, kMWaves * 64, sizeof(MultiLds), st>>>(store_dev, n_rows, queries_dev + (size_t)q0 * dim,
                                                                       nq, k_eff, cand, 1);
        HMM_LAUNCH_CHECK();
        topk_final_multi_kernel<<<nq, 1024, 0, st>>>(cand, grid, k_eff, k_eff, idx_out_dev + (size_t)q0 * k,
                                                     sim_out_dev + (size_t)q0 * k, n_out_dev ? n_out_dev + q0 : nullptr, k);
        HMM_LAUNCH_CHECK();
    }
    return HMM_OK;
}
