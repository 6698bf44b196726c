// Batched feature_search: top-k cosine rows for up to 16 queries in ONE pass over the (N,1024)
// fp32 store.  The reference scans once per question and per modality (top_k_cosine_similarity,
// hippomm/utils/vector_ops.py:151-188, called at hippocampal_memory.py:3153 / :3304); with several questions queued
// the store would be read once per question.  Here a row is read once for all of them: HBM-bound like the single-query
// scan (4096 B per row), i.e. up to 16 x the per-query throughput.
//
// The Q x rows similarity block is a GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: store rows are the
// A operand, queries the B operand, IEEE fp32 products and sums -- no precision is given up): per row 2 x 1024 x 16
// FLOP, ~50 TFLOP/s at the HBM-bound row rate against a ~157 TFLOP/s fp32 matrix peak, so the kernel stays on the
// memory roofline.
//
// Data movement (round 2; round 1 read 16 rows x 64 B per load instruction straight into registers and stopped at
// 5.0 TB/s of loads):
//   * the store is streamed by LDS-DMA in 512-B row pieces: one global_load_lds_dwordx4 fetches 2 rows x 512 B
//     (whole 128-B lines, two contiguous runs), a wave keeps a private ring of 4 slices (16 rows x 128 floats, 8 KiB)
//     with three slices always in flight behind a counted vmcnt -- no registers are spent on loads;
//   * the 16 queries live in REGISTERS: the B operand of lane (q, g) is q[.][32 g + 4 j + s], 256 values per lane for
//     the whole kernel (one wave per SIMD, 512 registers), so the main loop reads nothing but the A fragments from
//     LDS: one ds_read_b128 per four MFMAs;
//   * LDS image of a slice: piece i (rows 2i, 2i+1) at i x 1056 B, inside it 16-B unit 2c + (row & 1) holds chunk c of
//     the row -- the source address of each DMA lane is permuted accordingly.  A fragment read (the 16 rows of a
//     ds_read_b128 lane group, see a_lane below) then touches 16 different 16-B bank slots: conflict-free, measured.
// Selection is fused: every query has a candidate list of order keys in LDS and a threshold = its current k-th best;
// only keys above the threshold are appended (LDS atomic), lists are sorted down to k when they could overflow and at
// the end, and each workgroup leaves its best k keys per query.  (Sharing the thresholds chip-wide through one monotone
// word per query in global memory was built and measured: 0.75 -> 0.83 ms per pass -- the contended line and the atomics
// sit in the in-order VMEM queue in front of the slice waits.  Not kept.)  A second kernel (one workgroup per query) finishes
// exactly as topk_final_kernel does: the global top-k lies in the lists of the k workgroups with the largest maxima.
// Keys, total order (NaN first, higher row first on ties) and outputs are those of hmm_cosine_topk.
#include "hmm_common.h"
#include "topk_tournament.h"

namespace hmm {

constexpr int kMQ = 16;                 // queries per pass
constexpr int kMWaves = 4;              // one wave per SIMD: 512 registers each (256 of them hold the queries)
constexpr int kMTileRows = 16;          // rows per wave per round (one MFMA tile)
constexpr int kMRows = kMWaves * kMTileRows;      // rows per workgroup per round
constexpr int kMSlices = 8;             // K slices per tile: 128 floats each
constexpr int kMPiece = 1056;           // LDS bytes per DMA piece: 2 rows x 512 B + 32 B (bank rotation between pieces)
constexpr int kMSliceBytes = 8 * kMPiece;
constexpr int kMRing = 4;               // slices per wave: one being read, three in flight (96 KiB per CU).  Depth is not the limit:
                                        // 2 / 3 / 4 slices 0.648 / 0.654 / 0.651 ms per pass in one session (round 6).  Nor are the
                                        // MFMAs (VALU dot products instead: 0.653) or, mostly, the 512-B pieces: the bare DMA stream of
                                        // this kernel, nothing read or computed, runs 0.613 against 0.650.  1-KiB pieces (one row x 256
                                        // floats per instruction, 16-KiB slices) leave room for only two slices per wave: 0.683
constexpr int kMCap = 128;              // candidate keys per query and workgroup (>= k + kMRows, power of two)
constexpr int kMMaxK = 64;              // k*k <= 4096 for the one-kernel finish
constexpr int kMMaxBlocks = 2048;
constexpr int kMDmaAux = 2;             // cache policy of the row stream: 2 = non-temporal (each row is read once; 0: 0.724 ms)

struct MultiLds {
    char ring[kMWaves][kMRing][kMSliceBytes];     // 135168 B
    uint64_t keys[kMQ][kMCap];                     //  16384 B
    uint64_t tau[kMQ];
    int cnt[kMQ];
    int need;                                      // workgroup-uniform "sort now" flag (written by wave 0 between barriers)
};

#define HMM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define HMM_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

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

// A raw workgroup barrier that leaves LDS-DMA in flight (__syncthreads() would drain vmcnt to 0).
__device__ __forceinline__ void multi_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(kMWaves * 64) void scan_multi_kernel(const float* __restrict__ store, int64_t n_rows,
                                                                  const float* __restrict__ queries, int n_q, int k,
                                                                  uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    MultiLds& L = *reinterpret_cast<MultiLds*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;

    // B operand: lane (q = r16, g) holds q[128 s + 32 g + 4 j + 0..3] for every slice s and step j (zero rows past n_q).
    // The queries pass through LDS once (coalesced 16-B loads, rows padded by 16 B against bank conflicts; the ring is
    // still idle): fetching the 64 fragments per lane straight from global memory serialises 64 L2 round trips.
    {
        constexpr int QS = 1024 + 4;                              // floats per staged query row
        float* qs = reinterpret_cast<float*>(&L.ring[0][0][0]);   // 16 x 4112 B = 65792 B of the 101376-B ring
        for (int i = tid; i < kMQ * 256; i += kMWaves * 64) {
            const int qi = i >> 8, c = i & 255;
            const int src_row = qi < n_q ? qi : n_q - 1;          // clamped row + select: no branch around the load
            f32x4 v = *reinterpret_cast<const f32x4*>(queries + (size_t)src_row * 1024 + 4 * c);
            if (qi >= n_q) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(qs + qi * QS + 4 * c) = v;
        }
        __syncthreads();
    }
    f32x4 bq[kMSlices][8];
    float qss = 0.f;
    {
        constexpr int QS = 1024 + 4;
        const float* qs = reinterpret_cast<const float*>(&L.ring[0][0][0]) + r16 * QS + 32 * g;
#pragma unroll
        for (int sl = 0; sl < kMSlices; ++sl)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(qs + 128 * sl + 4 * j);
                bq[sl][j] = v;
                qss = fmaf(v[0], v[0], qss); qss = fmaf(v[1], v[1], qss); qss = fmaf(v[2], v[2], qss); qss = fmaf(v[3], v[3], qss);
            }
    }
    qss += __shfl_xor(qss, 16, 64);
    qss += __shfl_xor(qss, 32, 64);
    const float my_qlen = sqrtf(qss);                             // |query r16|, in all four lanes that hold a part of it
    if (tid < kMQ) { L.cnt[tid] = 0; L.tau[tid] = 0ull; }
    if (tid == 0) L.need = 0;
    __syncthreads();                                              // fragments are in registers: the ring may be overwritten

    const int64_t n_tiles = (n_rows + kMTileRows - 1) / kMTileRows;
    const int64_t n_waves = (int64_t)gridDim.x * kMWaves;
    const int64_t wave_gid = (int64_t)blockIdx.x * kMWaves + wave;
    const int64_t n_rounds = (n_tiles + n_waves - 1) / n_waves;   // same for every wave of the grid

    // DMA source of lane l for piece i of (tile, slice): row 16 tile + 2 i + (l & 1), floats 128 slice + 4 (l >> 1) .. + 3
    char* my_ring = L.ring[wave][0];
    auto issue_slice = [&](int64_t tile, int sl, int slot) {
        int64_t row0 = tile * kMTileRows;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t row = row0 + 2 * i + (lane & 1);
            row = row < n_rows ? row : n_rows - 1;                // clamp: rows past the end are masked at selection
            const float* src = store + row * 1024 + 128 * sl + (lane >> 1) * 4;
            __builtin_amdgcn_global_load_lds(HMM_GLB_PTR(src), HMM_LDS_PTR(my_ring + slot * kMSliceBytes + i * kMPiece),
                                             16, 0, kMDmaAux);
        }
    };
    // A fragment of lane (r16, g), step j: chunk c = 8 g + j of row r16 -> unit 2 c + (r16 & 1) of piece r16 >> 1.  k-slot g of
    // the MFMA owns the 32 floats 32 g .. 32 g + 31 of a slice (the B fragments above use the same assignment), so the four
    // k-slots sit 256 B = one full bank row apart: a ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, ...
    // (MI355X_MICROARCH.md, LDS), i.e. rows {0-3, 12-15} of one k-slot with rows {4-11} of the next -- 16 different rows, 16
    // different 16-B bank slots.  (With chunk 4 j + g, 32 B between k-slots, rows 10 / 11 of one slot met rows 12 / 13 of the
    // other: SQ_LDS_BANK_CONFLICT = 41 % of the kernel's LDS cycles.)
    const int a_lane = (r16 >> 1) * kMPiece + (r16 & 1) * 16 + g * 256;

    // flattened (round, slice) sequence n = 8 round + slice; slot n % kMRing; slices n + 1 .. n + LA are in flight while n is read
    constexpr int LA = kMRing - 1;
    auto tile_of = [&](int64_t round) { return wave_gid + round * n_waves; };
#pragma unroll
    for (int p = 0; p < LA; ++p) issue_slice(tile_of(0), p, p);
    int slot = 0;                                                 // slot of the slice being read
    for (int64_t round = 0; round < n_rounds; ++round) {
        const int64_t tile = tile_of(round);
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        float ss = 0.f;
#pragma unroll
        for (int sl = 0; sl < kMSlices; ++sl) {
            // refill the slot read in the previous step (its ds_reads were waited for before that step's MFMAs)
            const int nslot = slot == 0 ? kMRing - 1 : slot - 1;  // (slot + LA) % kMRing
            if (sl + LA < kMSlices) issue_slice(tile, sl + LA, nslot);
            else                    issue_slice(tile_of(round + 1), sl + LA - kMSlices, nslot);   // next round (clamped past the end)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LA * 8) : "memory");     // slice n has landed; n + 1 .. n + LA stay in flight
            __builtin_amdgcn_sched_barrier(0);
            const char* ap = my_ring + slot * kMSliceBytes + a_lane;
            f32x4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const f32x4*>(ap + j * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 b = bq[sl][j];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][1], b[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][2], b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][3], b[3], acc1, 0, 0, 0);
                ss = fmaf(x[j][0], x[j][0], ss); ss = fmaf(x[j][1], x[j][1], ss);
                ss = fmaf(x[j][2], x[j][2], ss); ss = fmaf(x[j][3], x[j][3], ss);
            }
            slot = slot == kMRing - 1 ? 0 : slot + 1;
        }
        // row norms: the four lanes (r16, g = 0..3) of a row hold its partial sums
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float norm = sqrtf(ss);                             // row r16 of the tile
        const f32x4 acc = acc0 + acc1;                            // D layout: query r16, rows 4 g + j
        if (tile < n_tiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rn = __shfl(norm, 4 * g + j, 64);
                const int64_t row = tile * kMTileRows + 4 * g + j;
                const float sim = acc[j] / (rn * my_qlen);
                const uint64_t key = ((uint64_t)order_bits(sim) << 32) | (uint64_t)(uint32_t)row;
                if (r16 < n_q && row < n_rows && key > L.tau[r16]) {
                    const int pos = atomicAdd(&L.cnt[r16], 1);
                    L.keys[r16][pos] = key;
                }
            }
        }
        // Sort the lists down to k: at the end, whenever one could overflow in the next round, and once after the very
        // first round -- that sets the thresholds early, so that from the second round on only rows that beat the
        // current k-th best are appended at all.  The decision is taken by ONE wave between two barriers and read by
        // the others after the second one: it is workgroup-uniform by construction (no wave can append again before
        // every wave has read the flag, because the flag is reset only behind the third barrier).
        multi_barrier();                                          // every append of this round is visible
        if (wave == 0) {
            bool need = round + 1 >= n_rounds || round == 0;
            if (lane < kMQ) {
                need |= L.cnt[lane] > kMCap - kMRows;
            }
            need = __any(need);
            if (lane == 0) L.need = need ? 1 : 0;
        }
        multi_barrier();
        if (L.need) {
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
            multi_barrier();                                      // new counts / thresholds visible before the next appends
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the look-ahead slices past the end
    __syncthreads();
    // this workgroup's best k per query (sorted, 0-padded); a workgroup with no tile leaves zeros
    for (int i = tid; i < n_q * k; i += kMWaves * 64) {
        const int qi = i / k, t = i - qi * k;
        out[((size_t)qi * gridDim.x + blockIdx.x) * k + t] = t < L.cnt[qi] ? L.keys[qi][t] : 0ull;
    }
}

// One workgroup per query: see topk_final_kernel.  Row r lies in tile r / 16, which wave (tile % (4 n_blocks)) of the
// grid scanned, i.e. workgroup (tile % (4 n_blocks)) / 4.
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
    top64_desc(mx, n2);                                      // k <= kMMaxK = 64: the best 64 maxima are enough
    const int n_win = n_blocks < k ? n_blocks : k;
    int m2 = 64;
    while (m2 < n_win * k) m2 <<= 1;
    for (int t = tid; t < m2; t += 1024) {
        uint64_t key = 0ull;
        if (t < n_win * k) {
            const uint64_t top = mx[t / k];
            if (top != 0ull) {
                const int64_t row = (int64_t)(top & 0xFFFFFFFFull);
                const int blk = (int)(((row / kMTileRows) % ((int64_t)n_blocks * kMWaves)) / kMWaves);
                key = c[(size_t)blk * k + (t % k)];
            }
        }
        s[t] = key;
    }
    __syncthreads();
    top64_desc(s, m2);
    if (tid == 0 && n_out) n_out[qi] = k_eff;
    for (int t = tid; t < k_eff; t += 1024) {
        idx_out[(size_t)qi * k_stride + t] = (int64_t)(s[t] & 0xFFFFFFFFull);
        sim_out[(size_t)qi * k_stride + t] = order_bits_inverse((uint32_t)(s[t] >> 32));
    }
}

static int multi_grid(int64_t n_rows) {                        // one workgroup per CU (its LDS fills the CU), 64 rows per round
    const int64_t chunks = (n_rows + kMRows - 1) / kMRows;
    return (int)(chunks < kNumCU ? chunks : kNumCU);
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
                                                                       nq, k_eff, cand);
        HMM_LAUNCH_CHECK();
        topk_final_multi_kernel<<<nq, 1024, 0, st>>>(cand, grid, k_eff, k_eff, idx_out_dev + (size_t)q0 * k,
                                                     sim_out_dev + (size_t)q0 * k, n_out_dev ? n_out_dev + q0 : nullptr, k);
        HMM_LAUNCH_CHECK();
    }
    return HMM_OK;
}
