// Consolidation similarity for gfx950.  Replaces HippocampalMemory._select_key_frames
// (reference hippomm/core/hippocampal_memory.py:944-967).
//
//   normalize_rows_kernel  Fn = F / ||F||_row           (:951)  fp64 sum of squares -> fp32 norm,
//                          fp32 division, exactly one rounding each.
//   gram_bits_kernel       S = Fn Fn^T                  (:952)  v_mfma_f64_16x16x4_f64: every dot
//                          product accumulated in fp64 and rounded to fp32 once, so the value
//                          is within 1 fp32 ulp of what ANY sgemm summation order produces; the
//                          kernel never materialises S: it compares (float)S < thr at once
//                          (:960) and emits one bit per pair, "not (S < thr)", which makes a NaN
//                          similarity block like `nan < thr == False` does in the reference.
//                          Only tile pairs (ti <= tj) are computed; the mirror bits are written
//                          from the same accumulator, so the relation is exactly symmetric.
//   greedy_select_kernel   the ordered scan of :955-961 on the bit matrix: keep 0; keep i iff
//                          no kept k has bit (k,i); one wave, `blocked` bitmap in LDS.
// HBM layout: features (n,1024) fp32 row-major (as stacked at :842); workspace = Fn (n_pad,1024)
// fp32 + adjacency bits (n_pad x n_pad/32 uint32), n_pad = n rounded up to 64.
#include "hmm_common.h"

namespace hmm {

constexpr int kGT = 64;        // gram tile edge
constexpr int kGK = 32;        // k-slab staged per step
constexpr int kGLd = kGK + 2;  // LDS row stride in floats: bank = 2*row + k -> conflict-free column reads

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ f, int n, int n_pad,
                                                             float* __restrict__ fn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    float4* dst = reinterpret_cast<float4*>(fn + (size_t)row * HMM_FEATURE_DIM);
    if (row >= n) {                                   // padding rows: zeros (their bits are masked anyway)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4* src = reinterpret_cast<const float4*>(f + (size_t)row * HMM_FEATURE_DIM);
    float4 x[4];
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] = src[j * 64 + lane];
        ss += (double)x[j].x * x[j].x + (double)x[j].y * x[j].y + (double)x[j].z * x[j].z + (double)x[j].w * x[j].w;
    }
    ss = wave_sum(ss);
    const float len = (float)sqrt(ss);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dst[j * 64 + lane] = make_float4(x[j].x / len, x[j].y / len, x[j].z / len, x[j].w / len);
}

// One block = one 64x64 tile pair (ti <= tj) of the adjacency relation.  4 waves, each a 32x32
// quadrant = 2x2 MFMA blocks of 16x16 (v_mfma_f64_16x16x4_f64: A[l&15][k=l>>4], B[k=l>>4][l&15],
// D col = l&15, row = (l>>4) + 4*reg).
__global__ __launch_bounds__(256) void gram_bits_kernel(const float* __restrict__ fn, int n, int n_pad,
                                                        float thr, uint32_t* __restrict__ adj) {
    // decode (ti, tj) with ti <= tj from the linear block id
    const int T = n_pad / kGT;
    int ti = 0, rem = blockIdx.x;
    while (rem >= T - ti) { rem -= T - ti; ++ti; }
    const int tj = ti + rem;

    __shared__ float sa[kGT * kGLd];
    __shared__ float sb[kGT * kGLd];
    __shared__ uint32_t bits_ij[kGT * 2];   // row i of tile ti -> 64 column bits of tile tj
    __shared__ uint32_t bits_ji[kGT * 2];   // row j of tile tj -> 64 column bits of tile ti

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int qr = (w >> 1) * 32, qc = (w & 1) * 32;      // quadrant origin inside the tile
    if (tid < kGT * 2) { bits_ij[tid] = 0; bits_ji[tid] = 0; }

    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    const float* ga = fn + (size_t)ti * kGT * HMM_FEATURE_DIM;
    const float* gb = fn + (size_t)tj * kGT * HMM_FEATURE_DIM;
    // staging: 64 rows x 32 floats per operand = 512 float4; 256 threads x 2
    const int srow = tid >> 3, scol = (tid & 7) * 4;

    for (int k0 = 0; k0 < HMM_FEATURE_DIM; k0 += kGK) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = srow + h * 32;
            const float4 va = *reinterpret_cast<const float4*>(ga + (size_t)r * HMM_FEATURE_DIM + k0 + scol);
            const float4 vb = *reinterpret_cast<const float4*>(gb + (size_t)r * HMM_FEATURE_DIM + k0 + scol);
            float* pa = sa + r * kGLd + scol;
            float* pb = sb + r * kGLd + scol;
            pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
            pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kGK; kk += 4) {
            const int kq = kk + (lane >> 4);
            double a0 = (double)sa[(qr + (lane & 15)) * kGLd + kq];
            double a1 = (double)sa[(qr + 16 + (lane & 15)) * kGLd + kq];
            double b0 = (double)sb[(qc + (lane & 15)) * kGLd + kq];
            double b1 = (double)sb[(qc + 16 + (lane & 15)) * kGLd + kq];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // compare and collect bits in LDS
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int li = qr + a * 16 + (lane >> 4) + 4 * reg;   // row inside tile ti
                const int lj = qc + b * 16 + (lane & 15);             // col inside tile tj
                const int gi = ti * kGT + li, gj = tj * kGT + lj;
                const float s = (float)acc[a][b][reg];
                const bool hit = !(s < thr) && gi < n && gj < n;
                if (hit) {
                    atomicOr(&bits_ij[li * 2 + (lj >> 5)], 1u << (lj & 31));
                    atomicOr(&bits_ji[lj * 2 + (li >> 5)], 1u << (li & 31));
                }
            }
    __syncthreads();
    const int W = n_pad / 32;
    if (tid < kGT * 2) {
        const int r = tid >> 1, h = tid & 1;
        adj[(size_t)(ti * kGT + r) * W + tj * 2 + h] = bits_ij[tid];
        if (ti != tj) adj[(size_t)(tj * kGT + r) * W + ti * 2 + h] = bits_ji[tid];
    }
}

// One wave.  blocked[] lives in LDS (n_pad/32 words).  Row k of adj is OR-ed in when k is kept.
__global__ __launch_bounds__(64) void greedy_select_kernel(const uint32_t* __restrict__ adj, int n, int n_pad,
                                                           int64_t* __restrict__ kept, int32_t* __restrict__ n_kept) {
    extern __shared__ __attribute__((aligned(16))) uint32_t blocked[];
    const int lane = threadIdx.x;
    const int W = n_pad / 32;
    for (int w = lane; w < W; w += 64) blocked[w] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int count = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t word = blocked[i >> 5];            // same address in every lane: broadcast
        const bool is_blocked = (word >> (i & 31)) & 1u;
        if (i == 0 || !is_blocked) {                       // wave-uniform branch
            if (lane == 0) kept[count] = i;
            ++count;
            const uint32_t* row = adj + (size_t)i * W;
            for (int w = (i >> 5) + lane; w < W; w += 64) blocked[w] |= row[w];
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) *n_kept = count;
}

__global__ void arange_kernel(int n, int64_t* kept, int32_t* n_kept) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) kept[t] = t;
    if (t == 0) *n_kept = n;
}

struct GramPlan { int n_pad; size_t off_fn, off_adj, total; };
static GramPlan gram_plan(int n) {
    GramPlan p{};
    p.n_pad = (n + kGT - 1) / kGT * kGT;
    p.off_fn = 0;
    p.off_adj = align_up((size_t)p.n_pad * HMM_FEATURE_DIM * sizeof(float), 256);
    p.total = p.off_adj + align_up((size_t)p.n_pad * (p.n_pad / 32) * sizeof(uint32_t), 256) + 256;
    return p;
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_gram_select_workspace_bytes(int n) {
    if (n < 1) return 256;
    return gram_plan(n).total;
}

extern "C" int hmm_gram_select(const float* features_dev, int n, int dim, float threshold,
                               int64_t* kept_out_dev, int32_t* n_kept_out_dev,
                               void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(dim == HMM_FEATURE_DIM, HMM_E_INVALID, "gram_select: dim must be %d, got %d", HMM_FEATURE_DIM, dim);
    HMM_REQUIRE(n >= 0 && n <= 1000000, HMM_E_INVALID, "gram_select: n=%d out of range", n);
    HMM_REQUIRE(kept_out_dev && n_kept_out_dev, HMM_E_INVALID, "gram_select: null output pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n <= 2) {                                           // hippocampal_memory.py:947-948
        arange_kernel<<<1, 64, 0, st>>>(n, kept_out_dev, n_kept_out_dev);
        HMM_LAUNCH_CHECK();
        return HMM_OK;
    }
    HMM_REQUIRE(features_dev && workspace_dev, HMM_E_INVALID, "gram_select: null pointer");
    HMM_REQUIRE(((uintptr_t)features_dev & 15) == 0, HMM_E_INVALID, "gram_select: features must be 16-byte aligned");
    const GramPlan p = gram_plan(n);
    HMM_REQUIRE(workspace_bytes >= p.total, HMM_E_WORKSPACE, "gram_select: workspace %zu < required %zu",
                workspace_bytes, p.total);
    const size_t lds = (size_t)(p.n_pad / 32) * sizeof(uint32_t);
    HMM_REQUIRE(lds <= 64 * 1024, HMM_E_INVALID, "gram_select: n=%d too large for the LDS bitmap", n);
    char* base = static_cast<char*>(workspace_dev);
    float* fn = reinterpret_cast<float*>(base + p.off_fn);
    uint32_t* adj = reinterpret_cast<uint32_t*>(base + p.off_adj);

    normalize_rows_kernel<<<(p.n_pad + 3) / 4, 256, 0, st>>>(features_dev, n, p.n_pad, fn);
    HMM_LAUNCH_CHECK();
    const int T = p.n_pad / kGT;
    gram_bits_kernel<<<T * (T + 1) / 2, 256, 0, st>>>(fn, n, p.n_pad, threshold, adj);
    HMM_LAUNCH_CHECK();
    greedy_select_kernel<<<1, 64, lds, st>>>(adj, n, p.n_pad, kept_out_dev, n_kept_out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
