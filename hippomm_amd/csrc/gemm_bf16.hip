// bf16 MFMA GEMM for gfx950 with fused epilogues: C[M,N] = A[M,K] x W[N,K]^T (+bias, GELU, residual).
// Carries 97 % of the encoder's FLOPs (QKV / out-proj / fc1 / fc2 / patch-embed / head).
//
// Structure
//   - block tile BM x BN, K step 64; WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile of
//     16x16 blocks on v_mfma_f32_16x16x32_bf16.
//   - both operands are K-contiguous ([M][K] activations, [N][K] nn.Linear weights), so both tiles
//     are staged the same way: global_load_lds_dwordx4 (16 B per lane, 1 KiB = 8 rows x 128 B
//     per wave-instruction) straight into a double-buffered LDS image.  The LDS destination of
//     LDS-DMA is lane-linear, so the bank swizzle chunk' = chunk ^ ((row>>1)&7) is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 side (conflict-free fragment reads).
//   - the MFMA is issued with W as the A-operand and the activations as the B-operand, so that a
//     lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row: the epilogue
//     stores 8 B (bf16) / 16 B (fp32) per lane instead of 2 B.
//   - tile order is remapped so that the blocks that land on one XCD (blockIdx % 8) walk
//     neighbouring tiles and share A / W panels in that XCD's L2.
#include "hmm_common.h"

namespace hmm {

#define HMM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define HMM_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    void* __restrict__ Cout, int M, int N, int K, int tiles_n) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int A_PER_WAVE = (BM / 8) / NW, W_PER_WAVE = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "staging must divide evenly over waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // bijective XCD-aware remap of the linear block id
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (swz / tiles_n) * BM, n0 = (swz % tiles_n) * BN;

    // per-lane staging sources (pre-swizzled), one per 1-KiB wave-instruction
    const bf16_t* a_src[A_PER_WAVE];
    const bf16_t* w_src[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave + i * NW) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;                       // clamp: rows past M are never stored
        a_src[i] = A + (size_t)gm * K + c * 8;
    }
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int row = (wave + i * NW) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        w_src[i] = W + (size_t)(n0 + row) * K + c * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds(HMM_GLB_PTR(a_src[i] + kt * 64),
                                             HMM_LDS_PTR(base + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds(HMM_GLB_PTR(w_src[i] + kt * 64),
                                             HMM_LDS_PTR(base + A_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
    };

    // fragment read offsets: row = tile_row0 + 16*blk + (lane&15); chunk' = (4*kh + (lane>>4)) ^ ((row>>1)&7)
    const int frow = lane & 15;
    const int fsw = frow >> 1;                           // == (row>>1)&7 because tile_row0 + 16*blk is a multiple of 16
    const int a_off0 = (wm * TM + frow) * 128;
    const int w_off0 = A_BYTES + (wn * TN + frow) * 128;
    const int c_k0 = ((lane >> 4) ^ fsw) * 16;           // kh = 0
    const int c_k1 = ((4 + (lane >> 4)) ^ fsw) * 16;     // kh = 1

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = K / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
        const char* base = smem + cur * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int ck = kh ? c_k1 : c_k0;
            bf16x8 af[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(base + a_off0 + mi * 2048 + ck);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *reinterpret_cast<const bf16x8*>(base + w_off0 + ni * 2048 + ck);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: lane holds C[m][n .. n+3], m = m0 + wm*TM + mi*16 + (lane&15), n = n0 + wn*TN + ni*16 + 4*(lane>>4)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * TM + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * TN + ni * 16 + 4 * (lane >> 4);
            f32x4 v = acc[mi][ni];
            if (bias != nullptr) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            const size_t o = (size_t)m * N + n;
            if constexpr (EPI == HMM_EPI_BIAS_BF16 || EPI == HMM_EPI_BIAS_GELU_BF16) {
                if constexpr (EPI == HMM_EPI_BIAS_GELU_BF16) {
                    v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
                }
                bf16x4 o4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(Cout) + o) = o4;
            } else if constexpr (EPI == HMM_EPI_BIAS_RESID_F32) {
                float4* p = reinterpret_cast<float4*>(static_cast<float*>(Cout) + o);
                float4 x = *p;
                x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
                *p = x;
            } else {
                *reinterpret_cast<float4*>(static_cast<float*>(Cout) + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
static int launch_gemm(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, hipStream_t st) {
    constexpr int LDS = 2 * (BM + BN) * 128;
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        HMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    kern<<<tiles_m * tiles_n, WM * WN * 64, LDS, st>>>(A, W, bias, C, M, N, K, tiles_n);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                           int epi, hipStream_t st) {
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return launch_gemm<BM, BN, WM, WN, HMM_EPI_BIAS_BF16>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_BIAS_GELU_BF16: return launch_gemm<BM, BN, WM, WN, HMM_EPI_BIAS_GELU_BF16>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_BIAS_RESID_F32: return launch_gemm<BM, BN, WM, WN, HMM_EPI_BIAS_RESID_F32>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_F32:            return launch_gemm<BM, BN, WM, WN, HMM_EPI_F32>(A, W, bias, C, M, N, K, st);
    }
    set_error("gemm: unknown epilogue %d", epi);
    return HMM_E_INVALID;
}

int g_gemm_default_variant = 0;

int gemm_bf16(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int epi,
              int variant, hipStream_t st) {
    HMM_REQUIRE(A && W && C, HMM_E_INVALID, "gemm: null pointer");
    HMM_REQUIRE(M >= 1 && N >= 128 && K >= 64 && K % 64 == 0 && N % 128 == 0, HMM_E_INVALID,
                "gemm: unsupported shape M=%d N=%d K=%d (need K%%64==0, N%%128==0)", M, N, K);
    HMM_REQUIRE(epi == HMM_EPI_F32 || bias != nullptr, HMM_E_INVALID, "gemm: epilogue %d needs a bias", epi);
    if (variant < 0) variant = g_gemm_default_variant;
    if (variant == 2 && N % 256 != 0) variant = 1;
    switch (variant) {
        case 0: return launch_gemm_epi<128, 128, 2, 2>(A, W, bias, C, M, N, K, epi, st);
        case 1: return launch_gemm_epi<256, 128, 2, 2>(A, W, bias, C, M, N, K, epi, st);
        case 2: return launch_gemm_epi<256, 256, 2, 4>(A, W, bias, C, M, N, K, epi, st);
    }
    set_error("gemm: unknown variant %d", variant);
    return HMM_E_INVALID;
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_gemm_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev, void* c_dev,
                                int M, int N, int K, int epilogue, hmm_stream_t stream) {
    return gemm_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev, c_dev,
                     M, N, K, epilogue, -1, static_cast<hipStream_t>(stream));
}

// Tuning hook (not part of the public header): run a specific tile geometry.
extern "C" int hmm_dev_gemm_bf16_variant(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                         void* c_dev, int M, int N, int K, int epilogue, int variant,
                                         hmm_stream_t stream) {
    return gemm_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev, c_dev,
                     M, N, K, epilogue, variant, static_cast<hipStream_t>(stream));
}
extern "C" void hmm_dev_set_gemm_variant(int variant) { g_gemm_default_variant = variant; }
