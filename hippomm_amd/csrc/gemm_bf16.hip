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
#include <string>
#include <type_traits>
#include "hmm_common.h"
#include "encoder_ops.h"
#include "gemm_pp_mainloop.h"

namespace hmm {

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26:
//   erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0,
// |error| <= 1.5e-7 absolute -- two orders below the bf16 rounding of the result -- with one v_rcp and
// one v_exp instead of libm erff's ~25-instruction expansion (the fc1 epilogue evaluates 337 M of
// these per transformer block at batch 256).  Written as  GELU(x) = max(x, 0) - |x| * (P(t)/2) * exp(-x^2/2),
// which is the same function on both signs of x (for x < 0 the 1 + erf cancels to P exp) and needs no sign
// handling: 13 full-rate operations + 2 transcendentals per element (the erf-then-combine form took ~20 + 2).
__device__ __forceinline__ float gelu_erf(float x) {
    const float u = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, u, 1.0f));
    float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(u * u * (-0.5f * 1.4426950408889634f));
    return fmaxf(x, 0.0f) - u * poly * e;
}

// Shared epilogue.  A lane holds C[m][n..n+3] for m = m_lane + 16*mi, n = n_lane + 16*ni.  The bias
// vectors are loaded once per lane (not per store), and the fp32 residual read-modify-write is
// software-pipelined one row-block ahead so that its load latency is not paid per element.
template <int EPI, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MI][NI], const float* __restrict__ bias,
                                              void* __restrict__ Cout, int M, int N, int m_lane, int n_lane) {
    float4 bv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        bv[ni] = bias ? *reinterpret_cast<const float4*>(bias + n_lane + ni * 16) : make_float4(0.f, 0.f, 0.f, 0.f);

    if constexpr (EPI == HMM_EPI_BIAS_RESID_F32) {
        float* C = static_cast<float*>(Cout);
        float4 xin[2][NI];
        auto load_row = [&](int mi, float4 (&dst)[NI]) {
            const int m = m_lane + mi * 16;
            if (m < M) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    dst[ni] = *reinterpret_cast<const float4*>(C + (size_t)m * N + n_lane + ni * 16);
            }
        };
        load_row(0, xin[0]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (mi + 1 < MI) load_row(mi + 1, xin[(mi + 1) & 1]);
            const int m = m_lane + mi * 16;
            if (m < M) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float4 x = xin[mi & 1][ni];
                    const f32x4 v = acc[mi][ni];
                    // (accumulator + bias) + residual: the order of the ping-pong kernel's epilogue, so that a row gets the
                    // same bits from every tile geometry
                    const float4 o = make_float4((v[0] + bv[ni].x) + x.x, (v[1] + bv[ni].y) + x.y, (v[2] + bv[ni].z) + x.z,
                                                 (v[3] + bv[ni].w) + x.w);
                    *reinterpret_cast<float4*>(C + (size_t)m * N + n_lane + ni * 16) = o;
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m_lane + mi * 16;
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f32x4 v = acc[mi][ni];
                v[0] += bv[ni].x; v[1] += bv[ni].y; v[2] += bv[ni].z; v[3] += bv[ni].w;
                const size_t o = (size_t)m * N + n_lane + ni * 16;
                if constexpr (EPI == HMM_EPI_F32) {
                    *reinterpret_cast<float4*>(static_cast<float*>(Cout) + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    if constexpr (EPI == HMM_EPI_BIAS_GELU_BF16) {
                        v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
                    }
                    bf16x4 o4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(Cout) + o) = o4;
                }
            }
        }
    }
}

// STAGES = 2: double buffer, the next K-tile is requested when the current one starts and drained before the barrier -- one
// memory latency per 64 columns of K when a workgroup is alone on its CU (few tiles: ~0.55 us per K-tile, the text tower's
// fc2 at K = 4096 is 40 us whatever M).  STAGES > 2: a ring with STAGES - 1 K-tiles in flight behind a counted vmcnt, one
// barrier per K-tile; same fragment reads and MFMA order, same bits.
// KSUB > 1 ("deep K"): a ring stage holds KSUB consecutive K-tiles, so the barrier, the counted wait and the DMA issue are paid
// once per KSUB x 64 columns of K -- for launches of few rows, where a workgroup is alone on its CU and every step is a chain
// of latencies, not of MFMAs.  K / 64 must be a multiple of KSUB (the launcher checks).  Same MFMA order, same bits.
// (A staggered loop for the eight-wave tiles -- waves 4-7 half a K-tile behind waves 0-3, the ping-pong kernel's schedule at ring
// size -- was built in round 5: bit-equal, 0 ... +4 % slower on the 128-row tiles; -14 % on a 256 x 128 tile, which is 25-30 % faster
// alone than anything else on 13-16 frames' fc2 / out-proj and buys nothing in the forwards; profiles/LABNOTES_r5.md 13, 18.)
// ABL (probe build, timing only -- results are wrong): 1 = no MFMAs, 2 = no fragment reads and no MFMAs, 3 = no LDS-DMA.
template <int BM, int BN, int WM, int WN, int EPI, int STAGES = 2, int KSUB = 1, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    void* __restrict__ Cout, int M, int N, int K, int tiles_n, int tiles_m, int col_major, int k_len) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, SUB = A_BYTES + W_BYTES, STAGE = KSUB * SUB;
    constexpr int A_PER_WAVE = (BM / 8) / NW, W_PER_WAVE = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "staging must divide evenly over waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // bijective XCD-aware remap of the linear block id: XCD x (= blockIdx % 8) takes a contiguous run of the work list.
    // The list is [K split][tile]; tiles row-major (a row tile's column tiles are neighbours: an XCD shares A panels) or
    // column-major (a column tile's row tiles are neighbours: an XCD shares W panels, and with few row tiles every weight
    // byte leaves HBM / the Infinity Cache once instead of once per row tile -- see launch_gemm).
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tiles = tiles_m * tiles_n;
    const int split = swz / tiles, tl = swz - split * tiles;      // split > 0 only in a split-K launch (grid = splits x tiles)
    const int m0 = (col_major ? tl % tiles_m : tl / tiles_n) * BM, n0 = (col_major ? tl / tiles_m : tl % tiles_n) * BN;
    A += (size_t)split * k_len;                                   // this workgroup's K range: columns [split k_len, + k_len)
    W += (size_t)split * k_len;
    if (nb > tiles) Cout = static_cast<float*>(Cout) + (size_t)split * M * N;   // split-K: fp32 partial slab of this split

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

    auto stage = [&](int kg, int buf) {                  // the KSUB K-tiles of group kg into ring buffer `buf`
        if constexpr (ABL == 3) return;
#pragma unroll
        for (int sub = 0; sub < KSUB; ++sub) {
            char* base = smem + buf * STAGE + sub * SUB;
            const int kt = kg * KSUB + sub;
#pragma unroll
            for (int i = 0; i < A_PER_WAVE; ++i)
                __builtin_amdgcn_global_load_lds(HMM_GLB_PTR(a_src[i] + kt * 64),
                                                 HMM_LDS_PTR(base + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < W_PER_WAVE; ++i)
                __builtin_amdgcn_global_load_lds(HMM_GLB_PTR(w_src[i] + kt * 64),
                                                 HMM_LDS_PTR(base + A_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
        }
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

    const int KT = k_len / (64 * KSUB);                  // ring steps (groups of KSUB K-tiles)
    // All fragment reads of the K-tile first, then the MFMAs in K order behind COUNTED waits: the second half's reads are in
    // flight under the first half's MFMAs (one exposed LDS latency per K-tile instead of two; with one wave per SIMD nothing else
    // hides it).  The reads are inline assembly because the compiler waits for lgkmcnt(0) before the first MFMA however the
    // loads are arranged; the empty asm statements tie each fragment to the wait in front of it (volatile asm keeps its order).
    const uint32_t lds0 = (uint32_t)(uintptr_t)HMM_LDS_PTR(smem);
    auto compute_sub = [&](int stage_off) {
        if constexpr (ABL == 2) return;
        bf16x8 af[2][MI], wf[2][NI];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const uint32_t pa = lds0 + stage_off + a_off0 + (kh ? c_k1 : c_k0);
            const uint32_t pw = lds0 + stage_off + w_off0 + (kh ? c_k1 : c_k0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kh][mi]) : "v"(pa), "n"(mi * 2048));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kh][ni]) : "v"(pw), "n"(ni * 2048));
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if (kh == 0) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");
            } else {
                __builtin_amdgcn_sched_barrier(0);     // the first half's MFMAs stay in front of this wait
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(af[kh][mi]));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[kh][ni]));
            if constexpr (ABL != 1) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][ni], af[kh][mi], acc[mi][ni], 0, 0, 0);
            }
        }
    };
    auto compute = [&](int stage_off) {
#pragma unroll
        for (int sub = 0; sub < KSUB; ++sub) compute_sub(stage_off + sub * SUB);
    };
    if constexpr (STAGES == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
            compute(cur * STAGE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // Ring of STAGES K-tiles.  Iteration kt: wait for this wave's own pieces of K-tile kt (the younger K-tiles stay in
        // flight), barrier (RAW: every wave's pieces of kt have landed; WAR: every wave has finished reading kt - 1), request
        // K-tile kt + STAGES - 1 into the buffer kt - 1 lived in, compute kt.
        constexpr int PER_STAGE = KSUB * (A_PER_WAVE + W_PER_WAVE);   // LDS-DMA instructions per wave and ring stage
        static_assert(STAGES <= 8 && PER_STAGE * (STAGES - 2) <= 63, "vmcnt is a 6-bit counter; the switch below has 7 cases");
#pragma unroll
        for (int st = 0; st < STAGES - 1; ++st)
            if (st < KT) stage(st, st);
        int buf = 0;
        for (int kt = 0; kt < KT; ++kt) {
            const int younger = KT - 1 - kt < STAGES - 2 ? KT - 1 - kt : STAGES - 2;   // K-tiles requested after kt so far
            switch (younger) {                                   // wave-uniform; the immediate must be a constant
                case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory"); break;
                case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * 2) : "memory"); break;
                case 3:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (STAGES > 4 ? 3 : 0)) : "memory"); break;
                case 4:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (STAGES > 5 ? 4 : 0)) : "memory"); break;
                case 5:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (STAGES > 6 ? 5 : 0)) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (STAGES > 7 ? 6 : 0)) : "memory"); break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int nxt = kt + STAGES - 1;
            if (nxt < KT) stage(nxt, buf == 0 ? STAGES - 1 : buf - 1);
            compute(buf * STAGE);
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // epilogue: lane holds C[m][n .. n+3]
    gemm_epilogue<EPI, MI, NI>(acc, bias, Cout, M, N, m0 + wm * TM + (lane & 15), n0 + wn * TN + 4 * (lane >> 4));
}


// ------------------------------------------------------------------------------------------------
// Few rows (one question through the text tower: 77 token rows; a handful of frames): the tiled kernels put
// ceil(M/128) x N/128 workgroups on 256 CUs and each of them walks K behind ONE LDS-DMA stage, i.e. one memory latency per
// 64 columns of K -- the text tower's fc2 (M = 77, N = 1024, K = 4096) is 8 workgroups x 64 steps = 60 us for 0.65 GFLOP,
// and a question's 24 blocks take 3.2 ms.  Here ONE WAVE owns a (16 MT) x 16 output sliver and streams its operands
// straight from L2 into MFMA fragments (a lane's fragment is 16 contiguous bytes of a row: no LDS, no barrier), DEPTH
// K-steps in flight per wave, a few waves per SIMD: N/16 x ceil(M / 16 MT) one-wave workgroups cover the chip and the
// kernel is bound by what L2 delivers, not by a latency chain.  The workgroups that share a 16-row slice of W sit on one
// XCD (same blockIdx % 8), so a weight byte leaves HBM once.  Same instruction, operand roles, fragment layout and K order
// as the tiled kernels: an output element gets the same bits (tests/test_gpu_ops.py compares them bitwise).
template <int EPI, int MT, int DEPTH>
__global__ __launch_bounds__(64) void gemm_bf16_sliver_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    void* __restrict__ Cout, int M, int N, int K, int tiles_m) {
    const int lane = threadIdx.x;
    const int frow = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int m0 = (j % tiles_m) * (16 * MT), n0 = ((j / tiles_m) * 8 + xcd) * 16;

    const bf16_t* wp = W + (size_t)(n0 + frow) * K + 8 * g;
    const bf16_t* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int gm = m0 + 16 * t + frow;
        gm = gm < M ? gm : M - 1;                       // clamp: rows past M are never stored
        ap[t] = A + (size_t)gm * K + 8 * g;
    }

    bf16x8 wf[DEPTH][2], af[DEPTH][MT][2];
    auto load = [&](int slot, int kt) {
        wf[slot][0] = *reinterpret_cast<const bf16x8*>(wp + kt * 64);
        wf[slot][1] = *reinterpret_cast<const bf16x8*>(wp + kt * 64 + 32);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            af[slot][t][0] = *reinterpret_cast<const bf16x8*>(ap[t] + kt * 64);
            af[slot][t][1] = *reinterpret_cast<const bf16x8*>(ap[t] + kt * 64 + 32);
        }
    };

    f32x4 acc[MT][1];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = K >> 6;
    // the first DEPTH steps, unconditionally (steps past the end of a short K re-load the last one and are never used): with a
    // branch here the compiler no longer knows how many loads are in flight at the loop header below
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load(s, s < KT ? s : KT - 1);
    auto step = [&](int s) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][kh], af[s][t][kh], acc[t][0], 0, 0, 0);
    };
    int kt0 = 0;
    // steady state without a branch in it, so that the compiler counts the loads in flight (vmcnt(4 (DEPTH-1)) style waits)
    // instead of draining them at every use
    for (; kt0 + 2 * DEPTH <= KT; kt0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            step(s);
            load(s, kt0 + s + DEPTH);
        }
    }
    for (; kt0 < KT; kt0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int kt = kt0 + s;
            if (kt < KT) {
                step(s);
                if (kt + DEPTH < KT) load(s, kt + DEPTH);
            }
        }
    }
    gemm_epilogue<EPI, MT, 1>(acc, bias, Cout, M, N, m0 + frow, n0 + 4 * g);
}

// LDS-transposed epilogue of the ping-pong kernel.  The MFMA leaves a lane with 4 consecutive
// columns of 16 different rows, so direct stores are 8-B (bf16) pieces of 16 rows per instruction:
// measured store-issue-bound at ~7 B/clk/CU (22 % of the qkv GEMM).  After the main loop the LDS is
// idle, so each wave parks its 128x64 sub-tile in a private LDS slab (row stride padded against
// bank conflicts), reads it back row-major and stores 16 B per lane: every wave-instruction then
// writes 8 whole 128-B lines (bf16) / 4 rows x 256 B (fp32), and the fp32 residual is read with the
// same shape.  fp32 outputs go in two 64-row halves to fit the slab.
constexpr int kEpiSlab = 18432;                      // per-wave LDS slab: 128 rows x 144 B (bf16) / 64 x 272 B (fp32)

// GUARD = false: the wave's 128 rows all exist (m_wave + 128 <= M, every tile but the last row tile): no per-row
// predicates, so the compiler batches the slab reads and keeps the loads / stores back to back.  Addresses are a
// wave-uniform base (C + m_wave * N + n_wave, scalar registers) plus a 32-bit per-lane byte offset; each row step adds a
// uniform amount.  (With per-row guards and 64-bit per-lane addresses the fp32 epilogue was 1,400 instructions per wave and
// waited for every slab read on its own.)
template <int EPI, bool GUARD>
__device__ __forceinline__ void gemm_epilogue_lds_body(f32x4 (&acc)[8][4], const float4 (&bv)[4], void* __restrict__ Cout,
                                                       int M, int N, int m_wave, int n_wave, char* slab, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int rows_left = M - m_wave;                            // GUARD: rows [0, rows_left) of the wave tile exist
    if constexpr (EPI == HMM_EPI_BIAS_BF16 || EPI == HMM_EPI_BIAS_GELU_BF16) {
        constexpr int RS = 144;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[mi][ni];
                v[0] += bv[ni].x; v[1] += bv[ni].y; v[2] += bv[ni].z; v[3] += bv[ni].w;
                if constexpr (EPI == HMM_EPI_BIAS_GELU_BF16) {
                    v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
                }
                bf16x4 o4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(slab + (mi * 16 + fr) * RS + (ni * 16 + 4 * fq) * 2) = o4;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        char* Cw = reinterpret_cast<char*>(static_cast<bf16_t*>(Cout) + (size_t)m_wave * N + n_wave);   // wave-uniform
        const int rsub = lane >> 3, chunk = lane & 7;
        const unsigned lane_off = (unsigned)(rsub * N + chunk * 8) * 2u;
        const unsigned row_step = (unsigned)N * 16u;            // 8 rows of bf16
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const uint4 v = *reinterpret_cast<const uint4*>(slab + (it * 8 + rsub) * RS + chunk * 16);
            if (!GUARD || it * 8 + rsub < rows_left)             // consumed once by the next kernel: non-temporal (+0.9 % on the forward)
                __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(Cw + it * row_step + lane_off));
        }
    } else {
        constexpr int RS = 272;
        constexpr bool RESID = EPI == HMM_EPI_BIAS_RESID_F32;
        char* Cw = reinterpret_cast<char*>(static_cast<float*>(Cout) + (size_t)m_wave * N + n_wave);      // wave-uniform
        const int rsub = lane >> 4, chunk = lane & 15;
        const unsigned lane_off = (unsigned)(rsub * N + chunk * 4) * 4u;
        const unsigned row_step = (unsigned)N * 16u;            // 4 rows of fp32
        // Software-pipelined over 16-row groups: the residual rows of a group are requested one 32-row quarter before they
        // are needed, into the registers the group of the previous quarter has just been added from -- a quarter never
        // waits a full HBM latency and only 32 registers of residual are live beside the accumulators (with both 64-row
        // halves in flight, 128 registers, a variant with more epilogue work spilled 70 registers to scratch).
        float4 xin[8];
        auto load_resid = [&](int q, int g) {                    // rows 32 q + 16 g .. + 15 of the wave tile
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = g * 4 + i, row = q * 32 + it * 4 + rsub;
                if constexpr (GUARD) xin[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!GUARD || row < rows_left) {                 // streamed once: non-temporal (out-proj -2 % in the tower, -12 % alone)
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Cw + (q * 8 + it) * row_step + lane_off));
                    xin[it] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        };
        auto park = [&](int half) {                              // accumulators + bias -> the wave's slab, row-major
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const f32x4 a = acc[half * 4 + mi][ni];
                    *reinterpret_cast<float4*>(slab + (mi * 16 + fr) * RS + (ni * 16 + 4 * fq) * 4) =
                        make_float4(a[0] + bv[ni].x, a[1] + bv[ni].y, a[2] + bv[ni].z, a[3] + bv[ni].w);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        auto drain = [&](int q) {                                // slab rows (+ residual) -> C, 256 B per 16 lanes
#pragma unroll
            for (int g = 0; g < 2; ++g) {                        // four rows at a time: reads batched, registers bounded
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = *reinterpret_cast<const float4*>(slab + ((q & 1) * 32 + (g * 4 + i) * 4 + rsub) * RS + chunk * 16);
                if constexpr (RESID) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 x = xin[g * 4 + i];
                        v[i].x += x.x; v[i].y += x.y; v[i].z += x.z; v[i].w += x.w;
                    }
                    if (q < 3) load_resid(q + 1, g);             // the same group of the next quarter, into the freed registers
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int it = g * 4 + i;
                    if (!GUARD || q * 32 + it * 4 + rsub < rows_left) {
                        if constexpr (RESID)
                            __builtin_nontemporal_store(f32x4{v[i].x, v[i].y, v[i].z, v[i].w},
                                                        reinterpret_cast<f32x4*>(Cw + (q * 8 + it) * row_step + lane_off));
                        else
                            *reinterpret_cast<float4*>(Cw + (q * 8 + it) * row_step + lane_off) = v[i];
                    }
                }
            }
        };
        auto slab_free = [&]() {                                 // the slab is re-used: its reads must be done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        if constexpr (RESID) { load_resid(0, 0); load_resid(0, 1); }
        park(0);
        drain(0);
        drain(1);
        slab_free();
        park(1);
        drain(2);
        drain(3);
    }
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_lds(f32x4 (&acc)[8][4], const float* __restrict__ bias,
                                                  void* __restrict__ Cout, int M, int N, int m_wave, int n_wave,
                                                  char* slab, int lane) {
    const int fq = lane >> 4;
    float4 bv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        bv[ni] = bias ? *reinterpret_cast<const float4*>(bias + n_wave + ni * 16 + 4 * fq) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (m_wave + 128 <= M)                                       // wave-uniform
        gemm_epilogue_lds_body<EPI, false>(acc, bv, Cout, M, N, m_wave, n_wave, slab, lane);
    else if (m_wave < M)
        gemm_epilogue_lds_body<EPI, true>(acc, bv, Cout, M, N, m_wave, n_wave, slab, lane);
}

// ------------------------------------------------------------------------------------------------
// Ping-pong kernel: 256x256x64 tiles, 8 waves; the K loop lives in gemm_pp_mainloop.h (shared with the fused
// in_proj + attention kernel).  Tiles are walked in plain strips (whole row tiles, XCD-contiguous); ablations, tile-walk and
// pipeline variants that were measured and rejected are recorded in profiles/LABNOTES.md section 4.3 / 4.5, not kept in the code.
// ------------------------------------------------------------------------------------------------
// Tile walk inside the XCD-contiguous order: 0 = strips (a row tile's column tiles are consecutive); (R << 8) | C = blocks of
// R row tiles x C column tiles.  Strips re-stream the whole weight matrix from the Infinity Cache once per row tile (fc1:
// 2.35 GB through the fabric per launch for 0.18 GB of operands); 6 x 5 blocks make that 1.65 GB and the fc1 GEMM 3 % faster
// (0.5 % on the forward), and do nothing for the K = 5120 and N = 1280 shapes (profiles/r3_walk_evidence.json,
// r3_walk_fetch.json).  -1 (product default) = 6 x 5 where the column tiles split into whole groups of five and there are
// enough row tiles, strips otherwise.  Which workgroup computes a tile does not change the tile: same bits.
HMM_TUNABLE(int, g_gemm_walk, -1)
static int pick_walk(int tiles_m, int tiles_n) {
    if (g_gemm_walk >= 0) return g_gemm_walk;
    return (tiles_n >= 10 && tiles_n % 5 == 0 && tiles_m >= 48) ? ((6 << 8) | 5) : 0;
}
#ifdef HMM_PROBE
// in-kernel stamps (s_memrealtime, 100 MHz): per workgroup {start, -, main loop done, stores retired, XCC id, HW id};
// written to a buffer of their own that nothing else reads
unsigned long long* g_gemm_stamps = nullptr;
extern "C" void hmm_probe_set_gemm_stamps(unsigned long long* p) { g_gemm_stamps = p; }
#define HMM_PROBE_ARG , unsigned long long* stamps
#define HMM_PROBE_VAL , g_gemm_stamps
#define HMM_STAMP(slot)                                                                              \
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define HMM_PROBE_ARG
#define HMM_PROBE_VAL
#define HMM_STAMP(slot)
#endif

HMM_TUNABLE(int, g_gemm_trunc_rounds, 0) // probe build: 1 = drop the tiles of a partly filled last round (quantisation-cost experiment; results wrong)
HMM_TUNABLE(int, g_gemm_skip_tail, 0)   // probe build: 1 = do not launch the peeled tail (what the tails cost in the forward)

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    void* __restrict__ Cout, int M, int N, int K, int tiles_n, int walk HMM_PROBE_ARG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int m0 = (swz / tiles_n) * 256, n0 = (swz % tiles_n) * 256;
    if (walk) {
        const int R = walk >> 8, C = walk & 255, tiles_m = nb / tiles_n;
        const int rg = swz / (R * tiles_n), rem = swz - rg * R * tiles_n;
        const int rows_g = tiles_m - R * rg < R ? tiles_m - R * rg : R;
        const int cg = rem / (rows_g * C), rem2 = rem - cg * rows_g * C;
        const int wc = tiles_n - C * cg < C ? tiles_n - C * cg : C;
        m0 = (rg * R + rem2 / wc) * 256;
        n0 = (cg * C + rem2 % wc) * 256;
    }
    HMM_STAMP(0)
#ifdef HMM_PROBE
    if (stamps && threadIdx.x == 0) stamps[(size_t)bid * 8 + 6] = __builtin_amdgcn_s_memtime();
    if (stamps && threadIdx.x == 0) {
        stamps[(size_t)bid * 8 + 4] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        stamps[(size_t)bid * 8 + 5] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    }
#endif

    // staging sources as 32-bit element offsets from A / W (launcher guarantees M*K, N*K < 2^31):
    // half-tile local row lr = (wave + 8j)*8 + (lane>>3), 16-B chunk (lane&7) un-swizzled
    PPSources src;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wave + 8 * j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int arow = (lr & 63) + (lr >> 6) * 128;           // rows 0..63 of wave-row 0 / wave-row 1
        int g0 = m0 + arow, g1 = m0 + arow + 64;
        g0 = g0 < M ? g0 : M - 1;
        g1 = g1 < M ? g1 : M - 1;
        src.alo[j] = g0 * K + c * 8;
        src.ahi[j] = g1 * K + c * 8;
        const int bcol = (lr >> 5) * 64 + (lr & 31);             // cols 0..31 of each wave-column
        src.blo[j] = (n0 + bcol) * K + c * 8;
        src.bhi[j] = (n0 + bcol + 32) * K + c * 8;
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    pp_mainloop(A, W, src, K >> 6, smem, lane, wave, acc);   // K/64 even, >= 2 (checked by the launcher)

    // every wave is past its last LDS read and every DMA has landed: the LDS is free
    HMM_STAMP(2)
#ifdef HMM_PROBE
    if (stamps && threadIdx.x == 0) stamps[(size_t)bid * 8 + 7] = __builtin_amdgcn_s_memtime();     // shader-clock ticks
#endif
    // the epilogue's lane-derived indices (lane >> 4, lane & 15, ...) are recomputed HERE from an opaque copy of the lane id:
    // hoisted in front of the main loop they were carried across it and two of them spilled to scratch in the fp32 epilogues
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    gemm_epilogue_lds<EPI>(acc, bias, Cout, M, N, m0 + wm * 128, n0 + wn * 64, smem + wave * kEpiSlab, lane_e);
#ifdef HMM_PROBE
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    HMM_STAMP(3)
#endif
}

template <int EPI>
static int launch_gemm_pp(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                          hipStream_t st) {
    constexpr int LDS = 8 * kEpiSlab > 2 * 4 * 16384 ? 8 * kEpiSlab : 2 * 4 * 16384;
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
    auto kern = gemm_bf16_pp_kernel<EPI>;
    HMM_ENSURE_DYN_LDS(kern, LDS);
    int grid = tiles_m * tiles_n;
    if (g_gemm_trunc_rounds && grid > 256 && grid % 256 > 64) grid = grid / 256 * 256;   // probe build, timing only: whole rounds
    kern<<<grid, 512, LDS, st>>>(A, W, bias, C, M, N, K, tiles_n, pick_walk(tiles_m, tiles_n) HMM_PROBE_VAL);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// Tile order of the tiled kernels (which workgroup computes a tile does not change the tile: same bits).  With fewer row
// tiles than column tiles -- few-row launches: one frame's qkv GEMM is 5 x 60 tiles of 64 x 64 -- the row-major list hands
// every XCD a run of column tiles of ONE row tile, so each W panel is pulled through the fabric by five different XCDs
// (49 MB for 9.8 MB of weights) while the tiny A panels are what gets shared; column-major hands an XCD whole columns.
// Built and measured in round 5 (profiles/r5_splitk_probe.json, r5_splitk_forward.json): alone with cold weights one
// frame's qkv / fc1 13.8 -> 13.3 / 15.3 -> 14.3 us, but IN the forwards 0 ... +4 % (one / two / four questions +1.5 / +2.8 /
// +4.3 %, one frame +0.9 %): the fabric is not what these launches wait for.  Row-major stays; 1 = column-major in the
// probe build only.
HMM_TUNABLE(int, g_gemm_col_major, 0)
static thread_local int t_gemm_tail_launch = 0;       // the peeled tail of a big launch keeps the row-major list (set by gemm_bf16)
static thread_local int t_gemm_splits = 1;            // > 1: the next launch_gemm is a split-K launch (set by gemm_bf16_splitk)

template <int BM, int BN, int WM, int WN, int EPI, int STAGES = 2, int KSUB = 1, int ABL = 0>
static int launch_gemm(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                       hipStream_t st) {
    constexpr int LDS = STAGES * KSUB * (BM + BN) * 128;
    static_assert(LDS <= 160 * 1024, "ring of the tiled GEMM: LDS");
    const int splits = t_gemm_splits;
    HMM_REQUIRE(K % splits == 0 && ((K / splits) / 64) % KSUB == 0 && (K / splits) % 64 == 0, HMM_E_INVALID,
                "gemm: K = %d / %d splits is not a multiple of %d", K, splits, 64 * KSUB);
    HMM_REQUIRE(splits == 1 || (EPI == HMM_EPI_F32 && bias == nullptr), HMM_E_INVALID, "gemm: split-K writes fp32 partials only");
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, STAGES, KSUB, ABL>;
    HMM_ENSURE_DYN_LDS(kern, LDS);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const int col_major = g_gemm_col_major > 0 && !t_gemm_tail_launch && 2 * tiles_m <= tiles_n;
    kern<<<tiles_m * tiles_n * splits, WM * WN * 64, LDS, st>>>(A, W, bias, C, M, N, K, tiles_n, tiles_m, col_major, K / splits);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The few-row dispatcher's rules, as data.  Every threshold of launch_gemm_small_epi / gemm_bf16 below is one row here: the
// constant's name (probe build: the run-time knob hmm_probe_set_<name>), the shipped limit, the value that switches the rule
// OFF, what it decides, the forwards on which it fires ("tower:batch,batch;...") and the measurement it came from.  FROZEN in
// round 6: no new rules.  `python tools/dispatch_audit_probe.py --check` re-times every row (rule on / off on its own forwards,
// interleaved in one process, bit equality checked) and writes profiles/r6_dispatch_recheck.json; a row that does not buy 2 %
// on any of its forwards on that run is deleted, not re-tuned.
struct DispatchRule { const char* knob; int limit; int off; const char* what; const char* forwards; const char* evidence; };
constexpr DispatchRule kDispatchRules[] = {
    {"g_gemm_small_64", 512, 0, "launches of at most this many 64x64 tiles run on them behind the 4-deep ring (else 128x128 ring / double buffer)",
     "text:1,4;vision:1,2;audio:1", "r3_sliver_probe.json, r3_text_latency.json"},
    {"g_gemm_small_32", 400, 0, "launches of at most this many 32x32 tiles run on them (one 16x16 block per wave)",
     "text:1,2;vision:1", "r3_text_latency.json"},
    {"g_gemm_deepk", 1, 0, "deep-K rings: 2 / 4 K-tiles per stage while most CUs have a workgroup of their own (32x32: <= 192 / 256 tiles, 64x64: <= 256)",
     "text:1,2;vision:1", "r4_deepk_probe.json"},
    {"g_gemm_ring8", 1, 0, "the 128-row ring tiles (128x128, 128x64) with eight waves instead of four",
     "vision:1,8;audio:1;text:9", "r5_rect_tile_probe2.json, r5_ring8_ab.json"},
    {"g_gemm_rect", 1, 0, "128x64 ring tiles when 64x64 tiles outnumber the CUs but 128x64 tiles do not, up to g_gemm_rect_rows rows",
     "vision:1;text:4,6;audio:1", "r5_rect_tile_probe.json, r5_rect_forward2.json"},
    {"g_gemm_rect_rows", 700, 320, "row limit of the 128x64 rule (320 = the four-wave tile's)",
     "text:5,6,7,8;audio:1", "r5_rect700_ab.json"},
    {"g_gemm_rect_rows_longk", 1536, 0, "the 128x64 rule's row limit for K >= 5120 (fc2 of three to five frames)",
     "vision:3,4,5", "r5_rect_longk_ab.json"},
    {"g_gemm_rect64_min_t64", 450, 0, "64x128 ring tiles from this many 64x64 tiles on, while 64x128 tiles do not outnumber the CUs",
     "vision:6;text:24,26", "r5_dispatch_audit_ab.json"},
    {"g_gemm_ring_peel_rows", 16, 0, "a last row tile of at most this many rows is peeled off a ring launch that it pushes past one tile per CU",
     "vision:3,4", "r5_ring_peel_ab_vision.json"},
};
constexpr bool rule_name_is(const char* a, const char* b) {
    while (*a && *a == *b) { ++a; ++b; }
    return *a == *b;
}
constexpr int rule_limit(const char* knob) {
    for (const DispatchRule& r : kDispatchRules)
        if (rule_name_is(r.knob, knob)) return r.limit;
    return -0x7fffffff;                                    // a misspelt name shows up as an absurd threshold in every test
}
#define HMM_RULE(name) HMM_TUNABLE(int, name, rule_limit(#name))
#ifdef HMM_PROBE
// the table as JSON, for tools/dispatch_audit_probe.py --check
extern "C" const char* hmm_probe_dispatch_rules() {
    static std::string text;
    if (text.empty()) {
        text = "[";
        for (const DispatchRule& r : kDispatchRules) {
            if (text.size() > 1) text += ", ";
            text += std::string("{\"knob\": \"") + r.knob + "\", \"limit\": " + std::to_string(r.limit) + ", \"off\": " + std::to_string(r.off) +
                    ", \"what\": \"" + r.what + "\", \"forwards\": \"" + r.forwards + "\", \"evidence\": \"" + r.evidence + "\"}";
        }
        text += "]";
    }
    return text.c_str();
}
#endif

// Rows per wave of the sliver kernel (16 MT) and whether it beats the tiled kernels, from two fitted lines
// (tools/sliver_probe.py, profiles/r3_sliver_probe.json): a wave reads its fragments from L2 unshared, so the kernel moves
// waves x K x (MT + 1) x 32 B at ~7.5 TB/s chip-wide (less when fewer than ~192 waves are in flight) after ~3 us of launch and
// first-load latency; tiles behind the 4-deep ring -- what a launch this small would otherwise use -- take ~2.8 us plus
// 0.225 us (64x64 tiles) or 0.19 us (32x32 tiles, when the launch fits them) per 64 columns of K.  The sliver kernel wins for
// one or two dozen rows and for the text tower's fc2 at 77 rows (13 vs 15 us).
HMM_TUNABLE(int, g_gemm_sliver_mt, 0)    // probe build: force 16 / 32 / 64 rows per wave (1 / 2 / 4)
HMM_TUNABLE(int, g_gemm_sliver_auto, 1)  // probe build: 0 = the dispatcher never picks the sliver kernel
HMM_RULE(g_gemm_small_64)
HMM_RULE(g_gemm_small_32)
HMM_RULE(g_gemm_deepk)
static bool ring32_fits(int M, int N) { return g_gemm_small_32 && (long)((M + 31) / 32) * (N / 32) <= g_gemm_small_32; }
static float sliver_us(int M, int N, int K, int mt) {
    const long waves = (long)((M + 16 * mt - 1) / (16 * mt)) * (N / 16);
    const float stream = (float)waves * K * (mt + 1) * 32.0f / 7.5e6f;
    return 3.0f + stream * (waves < 192 ? 192.0f / waves : 1.0f);
}
static int sliver_mt(int M, int N, int K) {
    if (g_gemm_sliver_mt > 0) return g_gemm_sliver_mt;
    int best = 1;
    for (int mt = 2; mt <= 4 && 16 * (mt / 2) < M; mt *= 2)
        if (sliver_us(M, N, K, mt) < sliver_us(M, N, K, best)) best = mt;
    return best;
}
static bool sliver_wins(int M, int N, int K, int epi) {
    if (!g_gemm_sliver_auto || epi > HMM_EPI_F32) return false;
    if ((long)((M + 63) / 64) * (N / 64) > 512) return false;               // that many rows: never
    const long t32 = (long)((M + 31) / 32) * (N / 32);
    float per_ktile = ring32_fits(M, N) ? 0.19f : 0.225f;                     // what the launch would use instead
    if (g_gemm_deepk && ring32_fits(M, N)) {                                  // the deep-K rings (launch_gemm_small_epi)
        if (K >= 2048 && (K >> 6) % 4 == 0 && t32 <= 256) per_ktile = 0.13f;
        else if (K >= 1024 && (K >> 6) % 2 == 0 && t32 <= 192) per_ktile = 0.155f;
    }
    return sliver_us(M, N, K, sliver_mt(M, N, K)) < 0.9f * (2.8f + per_ktile * (K >> 6));
}

template <int EPI>
static int launch_gemm_sliver(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                              hipStream_t st) {
    const int mt = sliver_mt(M, N, K);
    const int tiles_m = (M + 16 * mt - 1) / (16 * mt), grid = tiles_m * (N / 16);
    if (mt == 1) gemm_bf16_sliver_kernel<EPI, 1, 8><<<grid, 64, 0, st>>>(A, W, bias, C, M, N, K, tiles_m);
    else if (mt == 2) gemm_bf16_sliver_kernel<EPI, 2, 6><<<grid, 64, 0, st>>>(A, W, bias, C, M, N, K, tiles_m);
    else gemm_bf16_sliver_kernel<EPI, 4, 4><<<grid, 64, 0, st>>>(A, W, bias, C, M, N, K, tiles_m);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

static int launch_gemm_sliver_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                  int epi, hipStream_t st) {
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return launch_gemm_sliver<HMM_EPI_BIAS_BF16>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_BIAS_GELU_BF16: return launch_gemm_sliver<HMM_EPI_BIAS_GELU_BF16>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_BIAS_RESID_F32: return launch_gemm_sliver<HMM_EPI_BIAS_RESID_F32>(A, W, bias, C, M, N, K, st);
        case HMM_EPI_F32:            return launch_gemm_sliver<HMM_EPI_F32>(A, W, bias, C, M, N, K, st);
    }
    set_error("gemm: the sliver kernel has no epilogue %d", epi);
    return HMM_E_INVALID;
}

#define HMM_EPI_SWITCH(CALL)                                                              \
    switch (epi) {                                                                        \
        case HMM_EPI_BIAS_BF16:      return CALL(HMM_EPI_BIAS_BF16);                      \
        case HMM_EPI_BIAS_GELU_BF16: return CALL(HMM_EPI_BIAS_GELU_BF16);                 \
        case HMM_EPI_BIAS_RESID_F32: return CALL(HMM_EPI_BIAS_RESID_F32);                 \
        case HMM_EPI_F32:            return CALL(HMM_EPI_F32);                            \
    }                                                                                     \
    set_error("gemm: unknown epilogue %d", epi);                                          \
    return HMM_E_INVALID;

template <int BM, int BN, int WM, int WN>
static int launch_gemm_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                           int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<BM, BN, WM, WN, E>(A, W, bias, C, M, N, K, st)
    HMM_EPI_SWITCH(HMM_CALL)
#undef HMM_CALL
}

// 64x64 tiles behind the four-buffer ring (64 KiB of LDS: two workgroups per CU).  Measured and not kept, both bitwise equal:
// eight buffers (seven K-tiles in flight, one workgroup per CU) 0.29 us per K-tile instead of 0.225; five buffers with the
// fragments of the next K-tile read under the MFMAs of the current one 0.25 -- the step is the barrier, the DMA issue and the
// dependent read -> MFMA chain, not bytes in flight; eight waves of 32x16 per tile (two per SIMD) 0.25 (profiles/LABNOTES.md 4.8).
static int launch_gemm_ring64_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                  int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<64, 64, 2, 2, E, 4>(A, W, bias, C, M, N, K, st)
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return HMM_CALL(HMM_EPI_BIAS_BF16);
        case HMM_EPI_BIAS_GELU_BF16: return HMM_CALL(HMM_EPI_BIAS_GELU_BF16);
        case HMM_EPI_BIAS_RESID_F32: return HMM_CALL(HMM_EPI_BIAS_RESID_F32);
        case HMM_EPI_F32:            return HMM_CALL(HMM_EPI_F32);
    }
#undef HMM_CALL
    set_error("gemm: 64x64 tiles have no epilogue %d", epi);
    return HMM_E_INVALID;
}

static int launch_gemm_ring32_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                  int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<32, 32, 2, 2, E, 4>(A, W, bias, C, M, N, K, st)
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return HMM_CALL(HMM_EPI_BIAS_BF16);
        case HMM_EPI_BIAS_GELU_BF16: return HMM_CALL(HMM_EPI_BIAS_GELU_BF16);
        case HMM_EPI_BIAS_RESID_F32: return HMM_CALL(HMM_EPI_BIAS_RESID_F32);
        case HMM_EPI_F32:            return HMM_CALL(HMM_EPI_F32);
    }
#undef HMM_CALL
    set_error("gemm: 32x32 tiles have no epilogue %d", epi);
    return HMM_E_INVALID;
}

// Deep-K rings (KSUB K-tiles per stage): the epilogues of the plain rings
template <int BM, int STAGES, int KSUB>
static int launch_gemm_ringk_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                 int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<BM, BM, 2, 2, E, STAGES, KSUB>(A, W, bias, C, M, N, K, st)
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return HMM_CALL(HMM_EPI_BIAS_BF16);
        case HMM_EPI_BIAS_GELU_BF16: return HMM_CALL(HMM_EPI_BIAS_GELU_BF16);
        case HMM_EPI_BIAS_RESID_F32: return HMM_CALL(HMM_EPI_BIAS_RESID_F32);
        case HMM_EPI_F32:            return HMM_CALL(HMM_EPI_F32);
    }
#undef HMM_CALL
    set_error("gemm: deep-K tiles have no epilogue %d", epi);
    return HMM_E_INVALID;
}

// Rectangular tiles behind the ring (round 5): with few rows the chip is filled by COLUMN tiles, and what a launch then costs is
// its K walk times the workgroups a CU has to run one after (or beside) the other.  One frame's qkv GEMM is 300 tiles of 64 x 64
// on 256 CUs -- 44 CUs run two workgroups and the launch takes their time, 13.3 us; as 128 x 64 tiles it is 180 workgroups, one per
// CU, and the tile's 16 MFMAs per wave and K-tile hide the ring's latencies better than the 8 of a 64 x 64 tile.
template <int BM, int BN, int WM = 2, int WN = 2>
static int launch_gemm_ring_rect_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                     int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<BM, BN, WM, WN, E, 4>(A, W, bias, C, M, N, K, st)
    switch (epi) {
        case HMM_EPI_BIAS_BF16:      return HMM_CALL(HMM_EPI_BIAS_BF16);
        case HMM_EPI_BIAS_GELU_BF16: return HMM_CALL(HMM_EPI_BIAS_GELU_BF16);
        case HMM_EPI_BIAS_RESID_F32: return HMM_CALL(HMM_EPI_BIAS_RESID_F32);
        case HMM_EPI_F32:            return HMM_CALL(HMM_EPI_F32);
    }
#undef HMM_CALL
    set_error("gemm: rectangular ring tiles have no epilogue %d", epi);
    return HMM_E_INVALID;
}

static int launch_gemm_ring128_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                   int epi, hipStream_t st) {
#define HMM_CALL(E) launch_gemm<128, 128, 2, 2, E, 4>(A, W, bias, C, M, N, K, st)
    HMM_EPI_SWITCH(HMM_CALL)
#undef HMM_CALL
}

// What the dispatcher uses where it says "128 x 128 ring" / "128 x 64 ring": the EIGHT-wave instantiations (wave tiles 64 x 32 /
// 32 x 32, two waves per SIMD): one wave's LDS reads and DMA issue run under its partner's MFMAs.  Alone with cold weights
// (profiles/r5_rect_tile_probe2.json) 128 x 128: 17.1 -> 15.2 us (one frame's qkv), 13.3 -> 11.9 (one audio segment's fc1),
// 53.4 -> 46.8 (eight frames' fc2); 128 x 64: 12.2 -> 11.5, 13.0 -> 11.9.  Same MFMA sequence per output element: same bits.
HMM_RULE(g_gemm_ring8)
static int launch_gemm_ring128_auto_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                        int epi, hipStream_t st) {
    return g_gemm_ring8 ? launch_gemm_ring_rect_epi<128, 128, 2, 4>(A, W, bias, C, M, N, K, epi, st)
                        : launch_gemm_ring128_epi(A, W, bias, C, M, N, K, epi, st);
}

// Launches of few tiles (small batches, cls rows, the head, the peeled last row tile of the big launches): a workgroup is alone
// on its CU, so what counts is the latency of its own K walk.  64x64 tiles behind the 4-deep ring while there are at most 512 of
// them (64 KiB of LDS, two per CU: ~0.225 us per K-tile, and four times the workgroups of 128x128 tiles), 32x32 tiles (one
// 16x16 block per wave, ~0.19 us per K-tile) while there are at most 400 of those; 128x128 tiles behind
// the ring up to 256 tiles (128 KiB, one per CU); beyond that the double-buffered 128x128 kernel (two per CU).
HMM_TUNABLE(int, g_gemm_small_stages, 4)   // probe build: 2 = the double-buffered kernel only
static thread_local int t_gemm_small_tiles = 128;     // see gemm_set_small_tiles (encoder_ops.h)
int gemm_set_small_tiles(int tiles) {
    const int prev = t_gemm_small_tiles;
    t_gemm_small_tiles = tiles;
    return prev;
}
HMM_RULE(g_gemm_rect)
HMM_RULE(g_gemm_rect64_min_t64)
HMM_RULE(g_gemm_rect_rows)
HMM_RULE(g_gemm_rect_rows_longk)
HMM_RULE(g_gemm_ring_peel_rows)
static int launch_gemm_small_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                                 int epi, hipStream_t st, bool tail = false) {
    if (g_gemm_small_stages != 2 && !tail && g_gemm_ring_peel_rows > 0) {
        // A frame is 257 token rows = two 128-row tiles + ONE row: k frames end in a row tile of k rows.  When that sliver of a tile
        // is what pushes the launch past one 128 x 128 ring tile per CU (three frames' fc1: 7 x 40 = 280 tiles for 6 x 40 + 3 rows;
        // four frames' qkv: 9 x 30 = 270), peel it: the full row tiles in one round of the ring, the last rows through the
        // one-wave sliver kernel.  Same MFMA sequence per output element: same bits.
        const int tm = (M + 127) / 128, tail_rows = M - (tm - 1) * 128;
        const long cols = N / 128;
        if (tm > 1 && tail_rows <= g_gemm_ring_peel_rows && tm * cols > kNumCU && (tm - 1) * cols <= kNumCU) {
            const int m_main = (tm - 1) * 128;
            int rc = launch_gemm_small_epi(A, W, bias, C, m_main, N, K, epi, st, false);
            if (rc != HMM_OK) return rc;
            const bool c_bf16 = epi == HMM_EPI_BIAS_BF16 || epi == HMM_EPI_BIAS_GELU_BF16;
            return launch_gemm_sliver_epi(A + (size_t)m_main * K, W, bias, static_cast<char*>(C) + (size_t)m_main * N * (c_bf16 ? 2 : 4),
                                          tail_rows, N, K, epi, st);
        }
    }
    if (g_gemm_small_stages == 2 || (long)((M + 127) / 128) * (N / 128) > 256) {
        // past one ring tile per CU: the double-buffered 128 x 128 kernel, two per CU.  (Round 5 sent bias -> bf16 launches of at least
        // 80 ping-pong tiles to the 256 x 256 kernel here: +0.3 ... 0.5 % on five to seven frames in round 6's recheck, below the 2 % a
        // rule has to buy -- removed, profiles/r6_dispatch_recheck_midround.json.)
        return launch_gemm_epi<128, 128, 2, 2>(A, W, bias, C, M, N, K, epi, st);
    }
    if (!tail && epi <= HMM_EPI_F32 && ring32_fits(M, N)) {
        // deep-K rings while the launch leaves most CUs a workgroup of their own (tools/deepk_probe.py, cold weights): one
        // question's fc2 (96 tiles, K = 4096) 16.4 -> 10.5 us with four K-tiles per stage, its out-proj 6.3 -> 5.3 with two
        const long t32 = (long)((M + 31) / 32) * (N / 32);
        if (g_gemm_deepk && K >= 2048 && (K >> 6) % 4 == 0 && t32 <= 256) return launch_gemm_ringk_epi<32, 4, 4>(A, W, bias, C, M, N, K, epi, st);
        if (g_gemm_deepk && K >= 1024 && (K >> 6) % 2 == 0 && t32 <= 192) return launch_gemm_ringk_epi<32, 4, 2>(A, W, bias, C, M, N, K, epi, st);
        return launch_gemm_ring32_epi(A, W, bias, C, M, N, K, epi, st);
    }
    // (the peeled last row tile of a big launch used to take 64 x 64 tiles up to 128 of them: 0.0 % at 128 / 256 frames in round 6's
    // recheck -- removed; it runs on the 128 x 128 ring like any launch of few tiles)
    if (g_gemm_small_64 && !tail && (long)((M + 63) / 64) * (N / 64) <= g_gemm_small_64) {
        // a few hundred rows with more 64 x 64 tiles than CUs, but at most one 128 x 64 tile per CU: nobody runs two workgroups one
        // after (or beside) the other -- one frame's qkv / fc1 (300 / 400 tiles -> 180 / 240) 13.9 -> 12.1 us / 15.0 -> 13.0 us
        // alone with cold weights (profiles/r5_rect_tile_probe.json), one frame's forward 2.24 -> 2.12 ms, four questions
        // 1.234 -> 1.224.  Beyond ~300 rows it loses in the forwards (one / two audio segments +1 / +10 %, three frames +1 %,
        // profiles/r5_rect_forward.json: the 128-row tile's K-step is 0.47 us against 0.29, which only pays while it halves the
        // workgroups per CU of a SHORT K walk), hence the row limit.  Over the LONGEST K walk (the vision tower's fc2, K = 5120) it pays
        // (With the EIGHT-wave tile the general limit moved from 320 to 700 rows: 5 / 6 / 7 / 8 questions -6.3 / -8.4 / -2.1 / -1.4 %, one audio
        // segment -1.7 %, profiles/r5_rect700_ab.json; 771 rows = three frames' out-proj is where it starts to lose.)  For K = 5120
        // up to 1536 rows: three / four / five frames -3.8 / -2.6 / -4.1 % (profiles/r5_rect_longk_ab.json); the text tower's fc2
        // (K = 4096, 16 column tiles) loses 2-3 % at 14-18 questions with the same rule, so the limit is by K.
        if (g_gemm_rect && !tail && (M <= g_gemm_rect_rows || (M <= g_gemm_rect_rows_longk && K >= 5120)) && (long)((M + 63) / 64) * (N / 64) > kNumCU &&
            (long)((M + 127) / 128) * (N / 64) <= kNumCU)
            return g_gemm_ring8 ? launch_gemm_ring_rect_epi<128, 64, 4, 2>(A, W, bias, C, M, N, K, epi, st)
                                : launch_gemm_ring_rect_epi<128, 64>(A, W, bias, C, M, N, K, epi, st);
        // well past one 64 x 64 tile per CU (two workgroups side by side on most CUs) but at most one 64 x 128 tile per CU: the
        // 64-row x 128-column ring tile (four waves of 32 x 64) -- six frames' out-proj / fc2 (forward -6.3 %), 24-26 questions' (-2.4 /
        // -4 %); from 450 tiles on: at 432 (22 questions) it loses 3 % (profiles/r5_dispatch_audit_ab.json, LABNOTES_r5 17)
        if (g_gemm_rect64_min_t64 > 0 && !tail && (long)((M + 63) / 64) * (N / 64) >= g_gemm_rect64_min_t64 &&
            (long)((M + 63) / 64) * (N / 128) <= kNumCU)
            return launch_gemm_ring_rect_epi<64, 128>(A, W, bias, C, M, N, K, epi, st);
        // at most one 64x64 tile per CU: the deep-K ring (96 KiB, one workgroup per CU anyway) -- one frame's fc2 24.5 -> 21.9 us,
        // its out-proj 8.6 -> 8.0 (cold weights, tools/deepk_probe.py); with more tiles than CUs two plain-ring workgroups per CU win
        if (g_gemm_deepk && !tail && epi <= HMM_EPI_F32 && (K >> 6) % 2 == 0 && K >= 1024 && (long)((M + 63) / 64) * (N / 64) <= kNumCU)
            return launch_gemm_ringk_epi<64, 3, 2>(A, W, bias, C, M, N, K, epi, st);
        return launch_gemm_ring64_epi(A, W, bias, C, M, N, K, epi, st);
    }
    return launch_gemm_ring128_auto_epi(A, W, bias, C, M, N, K, epi, st);
}

static int launch_gemm_pp_epi(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K,
                              int epi, hipStream_t st) {
    if ((size_t)M * K >= (1ull << 31) || (size_t)N * K >= (1ull << 31))       // 32-bit staging offsets
        return launch_gemm_epi<256, 256, 2, 4>(A, W, bias, C, M, N, K, epi, st);
#define HMM_CALL(E) launch_gemm_pp<E>(A, W, bias, C, M, N, K, st)
    HMM_EPI_SWITCH(HMM_CALL)
#undef HMM_CALL
}
#undef HMM_EPI_SWITCH

int gemm_bf16(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int epi,
              int tile, hipStream_t st) {
    HMM_REQUIRE(A && W && C, HMM_E_INVALID, "gemm: null pointer");
    HMM_REQUIRE(M >= 1 && N >= 128 && K >= 64 && K % 64 == 0 && N % 128 == 0, HMM_E_INVALID,
                "gemm: unsupported shape M=%d N=%d K=%d (need K%%64==0, N%%128==0)", M, N, K);
    HMM_REQUIRE(epi == HMM_EPI_F32 || bias != nullptr, HMM_E_INVALID, "gemm: epilogue %d needs a bias", epi);
    const bool sliver_ok = tile == HMM_GEMM_TILE_AUTO;          // AUTO_TILED, and every named geometry: tiled kernels only
    if (tile < 0) tile = HMM_GEMM_TILE_PP_PEELED;
    const bool pp_ok = N % 256 == 0 && K % 128 == 0;
    if ((tile == HMM_GEMM_TILE_PP_PEELED || tile == HMM_GEMM_TILE_256x256_PP) && !pp_ok) tile = HMM_GEMM_TILE_256x256;
    if (tile == HMM_GEMM_TILE_256x256 && N % 256 != 0) tile = HMM_GEMM_TILE_256x128;
    if (tile == HMM_GEMM_TILE_PP_PEELED) {
        // Tile quantisation: with 256 CUs and one 256x256 tile per CU, T tiles take ceil(T/256) rounds.
        // ViT-H at batch 256 has 257 M-tiles (257 = 256 patches + cls per image), i.e. 5..20 tiles left
        // over for a whole extra round.  Peel the last M-tile(s) off into a small-tile launch when that
        // makes the main launch an exact number of rounds.
        const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
        const long tiles = (long)tiles_m * tiles_n;
        // few 256x256 tiles (cls-only last block, head, small and mid-size batches): smaller tiles put more CUs to work.
        // (Round 5 also sent every launch of at most one 128 x 128 tile per CU here, whatever its count of 256-row tiles -- 24 frames on
        // two chains -4.8 ... -6.2 % in five rechecks, -1.9 % in the one at round 6's final commit: under the 2 % a rule has to buy
        // ON THAT RUN, so it went, profiles/r6_dispatch_recheck.json.)
        if (tiles < t_gemm_small_tiles)
            return sliver_ok && sliver_wins(M, N, K, epi) ? launch_gemm_sliver_epi(A, W, bias, C, M, N, K, epi, st)
                                             : launch_gemm_small_epi(A, W, bias, C, M, N, K, epi, st);
        // peel p <= 2 row tiles when that leaves the main launch with a last round that is full or nearly full (>= 240 of 256
        // CUs) instead of a nearly empty one
        int peel = 0;
        if (tiles > 256 && tiles % 256 != 0 && tiles % 256 <= 64)
            for (int p = 1; p <= 2 && !peel; ++p) {
                const long r = ((long)(tiles_m - p) * tiles_n) % 256;
                if (r == 0 || r >= 240) peel = p;
            }
        if (!peel) return launch_gemm_pp_epi(A, W, bias, C, M, N, K, epi, st);
        const int m_main = (tiles_m - peel) * 256;
        int rc = launch_gemm_pp_epi(A, W, bias, C, m_main, N, K, epi, st);
        if (rc != HMM_OK) return rc;
        if (g_gemm_skip_tail) return rc;                         // probe build only (timing upper bound; results are wrong)
        const bool c_bf16 = epi == HMM_EPI_BIAS_BF16 || epi == HMM_EPI_BIAS_GELU_BF16;
        t_gemm_tail_launch = 1;
        rc = launch_gemm_small_epi(A + (size_t)m_main * K, W, bias,
                                   static_cast<char*>(C) + (size_t)m_main * N * (c_bf16 ? 2 : 4), M - m_main, N, K, epi, st, true);
        t_gemm_tail_launch = 0;
        return rc;
    }
    switch (tile) {
        case HMM_GEMM_TILE_SLIVER:     return launch_gemm_sliver_epi(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_128x128_RING: return launch_gemm_ring128_epi(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_64x64_RING:   return launch_gemm_ring64_epi(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_32x32_RING:   return launch_gemm_ring32_epi(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_64x64_RING_K2: return launch_gemm_ringk_epi<64, 3, 2>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_32x32_RING_K2: return launch_gemm_ringk_epi<32, 4, 2>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_32x32_RING_K4: return launch_gemm_ringk_epi<32, 4, 4>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_128x64_RING:   return launch_gemm_ring_rect_epi<128, 64>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_64x128_RING:   return launch_gemm_ring_rect_epi<64, 128>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_128x128_RING8: return launch_gemm_ring_rect_epi<128, 128, 2, 4>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_128x64_RING8:  return launch_gemm_ring_rect_epi<128, 64, 4, 2>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_64x128_RING8:  return launch_gemm_ring_rect_epi<64, 128, 2, 4>(A, W, bias, C, M, N, K, epi, st);
#ifdef HMM_PROBE
        // timing-only ablations of the eight-wave 128 x 128 ring (bias -> bf16 epilogue), five ring stages, and of the 64 x 64 ring
        case 101: return launch_gemm<128, 128, 2, 4, HMM_EPI_BIAS_BF16, 4, 1, 1>(A, W, bias, C, M, N, K, st);
        case 102: return launch_gemm<128, 128, 2, 4, HMM_EPI_BIAS_BF16, 4, 1, 2>(A, W, bias, C, M, N, K, st);
        case 103: return launch_gemm<128, 128, 2, 4, HMM_EPI_BIAS_BF16, 4, 1, 3>(A, W, bias, C, M, N, K, st);
        case 105: return launch_gemm<128, 128, 2, 4, HMM_EPI_BIAS_BF16, 5>(A, W, bias, C, M, N, K, st);
        case 106: return launch_gemm<128, 128, 2, 4, HMM_EPI_BIAS_BF16, 3>(A, W, bias, C, M, N, K, st);
        case 111: return launch_gemm<64, 64, 2, 2, HMM_EPI_BIAS_BF16, 4, 1, 1>(A, W, bias, C, M, N, K, st);
        case 112: return launch_gemm<64, 64, 2, 2, HMM_EPI_BIAS_BF16, 4, 1, 2>(A, W, bias, C, M, N, K, st);
        case 113: return launch_gemm<64, 64, 2, 2, HMM_EPI_BIAS_BF16, 4, 1, 3>(A, W, bias, C, M, N, K, st);
#endif
        case HMM_GEMM_TILE_256x256_PP: return launch_gemm_pp_epi(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_128x128:    return launch_gemm_epi<128, 128, 2, 2>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_256x128:    return launch_gemm_epi<256, 128, 2, 2>(A, W, bias, C, M, N, K, epi, st);
        case HMM_GEMM_TILE_256x256:    return launch_gemm_epi<256, 256, 2, 4>(A, W, bias, C, M, N, K, epi, st);
    }
    set_error("gemm: unknown tile geometry %d", tile);
    return HMM_E_INVALID;
}

// Deterministic split-K for launches of few rows x a long K (out-proj / fc2 of one frame, one question, one segment): the
// K walk is a chain of latencies -- one frame's fc2 is 100 tiles of 64 x 64 walking 80 K-tiles each on 256 CUs -- so `splits`
// workgroups per tile walk K / splits each and write their fp32 partial products to slab[split][M][N]; the consumer (the
// LayerNorm behind every residual GEMM: launch_layernorm_reduce_bf16) adds the slabs in split order, the bias and the
// residual, so no launch is added and an element's bits depend on (N, K, splits) only -- not on M, the tile geometry or timing.
HMM_TUNABLE(int, g_gemm_splitk_tile, -1)     // probe build: force a ring geometry for the split launches
int gemm_bf16_splitk(const bf16_t* A, const bf16_t* W, float* part, int M, int N, int K, int splits, int tile, hipStream_t st) {
    HMM_REQUIRE(A && W && part, HMM_E_INVALID, "gemm_splitk: null pointer");
    HMM_REQUIRE(M >= 1 && N >= 128 && N % 128 == 0 && splits >= 1 && splits <= 8 && K % (64 * splits) == 0, HMM_E_INVALID,
                "gemm_splitk: unsupported shape M=%d N=%d K=%d splits=%d", M, N, K, splits);
    if (g_gemm_splitk_tile >= 0) tile = g_gemm_splitk_tile;
    const int k_len = K / splits;
    if (tile < 0) {
        // by the number of 64 x 64 workgroups (profiles/r5_splitk_probe.json, cold weights): up to half a chip of them, 32 x 32
        // tiles put four times the workgroups to work (one question's fc2: 11.2 vs 14.3 us); up to one per CU, 64 x 64 tiles with
        // two K-tiles per ring stage (one frame's fc2: 12.6 us, 22 unsplit); beyond that the plain rings;
        const long w64 = (long)((M + 63) / 64) * (N / 64) * splits;
        const bool k2 = k_len % 128 == 0;
        // re-audited with the eight-wave tiles (profiles/r5_splitk_tile_audit.json, r5_splitk_rule2_ab.json): one question's 128 workgroups
        // belong to the 32 x 32 tiles (6.8 vs 7.2 us; the question 0.982 -> 0.972 ms); past ~450 workgroups of 64 x 64 one round of
        // 128 x 64 tiles (six questions 12.0 -> 10.9 us, -1.8 % in the forward; seven / eight questions would need two rounds: they keep
        // 64 x 64), past 640 the eight-wave 128 x 128 tile (nine questions 16.2 -> 14.6 us, -1.3 %)
        if (w64 <= 128) tile = k2 ? HMM_GEMM_TILE_32x32_RING_K2 : HMM_GEMM_TILE_32x32_RING;
        else if (w64 <= 256) tile = k2 ? HMM_GEMM_TILE_64x64_RING_K2 : HMM_GEMM_TILE_64x64_RING;
        else if (w64 > 448 && (long)((M + 127) / 128) * (N / 64) * splits <= kNumCU) tile = HMM_GEMM_TILE_128x64_RING;
        else if (w64 > 640) tile = HMM_GEMM_TILE_128x128_RING8;
        else tile = HMM_GEMM_TILE_64x64_RING;
    }
    t_gemm_splits = splits;
    int rc;
    switch (tile) {
        case HMM_GEMM_TILE_128x128_RING:  rc = launch_gemm<128, 128, 2, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_64x64_RING:    rc = launch_gemm<64, 64, 2, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_32x32_RING:    rc = launch_gemm<32, 32, 2, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_64x64_RING_K2: rc = launch_gemm<64, 64, 2, 2, HMM_EPI_F32, 3, 2>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_32x32_RING_K2: rc = launch_gemm<32, 32, 2, 2, HMM_EPI_F32, 4, 2>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_32x32_RING_K4: rc = launch_gemm<32, 32, 2, 2, HMM_EPI_F32, 4, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_128x64_RING:   rc = launch_gemm<128, 64, 2, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_64x128_RING:   rc = launch_gemm<64, 128, 2, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_128x128_RING8: rc = launch_gemm<128, 128, 2, 4, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_128x64_RING8:  rc = launch_gemm<128, 64, 4, 2, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        case HMM_GEMM_TILE_64x128_RING8:  rc = launch_gemm<64, 128, 2, 4, HMM_EPI_F32, 4>(A, W, nullptr, part, M, N, K, st); break;
        default:
            set_error("gemm_splitk: tile geometry %d has no split-K launch", tile);
            rc = HMM_E_INVALID;
    }
    t_gemm_splits = 1;
    return rc;
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_gemm_bf16_splitk(const uint16_t* a_dev, const uint16_t* w_dev, float* part_dev, int M, int N, int K,
                                       int splits, int tile, hmm_stream_t stream) {
    return gemm_bf16_splitk(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), part_dev, M, N, K,
                            splits, tile, static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_gemm_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev, void* c_dev,
                                int M, int N, int K, int epilogue, hmm_stream_t stream) {
    return gemm_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev, c_dev,
                     M, N, K, epilogue, HMM_GEMM_TILE_AUTO, static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_gemm_bf16_tile(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                     void* c_dev, int M, int N, int K, int epilogue, int tile, hmm_stream_t stream) {
    return gemm_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev, c_dev,
                     M, N, K, epilogue, tile, static_cast<hipStream_t>(stream));
}
