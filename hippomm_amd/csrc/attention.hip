// Multi-head self-attention core for the ImageBind towers on gfx950 (replaces the SDPA inside
// nn.MultiheadAttention; sequence is short: 257 vision tokens / 229+1 audio tokens).
//
// One workgroup (8 waves, two per SIMD) per (sample, head).  The whole K and V of the head are
// staged once into LDS, both ROW-major: K rows padded to an odd multiple of 16 B (conflict-free
// ds_read_b128 of the A-operand), V rows at a 192-B stride (the four rows a ds_read_b64_tr_b16
// block touches fall in disjoint bank windows).  Each wave owns 32-query tiles.
//
// The score tile is computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16, K as the A-operand),
// so a lane holds one query column with its keys in registers: max / sum are register-local plus
// one exchange with lane^32, and the exponentiated accumulator, converted to bf16 in place, IS the
// B-operand of the second product O^T = V^T P^T (no LDS round trip for P).  V^T fragments come
// straight from the row-major V image through the transposing LDS read.  Keys are processed in
// chunks of 3 tiles (96 keys) with an online softmax (running max / sum, O rescaled per chunk),
// which keeps a wave under 256 VGPRs so that two waves share each SIMD: while one is in its
// exp / max / sum VALU section, its partner issues MFMAs.
// fp32 scores, statistics and output accumulation; P and the output are rounded to bf16 once.
#include "hmm_common.h"
#include "encoder_ops.h"
#include "attention_core.h"

namespace hmm {

HMM_TUNABLE(int, g_attn_short_keys, 1)   // probe build: 0 = the text tower's 77 keys on the eight-tile instantiation (A/B, bit equality)
HMM_TUNABLE(int, g_attn_q_split, 1)      // probe build: 0 = never split a (sample, head)'s queries over workgroups (A/B)
HMM_TUNABLE(int, g_attn_even_map, 1)    // the (sample, head) list in eight equal contiguous runs, one per XCD, no padding workgroups (as qkv_attention.hip): 20 / 28 / 36 frames -1.1 / -0.7 / -0.8 %; the text tower (T = 77) keeps sample i on XCD i % 8 (+0.3 ... +0.6 % with the even map; profiles/r5_attn_even_map_ab.json)
HMM_TUNABLE(int, g_attn_q_split_wgs, 256) // the split is taken while twice the launch's workgroups are at most this many (512: 9-12 and 20-32 frames +2.4 ... +4.1 %)

template <int DH, int NKT>
__global__ __launch_bounds__(kAttnWaves * 64) void attention_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n_img, int T, int Lk, int H,
    const float* __restrict__ bias_k, const float* __restrict__ bias_v, float scale_log2e, int causal, int q_parts, int even_map) {
    using C = AttnCfg<DH, NKT>;
    constexpr int NT = kAttnWaves * 64;
    constexpr int CPR = DH / 8;                      // 16-B chunks per K/V row
    constexpr int NCHUNK = C::NKEY * CPR;
    constexpr int PER_THREAD = (NCHUNK + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + C::K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD x = blockIdx % 8 takes the images x, x+8, ...: the 16 heads of an image then run on one XCD at about the same
    // time, and a head's 160-B slice of a qkv row shares its 128-B lines with its neighbours' in that XCD's L2.
    // The images are walked backwards: the producer (QKV GEMM) wrote the last images last, so they are the
    // ones still resident in the 256-MiB Infinity Cache when this kernel starts.
    // causal (text tower): key j is visible to query i iff j <= i.
    // q_parts > 1 (few samples): consecutive blocks are the query parts of one (sample, head)
    const int q_part = blockIdx.x % q_parts;
    const int blk = blockIdx.x / q_parts;
    int b_lin, h;
    if (even_map) {                                      // q_parts == 1, grid = n_img x H: eight contiguous, equally long runs of the
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;     // (sample, head) list, one per XCD (qkv_attention.hip)
        const int pair = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
        b_lin = pair / H;
        h = pair % H;
    } else {
        const int n = blk >> 3;
        b_lin = (blk & 7) + 8 * (n / H);
        h = n % H;
        if (b_lin >= n_img) return;
    }
    const int b = n_img - 1 - b_lin;
    const int D = H * DH;
    const size_t row_stride = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * T * row_stride + h * DH;

    const int r = lane & 31, hh = lane >> 5;

    // this wave's first query fragments: issued before the K/V staging so that they ride along
    bf16x8 qf[C::KS];
    {
        const int qrow = (q_part + q_parts * wave) * 32 + r;
        const int qr = qrow < T ? qrow : T - 1;
        const bf16_t* qp = base + (size_t)qr * row_stride + hh * 8;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- stage K and V (row-major) into LDS: all global loads first, then the LDS writes ------
    {
        uint4 kv[PER_THREAD], vv[PER_THREAD];            // raw 16-B chunks (8 bf16): no per-element handling
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR, c = idx - row * CPR;
            const int rr = row < T ? row : T - 1;        // clamp: always a valid address; rows >= T are not written
            const bf16_t* p = base + (size_t)rr * row_stride + c * 8;
            kv[i] = *reinterpret_cast<const uint4*>(p + D);
            vv[i] = *reinterpret_cast<const uint4*>(p + 2 * D);
        }
        // rows T .. NKEY-1: zeros (the bias row, if any, is filled after the barrier)
        for (int idx = T * CPR + tid; idx < NCHUNK; idx += NT) {
            const int row = idx / CPR, c = idx - row * CPR;
            *reinterpret_cast<uint4*>(k_lds + row * C::KROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(v_lds + row * C::VROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR, c = idx - row * CPR;
            if (row < T) {
                *reinterpret_cast<uint4*>(k_lds + row * C::KROW + c * 16) = kv[i];
                *reinterpret_cast<uint4*>(v_lds + row * C::VROW + c * 16) = vv[i];
            }
        }
        __syncthreads();
        if (bias_k != nullptr && tid < DH) {             // the add_bias_kv position: row T
            *reinterpret_cast<bf16_t*>(k_lds + T * C::KROW + tid * 2) = (bf16_t)bias_k[h * DH + tid];
            *reinterpret_cast<bf16_t*>(v_lds + T * C::VROW + tid * 2) = (bf16_t)bias_v[h * DH + tid];
        }
    }
    __syncthreads();

    auto load_q = [&](int row, int ks) {
        return *reinterpret_cast<const bf16x8*>(base + (size_t)row * row_stride + hh * 8 + ks * 16);
    };
    attention_core<DH, NKT>(k_lds, v_lds, reinterpret_cast<float*>(smem + C::K_BYTES + C::V_BYTES), load_q, qf,
                            out + (size_t)b * T * D + h * DH, T, Lk, D, scale_log2e, causal, q_part, q_parts);
}

template <int DH, int NKT>
static int launch_attention(const bf16_t* qkv, bf16_t* out, int batch, int T, int Lk, int H,
                            const float* bias_k, const float* bias_v, hipStream_t st, bool causal) {
    using C = AttnCfg<DH, NKT>;
    auto kern = attention_kernel<DH, NKT>;
    HMM_ENSURE_DYN_LDS(kern, C::LDS);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    const int wgs = 8 * ((batch + 7) / 8) * H;
    // a handful of samples: the query tiles of a (sample, head) go to TWO workgroups while the launch still fits the chip (each
    // part stages the head's K and V itself: 82 KB from L2); same bits (attention_core.h).  Measured (tools/text_latency_probe.py):
    // one frame 2.46 -> 2.39 ms, one audio segment 0.712 -> 0.696; four parts are slower (2.63 ms), and so is any split of the
    // text tower's three query tiles (one question 1.04 -> 1.15 ms), hence T > 128.
    const int q_parts = (g_attn_q_split && T > 128 && wgs * 2 <= g_attn_q_split_wgs) ? 2 : 1;
    const int even_map = g_attn_even_map && q_parts == 1 && T > 128;
    kern<<<even_map ? batch * H : wgs * q_parts, kAttnWaves * 64, C::LDS, st>>>(qkv, out, batch, T, Lk, H, bias_k, bias_v, scale_log2e, causal ? 1 : 0,
                                                                                 q_parts, even_map);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}


// Single-query attention for the LAST transformer block: the head consumes only token 0
// (SelectElement(index=0)), so only the cls query of each image needs an attention output there.
// One 256-thread workgroup per (image, head): thread j scores key j (a 2*dh-byte row of K straight
// from HBM), block softmax through LDS, then the four waves accumulate disjoint key subsets of P.V.
// VALU only: 2*Lk*dh MACs per (image, head); bound by the one read of K and V (HBM).
template <int DH>
__global__ __launch_bounds__(256) void attention_cls_kernel(const bf16_t* __restrict__ q_cls, const bf16_t* __restrict__ kv,
                                                            bf16_t* __restrict__ out, int T, int Lk, int H,
                                                            const float* __restrict__ bias_k,
                                                            const float* __restrict__ bias_v, float scale) {
    constexpr int MAXK = 320;
    __shared__ float qs[DH];
    __shared__ float p[MAXK];
    __shared__ float red[8];
    __shared__ float opart[4 * (64 / (DH / 8))][DH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    const size_t kv_stride = (size_t)2 * D;
    const bf16_t* kbase = kv + (size_t)b * T * kv_stride + h * DH;
    if (tid < DH) qs[tid] = (float)q_cls[(size_t)b * D + h * DH + tid];
    __syncthreads();

    float local_max = -INFINITY;
    for (int j = tid; j < Lk; j += 256) {
        float s = 0.f;
        if (j < T) {
            const bf16_t* kr = kbase + (size_t)j * kv_stride;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                const bf16x8 kk = *reinterpret_cast<const bf16x8*>(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qs[c * 8 + e], (float)kk[e], s);
            }
        } else {
            for (int d = 0; d < DH; ++d) s = fmaf(qs[d], (float)(bf16_t)bias_k[h * DH + d], s);
        }
        s *= scale;
        p[j] = s;
        local_max = fmaxf(local_max, s);
    }
    local_max = wave_max(local_max);
    if (lane == 0) red[w] = local_max;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float local_sum = 0.f;
    for (int j = tid; j < Lk; j += 256) {
        const float e = __expf(p[j] - m);
        p[j] = e;
        local_sum += e;
    }
    local_sum = wave_sum(local_sum);
    if (lane == 0) red[4 + w] = local_sum;
    __syncthreads();
    const float l = (red[4] + red[5]) + (red[6] + red[7]);

    // P.V with 16-B loads: a lane owns one 8-wide d chunk (c) of one key slot; a wave covers SL keys per
    // step, the 4 waves 4*SL.  Partial sums are reduced over slots and waves through LDS.
    constexpr int CPR = DH / 8, SL = 64 / CPR;
    const int slot = lane / CPR, c = lane - slot * CPR;
    const bf16_t* vbase = kbase + D;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (slot < SL) {
#pragma unroll 4
        for (int j = w * SL + slot; j < Lk; j += 4 * SL) {
            const float pj = p[j];
            bf16x8 vv;
            if (j < T) {
                vv = *reinterpret_cast<const bf16x8*>(vbase + (size_t)j * kv_stride + c * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = (bf16_t)bias_v[h * DH + c * 8 + e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vv[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) opart[w * SL + slot][c * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < DH) {
        float o = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * SL; ++i) o += opart[i][tid];
        out[(size_t)b * D + h * DH + tid] = (bf16_t)(o / l);
    }
}

int attention_cls_bf16(const bf16_t* q_cls, const bf16_t* kv, bf16_t* out, int batch, int tokens, int heads,
                       int head_dim, const float* bias_k, const float* bias_v, hipStream_t st) {
    HMM_REQUIRE(q_cls && kv && out, HMM_E_INVALID, "attention_cls: null pointer");
    const int Lk = tokens + (bias_k ? 1 : 0);
    HMM_REQUIRE(Lk <= 320, HMM_E_INVALID, "attention_cls: %d keys exceed 320", Lk);
    const float scale = 1.0f / sqrtf((float)head_dim);
    if (head_dim == 80)
        attention_cls_kernel<80><<<batch * heads, 256, 0, st>>>(q_cls, kv, out, tokens, Lk, heads, bias_k, bias_v, scale);
    else if (head_dim == 64)
        attention_cls_kernel<64><<<batch * heads, 256, 0, st>>>(q_cls, kv, out, tokens, Lk, heads, bias_k, bias_v, scale);
    else {
        set_error("attention_cls: unsupported head_dim %d", head_dim);
        return HMM_E_INVALID;
    }
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int attention_bf16(const bf16_t* qkv, bf16_t* out, int batch, int tokens, int heads, int head_dim,
                   const float* bias_k, const float* bias_v, hipStream_t st, bool causal) {
    HMM_REQUIRE(qkv && out, HMM_E_INVALID, "attention: null pointer");
    HMM_REQUIRE((bias_k == nullptr) == (bias_v == nullptr), HMM_E_INVALID, "attention: bias_k and bias_v go together");
    HMM_REQUIRE(batch >= 1 && tokens >= 1 && heads >= 1, HMM_E_INVALID, "attention: bad shape");
    const int Lk = tokens + (bias_k ? 1 : 0);
    HMM_REQUIRE(!(causal && bias_k), HMM_E_INVALID, "attention: causal mask with bias_kv is not defined here");
    if (head_dim == 80 && Lk <= 288) return launch_attention<80, 9>(qkv, out, batch, tokens, Lk, heads, bias_k, bias_v, st, causal);
    // 77 text tokens fit three key tiles: the eight-tile instantiation would compute five fully masked tiles per query tile (their
    // probabilities are exactly 0 and their rescale factors exactly 1, so both instantiations give the same bits)
    if (g_attn_short_keys && head_dim == 64 && Lk <= 96) return launch_attention<64, 3>(qkv, out, batch, tokens, Lk, heads, bias_k, bias_v, st, causal);
    if (head_dim == 64 && Lk <= 256) return launch_attention<64, 8>(qkv, out, batch, tokens, Lk, heads, bias_k, bias_v, st, causal);
    set_error("attention: unsupported head_dim=%d / keys=%d (built: 80 x <=288 keys, 64 x <=256 keys)", head_dim, Lk);
    return HMM_E_INVALID;
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_attention_bf16(const uint16_t* qkv_dev, uint16_t* out_dev, int batch, int tokens, int heads,
                                     int head_dim, const float* bias_k_dev, const float* bias_v_dev,
                                     hmm_stream_t stream) {
    return attention_bf16(reinterpret_cast<const bf16_t*>(qkv_dev), reinterpret_cast<bf16_t*>(out_dev), batch, tokens,
                          heads, head_dim, bias_k_dev, bias_v_dev, static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_attention_causal_bf16(const uint16_t* qkv_dev, uint16_t* out_dev, int batch, int tokens, int heads,
                                             int head_dim, hmm_stream_t stream) {
    return attention_bf16(reinterpret_cast<const bf16_t*>(qkv_dev), reinterpret_cast<bf16_t*>(out_dev), batch, tokens,
                          heads, head_dim, nullptr, nullptr, static_cast<hipStream_t>(stream), true);
}
