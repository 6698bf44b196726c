// Multi-head self-attention core for the ImageBind towers on gfx950 (replaces the SDPA inside
// nn.MultiheadAttention; sequence is short: 257 vision tokens / 229+1 audio tokens).
//
// One workgroup (4 waves) per (sample, head).  The whole K and V of the head live in LDS
// (K row-major with a 16-B pad per row, V transposed [d][key]), each wave owns 32-query tiles.
// The score tile is computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16 with K as the
// A-operand), so a lane holds one query column and its keys sit in registers: the softmax
// max / sum are register-local plus one exchange with lane^32, and the exponentiated
// accumulator, converted to bf16 in place, IS the B-operand of the second product
// O^T = V^T P^T (no LDS round trip for P).  All 288 (256) keys fit in registers, so the softmax
// is exact two-pass, not online.  fp32 scores, statistics and output accumulation; P and the
// output are rounded to bf16 once.
#include "hmm_common.h"
#include "encoder_ops.h"

namespace hmm {

template <int DH, int NKT>
struct AttnCfg {
    static constexpr int KS = DH / 16;              // k-steps of QK^T
    static constexpr int DT = (DH + 31) / 32;       // 32-row d tiles of O^T
    static constexpr int NKEY = NKT * 32;
    static constexpr int KROW = DH * 2 + 16;        // K row stride, bytes: odd multiple of 16 -> conflict-free b128
    static constexpr int VROW = NKEY * 2 + 8;       // V^T row stride, bytes: conflict-free b64 column reads
    static constexpr int K_BYTES = NKEY * KROW;
    static constexpr int V_BYTES = DT * 32 * VROW;
    static constexpr int LDS = K_BYTES + V_BYTES;
};

template <int DH, int NKT>
__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                        int T, int Lk, int H,
                                                        const float* __restrict__ bias_k,
                                                        const float* __restrict__ bias_v, float scale_log2e) {
    using C = AttnCfg<DH, NKT>;
    constexpr int CPR = DH / 8;                      // 16-B chunks per K/V row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_lds = smem;
    char* vt_lds = smem + C::K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    const size_t row_stride = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * T * row_stride + h * DH;

    // ---- stage K (row-major) and V (transposed) into LDS; rows >= Lk are zero -------------
    for (int idx = tid; idx < C::NKEY * CPR; idx += 256) {
        const int row = idx / CPR, c = idx - row * CPR;
        bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (row < T) {
            const bf16_t* p = base + (size_t)row * row_stride + c * 8;
            kv = *reinterpret_cast<const bf16x8*>(p + D);
            vv = *reinterpret_cast<const bf16x8*>(p + 2 * D);
        } else if (row < Lk) {                       // the add_bias_kv position
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kv[e] = (bf16_t)bias_k[h * DH + c * 8 + e];
                vv[e] = (bf16_t)bias_v[h * DH + c * 8 + e];
            }
        }
        *reinterpret_cast<bf16x8*>(ks_lds + row * C::KROW + c * 16) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *reinterpret_cast<bf16_t*>(vt_lds + (c * 8 + e) * C::VROW + row * 2) = vv[e];
    }
    if constexpr (C::DT * 32 > DH) {                 // d rows DH .. 32*DT-1 of V^T: zeros
        constexpr int words = (C::DT * 32 - DH) * C::VROW / 4;
        uint32_t* z = reinterpret_cast<uint32_t*>(vt_lds + DH * C::VROW);
        for (int i = tid; i < words; i += 256) z[i] = 0u;
    }
    __syncthreads();

    const int r = lane & 31, hh = lane >> 5;
    const int nqt = (T + 31) >> 5;
    const bool mask_last_only = Lk > (NKT - 1) * 32;

    for (int qt = wave; qt < nqt; qt += 4) {
        const int qrow = qt * 32 + r;
        const int qr = qrow < T ? qrow : T - 1;
        const bf16_t* qp = base + (size_t)qr * row_stride + hh * 8;
        bf16x8 qf[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);

        // S^T tiles: rows = keys (registers), column = this lane's query
        f32x16 s[NKT];
        const char* k_base = ks_lds + r * C::KROW + hh * 16;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 acc = {};
            bf16x8 kf[C::KS];
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
                kf[ks] = *reinterpret_cast<const bf16x8*>(k_base + kt * 32 * C::KROW + ks * 32);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
            s[kt] = acc;
            __builtin_amdgcn_sched_barrier(0);      // keep one tile's K fragments live at a time
        }

        // mask padded keys, row max
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const bool need_mask = (kt == NKT - 1) || !mask_last_only;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (need_mask) {
                    const int key = kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    if (key >= Lk) s[kt][reg] = -INFINITY;
                }
                m = fmaxf(m, s[kt][reg]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));

        // p = exp((s - m) / sqrt(dh)), l = sum p
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float p = __builtin_amdgcn_exp2f((s[kt][reg] - m) * scale_log2e);
                s[kt][reg] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);

        // O^T = V^T P^T : the accumulator registers 8*st .. 8*st+7 of S^T tile kt are the B fragment
        // of k-step st; slot j of lane half hh is key 16*st + 8*(j>>2) + 4*hh + (j&3) of the tile.
        f32x16 o[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x16{};
        const char* v_base = vt_lds + r * C::VROW + hh * 8;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * st + j];
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const char* vp = v_base + dt * 32 * C::VROW + kt * 64 + st * 32;
                    const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vp);
                    const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vp + 16);
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if (qrow < T) {
            const float inv_l = 1.0f / l;
            bf16_t* op = out + ((size_t)b * T + qrow) * D + h * DH;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = dt * 32 + 8 * g + 4 * hh;
                    if (d0 < DH) {
                        bf16x4 o4 = {(bf16_t)(o[dt][4 * g + 0] * inv_l), (bf16_t)(o[dt][4 * g + 1] * inv_l),
                                     (bf16_t)(o[dt][4 * g + 2] * inv_l), (bf16_t)(o[dt][4 * g + 3] * inv_l)};
                        *reinterpret_cast<bf16x4*>(op + d0) = o4;
                    }
                }
        }
    }
}

template <int DH, int NKT>
static int launch_attention(const bf16_t* qkv, bf16_t* out, int batch, int T, int Lk, int H,
                            const float* bias_k, const float* bias_v, hipStream_t st) {
    using C = AttnCfg<DH, NKT>;
    auto kern = attention_kernel<DH, NKT>;
    static bool attr_set = false;
    if (!attr_set) {
        HMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_set = true;
    }
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    kern<<<batch * H, 256, C::LDS, st>>>(qkv, out, T, Lk, H, bias_k, bias_v, scale_log2e);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int attention_bf16(const bf16_t* qkv, bf16_t* out, int batch, int tokens, int heads, int head_dim,
                   const float* bias_k, const float* bias_v, hipStream_t st) {
    HMM_REQUIRE(qkv && out, HMM_E_INVALID, "attention: null pointer");
    HMM_REQUIRE((bias_k == nullptr) == (bias_v == nullptr), HMM_E_INVALID, "attention: bias_k and bias_v go together");
    HMM_REQUIRE(batch >= 1 && tokens >= 1 && heads >= 1, HMM_E_INVALID, "attention: bad shape");
    const int Lk = tokens + (bias_k ? 1 : 0);
    if (head_dim == 80 && Lk <= 288) return launch_attention<80, 9>(qkv, out, batch, tokens, Lk, heads, bias_k, bias_v, st);
    if (head_dim == 64 && Lk <= 256) return launch_attention<64, 8>(qkv, out, batch, tokens, Lk, heads, bias_k, bias_v, st);
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
