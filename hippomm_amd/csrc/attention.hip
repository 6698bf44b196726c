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

namespace hmm {

typedef __bf16 __attribute__((address_space(3))) * lds_bf16_ptr;

template <int DH, int NKT>
struct AttnCfg {
    static constexpr int KS = DH / 16;              // k-steps of QK^T
    static constexpr int DT = (DH + 31) / 32;       // 32-row d tiles of O^T
    static constexpr int NKEY = NKT * 32;
    static constexpr int KROW = DH * 2 + 16;        // K row stride, bytes
    static constexpr int VROW = 192;                // V row stride, bytes (>= 2*32*DT)
    static constexpr int K_BYTES = NKEY * KROW;
    static constexpr int V_BYTES = NKEY * VROW;
    static constexpr int LDS = K_BYTES + V_BYTES;
    static constexpr int CH = 3;                    // key tiles per online-softmax chunk
    static constexpr int NCH = (NKT + CH - 1) / CH;
    static_assert(DT * 64 <= VROW, "V row must cover every d tile a tr-read touches");
};

constexpr int kAttnWaves = 8;

template <int DH, int NKT>
__global__ __launch_bounds__(kAttnWaves * 64) void attention_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int Lk, int H,
    const float* __restrict__ bias_k, const float* __restrict__ bias_v, float scale_log2e) {
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
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    const size_t row_stride = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * T * row_stride + h * DH;

    const int r = lane & 31, hh = lane >> 5;
    const int nqt = (T + 31) >> 5;

    // this wave's first query fragments: issued before the K/V staging so that they ride along
    bf16x8 qf[C::KS];
    {
        const int qrow = wave * 32 + r;
        const int qr = qrow < T ? qrow : T - 1;
        const bf16_t* qp = base + (size_t)qr * row_stride + hh * 8;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- stage K and V (row-major) into LDS: all global loads first, then the LDS writes ------
    {
        bf16x8 kv[PER_THREAD], vv[PER_THREAD];
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR, c = idx - row * CPR;
            kv[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            vv[i] = kv[i];
            if (row < T) {                               // rows >= T (incl. idx >= NCHUNK) stay zero
                const bf16_t* p = base + (size_t)row * row_stride + c * 8;
                kv[i] = *reinterpret_cast<const bf16x8*>(p + D);
                vv[i] = *reinterpret_cast<const bf16x8*>(p + 2 * D);
            }
        }
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR, c = idx - row * CPR;
            if (idx < NCHUNK && !(bias_k != nullptr && row == T)) {
                *reinterpret_cast<bf16x8*>(k_lds + row * C::KROW + c * 16) = kv[i];
                *reinterpret_cast<bf16x8*>(v_lds + row * C::VROW + c * 16) = vv[i];
            }
        }
        if (bias_k != nullptr && tid < DH) {             // the add_bias_kv position: row T
            *reinterpret_cast<bf16_t*>(k_lds + T * C::KROW + tid * 2) = (bf16_t)bias_k[h * DH + tid];
            *reinterpret_cast<bf16_t*>(v_lds + T * C::VROW + tid * 2) = (bf16_t)bias_v[h * DH + tid];
        }
    }
    __syncthreads();

    // per-lane LDS bases
    const char* k_base = k_lds + r * C::KROW + hh * 16;
    // tr-read: 16-lane group g = lane>>4 reads a 4-key x 16-d block; lane (4q+p) of the group supplies
    // row q, columns 4p..4p+3, and receives column (lane&15) of the 4 rows.
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const char* v_base = v_lds + (4 * (g >> 1) + q4) * C::VROW + (16 * (g & 1) + 4 * p4) * 2;

    for (int qt = wave; qt < nqt; qt += kAttnWaves) {
        const int qrow = qt * 32 + r;
        if (qt != wave) {                                   // later tiles reload their queries
            const int qr = qrow < T ? qrow : T - 1;
            const bf16_t* qp = base + (size_t)qr * row_stride + hh * 8;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
        }
        f32x16 o[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x16{};
        float m_run = -INFINITY, l_run = 0.f;

#pragma unroll
        for (int ch = 0; ch < C::NCH; ++ch) {
            constexpr int CHT = C::CH;
            f32x16 s[CHT];
            // S^T tiles of this chunk
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
                    f32x16 acc = {};
                    bf16x8 kf[C::KS];
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
                        kf[ks] = *reinterpret_cast<const bf16x8*>(k_base + kt * 32 * C::KROW + ks * 32);
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
                    s[i] = acc;
                    __builtin_amdgcn_sched_barrier(0);     // one tile's K fragments live at a time
                }
            }
            // mask keys >= Lk (only tiles that can contain them), chunk max
            float mc = -INFINITY;
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
                    const bool may_mask = (kt + 1) * 32 > Lk;        // wave-uniform
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        if (may_mask) {
                            const int key = kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                            if (key >= Lk) s[i][reg] = -INFINITY;
                        }
                        mc = fmaxf(mc, s[i][reg]);
                    }
                }
            }
            mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
            const float m_new = fmaxf(m_run, mc);
            // a chunk whose keys are all masked (mc = -inf while m_run = -inf) cannot occur: chunk 0
            // always holds key 0.  Later all-masked chunks keep m_new = m_run (finite).
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            const float neg_m = -m_new * scale_log2e;
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[dt][reg] *= alpha;
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[i][reg], scale_log2e, neg_m));
                        s[i][reg] = p;
                        l_run += p;
                    }
                }
            }
            // O^T += V^T P^T for the chunk: accumulator registers 8*st..8*st+7 of tile kt are the B
            // fragment of k-step st (slot j of lane half hh = key 16*st + 8*(j>>2) + 4*hh + (j&3)).
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        bf16x8 pf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[i][8 * st + j];
#pragma unroll
                        for (int dt = 0; dt < C::DT; ++dt) {
                            const char* vp = v_base + (kt * 32 + 16 * st) * C::VROW + dt * 64;
                            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
                            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8 * C::VROW));
                            const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        const float l = l_run + __shfl_xor(l_run, 32, 64);
        if (qrow < T) {
            const float inv_l = 1.0f / l;
            bf16_t* op = out + ((size_t)b * T + qrow) * D + h * DH;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d0 = dt * 32 + 8 * gq + 4 * hh;
                    if (d0 < DH) {
                        bf16x4 o4 = {(bf16_t)(o[dt][4 * gq + 0] * inv_l), (bf16_t)(o[dt][4 * gq + 1] * inv_l),
                                     (bf16_t)(o[dt][4 * gq + 2] * inv_l), (bf16_t)(o[dt][4 * gq + 3] * inv_l)};
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
    kern<<<batch * H, kAttnWaves * 64, C::LDS, st>>>(qkv, out, T, Lk, H, bias_k, bias_v, scale_log2e);
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
