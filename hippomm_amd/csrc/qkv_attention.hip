// Fused QKV projection + attention for the ImageBind-huge VISION tower on gfx950 (D = 1280, 16 heads x 80, 257 tokens).
// Replaces, inside one transformer block, nn.MultiheadAttention's in_proj + scaled-dot-product attention
// (upstream imagebind/models/transformer.py, called through foundation_models.py:131); the out_proj stays a GEMM.
//
// Why: as two kernels the packed qkv matrix (505 MB at batch 256) is written by the projection's epilogue and read
// back by the attention's staging -- ~85 us of output burst plus ~80 us of staging per block -- although one
// (image, head) pair needs exactly a 257 x 240 slice of it, which fits in LDS.
//
// One workgroup (8 waves) per (image, head):
//   1. the 256 patch rows of the image times the head's 240 rows of in_proj_weight (q | k | v slices; the tile's last 16
//      columns do not exist and their MFMAs are skipped) is ONE 256 x 256 x 1280 tile of the ping-pong GEMM: pp_mainloop()
//      is the stand-alone kernel's K loop, bit for bit (gemm_pp_mainloop.h); only the staging source addresses differ;
//   2. accumulators + in_proj_bias -> bf16 -> the Q / K / V images in LDS (the staging buffers are dead by then), in
//      the layouts the attention core reads; the cls row (token 0), which does not fit the 256-row tile, comes from a
//      small GEMM over the cls rows of all images (qkv_cls, computed by the caller with the same weights);
//   3. attention_core() -- the stand-alone attention kernel's compute, bit for bit (attention_core.h) -- with the
//      query fragments read from the LDS image; the head's 257 x 80 output goes to HBM.
// Results are therefore bitwise equal to hmm_op_gemm_bf16(EPI_BIAS_BF16) followed by hmm_op_attention_bf16
// (tests/test_gpu_ops.py::test_fused_qkv_attention_equals_gemm_plus_attention).
// Blocks are dealt XCD-aware like the attention kernel's: the 16 heads of an image run on one XCD at about the same
// time and share the image's A panel (655 KB) in that XCD's L2.
#include "hmm_common.h"
#include "encoder_ops.h"
#include "gemm_pp_mainloop.h"
#include "attention_core.h"

namespace hmm {

constexpr int kFD = 1280, kFH = 16, kFDH = 80, kFT = 257, kFNKT = 9;
using FusedCfg = AttnCfg<kFDH, kFNKT>;
constexpr int kFQRow = kFDH * 2 + 16;                                  // Q image row stride (as K rows: conflict-free b128)
constexpr int kFQOff = FusedCfg::K_BYTES + FusedCfg::V_BYTES;          // Q image behind K and V
constexpr int kFXOff = kFQOff + ((kFT * kFQRow + 15) / 16) * 16;       // cooperative-query scratch
constexpr int kFLds = kFXOff + FusedCfg::X_BYTES;                      // 154,176 B (> the 131,072 B of GEMM staging)
static_assert(kFLds >= 2 * kPPTile && kFLds <= 160 * 1024, "LDS plan of the fused kernel");

// LNF (folded LayerNorm, encoder.hip): `a` is the bf16 image of the residual stream, `w` = bf16(gamma (.) in_proj_weight),
// `bias` = c2 and the projection value is fma(rs.x, acc, fma(-rs.y, c1, c2)) with rs[row] = (rstd, rstd * mean) -- the same
// expression as gemm_bf16's HMM_EPI_LN_BF16 epilogue, so the two routes stay bitwise equal.
template <bool LNF>
__global__ __launch_bounds__(512) void qkv_attention_kernel(
    const bf16_t* __restrict__ a /* [n_img*257][1280] LayerNorm output */, const bf16_t* __restrict__ w /* [3840][1280] */,
    const float* __restrict__ bias /* [3840] */, const bf16_t* __restrict__ qkv_cls /* [n_img][3840] */,
    bf16_t* __restrict__ out /* [n_img*257][1280] */, int n_img, float scale_log2e,
    const float2* __restrict__ rs /* [n_img*257] */, const float* __restrict__ c1 /* [3840] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n = blockIdx.x >> 3;
    const int b_lin = (blockIdx.x & 7) + 8 * (n / kFH);
    const int h = n % kFH;
    if (b_lin >= n_img) return;
    const int b = n_img - 1 - b_lin;

    // ---- 1. projection tile: rows = the image's 256 patch tokens, columns = [q_h | k_h | v_h | 16 unused] -------------
    PPSources src;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wave + 8 * j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int arow = (lr & 63) + (lr >> 6) * 128;
        const int tok = b * kFT + 1 + arow;                              // token 0 is the cls row (handled below)
        src.alo[j] = tok * kFD + c * 8;
        src.ahi[j] = (tok + 64) * kFD + c * 8;
        const int bcol = (lr >> 5) * 64 + (lr & 31);
        auto wrow = [&](int col) {                                       // tile column -> row of in_proj_weight
            const int part = col / kFDH, d = col - part * kFDH;
            return part < 3 ? part * kFD + h * kFDH + d : h * kFDH;      // columns 240..255: any valid row, results unused
        };
        src.blo[j] = wrow(bcol) * kFD + c * 8;
        src.bhi[j] = wrow(bcol + 32) * kFD + c * 8;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    pp_mainloop<true>(a, w, src, kFD >> 6, smem, lane, wave, acc);     // columns 240..255: no MFMAs

    // ---- 2. accumulators -> Q / K / V images (the staging buffers are dead: see pp_mainloop) --------------------------
    char* k_lds = smem;
    char* v_lds = smem + FusedCfg::K_BYTES;
    char* q_lds = smem + kFQOff;
    {
        const int fr = lane & 15, fq = lane >> 4;
        float2 rsv[8];
        if constexpr (LNF) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) rsv[mi] = rs[(size_t)b * kFT + 1 + wm * 128 + mi * 16 + fr];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + 4 * fq;                  // 4 consecutive tile columns, never across a part
            const int part = col / kFDH, d = col - part * kFDH;
            if (part < 3) {                                              // wave-uniform per (wn, ni) except the last group
                const float4 bv = *reinterpret_cast<const float4*>(bias + part * kFD + h * kFDH + d);
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (LNF) cv = *reinterpret_cast<const float4*>(c1 + part * kFD + h * kFDH + d);
                char* img = part == 0 ? q_lds : (part == 1 ? k_lds : v_lds);
                const int stride = part == 0 ? kFQRow : (part == 1 ? FusedCfg::KROW : FusedCfg::VROW);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int tok = 1 + wm * 128 + mi * 16 + fr;
                    const f32x4 v = acc[mi][ni];
                    bf16x4 o4;
                    if constexpr (LNF)
                        o4 = bf16x4{(bf16_t)ln_fold(v[0], rsv[mi], cv.x, bv.x), (bf16_t)ln_fold(v[1], rsv[mi], cv.y, bv.y),
                                    (bf16_t)ln_fold(v[2], rsv[mi], cv.z, bv.z), (bf16_t)ln_fold(v[3], rsv[mi], cv.w, bv.w)};
                    else
                        o4 = bf16x4{(bf16_t)(v[0] + bv.x), (bf16_t)(v[1] + bv.y), (bf16_t)(v[2] + bv.z), (bf16_t)(v[3] + bv.w)};
                    *reinterpret_cast<bf16x4*>(img + tok * stride + d * 2) = o4;
                }
            }
        }
        // token 0 (cls): its q | k | v slices come from the caller's small GEMM; 3 parts x 10 chunks of 16 B
        if (tid < 30) {
            const int part = tid / 10, ch = tid - part * 10;
            const uint4 v = *reinterpret_cast<const uint4*>(qkv_cls + (size_t)b * (3 * kFD) + part * kFD + h * kFDH + ch * 8);
            char* img = part == 0 ? q_lds : (part == 1 ? k_lds : v_lds);
            *reinterpret_cast<uint4*>(img + ch * 16) = v;
        }
        // key rows T .. NKEY-1: zeros
        constexpr int CPR = kFDH / 8;
        for (int idx = kFT * CPR + tid; idx < FusedCfg::NKEY * CPR; idx += 512) {
            const int row = idx / CPR, c = idx - row * CPR;
            *reinterpret_cast<uint4*>(k_lds + row * FusedCfg::KROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(v_lds + row * FusedCfg::VROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();

    // ---- 3. attention on the images ------------------------------------------------------------------------------------
    const int r = lane & 31, hh = lane >> 5;
    auto load_q = [&](int row, int ks) {
        return *reinterpret_cast<const bf16x8*>(q_lds + row * kFQRow + hh * 16 + ks * 32);
    };
    bf16x8 qf[FusedCfg::KS];
    {
        const int qrow = wave * 32 + r;
        const int qr = qrow < kFT ? qrow : kFT - 1;
#pragma unroll
        for (int ks = 0; ks < FusedCfg::KS; ++ks) qf[ks] = load_q(qr, ks);
    }
    attention_core<kFDH, kFNKT>(k_lds, v_lds, reinterpret_cast<float*>(smem + kFXOff), load_q, qf,
                                out + (size_t)b * kFT * kFD + h * kFDH, kFT, kFT, kFD, scale_log2e, false);
}

int qkv_attention_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* qkv_cls, bf16_t* out,
                       int n_img, hipStream_t st, const float2* rs, const float* c1) {
    HMM_REQUIRE(a && w && bias && qkv_cls && out, HMM_E_INVALID, "qkv_attention: null pointer");
    HMM_REQUIRE((rs == nullptr) == (c1 == nullptr), HMM_E_INVALID, "qkv_attention: row statistics and c1 go together");
    HMM_REQUIRE(n_img >= 1 && (int64_t)n_img * kFT * kFD < (1ll << 31), HMM_E_INVALID, "qkv_attention: n_img=%d out of range", n_img);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)kFDH);
    const int grid = 8 * ((n_img + 7) / 8) * kFH;
    if (rs) {
        HMM_ENSURE_DYN_LDS(qkv_attention_kernel<true>, kFLds);
        qkv_attention_kernel<true><<<grid, 512, kFLds, st>>>(a, w, bias, qkv_cls, out, n_img, scale_log2e, rs, c1);
    } else {
        HMM_ENSURE_DYN_LDS(qkv_attention_kernel<false>, kFLds);
        qkv_attention_kernel<false><<<grid, 512, kFLds, st>>>(a, w, bias, qkv_cls, out, n_img, scale_log2e, nullptr, nullptr);
    }
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_qkv_attention_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                         const uint16_t* qkv_cls_dev, uint16_t* out_dev, int n_img, hmm_stream_t stream) {
    return qkv_attention_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev,
                              reinterpret_cast<const bf16_t*>(qkv_cls_dev), reinterpret_cast<bf16_t*>(out_dev), n_img,
                              static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_qkv_attention_ln_bf16(const uint16_t* xb_dev, const uint16_t* wf_dev, const float* c2_dev,
                                            const uint16_t* qkv_cls_dev, uint16_t* out_dev, int n_img, const float* rs_dev,
                                            const float* c1_dev, hmm_stream_t stream) {
    HMM_REQUIRE(rs_dev && c1_dev, HMM_E_INVALID, "qkv_attention_ln: null statistics");
    return qkv_attention_bf16(reinterpret_cast<const bf16_t*>(xb_dev), reinterpret_cast<const bf16_t*>(wf_dev), c2_dev,
                              reinterpret_cast<const bf16_t*>(qkv_cls_dev), reinterpret_cast<bf16_t*>(out_dev), n_img,
                              static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(rs_dev), c1_dev);
}
