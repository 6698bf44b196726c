// Fused QKV projection + attention for the ImageBind-huge VISION tower (D = 1280, 16 heads x 80, 257 tokens) and AUDIO tower
// (D = 768, 12 heads x 64, 229 tokens + the add_bias_kv key) on gfx950; the text below describes the vision geometry, the
// audio one differs as `AudioGeo` says.
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

// Geometry of one tower's fused kernel.  The projection tile is always 256 token rows x 256 columns [q_h | k_h | v_h | unused]:
//   vision: 257 tokens, 16 heads x 80 -> rows = tokens 1..256 (token 0, the cls row, arrives through qkv_cls), 240 columns
//           used (pp_mainloop<1> skips the last 16);
//   audio:  229 tokens, 12 heads x 64, add_bias_kv -> rows = tokens 0..228 (+27 clamped rows whose results are dropped), 192
//           columns used: q, k and v are one wave-column each and the fourth computes nothing (pp_mainloop<2>); the learned
//           bias_k / bias_v row is appended to the K / V images as key 229.
struct VisionGeo { static constexpr int D = 1280, H = 16, DH = 80, T = 257, NKT = 9, ROW0 = 1, TAIL = 1; static constexpr bool CLS_OUTSIDE = true, BIAS_KV = false; };
struct AudioGeo  { static constexpr int D = 768,  H = 12, DH = 64, T = 229, NKT = 8, ROW0 = 0, TAIL = 2; static constexpr bool CLS_OUTSIDE = false, BIAS_KV = true; };

template <class G>
struct FusedPlan {
    using Cfg = AttnCfg<G::DH, G::NKT>;
    static constexpr int QRow = G::DH * 2 + 16;                          // Q image row stride (as K rows: conflict-free b128)
    static constexpr int QOff = Cfg::K_BYTES + Cfg::V_BYTES;             // Q image behind K and V
    static constexpr int XOff = QOff + ((G::T * QRow + 15) / 16) * 16;   // cooperative-query scratch
    static constexpr int Used = XOff + Cfg::X_BYTES;                     // vision: 154,176 B (> the 131,072 B of GEMM staging)
    static constexpr int Lds = Used > 2 * kPPTile ? Used : 2 * kPPTile;
    static_assert(Lds <= 160 * 1024, "LDS plan of the fused kernel");
    static_assert(G::T + (G::BIAS_KV ? 1 : 0) <= Cfg::NKEY && G::T - G::ROW0 <= 256 && 3 * G::DH <= 256, "tile geometry");
};

#ifdef HMM_PROBE
// in-kernel stamps per workgroup (16 slots): 0 start, 1 projection loop done, 2 Q/K/V images written, 3 main query tiles done,
// 4 end (stores retired); 6 / 7 / 8 shader-clock ticks at 0 / 1 / 4; 9 XCC id
unsigned long long* g_fused_stamps = nullptr;
extern "C" void hmm_probe_set_fused_stamps(unsigned long long* p) { g_fused_stamps = p; }
#define HMM_FUSED_PROBE_ARG , unsigned long long* stamps
#define HMM_FUSED_PROBE_VAL , g_fused_stamps
#define HMM_FSTAMP(slot, expr) if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (slot)] = (expr);
#else
#define HMM_FUSED_PROBE_ARG
#define HMM_FUSED_PROBE_VAL
#define HMM_FSTAMP(slot, expr)
#endif

template <class G>
__global__ __launch_bounds__(512) void qkv_attention_kernel(
    const bf16_t* __restrict__ a /* [n_img*T][D] LayerNorm output */, const bf16_t* __restrict__ w /* [3D][D] */,
    const float* __restrict__ bias /* [3D] */, const bf16_t* __restrict__ qkv_cls /* [n_img][3D], CLS_OUTSIDE only */,
    bf16_t* __restrict__ out /* [n_img*T][D] */, int n_img, float scale_log2e,
    const float* __restrict__ bias_k /* [D], BIAS_KV only */, const float* __restrict__ bias_v, int even_map HMM_FUSED_PROBE_ARG) {
    using P = FusedPlan<G>;
    using Cfg = typename P::Cfg;
    constexpr int kD = G::D, kH = G::H, kDH = G::DH, kT = G::T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // XCD x (= blockIdx % 8) takes a contiguous run of the (sample, head) list, sample-major: a sample's heads share its LayerNorm rows
    // through one L2, and the workgroups are spread evenly -- with the earlier "sample i on XCD i % 8" 18 clips put 36 workgroups on two
    // XCDs of 32 CUs each and the launch took two rounds (one chain of six audio segments: 44.5 -> 25 us per block).
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;
    const int pair = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    int b_lin = pair / kH;
    int h = pair % kH;
    if (!even_map) {                                   // probe build A/B: the earlier mapping (grid padded to 8 samples x heads)
        const int n = blockIdx.x >> 3;
        b_lin = (blockIdx.x & 7) + 8 * (n / kH);
        h = n % kH;
        if (b_lin >= n_img) return;
    }
    const int b = n_img - 1 - b_lin;
    HMM_FSTAMP(0, __builtin_amdgcn_s_memrealtime())
    HMM_FSTAMP(6, __builtin_amdgcn_s_memtime())
    HMM_FSTAMP(9, __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)))

    // ---- 1. projection tile: rows = the sample's tokens ROW0 .. ROW0+255, columns = [q_h | k_h | v_h | unused] ----------
    PPSources src;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wave + 8 * j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int arow = (lr & 63) + (lr >> 6) * 128;
        auto tok_row = [&](int r) {                                      // tile row -> row of `a` (clamped: results dropped)
            const int t = G::ROW0 + r;
            return b * kT + (t < kT ? t : kT - 1);
        };
        src.alo[j] = tok_row(arow) * kD + c * 8;
        src.ahi[j] = tok_row(arow + 64) * kD + c * 8;
        const int bcol = (lr >> 5) * 64 + (lr & 31);
        auto wrow = [&](int col) {                                       // tile column -> row of in_proj_weight
            const int part = col / kDH, d = col - part * kDH;
            return part < 3 ? part * kD + h * kDH + d : h * kDH;         // unused columns: any valid row, results unused
        };
        src.blo[j] = wrow(bcol) * kD + c * 8;
        src.bhi[j] = wrow(bcol + 32) * kD + c * 8;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    pp_mainloop<G::TAIL>(a, w, src, kD >> 6, smem, lane, wave, acc);   // the unused columns: no MFMAs
    HMM_FSTAMP(1, __builtin_amdgcn_s_memrealtime())
    HMM_FSTAMP(7, __builtin_amdgcn_s_memtime())

    // ---- 2. accumulators -> Q / K / V images (the staging buffers are dead: see pp_mainloop) --------------------------
    char* k_lds = smem;
    char* v_lds = smem + Cfg::K_BYTES;
    char* q_lds = smem + P::QOff;
    {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + 4 * fq;                  // 4 consecutive tile columns, never across a part
            const int part = col / kDH, d = col - part * kDH;
            if (part < 3) {                                              // wave-uniform per (wn, ni) except vision's last group
                const float4 bv = *reinterpret_cast<const float4*>(bias + part * kD + h * kDH + d);
                char* img = part == 0 ? q_lds : (part == 1 ? k_lds : v_lds);
                const int stride = part == 0 ? P::QRow : (part == 1 ? Cfg::KROW : Cfg::VROW);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int tok = G::ROW0 + wm * 128 + mi * 16 + fr;
                    const f32x4 v = acc[mi][ni];
                    const bf16x4 o4 = {(bf16_t)(v[0] + bv.x), (bf16_t)(v[1] + bv.y), (bf16_t)(v[2] + bv.z), (bf16_t)(v[3] + bv.w)};
                    if (G::ROW0 + 255 < kT || tok < kT)                  // tile rows past the last token are dropped
                        *reinterpret_cast<bf16x4*>(img + tok * stride + d * 2) = o4;
                }
            }
        }
        if constexpr (G::CLS_OUTSIDE) {
            // token 0 (cls): its q | k | v slices come from the caller's small GEMM; 3 parts x DH/8 chunks of 16 B
            constexpr int CH = kDH / 8;
            if (tid < 3 * CH) {
                const int part = tid / CH, ch = tid - part * CH;
                const uint4 v = *reinterpret_cast<const uint4*>(qkv_cls + (size_t)b * (3 * kD) + part * kD + h * kDH + ch * 8);
                char* img = part == 0 ? q_lds : (part == 1 ? k_lds : v_lds);
                *reinterpret_cast<uint4*>(img + ch * 16) = v;
            }
        }
        constexpr int LK = kT + (G::BIAS_KV ? 1 : 0);
        if constexpr (G::BIAS_KV) {                                      // the add_bias_kv position: key row T
            if (tid < kDH) {
                *reinterpret_cast<bf16_t*>(k_lds + kT * Cfg::KROW + tid * 2) = (bf16_t)bias_k[h * kDH + tid];
                *reinterpret_cast<bf16_t*>(v_lds + kT * Cfg::VROW + tid * 2) = (bf16_t)bias_v[h * kDH + tid];
            }
        }
        // key rows LK .. NKEY-1: zeros
        constexpr int CPR = kDH / 8;
        for (int idx = LK * CPR + tid; idx < Cfg::NKEY * CPR; idx += 512) {
            const int row = idx / CPR, c = idx - row * CPR;
            *reinterpret_cast<uint4*>(k_lds + row * Cfg::KROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(v_lds + row * Cfg::VROW + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    HMM_FSTAMP(2, __builtin_amdgcn_s_memrealtime())

    // ---- 3. attention on the images ------------------------------------------------------------------------------------
    const int r = lane & 31, hh = lane >> 5;
    auto load_q = [&](int row, int ks) {
        return *reinterpret_cast<const bf16x8*>(q_lds + row * P::QRow + hh * 16 + ks * 32);
    };
    bf16x8 qf[Cfg::KS];
    {
        const int qrow = wave * 32 + r;
        const int qr = qrow < kT ? qrow : kT - 1;
#pragma unroll
        for (int ks = 0; ks < Cfg::KS; ++ks) qf[ks] = load_q(qr, ks);
    }
    attention_core<kDH, G::NKT>(k_lds, v_lds, reinterpret_cast<float*>(smem + P::XOff), load_q, qf,
                                out + (size_t)b * kT * kD + h * kDH, kT, kT + (G::BIAS_KV ? 1 : 0), kD, scale_log2e, false
#ifdef HMM_PROBE
                                , 0, 1, stamps ? stamps + (size_t)blockIdx.x * 16 : nullptr
#endif
                                );
#ifdef HMM_PROBE
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#endif
    HMM_FSTAMP(4, __builtin_amdgcn_s_memrealtime())
    HMM_FSTAMP(8, __builtin_amdgcn_s_memtime())
}

HMM_TUNABLE(int, g_fused_even_map, 1)   // probe build: 0 = sample i on XCD i % 8 with a padded grid (the mapping before round 5's audio fix)
template <class G>
static int launch_fused(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* qkv_cls, bf16_t* out, int n_img,
                        hipStream_t st, const float* bias_k, const float* bias_v) {
    HMM_REQUIRE(n_img >= 1 && (int64_t)n_img * G::T * G::D < (1ll << 31), HMM_E_INVALID, "qkv_attention: n_img=%d out of range", n_img);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)G::DH);
    const int grid = g_fused_even_map ? n_img * G::H : 8 * ((n_img + 7) / 8) * G::H;
    auto kern = qkv_attention_kernel<G>;
    HMM_ENSURE_DYN_LDS(kern, FusedPlan<G>::Lds);
    kern<<<grid, 512, FusedPlan<G>::Lds, st>>>(a, w, bias, qkv_cls, out, n_img, scale_log2e, bias_k, bias_v, g_fused_even_map HMM_FUSED_PROBE_VAL);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int qkv_attention_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* qkv_cls, bf16_t* out,
                       int n_img, hipStream_t st) {
    HMM_REQUIRE(a && w && bias && qkv_cls && out, HMM_E_INVALID, "qkv_attention: null pointer");
    return launch_fused<VisionGeo>(a, w, bias, qkv_cls, out, n_img, st, nullptr, nullptr);
}

int qkv_attention_audio_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const float* bias_k, const float* bias_v,
                             bf16_t* out, int n_clips, hipStream_t st) {
    HMM_REQUIRE(a && w && bias && bias_k && bias_v && out, HMM_E_INVALID, "qkv_attention_audio: null pointer");
    return launch_fused<AudioGeo>(a, w, bias, nullptr, out, n_clips, st, bias_k, bias_v);
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_qkv_attention_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                         const uint16_t* qkv_cls_dev, uint16_t* out_dev, int n_img, hmm_stream_t stream) {
    return qkv_attention_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev,
                              reinterpret_cast<const bf16_t*>(qkv_cls_dev), reinterpret_cast<bf16_t*>(out_dev), n_img,
                              static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_qkv_attention_audio_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                               const float* bias_k_dev, const float* bias_v_dev, uint16_t* out_dev,
                                               int n_clips, hmm_stream_t stream) {
    return qkv_attention_audio_bf16(reinterpret_cast<const bf16_t*>(a_dev), reinterpret_cast<const bf16_t*>(w_dev), bias_dev,
                                    bias_k_dev, bias_v_dev, reinterpret_cast<bf16_t*>(out_dev), n_clips,
                                    static_cast<hipStream_t>(stream));
}
