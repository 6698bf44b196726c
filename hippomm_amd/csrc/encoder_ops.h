// Internal launchers shared between the encoder orchestration and the exported hmm_op_* entry points.
#pragma once
#include "hmm_common.h"

namespace hmm {

// Internal epilogue ids, beside the four public ones (include/hippomm_hip.h)
#define HMM_EPI_BIAS_RESID_XB 4
#define HMM_EPI_LN_BF16       5
#define HMM_EPI_LN_GELU_BF16  6

// Epilogues of the folded-LayerNorm path (encoder.hip, vision tower), beside the four public ones (include/hippomm_hip.h):
//   HMM_EPI_BIAS_RESID_XB   C_f32 = (A W^T + bias) + C_f32 as HMM_EPI_BIAS_RESID_F32, and aux.xb[m][n] = bf16(C_f32[m][n]):
//                           the bf16 image of the residual stream that the next GEMM takes as its A operand;
//   HMM_EPI_LN_BF16 /       C_bf16 = [gelu](rs[m].x * acc - rs[m].y * c1[n] + c2[n]) with rs[m] = (rstd, rstd * mean) of row m of
//   HMM_EPI_LN_GELU_BF16    the A operand, W = bf16(gamma (.) W0), c1[n] = sum_k W[n][k], c2[n] = sum_k W0[n][k] beta[k] + bias[n]:
//                           LayerNorm(A) W0^T + bias with the normalisation applied after the product (c2 travels as `bias`).
// `GemmAux` carries the extra pointers; evaluated as fma(rs.x, acc, fma(-rs.y, c1, c2)) everywhere (GEMM kernels and the
// fused in_proj + attention kernel), so that the same numbers come out of every tile geometry.
struct GemmAux {
    bf16_t* xb = nullptr;            // RESID_XB: [M][N] bf16
    float2* part = nullptr;          // RESID_XB: [M][N / 64] chunk statistics of the xb rows (may be null), see chunk_stat_*
    const float2* rs = nullptr;      // LN*: row statistics, row m at rs[m * rs_stride]
    const float* c1 = nullptr;       // LN*: [N]
    int rs_stride = 1;
};
// Row statistics of xb without a pass over it: the RESID_XB epilogues also emit, per row and per 64-column chunk, (s, q) =
// (sum, sum of squares about the chunk mean s/64) of the bf16 values they store, and launch_rowstat_finalize combines a row's
// chunks (Chan's formula, chunk order) into (rstd, rstd * mean).  A chunk is reduced in ONE order whatever the tile geometry:
// leaf j = columns 4j..4j+3 as (b0 + b1) + (b2 + b3) [squares: fma chain d3, d2, d1 onto d0 * d0], then an xor-butterfly over
// j = 1, 2, 4, 8 -- lanes of a DPP row in the LDS-transposed epilogue (hmm_common.h row16_sum), lanes 16 / 32 apart and the
// four 16-column blocks of a wave tile in the direct one -- so a row gets the same bits from every kernel.
__device__ __forceinline__ float chunk_leaf_sum(float b0, float b1, float b2, float b3) { return (b0 + b1) + (b2 + b3); }
__device__ __forceinline__ float chunk_leaf_sq(float b0, float b1, float b2, float b3, float mc) {
    const float d0 = b0 - mc, d1 = b1 - mc, d2 = b2 - mc, d3 = b3 - mc;
    return fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
}

__device__ __forceinline__ float ln_fold(float acc, float2 rs, float c1, float c2) {
    return fmaf(rs.x, acc, fmaf(-rs.y, c1, c2));
}

int gemm_bf16(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int epi,
              int variant, hipStream_t st, const GemmAux* aux = nullptr);
// Launches of fewer 256x256 tiles than this leave the ping-pong kernel for the small-tile kernels (default 128).  A forward
// that runs as two chains sets 64 for its duration: its launches share the chip with the other chain's, so half-full
// ping-pong launches pack well (tools/mid_batch_probe.py).  Per host thread; returns the previous value.
int gemm_set_small_tiles(int tiles);

int launch_layernorm_bf16(const float* x, size_t in_stride, const float* g, const float* b, bf16_t* y,
                          int rows, int D, float eps, hipStream_t st);
int launch_assemble_tokens(const float* patches, const float* cls, const float* pos,
                           const float* stem_g, const float* stem_b, float stem_eps,
                           const float* pre_g, const float* pre_b, float pre_eps,
                           float* x, int n_img, int T, int D, hipStream_t st);
int launch_im2col_vision(const float* frames, bf16_t* out, int n_img, hipStream_t st);
int launch_im2col_audio(const float* mels, bf16_t* out, int n_clip, hipStream_t st);
int launch_l2norm_rows(const float* v, float* out, int n_out, int clips, const float* log_scale, hipStream_t st);
int launch_gather_rows(const void* src, size_t src_row_stride_bytes, void* dst, int n_rows, int row_bytes, hipStream_t st);
int launch_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* x, int n_rows, int T, int vocab,
                        hipStream_t st);
int launch_select_eos(const int64_t* ids, int32_t* sel, int batch, int T, hipStream_t st);
int launch_gather_selected_rows(const void* src, const int32_t* sel, int T, void* dst, int n_rows, int row_bytes,
                                hipStream_t st);
int launch_cast_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t st);
int launch_rowstat_bf16(const bf16_t* xb, float2* rs, int rows, int D, float eps, hipStream_t st);
int launch_rowstat_finalize(const float2* part, float2* rs, int rows, int D, float eps, hipStream_t st);
int launch_fold_ln_weights(const float* w0, const float* gamma, const float* beta, const float* bias, bf16_t* wf,
                           float* c1, float* c2, int N, int D, hipStream_t st);
int launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st);
int launch_fold_conv3d(const float* w, bf16_t* dst, int D, hipStream_t st);

int attention_bf16(const bf16_t* qkv, bf16_t* out, int batch, int tokens, int heads, int head_dim,
                   const float* bias_k, const float* bias_v, hipStream_t st, bool causal = false);

// vision tower only (D 1280, 16 heads, 257 tokens): in_proj + attention in one kernel (qkv_attention.hip); qkv_cls is the
// [n_img][3D] projection of the cls rows, out is [n_img*257][D]
int qkv_attention_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* qkv_cls, bf16_t* out,
                       int n_img, hipStream_t st, const float2* rs = nullptr, const float* c1 = nullptr);

// audio tower (D 768, 12 heads of 64, 229 tokens per clip, add_bias_kv): the same fusion, all 229 rows inside the tile
int qkv_attention_audio_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const float* bias_k, const float* bias_v,
                             bf16_t* out, int n_clips, hipStream_t st);

// last block: one query (the cls token) per (image, head); kv is [rows][2D] = [k | v]
int attention_cls_bf16(const bf16_t* q_cls, const bf16_t* kv, bf16_t* out, int batch, int tokens, int heads,
                       int head_dim, const float* bias_k, const float* bias_v, hipStream_t st);

}  // namespace hmm
