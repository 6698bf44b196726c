// Internal launchers shared between the encoder orchestration and the exported hmm_op_* entry points.
#pragma once
#include "hmm_common.h"

namespace hmm {

int gemm_bf16(const bf16_t* A, const bf16_t* W, const float* bias, void* C, int M, int N, int K, int epi,
              int variant, hipStream_t st);
// Launches of fewer 256x256 tiles than this leave the ping-pong kernel for the small-tile kernels (default 128).  A forward
// that runs as two chains sets 64 for its duration: its launches share the chip with the other chain's, so half-full
// ping-pong launches pack well (tools/mid_batch_probe.py).  Per host thread; returns the previous value.
int gemm_set_small_tiles(int tiles);

// Split-K for few-row launches with a long K: `splits` fp32 partial products part[split][M][N] (no bias), added up in split
// order by launch_layernorm_reduce_bf16 -- the LayerNorm that follows every residual GEMM -- together with the bias and the
// residual.  tile < 0: by shape among the ring geometries.
int gemm_bf16_splitk(const bf16_t* A, const bf16_t* W, float* part, int M, int N, int K, int splits, int tile, hipStream_t st);
int launch_layernorm_reduce_bf16(float* x, const float* part, size_t part_stride, int splits, const float* bias,
                                 const float* g, const float* b, bf16_t* y, int rows, int D, float eps, hipStream_t st);

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
// y_bf16[b] = LN(x[b*T + first position of the largest id of sample b]): the text head's select + gather + LayerNorm in one launch
int launch_layernorm_eos_bf16(const float* x, const int64_t* ids, int T, const float* g, const float* b, bf16_t* y, int batch, int D,
                              float eps, hipStream_t st);
int launch_gather_selected_rows(const void* src, const int32_t* sel, int T, void* dst, int n_rows, int row_bytes,
                                hipStream_t st);
int launch_cast_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t st);
int launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st);
int launch_fold_conv3d(const float* w, bf16_t* dst, int D, hipStream_t st);

int attention_bf16(const bf16_t* qkv, bf16_t* out, int batch, int tokens, int heads, int head_dim,
                   const float* bias_k, const float* bias_v, hipStream_t st, bool causal = false);

// vision tower only (D 1280, 16 heads, 257 tokens): in_proj + attention in one kernel (qkv_attention.hip); qkv_cls is the
// [n_img][3D] projection of the cls rows, out is [n_img*257][D]
int qkv_attention_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* qkv_cls, bf16_t* out,
                       int n_img, hipStream_t st);

// audio tower (D 768, 12 heads of 64, 229 tokens per clip, add_bias_kv): the same fusion, all 229 rows inside the tile
int qkv_attention_audio_bf16(const bf16_t* a, const bf16_t* w, const float* bias, const float* bias_k, const float* bias_v,
                             bf16_t* out, int n_clips, hipStream_t st);

// last block: one query (the cls token) per (image, head); kv is [rows][2D] = [k | v]
int attention_cls_bf16(const bf16_t* q_cls, const bf16_t* kv, bf16_t* out, int batch, int tokens, int heads,
                       int head_dim, const float* bias_k, const float* bias_v, hipStream_t st);

}  // namespace hmm
