// HBM-bound building blocks of the ImageBind towers on gfx950 (everything that is not a GEMM or the
// attention core).  One wave per token row; a lane owns float4 columns 4*(64*j + lane), so each
// load/store instruction of a wave covers 1 KiB (fp32) / 512 B (bf16) contiguous.  LayerNorm
// statistics are fp32 two-pass (mean, then centred variance) as in torch.
#include "hmm_common.h"
#include "encoder_ops.h"

namespace hmm {

template <int NV>   // NV float4 per lane: D = 256*NV  (768 -> 3, 1280 -> 5)
__device__ __forceinline__ void row_layernorm(float4 (&x)[NV], const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float eps, int lane) {
    constexpr float inv_d = 1.0f / (256.0f * NV);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
    const float mean = wave_sum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
        v = fmaf(x[j].x, x[j].x, v); v = fmaf(x[j].y, x[j].y, v);
        v = fmaf(x[j].z, x[j].z, v); v = fmaf(x[j].w, x[j].w, v);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) * inv_d + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * (64 * j + lane));
        const float4 b = *reinterpret_cast<const float4*>(beta + 4 * (64 * j + lane));
        x[j].x = fmaf(x[j].x * rstd, g.x, b.x); x[j].y = fmaf(x[j].y * rstd, g.y, b.y);
        x[j].z = fmaf(x[j].z * rstd, g.z, b.z); x[j].w = fmaf(x[j].w * rstd, g.w, b.w);
    }
}

// y_bf16[r] = LN(x_f32[r * in_stride ...])          (norm_1 / norm_2 / head LN on the cls rows)
// <= 48 VGPRs on purpose: the ping-pong GEMM leaves 48 registers per SIMD lane free, so LayerNorm waves of
// one half-batch can co-reside on a CU that is busy with a GEMM tile of the other half (two-stream forward).
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void layernorm_bf16_kernel(const float* __restrict__ x, size_t in_stride,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             bf16_t* __restrict__ y, int rows, float eps, int nt) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {   // grid-stride
        const float* src = x + (size_t)row * in_stride;
        float4 v[NV];
        if (nt) {                                            // the fp32 stream is read once per LayerNorm: keep it out of the caches
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 4 * (64 * j + lane)));
                v[j] = make_float4(t[0], t[1], t[2], t[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(src + 4 * (64 * j + lane));
        }
        row_layernorm<NV>(v, gamma, beta, eps, lane);
        bf16_t* dst = y + (size_t)row * D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            bf16x4 o = {(bf16_t)v[j].x, (bf16_t)v[j].y, (bf16_t)v[j].z, (bf16_t)v[j].w};
            *reinterpret_cast<bf16x4*>(dst + 4 * (64 * j + lane)) = o;
        }
    }
}

// The consumer side of a split-K residual GEMM (gemm_bf16_splitk): x[r] = ((p_0[r] + p_1[r] + ... + p_{S-1}[r]) + bias) + x[r]
// -- the slabs in split order, then the bias, then the residual, the non-split epilogue's (accumulator + bias) + x -- written
// back to the fp32 stream, and y_bf16[r] = LN(x[r]).  A row's bits depend on S only, not on the launch.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_reduce_bf16_kernel(float* __restrict__ x, const float* __restrict__ part,
                                                                    size_t part_stride, int splits,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    bf16_t* __restrict__ y, int rows, float eps) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + (size_t)row * D;
    const float* pr = part + (size_t)row * D;
    float4 v[NV], acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j] = *reinterpret_cast<const float4*>(xr + 4 * (64 * j + lane));
        acc[j] = *reinterpret_cast<const float4*>(pr + 4 * (64 * j + lane));
    }
    for (int s = 1; s < splits; ++s) {
        const float* ps = pr + (size_t)s * part_stride;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 p = *reinterpret_cast<const float4*>(ps + 4 * (64 * j + lane));
            acc[j].x += p.x; acc[j].y += p.y; acc[j].z += p.z; acc[j].w += p.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(bias + 4 * (64 * j + lane));
        v[j] = make_float4((acc[j].x + b.x) + v[j].x, (acc[j].y + b.y) + v[j].y, (acc[j].z + b.z) + v[j].z, (acc[j].w + b.w) + v[j].w);
        *reinterpret_cast<float4*>(xr + 4 * (64 * j + lane)) = v[j];
    }
    row_layernorm<NV>(v, gamma, beta, eps, lane);
    bf16_t* dst = y + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        bf16x4 o = {(bf16_t)v[j].x, (bf16_t)v[j].y, (bf16_t)v[j].z, (bf16_t)v[j].w};
        *reinterpret_cast<bf16x4*>(dst + 4 * (64 * j + lane)) = o;
    }
}

// x[b*T + t] = pre_ln( (t == 0 ? cls : stem_ln(patch[b*P + t-1])) + pos[t] )
template <int NV>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(
    const float* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
    const float* __restrict__ stem_g, const float* __restrict__ stem_b, float stem_eps,
    const float* __restrict__ pre_g, const float* __restrict__ pre_b, float pre_eps,
    float* __restrict__ x, int n_img, int T) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_img * T) return;
    const int b = row / T, t = row % T;
    float4 v[NV];
    if (t == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(cls + 4 * (64 * j + lane));
    } else {
        const float* src = patches + ((size_t)b * (T - 1) + (t - 1)) * D;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(src + 4 * (64 * j + lane));
        if (stem_g != nullptr) row_layernorm<NV>(v, stem_g, stem_b, stem_eps, lane);
    }
    const float* pr = pos + (size_t)t * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 p = *reinterpret_cast<const float4*>(pr + 4 * (64 * j + lane));
        v[j].x += p.x; v[j].y += p.y; v[j].z += p.z; v[j].w += p.w;
    }
    if (pre_g != nullptr) row_layernorm<NV>(v, pre_g, pre_b, pre_eps, lane);
    float* dst = x + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) *reinterpret_cast<float4*>(dst + 4 * (64 * j + lane)) = v[j];
}

// vision: (B,3,224,224) fp32 -> bf16 [B*256][640]; column k = c*196 + dy*14 + dx (k >= 588 zero).
// One wave per patch row; both temporal taps of the Conv3d see this same frame (PadIm2Video
// "repeat"), which is why the weights were folded to K = 588 at load time.
__global__ __launch_bounds__(256) void im2col_vision_kernel(const float* __restrict__ frames,
                                                            bf16_t* __restrict__ out, int n_img) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_img * 256) return;
    const int b = row >> 8, p = row & 255, py = p >> 4, px = p & 15;
    const float* img = frames + (size_t)b * 3 * 224 * 224 + (size_t)(py * 14) * 224 + px * 14;
    bf16_t* dst = out + (size_t)row * 640;
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const int k = it * 64 + lane;
        float v = 0.f;
        if (k < 588) {
            const int c = k / 196, rem = k - c * 196, dy = rem / 14, dx = rem - dy * 14;
            v = img[(size_t)c * 224 * 224 + dy * 224 + dx];
        }
        dst[k] = (bf16_t)v;
    }
}

// audio: (N,1,128,204) fp32 -> bf16 [N*228][256]; patch (py<12, px<19), stride 10, kernel 16.
__global__ __launch_bounds__(256) void im2col_audio_kernel(const float* __restrict__ mels,
                                                           bf16_t* __restrict__ out, int n_clip) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_clip * 228) return;
    const int n = row / 228, p = row - n * 228, py = p / 19, px = p - py * 19;
    const float* src = mels + (size_t)n * 128 * 204 + (size_t)(py * 10) * 204 + px * 10;
    bf16_t* dst = out + (size_t)row * 256;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int k = it * 64 + lane, dy = k >> 4, dx = k & 15;
        dst[k] = (bf16_t)src[dy * 204 + dx];
    }
}

// out[b] = mean over `clips` consecutive rows of scale * v / max(||v||, 1e-12)   (Normalize,
// LearnableLogitScaling, clip mean).  scale = min(exp(*log_scale), 100) or 1 when log_scale is null.
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ v, float* __restrict__ out,
                                                          int n_out, int clips, const float* __restrict__ log_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_out) return;
    const float scale = log_scale ? fminf(expf(*log_scale), 100.0f) : 1.0f;
    float4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < clips; ++s) {
        const float* src = v + ((size_t)row * clips + s) * HMM_FEATURE_DIM;
        float4 x[4];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = *reinterpret_cast<const float4*>(src + 4 * (64 * j + lane));
            ss = fmaf(x[j].x, x[j].x, ss); ss = fmaf(x[j].y, x[j].y, ss);
            ss = fmaf(x[j].z, x[j].z, ss); ss = fmaf(x[j].w, x[j].w, ss);
        }
        const float f = scale / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j].x = fmaf(x[j].x, f, acc[j].x); acc[j].y = fmaf(x[j].y, f, acc[j].y);
            acc[j].z = fmaf(x[j].z, f, acc[j].z); acc[j].w = fmaf(x[j].w, f, acc[j].w);
        }
    }
    const float inv = 1.0f / (float)clips;
    float* dst = out + (size_t)row * HMM_FEATURE_DIM;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(dst + 4 * (64 * j + lane)) =
            make_float4(acc[j].x * inv, acc[j].y * inv, acc[j].z * inv, acc[j].w * inv);
}

// dst[i] = src[i * src_row_stride ... + D)   (cls rows of the token matrix; 16-B pieces)
__global__ void gather_rows_kernel(const char* __restrict__ src, size_t src_row_stride_bytes, char* __restrict__ dst,
                                   int n_rows, int row_bytes) {
    const int per_row = row_bytes / 16;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rows * per_row) return;
    const int r = (int)(i / per_row), c = (int)(i % per_row);
    *reinterpret_cast<uint4*>(dst + (size_t)r * row_bytes + c * 16) =
        *reinterpret_cast<const uint4*>(src + (size_t)r * src_row_stride_bytes + c * 16);
}

// text tower: x[b*T + t] = token_embedding[ids[b][t]] + pos_embed[t]    (D = 1024; one wave per token row)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ x,
                                                           int n_rows, int T, int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* src = table + (size_t)id * 1024;
    const float* pr = pos + (size_t)(row % T) * 1024;
    float* dst = x + (size_t)row * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(src + 4 * (64 * j + lane));
        const float4 p = *reinterpret_cast<const float4*>(pr + 4 * (64 * j + lane));
        *reinterpret_cast<float4*>(dst + 4 * (64 * j + lane)) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// sel[b] = first position of the largest token id of sample b (the EOS token has the largest id)
__global__ __launch_bounds__(64) void select_eos_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ sel, int T) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int64_t best = INT64_MIN;
    int pos = 0x7FFFFFFF;
    for (int t = lane; t < T; t += 64) {
        const int64_t v = ids[(size_t)b * T + t];
        if (v > best) { best = v; pos = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t ob = __shfl_xor(best, off, 64);
        const int op = __shfl_xor(pos, off, 64);
        if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
    }
    if (lane == 0) sel[b] = pos;
}

// Text head in ONE launch (was select_eos + gather_selected_rows + layernorm_bf16: three kernels at the launch floor each):
// y_bf16[b] = LN(x_f32[b*T + eos(b)]) with eos(b) = the first position of the largest token id of sample b.  One wave per sample;
// the same argmax as select_eos_kernel and the same row_layernorm as layernorm_bf16_kernel: same bits.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_eos_bf16_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids, int T,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 bf16_t* __restrict__ y, int batch, float eps) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    int64_t best = INT64_MIN;
    int pos = 0x7FFFFFFF;
    for (int t = lane; t < T; t += 64) {
        const int64_t v = ids[(size_t)b * T + t];
        if (v > best) { best = v; pos = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t ob = __shfl_xor(best, off, 64);
        const int op = __shfl_xor(pos, off, 64);
        if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
    }
    const float* src = x + ((size_t)b * T + pos) * D;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(src + 4 * (64 * j + lane));
    row_layernorm<NV>(v, gamma, beta, eps, lane);
    bf16_t* dst = y + (size_t)b * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        bf16x4 o = {(bf16_t)v[j].x, (bf16_t)v[j].y, (bf16_t)v[j].z, (bf16_t)v[j].w};
        *reinterpret_cast<bf16x4*>(dst + 4 * (64 * j + lane)) = o;
    }
}

// dst[b] = src[(b*T + sel[b])]   rows of row_bytes (16-B pieces)
__global__ void gather_selected_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ sel, int T,
                                            char* __restrict__ dst, int n_rows, int row_bytes) {
    const int per_row = row_bytes / 16;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rows * per_row) return;
    const int r = (int)(i / per_row), c = (int)(i % per_row);
    *reinterpret_cast<uint4*>(dst + (size_t)r * row_bytes + c * 16) =
        *reinterpret_cast<const uint4*>(src + ((size_t)r * T + sel[r]) * row_bytes + c * 16);
}

// ---- weight packing -------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (bf16_t)src[i];
}
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// Conv3d weight (D,3,2,14,14) -> folded bf16 [D][640]: w[d][c][0][..] + w[d][c][1][..], zero pad.
__global__ void fold_conv3d_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)D * 640) return;
    const int d = (int)(i / 640), k = (int)(i % 640);
    float v = 0.f;
    if (k < 588) {
        const int c = k / 196, rem = k - c * 196;
        const float* p = w + ((size_t)d * 3 + c) * 2 * 196 + rem;
        v = p[0] + p[196];
    }
    dst[i] = (bf16_t)v;
}

// ---- launchers ------------------------------------------------------------------------------
HMM_TUNABLE(int, g_ln_nt_loads, 1)       // non-temporal loads of the fp32 residual stream in LayerNorm (read once; keeps the bf16 output cache-resident for its consumer: forward -0.9 %, profiles/r3_forward_ab.json); probe build: A/B

int launch_layernorm_bf16(const float* x, size_t in_stride, const float* g, const float* b, bf16_t* y,
                          int rows, int D, float eps, hipStream_t st) {
    HMM_REQUIRE(D == 768 || D == 1024 || D == 1280, HMM_E_INVALID, "layernorm: D must be 768, 1024 or 1280, got %d", D);
    if (rows <= 0) return HMM_OK;
    int blocks = (rows + 3) / 4;
    const int nt = g_ln_nt_loads && rows >= 4096;            // small launches (cls rows) are re-read from L2 right away
    if (D == 768)       layernorm_bf16_kernel<3><<<blocks, 256, 0, st>>>(x, in_stride, g, b, y, rows, eps, nt);
    else if (D == 1024) layernorm_bf16_kernel<4><<<blocks, 256, 0, st>>>(x, in_stride, g, b, y, rows, eps, nt);
    else                layernorm_bf16_kernel<5><<<blocks, 256, 0, st>>>(x, in_stride, g, b, y, rows, eps, nt);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int launch_layernorm_reduce_bf16(float* x, const float* part, size_t part_stride, int splits, const float* bias,
                                 const float* g, const float* b, bf16_t* y, int rows, int D, float eps, hipStream_t st) {
    HMM_REQUIRE(D == 768 || D == 1024 || D == 1280, HMM_E_INVALID, "layernorm_reduce: D must be 768, 1024 or 1280, got %d", D);
    HMM_REQUIRE(splits >= 1 && splits <= 8, HMM_E_INVALID, "layernorm_reduce: splits=%d", splits);
    if (rows <= 0) return HMM_OK;
    const int blocks = (rows + 3) / 4;
    if (D == 768)       layernorm_reduce_bf16_kernel<3><<<blocks, 256, 0, st>>>(x, part, part_stride, splits, bias, g, b, y, rows, eps);
    else if (D == 1024) layernorm_reduce_bf16_kernel<4><<<blocks, 256, 0, st>>>(x, part, part_stride, splits, bias, g, b, y, rows, eps);
    else                layernorm_reduce_bf16_kernel<5><<<blocks, 256, 0, st>>>(x, part, part_stride, splits, bias, g, b, y, rows, eps);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int launch_assemble_tokens(const float* patches, const float* cls, const float* pos,
                           const float* stem_g, const float* stem_b, float stem_eps,
                           const float* pre_g, const float* pre_b, float pre_eps,
                           float* x, int n_img, int T, int D, hipStream_t st) {
    HMM_REQUIRE(D == 768 || D == 1280, HMM_E_INVALID, "assemble: D must be 768 or 1280, got %d", D);
    const int blocks = (n_img * T + 3) / 4;
    if (D == 768) assemble_tokens_kernel<3><<<blocks, 256, 0, st>>>(patches, cls, pos, stem_g, stem_b, stem_eps,
                                                                    pre_g, pre_b, pre_eps, x, n_img, T);
    else          assemble_tokens_kernel<5><<<blocks, 256, 0, st>>>(patches, cls, pos, stem_g, stem_b, stem_eps,
                                                                    pre_g, pre_b, pre_eps, x, n_img, T);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int launch_im2col_vision(const float* frames, bf16_t* out, int n_img, hipStream_t st) {
    im2col_vision_kernel<<<(n_img * 256 + 3) / 4, 256, 0, st>>>(frames, out, n_img);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_im2col_audio(const float* mels, bf16_t* out, int n_clip, hipStream_t st) {
    im2col_audio_kernel<<<(n_clip * 228 + 3) / 4, 256, 0, st>>>(mels, out, n_clip);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_l2norm_rows(const float* v, float* out, int n_out, int clips, const float* log_scale, hipStream_t st) {
    l2norm_rows_kernel<<<(n_out + 3) / 4, 256, 0, st>>>(v, out, n_out, clips, log_scale);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_gather_rows(const void* src, size_t src_row_stride_bytes, void* dst, int n_rows, int row_bytes, hipStream_t st) {
    const int64_t n = (int64_t)n_rows * (row_bytes / 16);
    gather_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(static_cast<const char*>(src), src_row_stride_bytes,
                                                                   static_cast<char*>(dst), n_rows, row_bytes);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* x, int n_rows, int T, int vocab,
                        hipStream_t st) {
    embed_tokens_kernel<<<(n_rows + 3) / 4, 256, 0, st>>>(ids, table, pos, x, n_rows, T, vocab);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_layernorm_eos_bf16(const float* x, const int64_t* ids, int T, const float* g, const float* b, bf16_t* y, int batch, int D,
                              float eps, hipStream_t st) {
    HMM_REQUIRE(D == 768 || D == 1024 || D == 1280, HMM_E_INVALID, "layernorm_eos: D must be 768, 1024 or 1280, got %d", D);
    if (batch <= 0) return HMM_OK;
    const int blocks = (batch + 3) / 4;
    if (D == 768)       layernorm_eos_bf16_kernel<3><<<blocks, 256, 0, st>>>(x, ids, T, g, b, y, batch, eps);
    else if (D == 1024) layernorm_eos_bf16_kernel<4><<<blocks, 256, 0, st>>>(x, ids, T, g, b, y, batch, eps);
    else                layernorm_eos_bf16_kernel<5><<<blocks, 256, 0, st>>>(x, ids, T, g, b, y, batch, eps);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

int launch_select_eos(const int64_t* ids, int32_t* sel, int batch, int T, hipStream_t st) {
    select_eos_kernel<<<batch, 64, 0, st>>>(ids, sel, T);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_gather_selected_rows(const void* src, const int32_t* sel, int T, void* dst, int n_rows, int row_bytes,
                                hipStream_t st) {
    const int64_t n = (int64_t)n_rows * (row_bytes / 16);
    gather_selected_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(static_cast<const char*>(src), sel, T,
                                                                            static_cast<char*>(dst), n_rows, row_bytes);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_cast_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t st) {
    cast_bf16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, n);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st) {
    copy_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, n);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
int launch_fold_conv3d(const float* w, bf16_t* dst, int D, hipStream_t st) {
    const int64_t n = (int64_t)D * 640;
    fold_conv3d_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w, dst, D);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}

}  // namespace hmm

using namespace hmm;

extern "C" int hmm_op_layernorm_bf16(const float* x_dev, const float* gamma_dev, const float* beta_dev,
                                     uint16_t* y_dev, int rows, int D, float eps, hmm_stream_t stream) {
    HMM_REQUIRE(x_dev && gamma_dev && beta_dev && y_dev, HMM_E_INVALID, "layernorm: null pointer");
    return launch_layernorm_bf16(x_dev, (size_t)D, gamma_dev, beta_dev, reinterpret_cast<bf16_t*>(y_dev), rows, D, eps,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int hmm_op_layernorm_reduce_bf16(float* x_dev, const float* part_dev, int splits, const float* bias_dev,
                                            const float* gamma_dev, const float* beta_dev, uint16_t* y_dev, int rows, int D,
                                            float eps, hmm_stream_t stream) {
    HMM_REQUIRE(x_dev && part_dev && bias_dev && gamma_dev && beta_dev && y_dev, HMM_E_INVALID, "layernorm_reduce: null pointer");
    return launch_layernorm_reduce_bf16(x_dev, part_dev, (size_t)rows * D, splits, bias_dev, gamma_dev, beta_dev,
                                        reinterpret_cast<bf16_t*>(y_dev), rows, D, eps, static_cast<hipStream_t>(stream));
}
