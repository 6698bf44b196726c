// Device-side vision preprocessing for gfx950: decoded RGB uint8 frames (B,H,W,3) ->
// (B,3,224,224) fp32, bit-identical to Pillow's BICUBIC resize of the short side to 224 + centre crop +
// ToTensor + CLIP Normalize (what upstream imagebind.data.load_and_transform_vision_data does through
// torchvision [recalled]; reference call site hippomm/models/foundation_models.py:87-90).
//
// Pillow's resample is two separable passes with an 8-bit intermediate image and 22-bit fixed-point
// coefficients: acc = 2^21 + sum(pixel * coef); out = clip8(acc >> 22).  The coefficient tables depend only on
// the frame size, are computed on the host in double precision exactly as Pillow's precompute_coeffs /
// normalize_coeffs_8bpc do (hippomm_amd/preprocess.py) and already restricted to the 224 cropped columns/rows.
// Pass 1 (horizontal) only touches the input rows the cropped output needs.  HBM-bound: one read of the frame.
#include "hmm_common.h"

namespace hmm {

constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr int kOut = 224;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[b][y - y0][xx][c], y in [y0, y1), xx in [0,224)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ frames, int in_h, int in_w,
                                                       const int32_t* __restrict__ kh, const int32_t* __restrict__ bh,
                                                       int ksize, int y0, int y1, uint8_t* __restrict__ tmp, int batch) {
    const int rows = y1 - y0;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)batch * rows * kOut) return;
    const int xx = (int)(t % kOut);
    const int yr = (int)((t / kOut) % rows);
    const int b = (int)(t / ((int64_t)kOut * rows));
    const int xmin = bh[xx * 2], xmax = bh[xx * 2 + 1];
    const uint8_t* line = frames + (((int64_t)b * in_h + (y0 + yr)) * in_w + xmin) * 3;
    const int32_t* k = kh + xx * ksize;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int c = k[x];
        s0 += line[x * 3 + 0] * c;
        s1 += line[x * 3 + 1] * c;
        s2 += line[x * 3 + 2] * c;
    }
    uint8_t* o = tmp + t * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// out[b][c][yy][xx] = ((u8 / 255) - mean_c) / std_c, u8 = vertical pass of tmp
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(const uint8_t* __restrict__ tmp, int rows, int y0,
                                                                 const int32_t* __restrict__ kv,
                                                                 const int32_t* __restrict__ bv, int ksize,
                                                                 float* __restrict__ out, int batch) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)batch * kOut * kOut) return;
    const int xx = (int)(t % kOut);
    const int yy = (int)((t / kOut) % kOut);
    const int b = (int)(t / (kOut * kOut));
    const int ymin = bv[yy * 2], ymax = bv[yy * 2 + 1];
    const uint8_t* col = tmp + (((int64_t)b * rows + (ymin - y0)) * kOut + xx) * 3;
    const int32_t* k = kv + yy * ksize;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
        const int c = k[y];
        const uint8_t* p = col + (int64_t)y * kOut * 3;
        s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const int s[3] = {s0, s1, s2};
    float* o = out + (int64_t)b * 3 * kOut * kOut + yy * kOut + xx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)clip8(s[c]) / 255.0f;
        o[(int64_t)c * kOut * kOut] = (v - mean[c]) / stdv[c];
    }
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_preprocess_vision_workspace_bytes(int batch, int rows_needed) {
    if (batch < 1 || rows_needed < 1) return 0;
    return align_up((size_t)batch * rows_needed * kOut * 3, 256) + 256;
}

extern "C" int hmm_preprocess_vision_u8(const uint8_t* frames_dev, int batch, int in_h, int in_w,
                                        const int32_t* kh_dev, const int32_t* bh_dev, int ksize_h,
                                        const int32_t* kv_dev, const int32_t* bv_dev, int ksize_v,
                                        int row_first, int row_last, float* out_dev,
                                        void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(frames_dev && kh_dev && bh_dev && kv_dev && bv_dev && out_dev && workspace_dev, HMM_E_INVALID,
                "preprocess_vision: null pointer");
    HMM_REQUIRE(batch >= 1 && in_h >= 1 && in_w >= 1 && ksize_h >= 1 && ksize_v >= 1, HMM_E_INVALID,
                "preprocess_vision: bad shape");
    HMM_REQUIRE(0 <= row_first && row_first < row_last && row_last <= in_h, HMM_E_INVALID,
                "preprocess_vision: row window [%d,%d) outside the frame height %d", row_first, row_last, in_h);
    const int rows = row_last - row_first;
    HMM_REQUIRE(workspace_bytes >= hmm_preprocess_vision_workspace_bytes(batch, rows), HMM_E_WORKSPACE,
                "preprocess_vision: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint8_t* tmp = static_cast<uint8_t*>(workspace_dev);
    const int64_t n1 = (int64_t)batch * rows * kOut;
    resize_h_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(frames_dev, in_h, in_w, kh_dev, bh_dev, ksize_h,
                                                                 row_first, row_last, tmp, batch);
    HMM_LAUNCH_CHECK();
    const int64_t n2 = (int64_t)batch * kOut * kOut;
    resize_v_normalize_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(tmp, rows, row_first, kv_dev, bv_dev, ksize_v,
                                                                            out_dev, batch);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
