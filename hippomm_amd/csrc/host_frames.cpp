// Host-side hand-over of one decoded frame to the pinned upload buffer (no GPU call in this file).
//
// The reference decodes every frame with Pillow inside imagebind.data.load_and_transform_vision_data
// (hippomm/models/foundation_models.py:87-90).  Pillow keeps an RGB image as four bytes per pixel (R,G,B,pad) and the ways to
// reach those pixels from Python -- Image.tobytes(), np.asarray(image) -- repack them while HOLDING the interpreter lock: 0.8 ms
// per 1280x720 frame, i.e. a ceiling of ~1 300 frames/s however many decode threads the host has, below what one MI355X embeds.
// Pillow >= 11.2 exports the pixel block without a copy through the Arrow C data interface (Image.__arrow_c_array__); this file
// reads that descriptor (a stable C ABI: struct ArrowArray) and packs R,G,B into the caller's buffer.  Called through ctypes the
// interpreter lock is released for the whole call, so the decode threads of hippomm_amd/preprocess.py scale with the cores.
#include "hmm_common.h"

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <tmmintrin.h>

namespace {

// The Arrow C data interface (https://arrow.apache.org/docs/format/CDataInterface.html), restated: the ABI Pillow's capsule holds.
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    ArrowArray** children;
    ArrowArray* dictionary;
    void (*release)(ArrowArray*);
    void* private_data;
};

void pack_scalar(const uint8_t* src, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; ++i) {
        dst[3 * i + 0] = src[4 * i + 0];
        dst[3 * i + 1] = src[4 * i + 1];
        dst[3 * i + 2] = src[4 * i + 2];
    }
}

// 16 pixels per step: four 16-byte loads, each shuffled to 12 payload bytes, stitched into three 16-byte stores.  Every store
// lies inside the 48 bytes of the step, so nothing is written beyond 3 * n (the next frame of the ring may be in another thread's hands).
__attribute__((target("ssse3"))) void pack_ssse3(const uint8_t* src, size_t n, uint8_t* dst) {
    const __m128i m = _mm_setr_epi8(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 4 * i)), m);
        const __m128i b = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 4 * i + 16)), m);
        const __m128i c = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 4 * i + 32)), m);
        const __m128i d = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 4 * i + 48)), m);
        // a = A0..A11 0000, b = B0..B11 0000, ...: out0 = A0..11 B0..3, out1 = B4..11 C0..7, out2 = C8..11 D0..11
        const __m128i o0 = _mm_or_si128(a, _mm_slli_si128(b, 12));
        const __m128i o1 = _mm_or_si128(_mm_srli_si128(b, 4), _mm_slli_si128(c, 8));
        const __m128i o2 = _mm_or_si128(_mm_srli_si128(c, 8), _mm_slli_si128(d, 4));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + 3 * i), o0);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + 3 * i + 16), o1);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + 3 * i + 32), o2);
    }
    pack_scalar(src + 4 * i, n - i, dst + 3 * i);
}

}  // namespace

extern "C" int hmm_host_rgbx_to_rgb(const uint8_t* src_rgbx, size_t n_pixels, uint8_t* dst_rgb) {
    HMM_REQUIRE(src_rgbx && dst_rgb, HMM_E_INVALID, "host_rgbx_to_rgb: null pointer");
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3");
    if (have_ssse3) pack_ssse3(src_rgbx, n_pixels, dst_rgb);
    else pack_scalar(src_rgbx, n_pixels, dst_rgb);
    return HMM_OK;
}

extern "C" int hmm_host_arrow_rgbx_to_rgb(const void* arrow_array, int width, int height, int x0, int y0, int roi_w, int roi_h,
                                          uint8_t* dst_rgb) {
    HMM_REQUIRE(arrow_array && dst_rgb, HMM_E_INVALID, "host_arrow_rgbx_to_rgb: null pointer");
    HMM_REQUIRE(width >= 1 && height >= 1 && x0 >= 0 && y0 >= 0 && roi_w >= 1 && roi_h >= 1 && x0 + roi_w <= width && y0 + roi_h <= height,
                HMM_E_INVALID, "host_arrow_rgbx_to_rgb: window %dx%d at (%d,%d) outside the %dx%d image", roi_w, roi_h, x0, y0, width, height);
    const ArrowArray* a = static_cast<const ArrowArray*>(arrow_array);
    // Pillow's layout for a 4-bytes-per-pixel mode: fixed_size_list<uint8>[4] of width*height entries over one uint8 child
    HMM_REQUIRE(a->release != nullptr, HMM_E_INVALID, "host_arrow_rgbx_to_rgb: the array has been released");
    HMM_REQUIRE(a->n_children == 1 && a->children && a->children[0], HMM_E_INVALID,
                "host_arrow_rgbx_to_rgb: expected a fixed-size list with one child, found %lld children", (long long)a->n_children);
    const int64_t n_pixels = (int64_t)width * height;
    HMM_REQUIRE(a->length == n_pixels && a->null_count <= 0, HMM_E_INVALID,
                "host_arrow_rgbx_to_rgb: %lld pixels in the array, %lld expected", (long long)a->length, (long long)n_pixels);
    const ArrowArray* c = a->children[0];
    HMM_REQUIRE(c->n_buffers == 2 && c->buffers && c->buffers[1], HMM_E_INVALID, "host_arrow_rgbx_to_rgb: the child has no data buffer");
    HMM_REQUIRE(c->length >= 4 * (a->offset + a->length), HMM_E_INVALID,
                "host_arrow_rgbx_to_rgb: the child holds %lld bytes, not four per pixel", (long long)c->length);
    const uint8_t* src = static_cast<const uint8_t*>(c->buffers[1]) + c->offset + 4 * a->offset;
    if (roi_w == width) return hmm_host_rgbx_to_rgb(src + (size_t)y0 * width * 4, (size_t)roi_w * roi_h, dst_rgb);
    for (int y = 0; y < roi_h; ++y) {
        const int rc = hmm_host_rgbx_to_rgb(src + ((size_t)(y0 + y) * width + x0) * 4, (size_t)roi_w, dst_rgb + (size_t)y * roi_w * 3);
        if (rc != HMM_OK) return rc;
    }
    return HMM_OK;
}
