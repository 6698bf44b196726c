// Waveform -> normalised 128-bin log-mel filterbank clips on gfx950 (audio half).
// Replaces, per 2-second clip, waveform2melspec + Normalize of imagebind.data.load_and_transform_audio_data
// [upstream, recalled; called at hippomm/models/foundation_models.py:106-109], i.e.
// torchaudio.compliance.kaldi.fbank(htk_compat=True, sample_frequency=16000, use_energy=False,
// window_type="hanning", num_mel_bins=128, dither=0, frame_length=25, frame_shift=10) with its defaults
// (snip_edges, remove_dc_offset, preemphasis 0.97, round_to_power_of_two, use_power, use_log_fbank, low_freq 20,
// high_freq = Nyquist), restated in oracle/audio_fbank_oracle.py.
//
// One workgroup per (clip, frame): 400 samples -> minus clip mean -> minus frame mean -> pre-emphasis -> Hann ->
// 512-point real DFT -> power -> 128 triangular mel filters -> log -> (x - mean) / std, written time-minor
// (clip, mel, frame) as the audio tower expects.  The DFT is the direct sum with a 512-entry twiddle table in LDS
// (thread k owns bin k, 400 x 2 FMA): 0.2 MFLOP per frame, microseconds for a whole video -- an FFT would not
// be measurably faster here, and the direct sum has no butterfly rounding accumulation.  HBM traffic is the
// waveform (2.5 reads per sample because frames overlap; L2 absorbs it) plus 128 x 204 floats out per clip.
#include <math.h>
#include "hmm_common.h"

namespace hmm {

constexpr int kWin = 400, kShift = 160, kFft = 512, kBins = 257, kMel = 128, kFrames = 204;

struct FbankTables {          // workspace layout (floats)
    static constexpr int WINDOW = 0;                    // [400]
    static constexpr int TWIDDLE = 512;                 // [512] float2 (cos, sin)
    static constexpr int BANKS = 512 + 1024;            // [128][257]
    static constexpr int MEANS = BANKS + kMel * kBins + 64;   // [n_clips]
};

// mel(f) = 1127 ln(1 + f / 700), evaluated in fp32 as torchaudio's mel_scale does
__device__ __forceinline__ float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

__global__ __launch_bounds__(256) void fbank_tables_kernel(float* __restrict__ ws) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kWin)                                        // torch.hann_window(400, periodic=False)
        ws[FbankTables::WINDOW + i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(kWin - 1)));
    if (i < kFft) {
        double s, c;
        sincospi(2.0 * (double)i / (double)kFft, &s, &c);
        reinterpret_cast<float2*>(ws + FbankTables::TWIDDLE)[i] = make_float2((float)c, (float)s);
    }
    if (i < kMel * kBins) {                              // get_mel_banks(128, 512, 16000, 20, 0, ...) + one zero column
        const int m = i / kBins, k = i - m * kBins;
        float w = 0.f;
        if (k < kFft / 2) {
            const float mel_low = mel_scale(20.0f), mel_high = mel_scale(8000.0f);
            const float delta = (mel_high - mel_low) / (float)(kMel + 1);
            const float left = mel_low + (float)m * delta, center = mel_low + ((float)m + 1.0f) * delta;
            const float right = mel_low + ((float)m + 2.0f) * delta;
            const float mel = mel_scale(31.25f * (float)k);              // fft_bin_width = 16000 / 512
            const float up = (mel - left) / (center - left), down = (right - mel) / (right - center);
            w = fmaxf(0.0f, fminf(up, down));
        }
        ws[FbankTables::BANKS + i] = w;
    }
}

// waveform -= waveform.mean(): one workgroup per clip
__global__ __launch_bounds__(256) void clip_mean_kernel(const float* __restrict__ clips, int clip_len, size_t clip_stride,
                                                        float* __restrict__ means) {
    __shared__ float red[4];
    const float* x = clips + (size_t)blockIdx.x * clip_stride;
    float s = 0.f;
    for (int i = threadIdx.x; i < clip_len; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) means[blockIdx.x] = clip_len > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)clip_len : 0.f;
}

__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ clips, int clip_len, size_t clip_stride,
                                                    int n_frames, const float* __restrict__ ws, float norm_mean,
                                                    float inv_std, float* __restrict__ out) {
    __shared__ float xs[kWin + 1];
    __shared__ float2 tw[kFft];
    __shared__ float power[kBins + 3];
    __shared__ float red[4];
    const int tid = threadIdx.x, clip = blockIdx.y, frame = blockIdx.x;
    float* dst = out + ((size_t)clip * kMel) * kFrames + frame;
    if (frame >= n_frames) {                             // F.pad(fbank, (0, p), value=0) happens before Normalize
        if (tid < kMel) dst[(size_t)tid * kFrames] = (0.0f - norm_mean) * inv_std;
        return;
    }
    const float* x = clips + (size_t)clip * clip_stride + (size_t)frame * kShift;
    const float clip_mean = ws[FbankTables::MEANS + clip];
    for (int i = tid; i < kFft; i += 256) tw[i] = reinterpret_cast<const float2*>(ws + FbankTables::TWIDDLE)[i];
    float part = 0.f;
    for (int i = tid; i < kWin; i += 256) {
        const float v = x[i] - clip_mean;
        xs[i + 1] = v;
        part += v;
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float frame_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)kWin;     // remove_dc_offset
    if (tid == 0) xs[0] = xs[1];                         // replicate padding for the pre-emphasis of sample 0
    __syncthreads();
    // pre-emphasis on the DC-free frame, then the window:  y[i] = (x[i] - 0.97 x[i-1]) w[i]
    float y0 = 0.f, y1 = 0.f;
    {
        const int i = tid;
        y0 = ((xs[i + 1] - frame_mean) - 0.97f * (xs[i] - frame_mean)) * ws[FbankTables::WINDOW + i];
        if (i + 256 < kWin)
            y1 = ((xs[i + 257] - frame_mean) - 0.97f * (xs[i + 256] - frame_mean)) * ws[FbankTables::WINDOW + i + 256];
    }
    __syncthreads();
    xs[tid] = y0;
    if (tid + 256 < kWin) xs[tid + 256] = y1;
    __syncthreads();
    // direct DFT: thread k -> bin k (and thread 0 also bin 256)
    {
        float re = 0.f, im = 0.f, re2 = 0.f;
        int idx = 0;
        for (int n = 0; n < kWin; ++n) {
            const float v = xs[n];
            const float2 t = tw[idx];
            re = fmaf(v, t.x, re);
            im = fmaf(v, t.y, im);                       // sign irrelevant for the power
            if (tid == 0) re2 = fmaf(v, (n & 1) ? -1.0f : 1.0f, re2);    // bin 256: cos(pi n), sin = 0
            idx = (idx + tid) & (kFft - 1);
        }
        power[tid] = re * re + im * im;
        if (tid == 0) power[256] = re2 * re2;
    }
    __syncthreads();
    if (tid < kMel) {
        const float* bank = ws + FbankTables::BANKS + tid * kBins;
        float e = 0.f;
        for (int k = 0; k < kBins; ++k) e = fmaf(power[k], bank[k], e);
        const float lg = logf(fmaxf(e, 1.1920928955078125e-07f));       // max(mel_energies, float32 eps).log()
        dst[(size_t)tid * kFrames] = (lg - norm_mean) * inv_std;
    }
}

}  // namespace hmm

using namespace hmm;

extern "C" size_t hmm_audio_fbank_workspace_bytes(int n_clips) {
    if (n_clips < 1) return 0;
    return ((size_t)FbankTables::MEANS + (size_t)n_clips + 64) * sizeof(float);
}

extern "C" int hmm_audio_fbank(const float* clips_dev, int n_clips, int clip_len, int64_t clip_stride,
                               const float* window_dev, const float* mel_banks_dev, float norm_mean,
                               float norm_std, float* out_dev, void* workspace_dev, size_t workspace_bytes,
                               hmm_stream_t stream) {
    HMM_REQUIRE(clips_dev && out_dev && workspace_dev, HMM_E_INVALID, "audio_fbank: null pointer");
    HMM_REQUIRE(n_clips >= 1 && n_clips <= 65535 && clip_len >= 0 && clip_stride >= clip_len, HMM_E_INVALID,
                "audio_fbank: n_clips=%d clip_len=%d clip_stride=%lld", n_clips, clip_len, (long long)clip_stride);
    HMM_REQUIRE(norm_std > 0.f, HMM_E_INVALID, "audio_fbank: std must be positive");
    HMM_REQUIRE(workspace_bytes >= hmm_audio_fbank_workspace_bytes(n_clips), HMM_E_WORKSPACE,
                "audio_fbank: workspace %zu < required %zu", workspace_bytes, hmm_audio_fbank_workspace_bytes(n_clips));
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace_dev);
    int n_frames = clip_len >= kWin ? 1 + (clip_len - kWin) / kShift : 0;       // snip_edges
    if (n_frames > kFrames) n_frames = kFrames;                                  // fbank[:, :target_length]
    fbank_tables_kernel<<<(kMel * kBins + 255) / 256, 256, 0, st>>>(ws);       // twiddles; window / banks unless supplied
    HMM_LAUNCH_CHECK();
    // Caller-supplied tables (e.g. computed with the very float32 ops torchaudio uses) replace the generated ones: the
    // weight of a triangle that barely touches an FFT bin is a cancelling difference of two mel values, so it is only
    // reproducible to the last bit with the same log() implementation.
    if (window_dev) HMM_HIP_CHECK(hipMemcpyAsync(ws + FbankTables::WINDOW, window_dev, kWin * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (mel_banks_dev) HMM_HIP_CHECK(hipMemcpyAsync(ws + FbankTables::BANKS, mel_banks_dev, (size_t)kMel * kBins * sizeof(float), hipMemcpyDeviceToDevice, st));
    clip_mean_kernel<<<n_clips, 256, 0, st>>>(clips_dev, clip_len, (size_t)clip_stride, ws + FbankTables::MEANS);
    HMM_LAUNCH_CHECK();
    fbank_kernel<<<dim3(kFrames, n_clips), 256, 0, st>>>(clips_dev, clip_len, (size_t)clip_stride, n_frames, ws, norm_mean,
                                                        1.0f / norm_std, out_dev);
    HMM_LAUNCH_CHECK();
    return HMM_OK;
}
