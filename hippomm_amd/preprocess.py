"""Device-side vision preprocessing: decoded RGB uint8 frames -> (B,3,224,224) fp32 on the GPU,
bit-identical to the reference's host path (Pillow BICUBIC resize of the short side to 224, centre crop,
ToTensor, CLIP Normalize -- ``imagebind.data.load_and_transform_vision_data`` [upstream, recalled], called at
hippomm/models/foundation_models.py:87-90).

The host only computes Pillow's coefficient tables (a few hundred rows of taps per frame size, cached);
``hmm_preprocess_vision_u8`` does the two resample passes and the normalisation.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

OUT = 224
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5                                     # Pillow's bicubic_filter
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for BICUBIC, in the same double-precision order:
    -> (kk int32 [out_size][ksize], bounds int32 [out_size][2] = (first input index, tap count))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def resized_shape(h: int, w: int) -> Tuple[int, int]:
    """torchvision Resize(224): short side -> 224, long side int(224 * long / short)."""
    if w <= h:
        return int(OUT * h / w), OUT
    return OUT, int(OUT * w / h)


@lru_cache(maxsize=64)
def _plan(h: int, w: int):
    nh, nw = resized_shape(h, w)
    left, top = int(round((nw - OUT) / 2.0)), int(round((nh - OUT) / 2.0))       # torchvision CenterCrop
    kh, bh = resample_coeffs(w, nw)
    kv, bv = resample_coeffs(h, nh)
    kh, bh = np.ascontiguousarray(kh[left:left + OUT]), np.ascontiguousarray(bh[left:left + OUT])
    kv, bv = np.ascontiguousarray(kv[top:top + OUT]), np.ascontiguousarray(bv[top:top + OUT])
    row_first = int(bv[:, 0].min())
    row_last = int((bv[:, 0] + bv[:, 1]).max())
    return kh, bh, kv, bv, row_first, row_last


_dev_plans = {}


def preprocess_frames_device(frames_u8: torch.Tensor) -> torch.Tensor:
    """frames_u8: (B,H,W,3) uint8 CUDA tensor of decoded RGB frames (all the same size, e.g. one video) ->
    (B,3,224,224) fp32 CUDA tensor, bit-identical to the Pillow/torchvision host pipeline."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise ValueError(f"frames must be (B,H,W,3) uint8, got {tuple(frames_u8.shape)} {frames_u8.dtype}")
    frames_u8 = frames_u8.contiguous()
    B, H, W, _ = frames_u8.shape
    dev = frames_u8.device
    key = (H, W, dev)
    if key not in _dev_plans:
        kh, bh, kv, bv, r0, r1 = _plan(H, W)
        _dev_plans[key] = (tuple(torch.from_numpy(a).to(dev) for a in (kh, bh, kv, bv)), kh.shape[1], kv.shape[1], r0, r1)
    (kh_d, bh_d, kv_d, bv_d), ksh, ksv, r0, r1 = _dev_plans[key]
    out = torch.empty(B, 3, OUT, OUT, dtype=torch.float32, device=dev)
    need = lib.hmm_preprocess_vision_workspace_bytes(B, r1 - r0)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.hmm_preprocess_vision_u8(frames_u8.data_ptr(), B, H, W, kh_d.data_ptr(), bh_d.data_ptr(), ksh,
                                            kv_d.data_ptr(), bv_d.data_ptr(), ksv, r0, r1, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "hmm_preprocess_vision_u8")
    return out


def decode_rgb(image_paths: Sequence[str], workers: int = 0) -> List[np.ndarray]:
    """Paths -> uint8 (H,W,3) arrays in the given order.  JPEG/PNG decoding is the one step left on the host; Pillow's
    decoders release the GIL, so a small thread pool scales it with the cores (the resize that used to dominate the
    host time -- 9 ms per 1080p frame -- now runs on the GPU).  workers = 0: min(8, cpu count); 1: sequential."""
    from PIL import Image

    def one(path):
        with open(path, "rb") as fh:
            return np.asarray(Image.open(fh).convert("RGB"), dtype=np.uint8)

    if workers <= 0:
        import os
        workers = min(8, os.cpu_count() or 1)
    if workers == 1 or len(image_paths) < 4:
        return [one(p) for p in image_paths]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(one, image_paths))


def load_and_transform_vision_data_device(image_paths: Sequence[str], device=None, workers: int = 0) -> torch.Tensor:
    """Decode on the host (PIL, thread pool), resize / crop / normalise on the GPU.  Frames are grouped by size so that
    the frames of one video go through a single launch."""
    dev = device or _lib.require_gpu()
    decoded = decode_rgb(image_paths, workers)
    out = torch.empty(len(decoded), 3, OUT, OUT, dtype=torch.float32, device=dev)
    groups = {}
    for i, a in enumerate(decoded):
        groups.setdefault(a.shape[:2], []).append(i)
    for _, idxs in groups.items():
        batch = torch.from_numpy(np.stack([decoded[i] for i in idxs])).to(dev)
        out[torch.tensor(idxs, device=dev)] = preprocess_frames_device(batch)
    return out


# =====================================================================================================
# Audio (audio half): wav samples -> (B,3,1,128,204) normalised log-mel clips on the GPU.
# Mirrors imagebind.data.load_and_transform_audio_data [upstream, recalled] (called at
# hippomm/models/foundation_models.py:106-109): three 2-second clips spread evenly over the file
# (pytorchvideo ConstantClipsPerVideoSampler), per clip `waveform -= waveform.mean()`, kaldi fbank (128 mel bins,
# 25 ms / 10 ms, Hann, 16 kHz), zero-pad to 204 frames, Normalize(-4.268, 9.138).  The host reads the wav and slices
# the clips; hmm_audio_fbank does everything else.
# =====================================================================================================
AUDIO_SAMPLE_RATE = 16000
AUDIO_CLIP_DURATION = 2
AUDIO_CLIPS_PER_VIDEO = 3
AUDIO_MEAN, AUDIO_STD = -4.268, 9.138
AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH = 128, 204


def audio_clip_bounds(n_samples: int, sample_rate: int = AUDIO_SAMPLE_RATE) -> List[Tuple[int, int]]:
    """Sample ranges of the 3 clips: starts spread evenly over [0, duration - 2 s] in exact rational arithmetic,
    then `int(t * sample_rate)` as upstream slices the waveform."""
    from fractions import Fraction
    duration = n_samples / sample_rate
    max_start = Fraction(max(duration - AUDIO_CLIP_DURATION, 0))
    step = Fraction(max_start, max(AUDIO_CLIPS_PER_VIDEO - 1, 1))
    out = []
    for i in range(AUDIO_CLIPS_PER_VIDEO):
        start, end = step * i, step * i + AUDIO_CLIP_DURATION
        out.append((int(start * sample_rate), min(int(end * sample_rate), n_samples)))
    return out


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (samples float32 in [-1, 1] shaped (channels, n), sample_rate).  The reference writes its segment files with
    scipy.io.wavfile (hippocampal_memory.py:1219, float32) or ffmpeg pcm_s16le (:1386-1394); torchaudio.load
    normalises integer PCM to [-1, 1), which is reproduced here."""
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return np.ascontiguousarray(x.T), int(rate)


@lru_cache(maxsize=8)
def _resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase windowed-sinc kernels of torchaudio.functional.resample (sinc_interp_hann, its defaults), which
    upstream's load_and_transform_audio_data applies when a file is not at 16 kHz [upstream, recalled:
    torchaudio/functional/functional.py _get_sinc_resample_kernel].  orig / new are already divided by their gcd.
    -> (kernels (new, 1, 2*width + orig) fp32, width)."""
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    return kernels.to(torch.float32), width


def resample_waveform(waveform: torch.Tensor, orig_freq: int, new_freq: int = AUDIO_SAMPLE_RATE) -> torch.Tensor:
    """(channels, n) fp32 at orig_freq -> (channels, ceil(n * new / orig)) at new_freq; runs where `waveform` lives.
    One strided conv1d against `new` phase kernels (polyphase form), as torchaudio's _apply_sinc_resample_kernel."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernels, width = _resample_kernel(orig, new)
    x = waveform.to(torch.float32)
    length = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels.to(x.device), stride=orig)      # (channels, new, frames)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    return y[..., : math.ceil(new * length / orig)]


@lru_cache(maxsize=4)
def _fbank_tables(device_str: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Hann window (400) and kaldi mel banks (128, 257) in the float32 torch CPU arithmetic torchaudio uses
    (torchaudio.compliance.kaldi.get_mel_banks / _feature_window_function [upstream, recalled]), uploaded once."""
    window = torch.hann_window(400, periodic=False)
    num_bins, n_fft_bins, fft_bin_width = AUDIO_MEL_BINS, 256, AUDIO_SAMPLE_RATE / 512
    mel_low = 1127.0 * math.log(1.0 + 20.0 / 700.0)
    mel_high = 1127.0 * math.log(1.0 + 0.5 * AUDIO_SAMPLE_RATE / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(n_fft_bins)) / 700.0).log()).unsqueeze(0)
    banks = torch.max(torch.zeros(1), torch.min((mel - left) / (center - left), (right - mel) / (right - center)))
    banks = torch.nn.functional.pad(banks, (0, 1)).contiguous()
    return window.to(device_str), banks.to(device_str)


def melspec_clips_device(clips: torch.Tensor, mean: float = AUDIO_MEAN, std: float = AUDIO_STD) -> torch.Tensor:
    """clips: (n_clips, clip_len) fp32 CUDA, mono 16 kHz -> (n_clips, 128, 204) fp32 CUDA (asynchronous)."""
    lib = _lib.load()
    _lib.require_gpu()
    if clips.dim() != 2 or clips.dtype != torch.float32 or not clips.is_cuda:
        raise ValueError("clips must be a 2-D float32 CUDA tensor (n_clips, clip_len)")
    clips = clips.contiguous()
    n, clip_len = clips.shape
    out = torch.empty(n, AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH, dtype=torch.float32, device=clips.device)
    ws = torch.empty(lib.hmm_audio_fbank_workspace_bytes(n), dtype=torch.uint8, device=clips.device)
    window, banks = _fbank_tables(str(clips.device))
    _lib.check(lib.hmm_audio_fbank(clips.data_ptr(), n, clip_len, clip_len, window.data_ptr(), banks.data_ptr(),
                                   float(mean), float(std), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               "hmm_audio_fbank")
    return out


def transform_waveforms_device(waveforms: Sequence, device, sample_rate: int = AUDIO_SAMPLE_RATE) -> torch.Tensor:
    """waveforms: per file a (channels, n) or (n,) float array / tensor -> (B,3,1,128,204) fp32 on `device`.  Input at
    another rate than 16 kHz is resampled first (torchaudio's windowed-sinc polyphase filter, resample_waveform), as
    upstream does.  Channel 0 is analysed (kaldi.fbank's default channel).  Upstream's `waveform -= waveform.mean()` runs over all
    channels, but any constant offset is removed again per frame (remove_dc_offset), so for multi-channel input the
    result differs from the mono rule only in rounding; the reference always writes mono (:1204-1207, ffmpeg -ac 1)."""
    rates = list(sample_rate) if isinstance(sample_rate, (list, tuple)) else [sample_rate] * len(waveforms)
    groups = {}                                      # clip length -> [(file index, clip index, samples)]
    for fi, w in enumerate(waveforms):
        w = torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w, dtype=torch.float32)
        if w.dim() == 1:
            w = w[None]
        if rates[fi] != AUDIO_SAMPLE_RATE:
            w = resample_waveform(w[:1].to(device), int(rates[fi]), AUDIO_SAMPLE_RATE).cpu()
        for ci, (s, e) in enumerate(audio_clip_bounds(w.shape[1], AUDIO_SAMPLE_RATE)):
            groups.setdefault(e - s, []).append((fi, ci, w[0, s:e]))
    out = torch.empty(len(waveforms), AUDIO_CLIPS_PER_VIDEO, 1, AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH,
                      dtype=torch.float32, device=device)
    for clip_len, items in groups.items():
        batch = torch.stack([c for _, _, c in items]).to(device)
        mel = melspec_clips_device(batch)
        fidx = torch.tensor([f for f, _, _ in items], device=device)
        cidx = torch.tensor([c for _, c, _ in items], device=device)
        out[fidx, cidx, 0] = mel
    return out


def load_and_transform_audio_data_device(audio_paths: Sequence[str], device) -> torch.Tensor:
    """Drop-in for imagebind.data.load_and_transform_audio_data(audio_paths, device) on wav files (any sample rate;
    the reference itself writes 16 kHz, hippocampal_memory.py:1219, :1386-1394)."""
    waves, rates = [], []
    for p in audio_paths:
        x, rate = read_wav(p)
        waves.append(x)
        rates.append(rate)
    return transform_waveforms_device(waves, device, rates)
