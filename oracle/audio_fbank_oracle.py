"""CPU oracle of the audio front end: wav samples -> (3,1,128,204) normalised log-mel clips.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

PARITY UNPINNED: the reference delegates this step to ``imagebind.data.load_and_transform_audio_data``
(hippomm/models/foundation_models.py:106-109), i.e. the upstream ImageBind package on top of torchaudio and
pytorchvideo -- none of which is under /root/reference or installed here (no torchaudio wheel, no network).  The
functions below restate the published algorithms from memory [upstream, recalled]:

* ``imagebind/data.py``: ``get_clip_timepoints`` + pytorchvideo ``ConstantClipsPerVideoSampler(clip_duration=2,
  clips_per_video=3)`` (clip starts spread evenly over [0, duration - 2]), ``waveform2melspec`` (``waveform -=
  waveform.mean()``; kaldi fbank; transpose; zero-pad / cut to 204 frames), ``Normalize(mean=-4.268, std=9.138)``;
* ``torchaudio.compliance.kaldi.fbank`` with the arguments ImageBind passes (htk_compat=True, sample_frequency=16000,
  use_energy=False, window_type="hanning", num_mel_bins=128, dither=0.0, frame_length=25, frame_shift=10) and its
  defaults: snip_edges, remove_dc_offset, preemphasis 0.97, round_to_power_of_two (512-point FFT), use_power,
  use_log_fbank, low_freq 20 Hz, high_freq 0 (= Nyquist), no VTLN warp, channel 0.

What pins it instead: closed-form known answers in tests/test_audio_fbank.py (frame count, Hann / mel-bank
identities, a pure tone lands in the analytically predicted mel bin with the analytically predicted energy,
a float64 evaluation of the same definition) -- properties of the published definition, not outputs of the
reference -- and an independent implementation: HuggingFace ``transformers.audio_utils.spectrogram`` configured as
its AST feature extractor's torchaudio-free path (kaldi-style) agrees with ``kaldi_fbank`` to 3e-5 on average.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import List, Tuple

import numpy as np
import torch

SAMPLE_RATE = 16000
NUM_MEL_BINS = 128
TARGET_LENGTH = 204
CLIP_DURATION = 2
CLIPS_PER_VIDEO = 3
AUDIO_MEAN = -4.268
AUDIO_STD = 9.138
WINDOW_SIZE, WINDOW_SHIFT, PADDED = 400, 160, 512          # 25 ms / 10 ms at 16 kHz, next power of two
EPSILON = torch.tensor(torch.finfo(torch.float32).eps)


def clip_timepoints(duration: float) -> List[Tuple[Fraction, Fraction]]:
    """pytorchvideo ConstantClipsPerVideoSampler(2, 3) driven by imagebind.data.get_clip_timepoints."""
    max_start = Fraction(max(duration - CLIP_DURATION, 0))
    step = Fraction(max_start, max(CLIPS_PER_VIDEO - 1, 1))
    return [(step * i, step * i + CLIP_DURATION) for i in range(CLIPS_PER_VIDEO)]


def mel_scale(freq):
    return 1127.0 * (1.0 + freq / 700.0).log()


def mel_banks() -> torch.Tensor:
    """kaldi get_mel_banks(128, 512, 16000, low 20, high 0): (128, 256) triangular weights in the mel domain."""
    num_fft_bins = PADDED // 2
    nyquist = 0.5 * SAMPLE_RATE
    low_freq, high_freq = 20.0, nyquist
    fft_bin_width = SAMPLE_RATE / PADDED
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (NUM_MEL_BINS + 1)
    b = torch.arange(NUM_MEL_BINS).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * torch.arange(num_fft_bins)).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def kaldi_fbank(waveform: torch.Tensor) -> torch.Tensor:
    """(channels, n) fp32 -> (frames, 128) log-mel energies, channel 0, snip_edges."""
    wave = waveform[0].to(torch.float32)
    n = wave.numel()
    if n < WINDOW_SIZE:
        return torch.empty(0, NUM_MEL_BINS)
    m = 1 + (n - WINDOW_SIZE) // WINDOW_SHIFT
    frames = wave.unfold(0, WINDOW_SIZE, WINDOW_SHIFT)[:m].clone()             # (m, 400)
    frames = frames - frames.mean(dim=1, keepdim=True)                        # remove_dc_offset
    prev = torch.nn.functional.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)
    frames = frames - 0.97 * prev[:, :-1]                                     # pre-emphasis
    frames = frames * torch.hann_window(WINDOW_SIZE, periodic=False).unsqueeze(0)
    frames = torch.nn.functional.pad(frames, (0, PADDED - WINDOW_SIZE))       # round_to_power_of_two
    spectrum = torch.fft.rfft(frames).abs().pow(2.0)                          # use_power, (m, 257)
    banks = torch.nn.functional.pad(mel_banks(), (0, 1))                      # (128, 257)
    energies = torch.mm(spectrum, banks.T)
    return torch.max(energies, EPSILON).log()                                 # use_log_fbank


def waveform2melspec(waveform_clip: torch.Tensor) -> torch.Tensor:
    """(channels, n) -> (1, 128, 204)."""
    clip = waveform_clip - waveform_clip.mean()
    fbank = kaldi_fbank(clip).transpose(0, 1)                                 # (128, frames)
    p = TARGET_LENGTH - fbank.size(1)
    if p > 0:
        fbank = torch.nn.functional.pad(fbank, (0, p), mode="constant", value=0)
    elif p < 0:
        fbank = fbank[:, :TARGET_LENGTH]
    return fbank.unsqueeze(0)


def load_and_transform_audio(waveform: torch.Tensor, sample_rate: int = SAMPLE_RATE) -> torch.Tensor:
    """One audio file's samples (channels, n) at 16 kHz -> (3, 1, 128, 204), normalised."""
    if sample_rate != SAMPLE_RATE:
        raise ValueError("resampling (torchaudio.functional.resample) is not restated: 16 kHz input only")
    clips = []
    for start, end in clip_timepoints(waveform.size(1) / sample_rate):
        clip = waveform[:, int(start * sample_rate): int(end * sample_rate)]
        clips.append((waveform2melspec(clip.clone()) - AUDIO_MEAN) / AUDIO_STD)
    return torch.stack(clips, dim=0)
