"""Audio front end: oracle known answers on the CPU, HIP parity on the GPU.

The oracle is unpinned by the reference (torchaudio / ImageBind absent, see its header); the CPU tests below pin it to
closed-form properties of the published kaldi-fbank definition instead."""
import math

import numpy as np
import pytest
import torch

from oracle import audio_fbank_oracle as fb


def test_clip_timepoints_match_constant_clips_sampler():
    assert [(float(a), float(b)) for a, b in fb.clip_timepoints(10.0)] == [(0.0, 2.0), (4.0, 6.0), (8.0, 10.0)]
    assert [(float(a), float(b)) for a, b in fb.clip_timepoints(2.0)] == [(0.0, 2.0)] * 3
    assert [(float(a), float(b)) for a, b in fb.clip_timepoints(1.0)] == [(0.0, 2.0)] * 3      # shorter than a clip
    t = fb.clip_timepoints(5.0)
    assert [float(a) for a, _ in t] == [0.0, 1.5, 3.0]
    from hippomm_amd.preprocess import audio_clip_bounds
    assert audio_clip_bounds(160000) == [(0, 32000), (64000, 96000), (128000, 160000)]
    assert audio_clip_bounds(16000) == [(0, 16000)] * 3
    assert audio_clip_bounds(80000) == [(0, 32000), (24000, 56000), (48000, 80000)]


def test_frame_count_and_padding():
    x = torch.randn(1, 32000)
    assert fb.kaldi_fbank(x).shape == (198, 128)                 # 1 + (32000 - 400) // 160
    assert fb.kaldi_fbank(torch.randn(1, 399)).shape == (0, 128)
    assert fb.kaldi_fbank(torch.randn(1, 400)).shape == (1, 128)
    m = fb.waveform2melspec(x)
    assert m.shape == (1, 128, 204) and torch.all(m[:, :, 198:] == 0)
    long = fb.waveform2melspec(torch.randn(1, 40000))            # 248 frames -> cut
    assert long.shape == (1, 128, 204)


def test_mel_banks_are_kaldi_triangles():
    banks = fb.mel_banks().double()
    assert banks.shape == (128, 256) and banks.min() >= 0 and banks.max() <= 1
    # interior FFT bins are covered by exactly two neighbouring triangles whose weights sum to 1
    mel = 1127.0 * np.log1p(31.25 * np.arange(256) / 700.0)
    lo, hi = 1127.0 * math.log1p(20 / 700.0), 1127.0 * math.log1p(8000 / 700.0)
    delta = (hi - lo) / 129
    inside = (mel > lo + delta) & (mel < hi - delta)
    np.testing.assert_allclose(banks.sum(0).numpy()[inside], 1.0, atol=2e-5)
    assert banks[:, 0].sum() == 0                                 # DC is below low_freq = 20 Hz
    # a filter narrower than one FFT bin can miss every bin: the known empty low bands of the 128-bin / 512-point setup
    assert int((banks.sum(1) == 0).sum()) >= 1


def test_pure_tone_lands_in_the_predicted_band_with_the_predicted_energy():
    """A sine at an exact FFT bin centre: the windowed, pre-emphasised power spectrum is known in closed form."""
    k0 = 64                                                       # 2000 Hz
    n = np.arange(32000)
    amp = 0.25
    x = torch.from_numpy((amp * np.sin(2 * np.pi * (k0 * 31.25) * n / 16000.0)).astype(np.float32))[None]
    got = fb.kaldi_fbank(x)                                       # (198, 128)
    # closed form in float64 for one frame (all frames are equal up to phase)
    frame = x[0, :400].double().numpy()
    frame = frame - frame.mean()
    pre = frame - 0.97 * np.concatenate([frame[:1], frame[:-1]])
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 399)
    spec = np.abs(np.fft.rfft(pre * win, 512)) ** 2
    banks = np.pad(fb.mel_banks().double().numpy(), ((0, 0), (0, 1)))
    want = np.log(np.maximum(banks @ spec, np.finfo(np.float32).eps))
    np.testing.assert_allclose(got[0].double().numpy(), want, rtol=0, atol=2e-3)
    top = int(got.mean(0).argmax())
    mel_k0 = 1127.0 * math.log1p(2000.0 / 700.0)
    lo, hi = 1127.0 * math.log1p(20 / 700.0), 1127.0 * math.log1p(8000 / 700.0)
    assert abs(top - ((mel_k0 - lo) / ((hi - lo) / 129) - 1)) <= 1.0   # centre of triangle `top` is nearest the tone


@pytest.mark.parametrize("kind", ["noise", "tones"])
def test_oracle_matches_transformers_audio_utils(kind):
    """Independent implementation of the same definition: HuggingFace `transformers.audio_utils.spectrogram` with the
    arguments its AST feature extractor uses when torchaudio is absent (kaldi-style: remove_dc_offset, preemphasis 0.97,
    non-periodic Hann, 512-point power spectrum, kaldi mel scale triangularised in mel space, log with float32-eps floor),
    evaluated in float64.  Not the reference's dependency, but written independently of this repository."""
    au = pytest.importorskip("transformers.audio_utils")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # "at least one mel filter has all zero values": true, see above
        mel_filters = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=8000,
                                         sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    ours = np.pad(fb.mel_banks().double().numpy(), ((0, 0), (0, 1))).T
    assert np.abs(mel_filters - ours).max() < 5e-5           # float32 vs float64 arithmetic
    w = _wave(2.0, kind)[0]
    hf = au.spectrogram(w.double().numpy(), au.window_function(400, "hann", periodic=False), frame_length=400,
                        hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97,
                        mel_filters=mel_filters, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    mine = fb.kaldi_fbank(w[None]).double().numpy()
    assert hf.shape == mine.shape == (198, 128)
    d = np.abs(hf - mine)
    assert d.mean() < 3e-5 and d.max() < 5e-3                # the max sits in bands ~70 dB below the frame maximum


def test_full_transform_shape_and_normalisation():
    w = torch.randn(1, 16000 * 7, generator=torch.Generator().manual_seed(0)) * 0.1
    out = fb.load_and_transform_audio(w)
    assert out.shape == (3, 1, 128, 204)
    assert torch.allclose(out[:, :, :, 198:], torch.full((3, 1, 128, 6), (0 + 4.268) / 9.138))
    with pytest.raises(ValueError):
        fb.load_and_transform_audio(w, 44100)


@pytest.mark.parametrize("orig", [44100, 48000, 22050, 8000])
def test_resample_to_16k(orig):
    """resample_waveform restates torchaudio.functional.resample (windowed sinc, Hann, lowpass_filter_width 6, rolloff
    0.99) [upstream, recalled; torchaudio is not in this image: unpinned].  Checked here by what the filter must do:
    length rule ceil(n * new / orig), a tone below the new Nyquist keeps amplitude and frequency, a tone above it is
    removed, and scipy's polyphase resampler (an independent design) agrees on band-limited input."""
    from scipy.signal import resample_poly
    from hippomm_amd.preprocess import resample_waveform
    n = orig * 2
    t = np.arange(n) / orig
    low = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    y = resample_waveform(torch.from_numpy(low)[None], orig, 16000)[0].numpy()
    assert y.shape[0] == int(np.ceil(n * 16000 / orig))
    tt = np.arange(y.shape[0]) / 16000.0
    mid = slice(200, -200)
    assert np.abs(y[mid] - np.sin(2 * np.pi * 1000.0 * tt)[mid]).max() < 2e-3
    if orig > 16000:
        # 2.5 kHz above the new Nyquist: past the transition band of the 6-zero-crossing Hann-windowed sinc (-44 dB)
        high = np.sin(2 * np.pi * (0.5 * 16000 + 2500.0) * t).astype(np.float32)
        assert np.abs(resample_waveform(torch.from_numpy(high)[None], orig, 16000)[0].numpy()[mid]).max() < 1e-2
    g = np.gcd(orig, 16000)
    rng = np.random.default_rng(orig)
    band = np.convolve(rng.standard_normal(n), np.hanning(64) / 32.0, mode="same").astype(np.float32)   # low-passed noise
    ours = resample_waveform(torch.from_numpy(band)[None], orig, 16000)[0].numpy()
    ref = resample_poly(band.astype(np.float64), 16000 // g, orig // g)[: ours.shape[0]]
    assert np.abs(ours[mid] - ref[mid]).max() < 0.02 * np.abs(ref).max()
    # identity when nothing has to change
    assert resample_waveform(torch.from_numpy(low)[None], 16000, 16000).shape[1] == n


def test_read_wav_round_trip(tmp_path):
    from scipy.io import wavfile
    from hippomm_amd.preprocess import read_wav
    x = np.clip(np.random.default_rng(0).standard_normal(5000) * 0.3, -0.99, 0.99).astype(np.float32)
    wavfile.write(tmp_path / "f32.wav", 16000, x)                 # as hippocampal_memory.py:1219
    got, rate = read_wav(str(tmp_path / "f32.wav"))
    assert rate == 16000 and got.shape == (1, 5000) and np.array_equal(got[0], x)
    wavfile.write(tmp_path / "s16.wav", 16000, (x * 32767).astype(np.int16))   # as ffmpeg pcm_s16le
    got, _ = read_wav(str(tmp_path / "s16.wav"))
    np.testing.assert_allclose(got[0], x, atol=2.0 / 32768 + 1e-5)       # truncation + the 32767/32768 scale


# ------------------------------------------------------------------------------------------ float64 evaluation
def f64_melspec(w: torch.Tensor):
    """The same definition evaluated in float64 (numpy FFT) with the float32 window-free formulas; returns the
    normalised (3,1,128,204) tensor and, per clip, the band energies (frames,128) for energy-aware tolerances."""
    banks = np.pad(fb.mel_banks().double().numpy(), ((0, 0), (0, 1)))
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 399)
    outs, energies = [], []
    x_all = w.double().numpy()
    for start, end in fb.clip_timepoints(x_all.shape[1] / 16000):
        x = x_all[0, int(start * 16000): int(end * 16000)]
        x = x - (x_all[:, int(start * 16000): int(end * 16000)]).mean()
        m = 1 + (len(x) - 400) // 160 if len(x) >= 400 else 0
        mel = np.zeros((128, 204))
        e = np.zeros((0, 128))
        if m > 0:
            fr = np.lib.stride_tricks.sliding_window_view(x, 400)[::160][:m]
            fr = fr - fr.mean(1, keepdims=True)
            fr = fr - 0.97 * np.concatenate([fr[:, :1], fr[:, :-1]], 1)
            spec = np.abs(np.fft.rfft(fr * win, 512, axis=1)) ** 2
            e = spec @ banks.T
            lg = np.log(np.maximum(e, np.finfo(np.float32).eps))
            mel[:, :min(m, 204)] = lg.T[:, :204]
        outs.append((mel[None] + 4.268) / 9.138)
        energies.append(e[:204])
    return torch.from_numpy(np.stack(outs)), energies


def assert_melspec_close(got: torch.Tensor, w: torch.Tensor, what: str):
    """fp32 transforms (the oracle's FFT as much as the kernel's direct DFT) carry an absolute error of ~1e-7 of the
    frame's LARGEST spectral line, so a band 60 dB below the frame maximum is only good to ~1e-3 relative.  Tolerance on
    the normalised log-mel: 1e-4 where the band holds >= 1e-5 of the frame's strongest band, 3e-3 below that
    (bands with no FFT bin at all sit at log(eps) exactly)."""
    ref, energies = f64_melspec(w)
    err = (got.double() - ref).abs()
    worst_solid = worst_weak = 0.0
    for c, e in enumerate(energies):
        m = e.shape[0]
        if m == 0:
            continue
        solid = torch.from_numpy(e >= 1e-5 * e.max(1, keepdims=True)).T            # (128, m)
        ec = err[c, 0, :, :m]
        worst_solid = max(worst_solid, ec[solid].max().item() if solid.any() else 0.0)
        worst_weak = max(worst_weak, ec[~solid].max().item() if (~solid).any() else 0.0)
        assert err[c, 0, :, m:].max().item() <= 1e-6 if m < 204 else True            # zero padding, exactly
    print(f"{what}: worst |diff| vs float64 -- solid bands {worst_solid:.2e}, weak bands {worst_weak:.2e}")
    assert worst_solid <= 1e-4 and worst_weak <= 3e-3


@pytest.mark.parametrize("seconds,kind", [(10.0, "noise"), (7.3, "tones"), (1.0, "noise")])
def test_oracle_matches_float64_evaluation(seconds, kind):
    w = _wave(seconds, kind)
    assert_melspec_close(fb.load_and_transform_audio(w), w, f"oracle {seconds}s {kind}")


def _wave(seconds, kind):
    n = int(seconds * 16000)
    g = torch.Generator().manual_seed(n)
    if kind == "noise":
        return torch.randn(1, n, generator=g) * 0.2 + 0.05               # with a DC offset
    t = torch.arange(n) / 16000.0
    return (0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.sin(2 * math.pi * 3333.0 * t)
            + 0.01 * torch.randn(n, generator=g))[None]


# ------------------------------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("seconds,kind", [(10.0, "noise"), (2.0, "noise"), (7.3, "tones"), (1.0, "noise"), (0.02, "noise")])
def test_hip_fbank_matches_oracle(seconds, kind):
    from hippomm_amd.preprocess import transform_waveforms_device
    w = _wave(seconds, kind)
    want = fb.load_and_transform_audio(w)
    got = transform_waveforms_device([w], torch.device("cuda"))[0].cpu()
    assert got.shape == (3, 1, 128, 204)
    assert_melspec_close(got, w, f"hip {seconds}s {kind}")
    assert (got - want).abs().max().item() <= 3e-3                    # and against the fp32 oracle itself


@pytest.mark.gpu
def test_hip_fbank_generated_tables_close_to_host_tables():
    """Null window / bank pointers: the library generates the tables on the device.  Same formulas, but the device
    log() differs from the host's in the last bit, which only matters for triangles that barely touch an FFT bin:
    compare on the bands whose energy is not ~0."""
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    clips = (torch.randn(2, 32000, generator=g) * 0.2).cuda()
    ws = torch.empty(lib.hmm_audio_fbank_workspace_bytes(2), dtype=torch.uint8, device="cuda")
    out = torch.empty(2, 128, 204, device="cuda")
    L.check(lib.hmm_audio_fbank(clips.data_ptr(), 2, 32000, 32000, None, None, -4.268, 9.138, out.data_ptr(),
                                ws.data_ptr(), ws.numel(), L.stream_ptr()), "fbank")
    from hippomm_amd.preprocess import melspec_clips_device
    ref = melspec_clips_device(clips)
    solid = fb.mel_banks().sum(1) > 0.05                         # bands with real support
    assert (out[:, solid] - ref[:, solid]).abs().max().item() <= 1e-4
    assert torch.isfinite(out).all()
    # argument checks
    assert lib.hmm_audio_fbank(clips.data_ptr(), 2, 32000, 100, None, None, -4.268, 9.138, out.data_ptr(),
                               ws.data_ptr(), ws.numel(), L.stream_ptr()) != 0
    assert lib.hmm_audio_fbank(clips.data_ptr(), 2, 32000, 32000, None, None, -4.268, 9.138, out.data_ptr(),
                               ws.data_ptr(), 16, L.stream_ptr()) != 0 and b"workspace" in lib.hmm_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("clip_len", [64000, 32123, 400, 399, 33000])
def test_hip_fbank_clip_lengths(clip_len):
    """Clips that are not 2 s long: more than 204 frames are cut, fewer are zero-padded, < 400 samples give no frame."""
    from hippomm_amd.preprocess import melspec_clips_device
    g = torch.Generator().manual_seed(clip_len)
    clips = torch.randn(5, clip_len, generator=g) * 0.3
    got = melspec_clips_device(clips.cuda()).cpu()
    for c in range(5):
        want = (fb.waveform2melspec(clips[c:c + 1].clone()) + 4.268) / 9.138
        assert (got[c] - want[0]).abs().max().item() <= 3e-3
    n_frames = max(0, 1 + (clip_len - 400) // 160) if clip_len >= 400 else 0
    if n_frames < 204:
        assert torch.allclose(got[:, :, n_frames:], torch.full((5, 128, 204 - n_frames), 4.268 / 9.138))


@pytest.mark.gpu
def test_hip_fbank_many_clips():
    from hippomm_amd.preprocess import melspec_clips_device
    clips = torch.randn(3000, 32000, generator=torch.Generator().manual_seed(1)) * 0.1
    got = melspec_clips_device(clips.cuda())
    assert got.shape == (3000, 128, 204) and torch.isfinite(got).all()
    want = (fb.waveform2melspec(clips[2999:3000].clone()) + 4.268) / 9.138
    assert (got[2999].cpu() - want[0]).abs().max().item() <= 3e-3


@pytest.mark.gpu
def test_hip_fbank_silence_and_batching():
    """All-zero input takes the max(energy, eps) branch; several files of different lengths in one call."""
    from hippomm_amd.preprocess import transform_waveforms_device
    g = torch.Generator().manual_seed(3)
    waves = [torch.zeros(1, 32000), torch.randn(2, 48000, generator=g) * 0.1, torch.randn(20000, generator=g) * 0.1]
    got = transform_waveforms_device(waves, torch.device("cuda")).cpu()
    assert got.shape == (3, 3, 1, 128, 204)
    want0 = fb.load_and_transform_audio(waves[0])
    assert torch.allclose(got[0], want0, atol=1e-6)              # log(eps) rows, zero padding
    assert_melspec_close(got[1], waves[1][:1], "batched file 1 (channel 0)")
    assert_melspec_close(got[2], waves[2][None], "batched file 2")


@pytest.mark.gpu
def test_imagebind_load_data_accepts_wav_paths(tmp_path):
    """foundation_models.py:93-109: {'audio': [path]} -> (B,3,1,128,204) on the device."""
    from scipy.io import wavfile
    from hippomm_amd.encoder import ImageBind, synthetic_state_dict
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(16000 * 4, generator=g) * 0.2).numpy().astype(np.float32)
    wavfile.write(tmp_path / "seg.wav", 16000, w)
    model = ImageBind(state_dict=synthetic_state_dict(("audio",), depth={"audio": 1}), towers=("audio",), depth={"audio": 1})
    data = model.load_data({"audio": [str(tmp_path / "seg.wav")]}, ["audio"])
    assert data["audio"].shape == (1, 3, 1, 128, 204) and data["audio"].is_cuda
    assert_melspec_close(data["audio"][0].cpu(), torch.from_numpy(w)[None], "load_data wav path")
    feats = model.extract_features({"audio": [str(tmp_path / "seg.wav")]}, ["audio"])["audio"]
    assert feats.shape == (1, 1024) and torch.isfinite(feats).all()
    # another sample rate: resampled to 16 kHz first, as upstream's load_and_transform_audio_data does
    from hippomm_amd.preprocess import resample_waveform
    w44 = (torch.randn(44100 * 4, generator=g) * 0.2).numpy().astype(np.float32)
    wavfile.write(tmp_path / "cd.wav", 44100, w44)
    data44 = model.load_data({"audio": [str(tmp_path / "cd.wav")]}, ["audio"])
    assert data44["audio"].shape == (1, 3, 1, 128, 204)
    assert_melspec_close(data44["audio"][0].cpu(), resample_waveform(torch.from_numpy(w44)[None], 44100, 16000),
                         "load_data 44.1 kHz wav path")
    # an unreadable file: logged and skipped like any per-modality failure (:110-112)
    (tmp_path / "bad.wav").write_bytes(b"not a wav file")
    assert "audio" not in model.load_data({"audio": [str(tmp_path / "bad.wav")]}, ["audio"])


@pytest.mark.gpu
def test_wav_paths_fast_path_equals_clip_by_clip(tmp_path):
    """load_and_transform_audio_data_device: samples as stored -> pinned buffer -> one upload -> one launch writing the
    (B,3,1,128,204) result in place.  Bitwise equal to the fbank of every clip launched on its own, for float32 and int16 files,
    stereo, and a call that mixes a file shorter than a clip with full-length ones (the grouped route)."""
    from scipy.io import wavfile
    from hippomm_amd.preprocess import (audio_clip_bounds, load_and_transform_audio_data_device, melspec_clips_device, read_wav)
    rng = np.random.default_rng(21)
    files = {"a_f32.wav": (rng.standard_normal(160000) * 0.1).astype(np.float32),
             "b_i16.wav": (rng.standard_normal(112345) * 3000).astype(np.int16),
             "c_stereo_i16.wav": (rng.standard_normal((48000, 2)) * 3000).astype(np.int16),
             "d_short.wav": (rng.standard_normal(20000) * 0.1).astype(np.float32)}
    for name, x in files.items():
        wavfile.write(tmp_path / name, 16000, x)

    def clip_by_clip(names):
        rows = []
        for name in names:
            w, rate = read_wav(str(tmp_path / name))
            assert rate == 16000
            for s, e in audio_clip_bounds(w.shape[1]):
                rows.append(melspec_clips_device(torch.from_numpy(w[0, s:e].copy())[None].cuda())[0])
        return torch.stack(rows).view(len(names), 3, 1, 128, 204)

    for names in (["a_f32.wav"], ["b_i16.wav", "a_f32.wav", "c_stereo_i16.wav"], ["a_f32.wav", "d_short.wav", "b_i16.wav"]):
        for _ in range(2):                                       # the second call reuses the pinned buffer
            got = load_and_transform_audio_data_device([str(tmp_path / n) for n in names], torch.device("cuda"))
            assert got.shape == (len(names), 3, 1, 128, 204) and torch.equal(got, clip_by_clip(names)), names


def test_fast_wav_reader_equals_scipy(tmp_path):
    """_read_wav_raw: the RIFF walk returns exactly what scipy.io.wavfile.read does (dtype, shape, values, rate) for the formats it
    takes itself -- 8 / 16 / 32-bit PCM and float32, mono and multi-channel, an odd-sized LIST chunk before the data -- and hands
    everything else (float64, a non-wav file) to scipy."""
    import struct
    import warnings
    from scipy.io import wavfile
    from hippomm_amd.preprocess import _read_wav_raw
    rng = np.random.default_rng(2)
    cases = {"f32.wav": (16000, (rng.standard_normal(5001) * 0.2).astype(np.float32)),
             "i16.wav": (16000, (rng.standard_normal(4000) * 3000).astype(np.int16)),
             "i16_stereo.wav": (44100, (rng.standard_normal((3000, 2)) * 3000).astype(np.int16)),
             "i32.wav": (8000, (rng.standard_normal(1000) * 1e8).astype(np.int32)),
             "u8.wav": (8000, rng.integers(0, 256, 999, dtype=np.uint8)),
             "f64.wav": (16000, rng.standard_normal(100)),                        # not taken by the fast route
             "f32_3ch.wav": (48000, (rng.standard_normal((777, 3)) * 0.1).astype(np.float32))}
    for name, (rate, x) in cases.items():
        wavfile.write(tmp_path / name, rate, x)
        got, got_rate = _read_wav_raw(str(tmp_path / name))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_rate, want = wavfile.read(tmp_path / name)
        assert got_rate == want_rate == rate and got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), name
    # a LIST chunk of odd size (padded to even) between fmt and data, as ffmpeg writes
    raw = (tmp_path / "i16.wav").read_bytes()
    at = raw.index(b"data")
    padded = raw[:at] + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" + raw[at:]
    padded = padded[:4] + struct.pack("<I", len(padded) - 8) + padded[8:]
    (tmp_path / "list.wav").write_bytes(padded)
    got, _ = _read_wav_raw(str(tmp_path / "list.wav"))
    assert np.array_equal(got, cases["i16.wav"][1])
    (tmp_path / "bad.wav").write_bytes(b"not a wav file")
    with pytest.raises(Exception):
        _read_wav_raw(str(tmp_path / "bad.wav"))
