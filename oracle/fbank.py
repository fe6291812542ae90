"""Oracle (TEST INFRASTRUCTURE): Kaldi-compatible log-mel filterbank front-end of the BEATs audio processor.

The reference calls `torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins=128, sample_frequency=16000,
frame_length=25, frame_shift=10)` and normalises with (x - 15.41663) / (2 * 6.55582)
(modelcompose/model/multimodal_encoder/beats/audio_processor.py:9-22, :143-152), then pads / cuts to n_frames * frame_length
rows (:154-170).  torchaudio==2.0.2 is a third-party dependency that is NOT under /root/reference and is not installed here:
**parity unpinned by the reference**.  This file restates the published algorithm of torchaudio's kaldi.fbank with its default
arguments (dither 0, snip_edges, remove_dc_offset, pre-emphasis 0.97, povey window, round_to_power_of_two, power spectrum,
HTK-free kaldi mel scale 1127 ln(1 + f/700), low_freq 20 Hz, high_freq = Nyquist, log of max(energy, float32 eps));
tests/test_oracle_golden.py cross-checks it against the independent numpy implementation in transformers.audio_utils
(spectrogram + mel_filter_bank(mel_scale='kaldi', triangularize_in_mel_space=True)), which is the strongest pin available."""
from __future__ import annotations

import math

import numpy as np

FBANK_MEAN, FBANK_STD = 15.41663, 6.55582
EPS = float(np.finfo(np.float32).eps)


def mel_banks(num_bins=128, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0) -> np.ndarray:
    """kaldi get_mel_banks: [num_bins, padded/2 + 1] (the Nyquist column is zero: torchaudio pads it)."""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    num_fft_bins = padded // 2
    fft_bin_width = sample_freq / padded
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
    m = mel(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def povey_window(n=400) -> np.ndarray:
    i = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * math.pi * i / (n - 1))) ** 0.85).astype(np.float32)


def kaldi_fbank(waveform: np.ndarray, num_mel_bins=128, sample_frequency=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0,
                preemphasis=0.97) -> np.ndarray:
    """waveform [n_samples] float32 (already scaled by 2**15) -> [n_frames, num_mel_bins] float32 log-mel energies."""
    x = np.asarray(waveform, dtype=np.float32)
    win = int(sample_frequency * frame_length_ms * 0.001)
    shift = int(sample_frequency * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    if len(x) < win:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    m = 1 + (len(x) - win) // shift
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx].astype(np.float32)
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)                 # replicate padding on the left
    fr = fr - np.float32(preemphasis) * prev
    fr = fr * povey_window(win)[None, :]
    fr = np.pad(fr, ((0, 0), (0, padded - win)))
    spec = np.abs(np.fft.rfft(fr.astype(np.float32), axis=1)).astype(np.float32) ** 2
    e = spec @ mel_banks(num_mel_bins, padded, sample_frequency).T
    return np.log(np.maximum(e, EPS)).astype(np.float32)


def beats_process_waveform(waveform: np.ndarray, n_frames=2, frame_length=512, is_eval=False):
    """BeatsAudioProcessor.__call__ after decoding (audio_processor.py:133-175): waveform [n_samples] in [-1, 1] at 16 kHz ->
    (fbank [n_frames * frame_length, 128] normalised and zero padded, padding_mask of zeros)."""
    fb = kaldi_fbank(np.asarray(waveform, dtype=np.float32) * np.float32(2 ** 15))
    fb = (fb - np.float32(FBANK_MEAN)) / np.float32(2 * FBANK_STD)
    if not is_eval:
        tot = frame_length * n_frames
        if fb.shape[0] < tot:
            fb = np.pad(fb, ((0, tot - fb.shape[0]), (0, 0)))
        fb = fb[:tot]
    else:
        pad = fb.shape[0] % frame_length          # the reference pads by the remainder (not to a multiple), :164-168
        if pad > 0:
            fb = np.pad(fb, ((0, pad), (0, 0)))
        fb = fb[:(fb.shape[0] // frame_length) * frame_length]
    return fb.astype(np.float32), np.zeros(fb.shape[0], dtype=bool)
