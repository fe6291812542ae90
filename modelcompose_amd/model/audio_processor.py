"""BeatsAudioProcessor on the HIP path — host-side mirror of
modelcompose/model/multimodal_encoder/beats/audio_processor.py:37-175 for waveforms that are already decoded.

The reference decodes a file (torchaudio / moviepy), resamples to 16 kHz, scales by 2**15, runs torchaudio's Kaldi fbank on the
CPU, normalises with the BEATs statistics and pads / cuts to n_frames x frame_length rows.  Here everything after decoding is one
kernel (csrc/preprocess.hip); file decoding stays with the caller's data pipeline (out of scope)."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib

FBANK_MEAN, FBANK_STD = 15.41663, 6.55582           # audio_processor.py:47-48


def _kaldi_mel_banks(num_bins=128, padded=512, sample_freq=16000.0, low_freq=20.0) -> np.ndarray:
    """Triangular filters on the Kaldi mel scale 1127 ln(1 + f/700) between low_freq and Nyquist: [num_bins, padded/2 + 1]."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(low_freq), mel(0.5 * sample_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
    m = mel((sample_freq / padded) * np.arange(padded // 2, dtype=np.float64))[None, :]
    w = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


class HipBeatsAudioProcessor:
    def __init__(self, sampling_rate=16000, n_frames=2, frame_length=512, is_eval=False, device="cuda"):
        if sampling_rate != 16000:
            raise ValueError("the BEATs front-end is defined for 16 kHz audio")
        self.sampling_rate, self.n_frames, self.frame_length, self.is_eval = sampling_rate, n_frames, frame_length, is_eval
        self.fbank_mean, self.fbank_std = FBANK_MEAN, FBANK_STD
        self.device = torch.device(device)
        i = np.arange(400, dtype=np.float64)
        win = ((0.5 - 0.5 * np.cos(2.0 * math.pi * i / 399.0)) ** 0.85).astype(np.float32)           # povey window
        mel = _kaldi_mel_banks()
        nz = mel > 0
        lo = np.where(nz.any(1), nz.argmax(1), 1).astype(np.int32)
        hi = np.where(nz.any(1), mel.shape[1] - 1 - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        self._win = torch.from_numpy(win).to(self.device)
        self._mel = torch.from_numpy(mel).contiguous().to(self.device)
        self._lo, self._hi = torch.from_numpy(lo).to(self.device), torch.from_numpy(hi).to(self.device)

    def fbank(self, waveforms: torch.Tensor, n_samples: torch.Tensor, frames_out: int, out_dtype=None) -> torch.Tensor:
        """waveforms [B, T] fp32 in [-1, 1] on the device, n_samples [B] int32 -> normalised log-mel [B, frames_out, 128]."""
        if not waveforms.is_cuda:
            raise ValueError("waveforms must be a device (HIP) tensor; this path has no CPU fallback")
        w = waveforms.to(torch.float32).contiguous()
        B = w.shape[0]
        out_dtype = _lib.storage_dtype() if out_dtype is None else out_dtype
        out = torch.empty(B, frames_out, 128, dtype=out_dtype, device=w.device)
        p16 = out.data_ptr() if out_dtype == _lib.storage_dtype() else None
        p32 = out.data_ptr() if out_dtype == torch.float32 else None
        if p16 is None and p32 is None:
            raise ValueError("out_dtype must be the library's storage dtype (bfloat16; float16 in the fp16 build) or float32")
        _lib.check(_lib.lib().mc_fbank_f32(w.data_ptr(), n_samples.to(w.device, torch.int32).data_ptr(), w.stride(0), B, self._win.data_ptr(),
                                           self._mel.data_ptr(), self._lo.data_ptr(), self._hi.data_ptr(), float(2 ** 15), self.fbank_mean,
                                           self.fbank_std, p16, p32, frames_out, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "mc_fbank_f32")
        return out

    def __call__(self, audio: Union[torch.Tensor, Sequence[torch.Tensor], str], start_sec=None, end_sec=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio: a decoded mono waveform [T] (or [1, T] / [2, T]: stereo is averaged, :136-137) at 16 kHz in [-1, 1], or a list of them
        (:94-108).  Returns (frames [N, 128], padding_mask [N] all False) like the reference; N = n_frames * frame_length in training
        mode, a multiple of frame_length in eval mode (:154-175)."""
        if isinstance(audio, str):
            raise NotImplementedError("file decoding (torchaudio.load / moviepy, audio_processor.py:52-66) is left to the data pipeline: "
                                      "pass the decoded 16 kHz waveform")
        if isinstance(audio, (list, tuple)):
            outs = [self(a) for a in audio]
            frames = torch.nn.utils.rnn.pad_sequence([o[0] for o in outs], batch_first=True, padding_value=0)
            masks = torch.nn.utils.rnn.pad_sequence([o[1] for o in outs], batch_first=True, padding_value=1)
            return frames, masks
        wav = audio
        if wav.dim() == 2:
            wav = wav.mean(0) if wav.shape[0] == 2 else wav[0]
        if wav.dim() != 1 or wav.numel() == 0:
            return torch.zeros(self.n_frames * self.frame_length, 128, dtype=_lib.storage_dtype(), device=self.device), \
                torch.zeros(self.n_frames * self.frame_length, dtype=torch.bool, device=self.device)
        wav = wav.to(self.device, torch.float32)
        T = wav.numel()
        m = 1 + (T - 400) // 160 if T >= 400 else 0
        if not self.is_eval:
            rows = self.frame_length * self.n_frames
        else:
            rows = ((m + (m % self.frame_length)) // self.frame_length) * self.frame_length        # pads by the remainder (:164-168)
        if rows == 0:
            return torch.zeros(0, 128, dtype=_lib.storage_dtype(), device=self.device), torch.zeros(0, dtype=torch.bool, device=self.device)
        fb = self.fbank(wav.view(1, -1), torch.tensor([T], dtype=torch.int32), rows)[0]
        return fb, torch.zeros(rows, dtype=torch.bool, device=self.device)
