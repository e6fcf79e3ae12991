"""CPU restatement of the prompt feature path of the reference frontend — TEST INFRASTRUCTURE ONLY (tests/, smoke, bench's
cpu_baseline); the product path (cv2amd/prompt.py -> csrc/frontend.hip) never imports this.

  mel_spectrogram      third_party/Matcha-TTS/matcha/utils/audio.py:45-82 with the feat_extractor settings of
                       examples/libritts/cosyvoice2/conf/cosyvoice2.yaml:152-160 (n_fft 1920, 80 mels, 24 kHz, hop 480, win 1920,
                       fmin 0, fmax 8000, center False)
  mel_filterbank       librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm='slaney'), the call at audio.py:55
  resample_kernel      torchaudio.functional.functional._get_sinc_resample_kernel / _apply_sinc_resample_kernel behind
                       torchaudio.transforms.Resample(16000, 24000) (cli/frontend.py:497), defaults sinc_interp_hann,
                       lowpass_filter_width 6, rolloff 0.99

Pins: the STFT half is checked against torch.stft (the reference's own call, importable here).  librosa and torchaudio are third-party
dependencies absent from /root/reference and from this image (requirements.txt pins librosa==0.10.2, torchaudio==2.3.1): the
filterbank and the resampling kernel restate their published algorithms and are PARITY-UNPINNED beyond the structural properties in
tests/test_oracle_golden.py (filter areas, partition of unity of the polyphase kernel, DC gain).
"""
import math

import numpy as np
import torch


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0), lin)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))


def mel_filterbank(sr=24000, n_fft=1920, n_mels=80, fmin=0.0, fmax=8000.0):
    """Slaney-style triangular filters, area-normalised: [n_mels][n_fft // 2 + 1] float32."""
    fftfreqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    fb = np.maximum(0, np.minimum(lower, upper)) * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return fb.astype(np.float32)


def stft_mag(y, n_fft=1920, hop=480):
    """audio.py:60-77: reflect-pad (n_fft - hop) / 2, hann STFT (center False), sqrt(re^2 + im^2 + 1e-9).  y [1, n] -> [1, bins, frames]."""
    p = (n_fft - hop) // 2
    y = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode='reflect').squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                                         pad_mode='reflect', normalized=False, onesided=True, return_complex=True))
    return torch.sqrt(spec.pow(2).sum(-1) + 1e-9)


def stft_mag_direct(y, n_fft=1920, hop=480):
    """The same by the definition of the DFT in float64 (what the device kernel evaluates): pins stft_mag's conventions (window
    periodicity, sign, framing) independently of the FFT library."""
    p = (n_fft - hop) // 2
    yp = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode='reflect').squeeze(1)[0].double()
    frames = yp.unfold(0, n_fft, hop) * torch.hann_window(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None] * torch.arange(n_fft, dtype=torch.float64)[None, :]
    ang = 2 * math.pi * torch.remainder(k, n_fft) / n_fft
    re, im = frames @ torch.cos(ang).t(), -(frames @ torch.sin(ang).t())
    return torch.sqrt(re.float().pow(2) + im.float().pow(2) + 1e-9).t().unsqueeze(0)


def mel_spectrogram(y, n_fft=1920, n_mels=80, sr=24000, hop=480, fmin=0.0, fmax=8000.0, exact_dft=False):
    """y fp32 [1, n] in [-1, 1] -> log-mel [1, n_mels, frames] (audio.py:45-82).  exact_dft: the DFT by its definition in float64
    instead of torch.stft's fp32 FFT, whose round-off (~1e-7 of the frame's largest bin in EVERY bin) is what limits the agreement
    of any two implementations in the quiet bands of a frame."""
    fb = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    mag = stft_mag_direct(y, n_fft, hop) if exact_dft else stft_mag(y, n_fft, hop)
    return torch.log(torch.clamp(torch.matmul(fb, mag), min=1e-5))


def resample_kernel(orig_freq=16000, new_freq=24000, lowpass_filter_width=6, rolloff=0.99):
    """(kernel [new][2 * width + orig] float32, width, orig, new) after reduction by the gcd."""
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base_freq / orig)
    return kern.to(torch.float32)[:, 0, :].contiguous(), width, orig, new


def resample(x, orig_freq=16000, new_freq=24000):
    """x [1, n] -> [1, ceil(new * n / orig)]: zero-pad (width, width + orig), strided convolution, interleave the phases."""
    kern, width, orig, new = resample_kernel(orig_freq, new_freq)
    n = x.shape[1]
    xp = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(xp[:, None], kern[:, None], stride=orig)          # [1, new, frames]
    y = y.transpose(1, 2).reshape(1, -1)
    return y[:, :math.ceil(new * n / orig)]


def prompt_feat(speech_16k):
    """cli/frontend.py:497-498: resample to 24 kHz, mel, time-major: [1, frames, 80]."""
    return mel_spectrogram(resample(speech_16k)).squeeze(0).transpose(0, 1).unsqueeze(0)
