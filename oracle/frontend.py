"""CPU restatement of the prompt feature path of the reference frontend — TEST INFRASTRUCTURE ONLY (tests/, smoke, bench's
cpu_baseline); the product path (cv2amd/prompt.py -> csrc/frontend.hip) never imports this.

  mel_spectrogram      third_party/Matcha-TTS/matcha/utils/audio.py:45-82 with the feat_extractor settings of
                       examples/libritts/cosyvoice2/conf/cosyvoice2.yaml:152-160 (n_fft 1920, 80 mels, 24 kHz, hop 480, win 1920,
                       fmin 0, fmax 8000, center False)
  mel_filterbank       librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm='slaney'), the call at audio.py:55
  resample_kernel      torchaudio.functional.functional._get_sinc_resample_kernel / _apply_sinc_resample_kernel behind
                       torchaudio.transforms.Resample(16000, 24000) (cli/frontend.py:497), defaults sinc_interp_hann,
                       lowpass_filter_width 6, rolloff 0.99
  whisper_log_mel      openai-whisper audio.py log_mel_spectrogram(audio, n_mels=128), the call of cli/frontend.py:264
  kaldi_fbank          torchaudio.compliance.kaldi.fbank(num_mel_bins=80, dither=0, sample_frequency=16000) (+ get_mel_banks), the call of
                       cli/frontend.py:277

Pins: the STFT half is checked against torch.stft (the reference's own call, importable here).  librosa and torchaudio are third-party
dependencies absent from /root/reference and from this image (requirements.txt pins librosa==0.10.2, torchaudio==2.3.1): the
filterbank and the resampling kernel restate their published algorithms and are PARITY-UNPINNED beyond the structural properties in
tests/test_oracle_golden.py (filter areas, partition of unity of the polyphase kernel, DC gain).  openai-whisper is absent too: whisper_log_mel
and kaldi_fbank restate the packages' published code with the same torch ops and are PARITY-UNPINNED as well (self-consistency of the
torch.stft / torch.fft form against the float64 DFT by definition is what the tests hold).
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


# ---- the feature extractors in front of the ONNX prompt models (cli/frontend.py:262-283): third-party algorithms, PARITY-UNPINNED ----
def whisper_log_mel(audio, n_mels=128, exact_dft=False):
    """openai-whisper audio.py log_mel_spectrogram (the call of cli/frontend.py:264; package absent here, its published code
    restated): hann(400) STFT at hop 160 (center, reflect), |X|^2 without the last frame, librosa Slaney filters (mel_filterbank
    above), log10(clamp 1e-10), max(x, max - 8), (x + 4) / 4.  audio fp32 [1, n] -> [1, n_mels, n // 160].  The STFT is torch.stft as
    in the package (exact_dft: the DFT by its definition in float64, what the device evaluates)."""
    a = audio.reshape(-1)
    if exact_dft:
        ap = torch.nn.functional.pad(a[None, None].double(), (200, 200), mode='reflect')[0, 0]
        fr = ap.unfold(0, 400, 160) * torch.hann_window(400, dtype=torch.float64)
        k = torch.arange(201, dtype=torch.float64)[:, None] * torch.arange(400, dtype=torch.float64)[None, :]
        ang = 2 * math.pi * torch.remainder(k, 400) / 400
        re, im = (fr @ torch.cos(ang).t()).float(), (-(fr @ torch.sin(ang).t())).float()
        mag = (re * re + im * im).t()[:, :-1]
    else:
        st = torch.stft(a, 400, 160, window=torch.hann_window(400), return_complex=True)
        mag = st[..., :-1].abs() ** 2
    fb = torch.from_numpy(mel_filterbank(16000, 400, n_mels, 0.0, 8000.0))
    log_spec = torch.clamp(fb @ mag, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).unsqueeze(0)


def kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN), in torch fp32 as the package computes it."""
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)       # noqa: E731
    lo, hi = 1127.0 * math.log(1.0 + low / 700.0), 1127.0 * math.log(1.0 + high / 700.0)
    delta = (hi - lo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    m = mel((sr / padded) * torch.arange(padded // 2)).unsqueeze(0)
    return torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))


def kaldi_fbank(waveform, num_mel_bins=80, sr=16000.0, exact_dft=False):
    """torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80, dither=0, sample_frequency=16000) with its defaults (the call of
    cli/frontend.py:277; package absent here, its published code restated with the same torch ops): snip_edges frames of 400 at stride
    160, remove_dc_offset, pre-emphasis 0.97 with the first sample replicated, povey window, zero padding to 512, |rfft|^2, mel banks
    above (+ a zero Nyquist column), log(max(., eps)).  [1, n] -> [frames, 80]."""
    x = waveform[0]
    m = 1 + (x.numel() - 400) // 160
    fr = x.unfold(0, 400, 160)[:m].clone()
    fr = fr - fr.mean(dim=1, keepdim=True)
    off = torch.nn.functional.pad(fr.unsqueeze(0), (1, 0), mode='replicate').squeeze(0)
    fr = fr - 0.97 * off[:, :-1]
    fr = fr * torch.hann_window(400, periodic=False).pow(0.85).unsqueeze(0)
    fr = torch.nn.functional.pad(fr, (0, 112))
    if exact_dft:
        k = torch.arange(257, dtype=torch.float64)[:, None] * torch.arange(512, dtype=torch.float64)[None, :]
        ang = 2 * math.pi * torch.remainder(k, 512) / 512
        f64 = fr.double()
        # the device applies window etc. in float64: redo the frame arithmetic there
        x64 = x.double().unfold(0, 400, 160)[:m]
        x64 = x64 - x64.mean(dim=1, keepdim=True)
        o64 = torch.cat([x64[:, :1], x64[:, :-1]], 1)
        f64 = torch.nn.functional.pad((x64 - 0.97 * o64) * torch.hann_window(400, periodic=False, dtype=torch.float64).pow(0.85), (0, 112))
        re, im = (f64 @ torch.cos(ang).t()).float(), (-(f64 @ torch.sin(ang).t())).float()
        power = re * re + im * im
    else:
        power = torch.fft.rfft(fr).abs().pow(2.0)
    banks = torch.nn.functional.pad(kaldi_mel_banks(num_mel_bins, 512, sr), (0, 1))
    return torch.max(power @ banks.t(), torch.tensor(torch.finfo(torch.float).eps)).log()
