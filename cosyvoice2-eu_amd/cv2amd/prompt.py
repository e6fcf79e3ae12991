"""Host side of the prompt feature extraction on the device (csrc/frontend.hip): tables for cv2_melspec / cv2_resample and the
call `prompt_feat(speech_16k) -> [1, frames, 80]` that replaces `Resample(16000, 24000)` + `feat_extractor` of the reference
frontend (cli/frontend.py:497-498; matcha/utils/audio.py:45-82).  No CPU fallback: without libcv2amd.so and a GPU this raises.

The tables are plain numbers (Hann window, DFT twiddles, Slaney mel filterbank as librosa.filters.mel defines it, torchaudio's
sinc_interp_hann polyphase kernel), computed here in float64 and rounded once to fp32."""
import ctypes as C
import math

import numpy as np
import torch

from . import lib as L


class MelCfg(C.Structure):
    _fields_ = [('n_fft', C.c_int32), ('hop', C.c_int32), ('n_mels', C.c_int32), ('n_bins', C.c_int32), ('window', C.c_void_p),
                ('twiddle', C.c_void_p), ('mel_fb', C.c_void_p), ('fb_lo', C.c_void_p), ('fb_hi', C.c_void_p), ('clamp_min', C.c_float)]


def _bind(lib):
    if getattr(lib, '_prompt_bound', False):
        return
    lib.cv2_melspec.argtypes = [C.POINTER(MelCfg), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.cv2_resample.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                 C.c_void_p]
    lib._prompt_bound = True


def _slaney_fb(sr, n_fft, n_mels, fmin, fmax):
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3))

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))
    freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    pts = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    d = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    fb = np.maximum(0, np.minimum(-ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]))
    return (fb * (2.0 / (pts[2:] - pts[:-2]))[:, None]).astype(np.float32)


def _sinc_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.clip((np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base, -lowpass_filter_width, lowpass_filter_width)
    win = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid='ignore', divide='ignore'):
        k = np.where(t == 0, 1.0, np.sin(t) / t) * win * (base / orig)
    return k.astype(np.float32), width, orig, new


class PromptFeatures:
    """feat_extractor + resampler of the reference frontend on one device."""

    def __init__(self, device='cuda:0', sr=24000, n_fft=1920, hop=480, n_mels=80, fmin=0.0, fmax=8000.0, src_sr=16000):
        if not torch.cuda.is_available():
            raise L.Cv2Error('PromptFeatures needs a GPU: the prompt features have no CPU fallback in this build')
        self.device = dev = torch.device(device)
        self.lib = L.lib()
        _bind(self.lib)
        self.n_fft, self.hop, self.n_mels = n_fft, hop, n_mels
        j = np.arange(n_fft, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2 * math.pi * j / n_fft)                       # torch.hann_window(n): periodic
        tw = np.stack([np.cos(2 * math.pi * j / n_fft), np.sin(2 * math.pi * j / n_fft)], 1)
        fb = _slaney_fb(sr, n_fft, n_mels, fmin, fmax)
        nz = fb > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), fb.shape[1] - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        kern, self._width, self._down, self._up = _sinc_kernel(src_sr, sr)
        self._t = dict(window=torch.from_numpy(window).to(dev), tw=torch.from_numpy(np.ascontiguousarray(tw)).to(dev),
                       fb=torch.from_numpy(fb).contiguous().to(dev), lo=torch.from_numpy(lo).to(dev), hi=torch.from_numpy(hi).to(dev),
                       kern=torch.from_numpy(kern).contiguous().to(dev))
        t = self._t
        self._cfg = MelCfg(n_fft, hop, n_mels, n_fft // 2 + 1, t['window'].data_ptr(), t['tw'].data_ptr(), t['fb'].data_ptr(),
                           t['lo'].data_ptr(), t['hi'].data_ptr(), 1e-5)

    def resample(self, speech):
        """[1, n] at the source rate -> [1, ceil(up n / down)] (device)."""
        x = speech.reshape(-1).to(self.device, torch.float32).contiguous()
        n_out = -(-self._up * x.numel() // self._down)
        out = torch.empty(n_out, dtype=torch.float32, device=self.device)
        k = self._t['kern']
        L.check(self.lib.cv2_resample(x.data_ptr(), x.numel(), k.data_ptr(), self._up, self._down, k.shape[1], self._width, out.data_ptr(),
                                      n_out, L.stream_ptr()))
        return out.unsqueeze(0)

    def mel(self, speech_24k):
        """[1, n] -> log-mel [1, frames, n_mels] time-major (the layout of prompt_speech_feat), device."""
        x = speech_24k.reshape(-1).to(self.device, torch.float32).contiguous()
        pad = (self.n_fft - self.hop) // 2
        if x.numel() <= pad or x.numel() + 2 * pad < self.n_fft:
            raise ValueError(f'prompt of {x.numel()} samples is too short for a {self.n_fft}-point frame')
        frames = 1 + (x.numel() + 2 * pad - self.n_fft) // self.hop
        out = torch.empty(frames, self.n_mels, dtype=torch.float32, device=self.device)
        L.check(self.lib.cv2_melspec(C.byref(self._cfg), x.data_ptr(), x.numel(), out.data_ptr(), frames, L.stream_ptr()))
        self._keep = x
        return out.unsqueeze(0)

    def prompt_feat(self, speech_16k):
        """cli/frontend.py:497-498: Resample(16000, 24000) then feat_extractor, [1, frames, 80]."""
        return self.mel(self.resample(speech_16k))
