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


def _on_own_device(fn):
    """run the method with the object's device current: L.stream_ptr() is the current device's stream, and a rank that lives on
    cuda:k must not launch on cuda:0's"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


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

    @_on_own_device
    def resample(self, speech):
        """[1, n] at the source rate -> [1, ceil(up n / down)] (device)."""
        x = speech.reshape(-1).to(self.device, torch.float32).contiguous()
        n_out = -(-self._up * x.numel() // self._down)
        out = torch.empty(n_out, dtype=torch.float32, device=self.device)
        k = self._t['kern']
        L.check(self.lib.cv2_resample(x.data_ptr(), x.numel(), k.data_ptr(), self._up, self._down, k.shape[1], self._width, out.data_ptr(),
                                      n_out, L.stream_ptr()))
        return out.unsqueeze(0)

    @_on_own_device
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


# ---- the feature extractors in front of the ONNX prompt models (cli/frontend.py:262-283) ------------------------------------------
class FrameFeatCfg(C.Structure):
    _fields_ = [('n_fft', C.c_int32), ('win', C.c_int32), ('hop', C.c_int32), ('n_bins', C.c_int32), ('n_mels', C.c_int32),
                ('window', C.c_void_p), ('twiddle', C.c_void_p), ('mel_fb', C.c_void_p), ('fb_lo', C.c_void_p), ('fb_hi', C.c_void_p),
                ('center', C.c_int32), ('remove_dc', C.c_int32), ('preemph', C.c_float), ('log10', C.c_int32), ('floor', C.c_float)]


def _kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks without VTLN: triangles in the mel domain 1127 ln(1 + f / 700) over the first
    padded / 2 FFT bins, one zero column appended (fbank pads the Nyquist bin): [num_bins][padded / 2 + 1].  Evaluated in torch fp32
    with the package's own expression order: its table carries fp32 round-off of ~1e-5 in the weights, which a float64 table would
    not reproduce (visible in bins that only receive the skirt of a loud neighbour)."""
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    lo, hi = 1127.0 * math.log(1.0 + low / 700.0), 1127.0 * math.log(1.0 + high / 700.0)
    delta = (hi - lo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    m = (1127.0 * (1.0 + ((sr / padded) * torch.arange(padded // 2)) / 700.0).log()).unsqueeze(0)
    bins = torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))
    return torch.nn.functional.pad(bins, (0, 1)).numpy().astype(np.float32)


class SpeechFeatures:
    """whisper.log_mel_spectrogram(speech, n_mels=128) and kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000)
    (- column mean) of the reference frontend (cli/frontend.py:264, 277-278) on one device: what feeds speech_tokenizer_v2.onnx and
    campplus.onnx.  16 kHz input, [1, n] in [-1, 1]."""

    def __init__(self, device='cuda:0'):
        if not torch.cuda.is_available():
            raise L.Cv2Error('SpeechFeatures needs a GPU: the prompt features have no CPU fallback in this build')
        self.device = dev = torch.device(device)
        self.lib = L.lib()
        self.lib.cv2_framefeat.argtypes = [C.POINTER(FrameFeatCfg), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        self.lib.cv2_whisper_post.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        self.lib.cv2_sub_col_mean.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        self._keep = []

        def tables(n_fft, window, fb):
            j = np.arange(n_fft, dtype=np.float64)
            tw = np.stack([np.cos(2 * math.pi * j / n_fft), np.sin(2 * math.pi * j / n_fft)], 1)
            nz = fb > 0
            lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
            hi = np.where(nz.any(1), fb.shape[1] - nz[:, ::-1].argmax(1), 0).astype(np.int32)
            t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (window, tw, fb, lo, hi)]
            self._keep.append(t)
            return t
        j = np.arange(400, dtype=np.float64)
        # whisper audio.py: torch.hann_window(400) (periodic), N_FFT 400, HOP 160, mel_filters(n_mels) = librosa.filters.mel(sr=16000, n_fft=400)
        w, tw, fb, lo, hi = tables(400, 0.5 - 0.5 * np.cos(2 * math.pi * j / 400), _slaney_fb(16000, 400, 128, 0.0, 8000.0))
        self._whisper = FrameFeatCfg(400, 400, 160, 201, 128, w.data_ptr(), tw.data_ptr(), fb.data_ptr(), lo.data_ptr(), hi.data_ptr(), 1, 0, 0.0, 1, 1e-10)
        # kaldi.py: povey window = hann(400, periodic=False) ** 0.85, round_to_power_of_two -> 512, preemphasis 0.97, remove_dc_offset,
        # log(max(., torch.finfo(float).eps))
        w, tw, fb, lo, hi = tables(512, (0.5 - 0.5 * np.cos(2 * math.pi * j / 399)) ** 0.85, _kaldi_mel_banks())
        self._kaldi = FrameFeatCfg(512, 400, 160, 257, 80, w.data_ptr(), tw.data_ptr(), fb.data_ptr(), lo.data_ptr(), hi.data_ptr(), 0, 1, 0.97, 0,
                                   float(np.finfo(np.float32).eps))

    @_on_own_device
    def _run(self, cfg, x, frames):
        out = torch.empty(frames, cfg.n_mels, dtype=torch.float32, device=self.device)
        L.check(self.lib.cv2_framefeat(C.byref(cfg), x.data_ptr(), x.numel(), out.data_ptr(), frames, L.stream_ptr()))
        return out

    @_on_own_device
    def whisper_log_mel(self, speech_16k):
        """[1, n] -> [1, 128, n // 160] (device), the tensor the reference hands to speech_tokenizer_v2.onnx."""
        x = speech_16k.reshape(-1).to(self.device, torch.float32).contiguous()
        if x.numel() <= 200:
            raise ValueError(f'{x.numel()} samples are too short for whisper features')
        frames = x.numel() // 160
        raw = self._run(self._whisper, x, frames)
        out = torch.empty(128, frames, dtype=torch.float32, device=self.device)
        L.check(self.lib.cv2_whisper_post(raw.data_ptr(), out.data_ptr(), frames, 128, L.stream_ptr()))
        return out.unsqueeze(0)

    @_on_own_device
    def kaldi_fbank(self, speech_16k, subtract_mean=True):
        """[1, n] -> [1 + (n - 400) // 160, 80] (device); subtract_mean: the `feat - feat.mean(dim=0)` of frontend.py:278."""
        x = speech_16k.reshape(-1).to(self.device, torch.float32).contiguous()
        if x.numel() < 400:
            raise ValueError(f'{x.numel()} samples are shorter than one 25 ms frame')
        frames = 1 + (x.numel() - 400) // 160
        out = self._run(self._kaldi, x, frames)
        if subtract_mean:
            L.check(self.lib.cv2_sub_col_mean(out.data_ptr(), frames, 80, L.stream_ptr()))
        return out
