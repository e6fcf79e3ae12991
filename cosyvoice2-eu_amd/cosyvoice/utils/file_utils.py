"""cosyvoice/utils/file_utils.py:44-50 — load_wav(wav, target_sr): mono, resampled down to target_sr, float32 [1, n].

The reference uses torchaudio (soundfile backend + sinc resampler).  torchaudio is optional here: it is used when present,
otherwise 16-bit / 32-bit PCM and float WAV files are read with the standard library and resampled with the same windowed-sinc
polyphase kernel torchaudio's Resample builds (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; cv2amd/prompt.py:_sinc_kernel,
the table the device resampler uses), applied as a strided convolution.
"""
import logging
import wave

import numpy as np
import torch

logging.getLogger('matplotlib').setLevel(logging.WARNING)


def _read_wav(path):
    try:
        import soundfile as sf
        data, sr = sf.read(path, dtype='float32', always_2d=True)
        return torch.from_numpy(data.T.copy()), sr
    except ImportError:
        pass
    with wave.open(path, 'rb') as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype='<i4').astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError('unsupported WAV sample width {}'.format(sw))
    return torch.from_numpy(x.reshape(-1, nch).T.copy()), sr


def load_wav(wav, target_sr):
    try:
        import torchaudio
        speech, sample_rate = torchaudio.load(wav, backend='soundfile')
        speech = speech.mean(dim=0, keepdim=True)
        if sample_rate != target_sr:
            assert sample_rate > target_sr, 'wav sample rate {} must be greater than {}'.format(sample_rate, target_sr)
            speech = torchaudio.transforms.Resample(orig_freq=sample_rate, new_freq=target_sr)(speech)
        return speech
    except ImportError:
        pass
    speech, sample_rate = _read_wav(wav)
    speech = speech.mean(dim=0, keepdim=True)
    if sample_rate != target_sr:
        assert sample_rate > target_sr, 'wav sample rate {} must be greater than {}'.format(sample_rate, target_sr)
        from cv2amd.prompt import _sinc_kernel
        kern, width, orig, new = _sinc_kernel(int(sample_rate), int(target_sr))
        n = speech.shape[1]
        xp = torch.nn.functional.pad(speech, (width, width + orig))
        y = torch.nn.functional.conv1d(xp[:, None], torch.from_numpy(kern)[:, None], stride=orig)
        speech = y.transpose(1, 2).reshape(1, -1)[:, :-(-new * n // orig)].contiguous()
    return speech
