"""GPU parity of the prompt feature path (csrc/frontend.hip through cv2_melspec / cv2_resample) against oracle/frontend.py, the CPU
restatement of matcha.utils.audio.mel_spectrogram (audio.py:45-82, cosyvoice2.yaml:152-160) and of torchaudio's
Resample(16000, 24000) (cli/frontend.py:497).  Tolerances: the device DFT accumulates in fp64 (exact to fp32 round-off of the
inputs), torch.stft is an fp32 FFT: magnitudes agree to ~1e-6 of the frame's largest bin, so log-mels agree to 2e-4 wherever the mel
energy is above the clamp; the resampler is a 16-tap fp32 dot product (1e-6).  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pf():
    assert torch.cuda.is_available(), 'needs a GPU'
    from cv2amd.prompt import PromptFeatures
    return PromptFeatures('cuda:0')


def _signals():
    g = torch.Generator().manual_seed(3)
    n = 24000 * 3 + 217                                   # not a multiple of the hop
    t = torch.arange(n, dtype=torch.float64) / 24000
    chirp = (0.5 * torch.sin(2 * np.pi * (80 * t + 1500 * t * t))).float()[None]
    noise = (torch.randn(1, 24000 * 2, generator=g) * 0.1)
    speechy = (0.3 * torch.sin(2 * np.pi * 140 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 3 * t))).float()[None] + 0.01 * torch.randn(1, n, generator=g)
    short = torch.randn(1, 1920, generator=g) * 0.05      # two frames after padding... (1920 + 1440 - 1920) / 480 + 1 = 4
    return dict(chirp=chirp, noise=noise, speechy=speechy, short=short)


@pytest.mark.parametrize('name', ['chirp', 'noise', 'speechy', 'short'])
def test_mel_matches_oracle(pf, name):
    from oracle import frontend as OF
    x = _signals()[name]
    want = OF.mel_spectrogram(x, exact_dft=True).squeeze(0).transpose(0, 1)  # [frames, 80], DFT by definition in float64
    fft32 = OF.mel_spectrogram(x).squeeze(0).transpose(0, 1)                  # the reference's torch.stft (fp32 FFT)
    got = pf.mel(x)[0].cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    live = want > np.log(1e-5) + 1.0                                          # well above the clamp
    assert live.any()
    err = (got - want).abs()
    assert err[live].max().item() < 2e-5, f'{name}: {err[live].max().item():.3e}'
    assert err.max().item() < 5e-2                                            # at the clamp a round-off of 1e-7 in the energy moves the log more
    assert (got >= np.log(1e-5) - 2e-6).all()                                  # fp32 log of the fp32 clamp value
    # against the reference's fp32 FFT: the device result is never further from it than the float64 DFT is (the FFT's own round-off,
    # ~1e-7 of the frame's largest bin in every bin, is what separates them in the quiet bands of a loud frame)
    assert (got - fft32).abs()[live].max().item() <= (want - fft32).abs()[live].max().item() + 2e-5


def test_resample_matches_oracle(pf):
    from oracle import frontend as OF
    g = torch.Generator().manual_seed(4)
    for n in (16000, 16000 * 3 + 1, 7):
        x = torch.randn(1, n, generator=g) * 0.2
        want = OF.resample(x)
        got = pf.resample(x).cpu()
        assert got.shape == want.shape == (1, -(-3 * n // 2))
        assert (got - want).abs().max().item() < 1e-6


def test_prompt_feat_end_to_end_and_errors(pf):
    from oracle import frontend as OF
    g = torch.Generator().manual_seed(5)
    t = torch.arange(16000 * 4, dtype=torch.float64) / 16000
    x = (0.4 * torch.sin(2 * np.pi * 220 * t) * torch.sin(2 * np.pi * 1.5 * t)).float()[None] + 0.02 * torch.randn(1, 64000, generator=g)
    want = OF.mel_spectrogram(OF.resample(x), exact_dft=True).squeeze(0).transpose(0, 1).unsqueeze(0)
    got = pf.prompt_feat(x).cpu()
    assert got.shape == want.shape == OF.prompt_feat(x).shape == (1, 200, 80) # 4 s -> 96 000 samples -> 200 frames
    live = want > np.log(1e-5) + 1.0
    assert (got - want).abs()[live].max().item() < 1e-4
    with pytest.raises(ValueError):
        pf.mel(torch.zeros(1, 300))                                           # shorter than the reflect padding


# ---- whisper log-mel and kaldi fbank (cli/frontend.py:262-283) -------------------------------------------------------------------
@pytest.fixture(scope='module')
def sf():
    from cv2amd.prompt import SpeechFeatures
    return SpeechFeatures('cuda:0')


def _signals16():
    g = torch.Generator().manual_seed(13)
    n = 16000 * 3 + 77
    t = torch.arange(n, dtype=torch.float64) / 16000
    return dict(chirp=(0.5 * torch.sin(2 * np.pi * (60 * t + 900 * t * t))).float()[None],
                noise=torch.randn(1, 16000 * 2, generator=g) * 0.1,
                speechy=(0.3 * torch.sin(2 * np.pi * 140 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 3 * t))).float()[None] + 0.01 * torch.randn(1, n, generator=g),
                short=torch.randn(1, 800, generator=g) * 0.05)


@pytest.mark.parametrize('name', ['chirp', 'noise', 'speechy', 'short'])
def test_whisper_log_mel_matches_oracle(sf, name):
    """Device features against the restated whisper.log_mel_spectrogram: the float64-DFT form within 1e-5 (after the /4 scaling, where the
    energy is well above the 1e-10 clamp), and never further from the package's fp32 torch.stft form than that form's own FFT round-off."""
    from oracle import frontend as OF
    x = _signals16()[name]
    want, fft32 = OF.whisper_log_mel(x, exact_dft=True), OF.whisper_log_mel(x)
    got = sf.whisper_log_mel(x).cpu()
    assert got.shape == want.shape == (1, 128, x.shape[1] // 160) and torch.isfinite(got).all()
    live = want > want.max() - 1.5                      # (x + 4) / 4: within 6 decades of the loudest bin, away from the max - 8 clamp
    assert live.any()
    assert (got - want).abs()[live].max().item() < 1e-5
    assert (got - want).abs().max().item() < 2e-2        # at the clamps a 1e-7 round-off of the energy moves the log more
    assert (got - fft32).abs()[live].max().item() <= (want - fft32).abs()[live].max().item() + 1e-5


@pytest.mark.parametrize('name', ['chirp', 'noise', 'speechy', 'short'])
def test_kaldi_fbank_matches_oracle(sf, name):
    from oracle import frontend as OF
    x = _signals16()[name]
    want, fft32 = OF.kaldi_fbank(x, exact_dft=True), OF.kaldi_fbank(x)
    got = sf.kaldi_fbank(x, subtract_mean=False).cpu()
    assert got.shape == want.shape == (1 + (x.shape[1] - 400) // 160, 80) and torch.isfinite(got).all()
    live = want > want.max() - 11.5                     # natural log: within 5 decades of the loudest bin
    assert (got - want).abs()[live].max().item() < 1e-4
    assert (got - fft32).abs()[live].max().item() <= (want - fft32).abs()[live].max().item() + 1e-4
    # frontend.py:278: feat - feat.mean(dim=0, keepdim=True)
    got_m = sf.kaldi_fbank(x).cpu()
    assert (got_m - (got - got.mean(dim=0, keepdim=True))).abs().max().item() < 2e-5
    with pytest.raises(ValueError):
        sf.kaldi_fbank(torch.zeros(1, 399))
