"""CPU fp32 restatement of stage 3 (HiFT vocoder: F0 predictor -> NSF source -> ISTFTNet).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Plain functions over a state dict in the reference's `hift.pt` key schema:
  * HiFTGenerator.inference / decode / _stft / _istft   cosyvoice/hifigan/generator.py:570-582, 520-552, 504-518
  * ResBlock.forward :94-101 ; Snake cosyvoice/transformer/activation.py:73-84
  * ConvRNNF0Predictor.forward   cosyvoice/hifigan/f0_predictor.py:55-58
  * SourceModuleHnNSF2.forward :375-389 ; SineGen2.forward/_f02sine :256-339
The three RNG draws of the reference (`torch.rand(B,9)`, `randn_like(sine_waves)`, `randn_like(uv)` —
generator.py:270,334,388) are INJECTED as arguments (the third is discarded by the reference itself).
Pinned against the real reference modules by tests/golden/make_golden.py -> tests/golden/hift_*.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

UPS = [(8, 16), (5, 11), (3, 7)]
SCALE = 480


def wn(sd, name):
    """weight_norm parametrisation: w = g * v / ||v|| (norm over all dims but 0); plain weight if not parametrised."""
    if name + '.weight' in sd:
        return sd[name + '.weight']
    g = sd[name + '.parametrizations.weight.original0']
    v = sd[name + '.parametrizations.weight.original1']
    return torch._weight_norm(v, g, 0)      # v * (g / ||v||), the op torch.nn.utils.parametrizations uses


def snake(x, alpha):
    a = alpha.view(1, -1, 1)
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def resblock(sd, p, x, k, dils=(1, 3, 5)):
    for i, d in enumerate(dils):
        xt = snake(x, sd[f'{p}.activations1.{i}.alpha'])
        xt = F.conv1d(xt, wn(sd, f'{p}.convs1.{i}'), sd[f'{p}.convs1.{i}.bias'], padding=(k * d - d) // 2, dilation=d)
        xt = snake(xt, sd[f'{p}.activations2.{i}.alpha'])
        xt = F.conv1d(xt, wn(sd, f'{p}.convs2.{i}'), sd[f'{p}.convs2.{i}.bias'], padding=(k - 1) // 2)
        x = xt + x
    return x


def f0_predictor(sd, mel):
    x = mel
    for i in (0, 2, 4, 6, 8):
        x = F.elu(F.conv1d(x, wn(sd, f'f0_predictor.condnet.{i}'), sd[f'f0_predictor.condnet.{i}.bias'], padding=1))
    x = x.transpose(1, 2)
    return torch.abs(F.linear(x, sd['f0_predictor.classifier.weight'], sd['f0_predictor.classifier.bias']).squeeze(-1))


def source(sd, f0, rand_ini, noise):
    """f0 [B,T] -> s [B,1,480T].  rand_ini [B,9] (col 0 forced to 0), noise [B,480T,9] ~ N(0,1)."""
    f0u = f0[:, None].repeat_interleave(SCALE, dim=2).transpose(1, 2)          # nn.Upsample(nearest) -> [B, L, 1]
    fn = f0u * torch.arange(1, 10, dtype=torch.float32).view(1, 1, 9)
    rad = (fn / 24000) % 1
    ri = rand_ini.clone()
    ri[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ri
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / SCALE, mode='linear').transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * np.pi
    phase = F.interpolate(phase.transpose(1, 2) * SCALE, scale_factor=SCALE, mode='linear').transpose(1, 2)
    sines = torch.sin(phase) * 0.1
    uv = (f0u > 10).float()
    namp = uv * 0.003 + (1 - uv) * 0.1 / 3
    sw = sines * uv + namp * noise
    s = torch.tanh(F.linear(sw, sd['m_source.l_linear.weight'], sd['m_source.l_linear.bias']))
    return s.transpose(1, 2)


def stft(s):
    w = torch.from_numpy(_hann16())
    spec = torch.view_as_real(torch.stft(s, 16, 4, 16, window=w, return_complex=True))
    return spec[..., 0], spec[..., 1]


def _hann16():
    # scipy.signal.get_window('hann', 16, fftbins=True) == periodic hann
    n = np.arange(16)
    return (0.5 - 0.5 * np.cos(2 * np.pi * n / 16)).astype(np.float32)


def istft(mag, phase):
    mag = torch.clip(mag, max=1e2)
    w = torch.from_numpy(_hann16())
    return torch.istft(torch.complex(mag * torch.cos(phase), mag * torch.sin(phase)), 16, 4, 16, window=w)


def decode(sd, mel, s, return_pre=False):
    sr, si = stft(s.squeeze(1))
    s_stft = torch.cat([sr, si], dim=1)
    x = F.conv1d(mel, wn(sd, 'conv_pre'), sd['conv_pre.bias'], padding=3)
    sds = [(15, 7), (3, 1), (1, 0)]
    for i, (u, k) in enumerate(UPS):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, wn(sd, f'ups.{i}'), sd[f'ups.{i}.bias'], stride=u, padding=(k - u) // 2)
        if i == 2:
            x = F.pad(x, (1, 0), mode='reflect')
        st, pd = sds[i]
        si_ = F.conv1d(s_stft, sd[f'source_downs.{i}.weight'], sd[f'source_downs.{i}.bias'], stride=st, padding=pd)
        si_ = resblock(sd, f'source_resblocks.{i}', si_, [7, 7, 11][i])
        x = x + si_
        xs = None
        for j, kk in enumerate((3, 7, 11)):
            r = resblock(sd, f'resblocks.{i * 3 + j}', x, kk)
            xs = r if xs is None else xs + r
        x = xs / 3
    x = F.leaky_relu(x)
    x = F.conv1d(x, wn(sd, 'conv_post'), sd['conv_post.bias'], padding=3)
    if return_pre:
        return x
    mag = torch.exp(x[:, :9])
    ph = torch.sin(x[:, 9:])
    return torch.clamp(istft(mag, ph), -0.99, 0.99)


def inference(sd, mel, cache_source, rand_ini, noise):
    """HiFTGenerator.inference: mel [1,80,T], cache_source [1,1,k] -> (wav [1,480T], source [1,1,480T])."""
    f0 = f0_predictor(sd, mel)
    s = source(sd, f0, rand_ini, noise)
    if cache_source.shape[2] != 0:
        s[:, :, :cache_source.shape[2]] = cache_source
    return decode(sd, mel, s), s
