"""The reference's own waveform-distance measures, restated for the chain-level parity test (test infrastructure; librosa is absent here).

lsd_mel_db      evaluation/metrics_computer.py:311-360 `compute_lsd_mel_db`: 25 ms Hann window / 10 ms hop / n_fft 2048 power spectrogram ->
                80 Slaney mel bands (librosa.feature.melspectrogram defaults: fmin 0, fmax sr / 2, area-normalised, centred frames) ->
                power_to_db(ref=max, amin 1e-10, top_db 80) -> per frame sqrt(mean_k (ref_dB - syn_dB)^2) -> mean over frames.  The reference
                aligns the two signals by DTW first (they come from different systems); here both waveforms are renderings of the SAME
                tokens with the SAME noise, sample-aligned by construction, so the path is the diagonal.
pitch_metrics   evaluation/metrics_computer.py:550-650 `compute_pitch_metrics` on two f0 tracks (10 ms frames there; here the vocoder's own
                20 ms frames): gross pitch error % (|df0| / f0_ref > 20 % on frames voiced in both), RMSE in Hz and Pearson correlation on
                those frames, voiced / unvoiced mismatch %.  Voiced = f0 > 10 Hz, the rule of SineGen2 (hifigan/generator.py:322).
"""
import numpy as np
import torch


def _mel_db(y, sr=24000, n_mels=80):
    from oracle.frontend import mel_filterbank
    win, hop = int(0.025 * sr), int(0.010 * sr)
    n_fft = 2048 if win < 2048 else win
    w = torch.zeros(n_fft, dtype=torch.float64)
    w0 = (n_fft - win) // 2
    w[w0:w0 + win] = torch.hann_window(win, periodic=True, dtype=torch.float64)          # librosa pads the window to n_fft, centred
    yp = torch.nn.functional.pad(y.double().reshape(1, -1), (n_fft // 2, n_fft // 2))    # center=True, pad_mode='constant'
    spec = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=w, center=False, return_complex=True)[0]
    S = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2.0)).double() @ spec.abs().pow(2)
    db = 10.0 * torch.log10(S.clamp_min(1e-10)) - 10.0 * torch.log10(S.max().clamp_min(1e-10))
    return db.clamp_min(db.max() - 80.0)


def lsd_mel_db(ref, syn, sr=24000):
    a, b = _mel_db(ref, sr), _mel_db(syn, sr)
    n = min(a.shape[1], b.shape[1])
    return float((a[:, :n] - b[:, :n]).pow(2).mean(0).sqrt().mean())


def pitch_metrics(f0_ref, f0_syn, voiced_hz=10.0):
    r, s = np.asarray(f0_ref, dtype=np.float64).reshape(-1), np.asarray(f0_syn, dtype=np.float64).reshape(-1)
    vr, vs = r > voiced_hz, s > voiced_hz
    both = vr & vs
    out = {'vuv': float(np.mean(vr != vs) * 100.0), 'voiced_pairs': int(both.sum())}
    if both.sum() < 2:
        out.update(gpe=float('nan'), f0_rmse_hz=float('nan'), f0_corr=float('nan'))
        return out
    rr, ss = r[both], s[both]
    out['gpe'] = float(np.mean(np.abs(ss - rr) / rr > 0.2) * 100.0)
    out['f0_rmse_hz'] = float(np.sqrt(np.mean((ss - rr) ** 2)))
    out['f0_corr'] = float(np.corrcoef(rr, ss)[0, 1])
    return out
