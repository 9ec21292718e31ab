"""On-GPU linear / mel spectrograms: what the reference gets from torchaudio in
``utils/audio/mel_processing.py:15-38`` (MelSpectrogram / Spectrogram with n_fft 2048, win 1200, hop 300, power 2,
centre + reflect padding, HTK mel scale, ``log(x + 1e-3)``, last frame dropped), restated on ``torch.stft``.

PARITY UNPINNED: torchaudio is a third-party dependency that is neither vendored under the reference tree nor
installed here, and the reference ships no test vector for it; the restatement follows torchaudio 0.11's documented
defaults (hann window, center=True, pad_mode='reflect', normalized=False, onesided, mel_scale='htk', norm=None).
Plumbing over PyTorch-ROCm ops; feeds the posterior encoder and the mel loss of the training step."""
import math

import torch


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(mel_scale='htk', norm=None) -> [n_freqs, n_mels]"""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def linear_spectrogram(wav, n_fft=2048, win_length=1200, hop_length=300, power=2.0):
    """wav [B, L] -> [B, T, n_fft/2+1] (power spectrogram, last frame dropped as the reference does)."""
    window = torch.hann_window(win_length, device=wav.device, dtype=wav.dtype)
    spec = torch.stft(wav, n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = spec.abs().pow(power)
    return spec[..., :-1].transpose(1, 2)


def mel_spectrogram(wav, sample_rate=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=128, f_min=20.0,
                    f_max=12000.0, eps=1e-3):
    """wav [B, L] -> log-mel [B, T, n_mels]"""
    lin = linear_spectrogram(wav, n_fft, win_length, hop_length)            # [B, T, F]
    fb = mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate).to(device=wav.device, dtype=wav.dtype)
    return torch.log(lin @ fb + eps)
