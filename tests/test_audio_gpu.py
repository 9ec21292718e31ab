"""SURVEY.md 8f-2 on the device: visinger_amd/audio.py (torch.stft -> rocFFT on the MI355X, the spectrograms that feed the posterior
encoder and the mel loss of the training step) against the fp64 framed-DFT + HTK-mel oracle.  PARITY UNPINNED w.r.t. the
reference's torchaudio transforms (utils/audio/mel_processing.py:15-38): torchaudio is absent and nothing in the reference pins
it; this test catches device-side regressions of the restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_fft,win,hop,n_mels,sr,fmin,fmax,T,B", [(2048, 1200, 300, 128, 24000, 20.0, 12000.0, 64, 3),
                                                                  (2048, 1200, 256, 128, 22050, 20.0, 11025.0, 32, 16),
                                                                  (64, 32, 8, 16, 8000, 0.0, 4000.0, 48, 2)])
def test_device_spectrograms_match_fp64_oracle(oracle, n_fft, win, hop, n_mels, sr, fmin, fmax, T, B):
    from visinger_amd import audio
    g = torch.Generator().manual_seed(n_fft + hop + T)
    wav = torch.rand(B, T * hop, generator=g) - 0.5          # synthetic training waveforms ~ U(-0.5, 0.5)
    wav[0] = torch.sin(2 * np.pi * 440.0 * torch.arange(T * hop) / sr) * 0.8
    lin_ref = oracle.linear_spectrogram_f64(wav.numpy(), n_fft, win, hop)
    mel_ref = oracle.mel_spectrogram_f64(wav.numpy(), sr, n_fft, win, hop, n_mels, fmin, fmax)
    lin = audio.linear_spectrogram(wav.cuda(), n_fft, win, hop)
    mel = audio.mel_spectrogram(wav.cuda(), sr, n_fft, win, hop, n_mels, fmin, fmax)
    assert lin.is_cuda and lin.shape == (B, T, n_fft // 2 + 1) and mel.shape == (B, T, n_mels)
    assert np.abs(lin.cpu().double().numpy() - lin_ref).max() <= 1e-4 * lin_ref.max()
    assert np.abs(mel.cpu().double().numpy() - mel_ref).max() <= 2e-3
    # the masked mel-L1 of the training step (tasks/base.py:232-238) computed from device spectrograms equals the oracle's
    tgt = torch.from_numpy(mel_ref).float().cuda()
    assert float((mel - tgt).abs().mean()) <= 5e-4
