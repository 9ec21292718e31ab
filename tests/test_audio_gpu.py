"""SURVEY.md 8f-2 on the device: visinger_amd/audio.py (the framed DFT and the mel projection as convs on the HIP MFMA engine + the
power kernel of csrc/audio_ops.hip: the spectrograms that feed the posterior encoder and the mel loss of the training step) against
the fp64 framed-DFT + HTK-mel oracle, and against torch.stft (rocFFT) as a second, independent statement.  PARITY UNPINNED w.r.t. the
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
    # the transforms ran on the conv engine (no transform library on the product path) ...
    plan = audio._plan(n_fft, win, hop, lin.device)
    assert plan.fwd.kernel_instance().startswith("conv_"), plan.fwd.kernel_instance()
    # ... and agree with torch.stft on the device
    lin2 = audio.stft_spectrogram(wav.cuda(), n_fft, win, hop)
    assert float((lin - lin2).abs().max()) <= 1e-4 * float(lin2.max())
    with pytest.raises(RuntimeError):
        audio.linear_spectrogram(wav, n_fft, win, hop)               # CPU tensors are refused: no CPU path


@pytest.mark.parametrize("n_fft,win,hop,n_mels,sr,fmin,fmax,T,B", [(2048, 1200, 300, 128, 24000, 20.0, 12000.0, 27, 4),
                                                                  (1024, 1024, 256, 80, 22050, 0.0, 11025.0, 32, 16),
                                                                  (2048, 1200, 256, 128, 22050, 20.0, 11025.0, 130, 2),
                                                                  (64, 32, 8, 16, 8000, 0.0, 4000.0, 48, 2)])
def test_mel_loss_gradient_through_the_engine(n_fft, win, hop, n_mels, sr, fmin, fmax, T, B):
    """The mel-L1 of the training step is taken on GENERATED audio (tasks/base.py:232-238): d loss / d wav through the engine's DFT
    (grad-input conv with the transposed basis), vs_spec_power_bwd and the mel conv, against autograd through torch.stft; both the
    short-item (segments laid end to end) and the per-item layout."""
    from visinger_amd import audio
    g = torch.Generator().manual_seed(n_fft + hop + T)
    wav = ((torch.rand(B, T * hop, generator=g) - 0.5) * 0.9).cuda()
    tgt = torch.randn(B, T, n_mels, generator=g).cuda()
    grads = []
    for fn in (audio.mel_spectrogram, audio.stft_mel_spectrogram):
        w = wav.clone().requires_grad_(True)
        mel = fn(w, sr, n_fft, win, hop, n_mels, fmin, fmax)
        loss = (mel - tgt).abs().mean()
        grads.append((torch.autograd.grad(loss, w)[0], float(loss.detach())))
    (g1, l1), (g2, l2) = grads
    assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l2))
    assert torch.isfinite(g1).all()
    assert float((g1 - g2).abs().max()) <= 2e-4 * float(g2.abs().max()), (float((g1 - g2).abs().max()), float(g2.abs().max()))


def test_device_spectrograms_match_the_reference_fixture_when_there_is_one():
    """utils/audio/mel_processing.py wraps torchaudio, which neither the image nor /root/reference holds, so nothing reference-held pins
    visinger_amd/audio.py ("parity unpinned").  tests/golden/make_golden.py::gen_mel_processing writes mel_processing.npz the day torchaudio is importable in
    the build container; this test then holds the device transforms to it (linear power spectrogram: 1e-4 of its largest value; log-mel: 1e-3 abs)."""
    import os
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "mel_processing.npz")
    if not os.path.exists(path):
        pytest.skip("no mel_processing.npz: torchaudio was not importable when the fixtures were generated (parity unpinned, by declaration)")
    from visinger_amd import audio
    z = np.load(path)
    for tag in ("hop300", "hop256"):
        sr, n_fft, win, hop, n_mels, fmin, fmax = z[f"{tag}.params"].tolist()
        wav = torch.from_numpy(z[f"{tag}.wav"]).cuda()
        spec = audio.linear_spectrogram(wav, n_fft=int(n_fft), win_length=int(win), hop_length=int(hop)).cpu().numpy()
        mel = audio.mel_spectrogram(wav, sample_rate=int(sr), n_fft=int(n_fft), win_length=int(win), hop_length=int(hop), n_mels=int(n_mels), f_min=fmin,
                                    f_max=fmax).cpu().numpy()
        assert spec.shape == z[f"{tag}.spec"].shape and mel.shape == z[f"{tag}.mel"].shape
        assert np.abs(spec - z[f"{tag}.spec"]).max() <= 1e-4 * np.abs(z[f"{tag}.spec"]).max()
        assert np.abs(mel - z[f"{tag}.mel"]).max() <= 1e-3
