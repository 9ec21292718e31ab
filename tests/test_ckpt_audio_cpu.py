"""Reference-format checkpoint round trip (legacy pickle, {'state_dict': {'model': ...}}) and shape/consistency checks
of the spectrogram restatement (parity of the latter is unpinned: torchaudio is absent, see visinger_amd/audio.py)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def test_checkpoint_roundtrip_reference_format(tmp_path):
    from visinger_amd import ckpt
    from visinger_amd.models.visinger import VISinger
    hp = json.load(open(os.path.join(GOLDEN, "visinger_tiny_hparams.json")))
    m = VISinger(13, 9, 7, hp)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, betas=(0.8, 0.99), eps=1e-9)
    p = ckpt.save_checkpoint(tmp_path / "model_ckpt_steps_1200.ckpt", {"model": m}, [opt], epoch=3, global_step=1200)
    ckpt.save_checkpoint(tmp_path / "model_ckpt_steps_800.ckpt", {"model": m}, [opt], epoch=2, global_step=800)
    assert not os.path.exists(str(p) + ".part")
    with open(p, "rb") as f:
        assert f.read(2) != b"PK", "the reference writes the legacy (non-zip) serialisation"
    assert [os.path.basename(q) for q in ckpt.all_checkpoints(str(tmp_path))] == ["model_ckpt_steps_1200.ckpt",
                                                                                  "model_ckpt_steps_800.ckpt"]
    raw, path = ckpt.read_checkpoint(str(tmp_path))
    assert set(raw) == {"epoch", "global_step", "checkpoint_callback_best", "optimizer_states", "state_dict"}
    assert raw["global_step"] == 1200 and path.endswith("1200.ckpt")
    m2 = VISinger(13, 9, 7, hp)
    step, _ = ckpt.load_model(m2, str(tmp_path), child="model", strict=True)
    assert step == 1200
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_spectrogram_shapes_and_energy():
    from visinger_amd import audio
    torch.manual_seed(0)
    hop, T = 300, 37
    wav = torch.randn(2, T * hop) * 0.1
    lin = audio.stft_spectrogram(wav, 2048, 1200, hop)
    assert lin.shape == (2, T, 1025) and (lin >= 0).all()          # T frames: center padding gives T+1, last dropped
    mel = audio.stft_mel_spectrogram(wav, 24000, 2048, 1200, hop, 128, 20.0, 12000.0)
    assert mel.shape == (2, T, 128) and torch.isfinite(mel).all()
    fb = audio.mel_filterbank(1025, 20.0, 12000.0, 128, 24000)
    assert fb.shape == (1025, 128) and (fb >= 0).all() and (fb.sum(0) > 0).all()
    # a pure tone lands in the bin torch.stft puts it in and in the matching mel band
    t = torch.arange(24000) / 24000.0
    tone = torch.sin(2 * np.pi * 3000.0 * t)[None]
    lin = audio.stft_spectrogram(tone, 2048, 1200, hop)
    assert abs(int(lin[0, 20].argmax()) - round(3000.0 / (24000 / 2048))) <= 1


def test_wav_writer_and_bucketing(tmp_path):
    """utils/audio/io.py:8-15 restated (visinger_amd/synth.py): int16 scaling with / without peak normalisation, the file
    scipy reads back, and the length bucketing of the batched synthesis driver."""
    import numpy as np
    from scipy.io import wavfile
    from visinger_amd.synth import bucket_by_length, save_wav, to_int16
    wav = np.array([0.0, 0.25, -0.5, 0.125], np.float32)
    assert to_int16(wav, norm=False).tolist() == [0, 8191, -16383, 4095]            # truncation toward zero, as astype does
    assert to_int16(wav, norm=True).tolist() == [0, 16383, -32767, 8191]
    path = str(tmp_path / "a.wav")
    save_wav(wav, path, 22050, norm=True)
    sr, data = wavfile.read(path)
    assert sr == 22050 and data.dtype == np.int16 and data.tolist() == [0, 16383, -32767, 8191]
    lengths = [100, 900, 400, 1000, 50, 410]
    batches = bucket_by_length(lengths, max_frames_per_batch=2000)
    assert sorted(i for b in batches for i in b) == list(range(6))                       # a partition
    for b in batches:
        assert max(lengths[i] for i in b) * len(b) <= 2000 or len(b) == 1               # padded-frame budget
        assert [lengths[i] for i in b] == sorted((lengths[i] for i in b), reverse=True)  # longest first


def test_spectrogram_against_fp64_dft_oracle(oracle):
    """The fp64 framed-DFT oracle (oracle/visinger_oracle.py) against torch.stft (visinger_amd/audio.py `stft_spectrogram`: the second
    statement of the same definition; the product path is the HIP conv engine and runs in tests/test_audio_gpu.py).  Still PARITY
    UNPINNED w.r.t. the reference's torchaudio transforms (utils/audio/mel_processing.py:15-38; torchaudio is absent): this pins
    the oracle the GPU test checks the engine against, and the DFT basis the engine multiplies with."""
    from visinger_amd import audio
    torch.manual_seed(3)
    for (n_fft, win, hop, n_mels, sr, fmin, fmax, T) in ((2048, 1200, 300, 128, 24000, 20.0, 12000.0, 21), (64, 32, 8, 16, 8000, 0.0, 4000.0, 40),
                                                         (1024, 1024, 256, 80, 22050, 0.0, 11025.0, 17)):
        wav = (torch.randn(2, T * hop) * 0.3).clamp(-1, 1)
        lin_ref = oracle.linear_spectrogram_f64(wav.numpy(), n_fft, win, hop)
        lin = audio.stft_spectrogram(wav, n_fft, win, hop).double().numpy()
        assert lin.shape == lin_ref.shape == (2, T, n_fft // 2 + 1)
        assert np.abs(lin - lin_ref).max() <= 1e-4 * lin_ref.max()
        fb_ref = oracle.mel_filterbank_f64(n_fft // 2 + 1, fmin, fmax, n_mels, sr)
        assert np.abs(audio.mel_filterbank(n_fft // 2 + 1, fmin, fmax, n_mels, sr).double().numpy() - fb_ref).max() <= 5e-5   # fp32 triangles (as torchaudio builds them) vs fp64
        mel_ref = oracle.mel_spectrogram_f64(wav.numpy(), sr, n_fft, win, hop, n_mels, fmin, fmax)
        mel = audio.stft_mel_spectrogram(wav, sr, n_fft, win, hop, n_mels, fmin, fmax).double().numpy()
        assert np.abs(mel - mel_ref).max() <= 2e-3                    # log(x + 1e-3): fp32 bins of ~1e-3 relative to eps
        # the engine's basis (host-built, fp64 -> fp32) reproduces the oracle's transform: frames x basis in fp64
        basis = audio.dft_basis(n_fft, win).double().numpy()                        # [2F, win]
        Fq, o = n_fft // 2 + 1, (n_fft - win) // 2
        xp = np.pad(wav.numpy().astype(np.float64), ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        frames = np.stack([xp[:, t * hop + o:t * hop + o + win] for t in range(T)], 1)          # [B, T, win]
        y = frames @ basis.T
        assert np.abs(y[..., :Fq] ** 2 + y[..., Fq:] ** 2 - lin_ref).max() <= 1e-6 * lin_ref.max()


def test_trainer_learning_rate_follows_the_reference_schedule():
    """tasks/visinger.py:221-227 with config/models/visinger.yaml:106 (`endless_ds: false`): both optimizers' rates are
    base * gamma ** EPOCH after every optimizer step -- not gamma ** step (0.999875 ** 100000 = 3.7e-6 would stall training);
    with `endless_ds: true` the exponent is global_step // accumulate_grad_batches."""
    from visinger_amd.train import VISingerTrainer
    hp = json.load(open(os.path.join(GOLDEN, "visinger_tiny_hparams.json")))
    tr = VISingerTrainer(13, 9, 7, hp).configure()
    tr.global_step = 5000
    tr.on_after_optimization()
    assert tr.opt_gen.param_groups[0]["lr"] == 2e-4 and tr.opt_disc.param_groups[0]["lr"] == 2e-4     # still epoch 0
    for _ in range(3):
        tr.on_epoch_end()
    tr.on_after_optimization()
    for o in (tr.opt_gen, tr.opt_disc):
        assert abs(o.param_groups[0]["lr"] - 2e-4 * 0.999875 ** 3) < 1e-18
    assert tr.sched[0].get_last_lr()[0] == tr.opt_gen.param_groups[0]["lr"]
    tr2 = VISingerTrainer(13, 9, 7, hp, dict(endless_ds=True, accumulate_grad_batches=2)).configure()
    tr2.global_step = 21
    tr2.on_after_optimization()
    assert abs(tr2.opt_disc.param_groups[0]["lr"] - 2e-4 * 0.999875 ** 10) < 1e-18
