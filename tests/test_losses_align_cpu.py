"""The training-step loss functions, get_note2dur and save_wav against the REFERENCE's own functions (fixtures written by
tests/golden/make_golden.py::gen_losses / gen_align_io, which imports tasks/visinger.py, tasks/base.py, utils/audio/align.py and
utils/audio/io.py and calls them on seeded inputs).  VERDICT r3 missing #2 / next #4, #9."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _losses(device):
    from visinger_amd import train
    z = np.load(os.path.join(GOLDEN, "losses.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(device)      # noqa: E731
    hp = json.loads(bytes(z["hparams_json"]).decode())
    got = {}
    got["mel_l1_unweighted"] = train.masked_l1(t("mel_out"), t("mel_tgt"))
    got["mel_l1"] = got["mel_l1_unweighted"] * train.TRAIN_HPARAMS["lambda_mel"]
    got["uv_loss"], got["f0_loss"] = train.pitch_losses(t("p_pred"), t("p_f0"), t("p_uv"), t("p_mel2ph"), train.TRAIN_HPARAMS["lambda_pitch"],
                                                        train.TRAIN_HPARAMS["lambda_pitch"])
    got["ctc_loss"] = train.ctc_loss(t("c_ph_pred"), t("c_text"), t("c_mel_len"), t("c_txt_len"), train.TRAIN_HPARAMS["lambda_ctc"])
    d_tgt, d_gen = [t(f"d_tgt{i}") for i in range(6)], [t(f"d_gen{i}") for i in range(6)]
    f_tgt = [[t(f"f_tgt{i}_{j}") for j in range(3)] for i in range(6)]
    f_gen = [[t(f"f_gen{i}_{j}") for j in range(3)] for i in range(6)]
    got["disc_loss"] = train.discriminator_loss(d_tgt, d_gen)
    got["gen_loss"] = train.generator_loss(d_gen)
    got["fm_loss"] = train.feature_matching_loss(f_tgt, f_gen)
    return z, hp, got


def _check_losses(device, rtol):
    from visinger_amd import train
    z, hp, got = _losses(device)
    for k, v in got.items():
        ref = float(z[k])
        assert abs(float(v) - ref) <= rtol * abs(ref), (k, float(v), ref)
    # the constants train.py carries are the YAML's (config/models/visinger.yaml:51-64), as recorded with the fixture
    T = train.TRAIN_HPARAMS
    assert hp["mel_losses"] == "l1:45.0" and T["lambda_mel"] == 45.0
    for k in ("lambda_pitch", "lambda_ctc", "lambda_mel_adv", "lambda_kl", "lambda_fm", "kl_start_steps", "kl_min"):
        assert T[k] == hp[k], k
    # KL weighting (tasks/visinger.py:103-107): clamp at kl_min, warm-up min(step / kl_start_steps, 1), weight lambda_kl
    kl = torch.tensor(-0.3, device=device)
    assert float(train.kl_loss(kl, 0, 1, 0.0, 1.0)) == 0.0 and float(train.kl_loss(kl + 1, 5, 10, 0.0, 2.0)) == pytest.approx(0.7, rel=1e-6)


def test_training_losses_match_the_reference_task_cpu():
    _check_losses("cpu", 1e-6)


@pytest.mark.gpu
def test_training_losses_match_the_reference_task_gpu():
    _check_losses("cuda", 2e-6)


def test_get_note2dur_bit_exact():
    from visinger_amd.align import get_note2dur
    cases = json.load(open(os.path.join(GOLDEN, "note2dur.json")))
    assert set(cases) == {"a", "b", "c"}
    for name, c in cases.items():
        rows = copy.deepcopy(c["midi_info"])
        mel2phone, mel2note, duration, ph_list, merged = get_note2dur(rows, c["hop_size"], c["sample_rate"], min_sil_duration=c["min_sil_duration"])
        assert mel2phone == c["mel2phone"] and mel2note == c["mel2note"] and duration == c["duration"], name
        assert ph_list == c["ph_list"] and merged == c["midi_info_out"], name
        assert sum(duration) == len(mel2phone) and all(isinstance(v, int) for v in duration)
    assert len(cases["b"]["midi_info_out"]) < len(cases["b"]["midi_info"])       # the consecutive rests were merged
    with pytest.raises(AssertionError):      # a gap that min_sil_duration does not close leaves frames without a note: the reference asserts
        rows = copy.deepcopy(cases["a"]["midi_info"])
        rows[3][4] += 0.2
        rows[3][5] += 0.2
        get_note2dur(rows[:5], 300, 24000)


def test_save_wav_writes_the_reference_bytes(tmp_path):
    from visinger_amd.synth import save_wav, to_int16
    z = np.load(os.path.join(GOLDEN, "save_wav.npz"))
    for key, wav, norm in (("f32", z["wav32"], False), ("f32_norm", z["wav32"], True), ("f64_norm", z["wav64"], True)):
        assert np.array_equal(to_int16(wav, norm=norm), z["pcm_" + key]), key
        path = str(tmp_path / (key + ".wav"))
        save_wav(wav, path, 22050, norm=norm)
        assert open(path, "rb").read() == bytes(z["bytes_" + key]), key
