"""Host-side mirror of the reference's model glue ``models/visinger.py:18-135`` (class VISinger) built on the
MI355X-native modules.  Same constructor signature, same sub-module names (so ``state_dict()`` keys equal the
reference's, tests/golden/visinger_state_dict_manifest.json), same ``forward`` arguments and return dict.

The reference's glue cannot travel to the GPU box, and as shipped it raises whenever ``use_pitch_embed`` is true
(``FramePriorNetwork.forward`` transposes the already-[B,1,T] pitch condition).  This mirror keeps
every module call identical and only hands the pitch condition to ``frame_prior`` as [B, T, 1] so that the
reference's transpose restores the intended [B, 1, T]; with ``use_pitch_embed=False`` it is call-for-call the
reference (and is parity-tested against its golden output).

RNG: ``noise`` / ``noise_q`` / ``u_slice`` may be injected for reproducible parity (CPU and GPU generators differ);
when omitted they are drawn exactly where the reference draws them.
"""
from copy import deepcopy

import torch
import torch.nn as nn

from ..modules.commons.utils import Embedding, rand_slice_segments, slice_segments
from ..modules.discriminator import DiscriminatorP, DiscriminatorS
from ..modules.rel_transformer import SinusoidalPositionalEmbedding
from ..modules.visinger.decoder import Generator
from ..modules.visinger.encoder import TextEncoder, PosteriorEncoder, FramePriorNetwork
from ..modules.visinger.flow import ResidualCouplingBlock
from ..modules.visinger.predictor import PitchPredictor, PhonemePredictor

DEFAULT_MAX_TARGET_POSITIONS = 2000


class VISinger(nn.Module):
    """models/visinger.py:18-135"""

    def __init__(self, ph_dict_size, pitch_size, dur_size, hparams, out_dims=None):
        super().__init__()
        self.hparams = deepcopy(hparams)
        self.enc_layers = hparams["enc_layers"]
        self.dec_blocks = hparams["dec_blocks"]
        self.hidden_size = hparams["hidden_size"]
        self.use_pos_embed = hparams["use_pos_embed"]
        self.segment_size = hparams["segment_size"]
        self.out_dims = hparams["num_mel_bins"] if out_dims is None else out_dims
        if hparams["use_spk_id"]:
            self.spk_id_proj = Embedding(hparams["num_spk"], hparams["gin_channels"])
        if hparams['use_spk_embed']:
            self.spk_embed_proj = nn.Linear(256, hparams["gin_channels"], bias=True)
        self.text_encoder = TextEncoder(ph_dict_size, pitch_size, dur_size, self.hidden_size,
                                        hparams["ffn_filter_channels"], hparams["num_heads"], self.enc_layers,
                                        hparams["ffn_kernel_size"], hparams["p_dropout"], True)
        self.embed_positions = SinusoidalPositionalEmbedding(self.hidden_size, 0, init_size=DEFAULT_MAX_TARGET_POSITIONS)
        if hparams["use_pitch_embed"]:
            self.pitch_predictor = PitchPredictor(self.hidden_size, hparams["ffn_filter_channels"], hparams["num_heads"],
                                                  n_layers=hparams["pitch_predictor_layers"],
                                                  kernel_size=hparams['ffn_kernel_size'], p_dropout=hparams["p_dropout"],
                                                  gin_channels=hparams["gin_channels"], out_dim=2)
        if hparams["use_phoneme_pred"]:
            self.phoneme_predictor = PhonemePredictor(ph_dict_size, self.hidden_size, hparams["ffn_filter_channels"],
                                                      hparams["num_heads"], n_layers=hparams["phoneme_predictor_layers"],
                                                      kernel_size=hparams["ffn_kernel_size"], p_dropout=hparams["p_dropout"])
        self.frame_prior = FramePriorNetwork(self.hidden_size, hparams["ffn_filter_channels"], hparams["num_heads"],
                                             hparams["frame_prior_layers"], hparams["ffn_kernel_size"],
                                             p_dropout=hparams["p_dropout"], gin_channels=1)
        self.posterior_encoder = PosteriorEncoder(hparams["num_linear_bins"], self.hidden_size, self.hidden_size, 5, 1, 16,
                                                  gin_channels=hparams["gin_channels"])
        self.flow = ResidualCouplingBlock(self.hidden_size, self.hidden_size, 5, 1, 4, gin_channels=hparams["gin_channels"])
        self.decoder = Generator(self.hidden_size, hparams["dec_blocks"], hparams["dec_kernel_size"],
                                 hparams["dec_dilation_sizes"], hparams["upsample_rates"],
                                 hparams["initial_upsample_channels"], hparams["upsample_kernel_sizes"],
                                 gin_channels=hparams["gin_channels"])

    def forward(self, text_tokens, pitch_tokens, dur_tokens, mel2ph, spk_embed=None, spk_id=None, f0=None, uv=None,
                mel=None, infer=False, noise=None, noise_q=None, u_slice=None, **kwargs):
        ret = {}
        tgt_nonpadding = (mel2ph > 0).float().unsqueeze(1)
        prior_inp = self.text_encoder(text_tokens, pitch_tokens, dur_tokens, mel2ph)  # [B, H, T]
        prior_inp = prior_inp * tgt_nonpadding
        if self.use_pos_embed:
            pos_in = prior_inp.transpose(1, 2)[..., 0]
            positions = self.embed_positions(prior_inp.shape[0], prior_inp.shape[2], pos_in)
            prior_inp = prior_inp + positions.transpose(1, 2)
        spk_emb = self.speaker_embedding(spk_embed, spk_id).transpose(1, 2)
        cond_pitch = None
        if self.hparams["use_pitch_embed"]:
            cond_pitch = self.forward_pitch(prior_inp, f0, uv, spk_emb, tgt_nonpadding, ret)  # [B, 1, T]
            cond_pitch = cond_pitch.transpose(1, 2)   # see module docstring: FramePriorNetwork transposes it back
        mu_p, logs_p = self.frame_prior(prior_inp, tgt_nonpadding, cond_pitch)
        if not infer:
            z_q, _, logs_q = self.posterior_encoder(mel.transpose(1, 2), tgt_nonpadding, g=spk_emb, noise=noise_q)
            if self.hparams["use_phoneme_pred"]:
                ret["ph_pred"] = self.phoneme_predictor(z_q, tgt_nonpadding) * tgt_nonpadding
            z_p = ret["z_p"] = self.flow(z_q, tgt_nonpadding, g=spk_emb) * tgt_nonpadding
            kl = (logs_p - logs_q - 0.5) + 0.5 * ((z_p - mu_p) ** 2) * torch.exp(-2. * logs_p)
            ret["kl"] = (kl * tgt_nonpadding).sum() / tgt_nonpadding.sum()
            if u_slice is None:
                z_slice, ret["ids_slice"] = rand_slice_segments(z_q, self.segment_size)
            else:  # injected uniform draws: the same fp32 product + truncation as modules/commons/utils.py:97-98
                ids = (u_slice.to(device=z_q.device) * (z_q.size(2) - self.segment_size + 1)).to(dtype=torch.long)
                z_slice, ret["ids_slice"] = slice_segments(z_q, ids, self.segment_size), ids
            ret["wav_out"] = self.decoder(z_slice, g=spk_emb).squeeze(1)
        else:
            if noise is None:
                noise = torch.randn_like(mu_p)
            z_p = (mu_p + noise * torch.exp(logs_p)) * tgt_nonpadding
            z_q = self.flow(z_p, tgt_nonpadding, g=spk_emb, reverse=True) * tgt_nonpadding
            ret["wav_out"] = self.decoder(z_q * tgt_nonpadding, g=spk_emb).squeeze(1)
        return ret

    def speaker_embedding(self, spk_embed=None, spk_id=None):
        speaker_embed = 0
        if self.hparams['use_spk_embed']:
            speaker_embed = speaker_embed + self.spk_embed_proj(spk_embed)[:, None, :]
        if self.hparams['use_spk_id']:
            speaker_embed = speaker_embed + self.spk_id_proj(spk_id)[:, None, :]
        return speaker_embed

    def forward_pitch(self, pitch_inp, f0, uv, spk_emb, tgt_nonpadding, ret):
        if self.hparams['predictor_grad'] != 1:
            pitch_inp = pitch_inp.detach() + self.hparams['predictor_grad'] * (pitch_inp - pitch_inp.detach())
        ret['f0_pred'] = pitch_pred = self.pitch_predictor(pitch_inp, tgt_nonpadding, spk_emb)
        if f0 is None:
            f0 = pitch_pred[:, :, 0]
            v = (pitch_pred[:, :, 1] <= 0)
        else:
            v = (uv == 0)
        f0 = (f0 * v).unsqueeze(1) * tgt_nonpadding
        return f0


class MultiPeriodDiscriminator(nn.Module):
    """models/visinger.py:138-158: one scale discriminator + period discriminators (2, 3, 5, 7, 11) applied to the real
    and the generated waveform."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        discs = [DiscriminatorS(use_spectral_norm=use_spectral_norm)]
        discs += [DiscriminatorP(p, use_spectral_norm=use_spectral_norm) for p in (2, 3, 5, 7, 11)]
        self.discriminators = nn.ModuleList(discs)

    def forward(self, y, y_hat):
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for d in self.discriminators:
            y_d_r, fmap_r = d(y)
            y_d_g, fmap_g = d(y_hat)
            y_d_rs.append(y_d_r)
            y_d_gs.append(y_d_g)
            fmap_rs.append(fmap_r)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


# Hyper-parameters of config/models/visinger.yaml:8-45 (+ datasets/svs/csd/preprocess.yaml), as a plain dict.
REFERENCE_HPARAMS = dict(
    enc_layers=6, dec_blocks="1", hidden_size=192, use_pos_embed=True, segment_size=32, num_mel_bins=128,
    use_spk_id=True, use_spk_embed=False, num_spk=1, gin_channels=256, ffn_filter_channels=768, num_heads=2,
    ffn_kernel_size=9, p_dropout=0.1, use_pitch_embed=True, pitch_predictor_layers=6, use_phoneme_pred=True,
    phoneme_predictor_layers=2, frame_prior_layers=4, num_linear_bins=1025, dec_kernel_size=[3, 7, 11],
    dec_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[5, 5, 3, 2, 2], initial_upsample_channels=512,
    upsample_kernel_sizes=[11, 11, 7, 4, 4], predictor_grad=1.0)


def hop256_hparams(**over):
    """The BASELINE.json benchmark variant: hop 256 / 22.05 kHz -> upsample [8,8,2,2], kernels [16,16,4,4]."""
    hp = dict(REFERENCE_HPARAMS, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4])
    hp.update(over)
    return hp
