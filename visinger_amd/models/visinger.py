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
    """Drop-in for the reference model class (models/visinger.py:18-135).  The sub-modules are registered under the
    reference's attribute names (text_encoder, embed_positions, pitch_predictor, phoneme_predictor, frame_prior,
    posterior_encoder, flow, decoder, spk_id_proj, spk_embed_proj), which fixes the state-dict layout; `forward` keeps the
    reference's arguments and result keys and is organised as prior -> (posterior | sampling) -> decode."""

    def __init__(self, ph_dict_size, pitch_size, dur_size, hparams, out_dims=None):
        super().__init__()
        hp = self.hparams = deepcopy(hparams)
        width = self.hidden_size = hp["hidden_size"]
        self.enc_layers, self.dec_blocks = hp["enc_layers"], hp["dec_blocks"]
        self.use_pos_embed, self.segment_size = hp["use_pos_embed"], hp["segment_size"]
        self.out_dims = hp["num_mel_bins"] if out_dims is None else out_dims
        gin = hp["gin_channels"]
        # transformer settings shared by the text encoder, the predictors and the frame prior
        tf = dict(filter_channels=hp["ffn_filter_channels"], n_heads=hp["num_heads"], kernel_size=hp["ffn_kernel_size"],
                  p_dropout=hp["p_dropout"])

        if hp["use_spk_id"]:
            self.spk_id_proj = Embedding(hp["num_spk"], gin)
        if hp["use_spk_embed"]:
            self.spk_embed_proj = nn.Linear(256, gin, bias=True)
        self.text_encoder = TextEncoder(ph_dict_size, pitch_size, dur_size, width, tf["filter_channels"], tf["n_heads"],
                                        self.enc_layers, tf["kernel_size"], tf["p_dropout"], True)
        self.embed_positions = SinusoidalPositionalEmbedding(width, 0, init_size=DEFAULT_MAX_TARGET_POSITIONS)
        if hp["use_pitch_embed"]:
            self.pitch_predictor = PitchPredictor(width, n_layers=hp["pitch_predictor_layers"], gin_channels=gin, out_dim=2, **tf)
        if hp["use_phoneme_pred"]:
            self.phoneme_predictor = PhonemePredictor(ph_dict_size, width, n_layers=hp["phoneme_predictor_layers"], **tf)
        self.frame_prior = FramePriorNetwork(width, tf["filter_channels"], tf["n_heads"], hp["frame_prior_layers"],
                                             tf["kernel_size"], p_dropout=tf["p_dropout"], gin_channels=1)
        # posterior WaveNet: kernel 5, dilation rate 1, 16 layers; flow: 4 couplings of 4 WaveNet layers (visinger.py:62-66)
        self.posterior_encoder = PosteriorEncoder(hp["num_linear_bins"], width, width, 5, 1, 16, gin_channels=gin)
        self.flow = ResidualCouplingBlock(width, width, 5, 1, 4, gin_channels=gin)
        self.decoder = Generator(width, hp["dec_blocks"], hp["dec_kernel_size"], hp["dec_dilation_sizes"], hp["upsample_rates"],
                                 hp["initial_upsample_channels"], hp["upsample_kernel_sizes"], gin_channels=gin)

    # ---- pieces of forward ----------------------------------------------------------------------------------------------
    def _prior(self, text_tokens, pitch_tokens, dur_tokens, mel2ph, spk_embed, spk_id, f0, uv, ret):
        """frame mask, speaker condition and the prior's (mu_p, logs_p): visinger.py:73-90"""
        frame_mask = (mel2ph > 0).float().unsqueeze(1)                                     # [B, 1, T]
        h = self.text_encoder(text_tokens, pitch_tokens, dur_tokens, mel2ph) * frame_mask  # [B, H, T]
        if self.use_pos_embed:
            pos = self.embed_positions(h.shape[0], h.shape[2], h.transpose(1, 2)[..., 0])
            h = h + pos.transpose(1, 2)
        spk = self.speaker_embedding(spk_embed, spk_id).transpose(1, 2)
        cond = None
        if self.hparams["use_pitch_embed"]:
            # [B, 1, T] -> [B, T, 1]: FramePriorNetwork transposes its condition back (see the module docstring)
            cond = self.forward_pitch(h, f0, uv, spk, frame_mask, ret).transpose(1, 2)
        mu_p, logs_p = self.frame_prior(h, frame_mask, cond)
        return frame_mask, spk, mu_p, logs_p

    def _posterior_branch(self, mel, frame_mask, spk, mu_p, logs_p, noise_q, u_slice, ret):
        """training side: posterior sample, phoneme CTC head, flow forward, KL, random segment decode (visinger.py:91-104)"""
        z_q, _, logs_q = self.posterior_encoder(mel.transpose(1, 2), frame_mask, g=spk, noise=noise_q)
        if self.hparams["use_phoneme_pred"]:
            ret["ph_pred"] = self.phoneme_predictor(z_q, frame_mask) * frame_mask
        z_p = ret["z_p"] = self.flow(z_q, frame_mask, g=spk) * frame_mask
        kl = (logs_p - logs_q - 0.5) + 0.5 * (z_p - mu_p) ** 2 * torch.exp(-2.0 * logs_p)
        ret["kl"] = (kl * frame_mask).sum() / frame_mask.sum()
        if u_slice is None:
            z_seg, ret["ids_slice"] = rand_slice_segments(z_q, self.segment_size)
        else:   # injected uniform draws: the same fp32 product + truncation as modules/commons/utils.py:97-98
            ids = (u_slice.to(device=z_q.device) * (z_q.size(2) - self.segment_size + 1)).to(dtype=torch.long)
            z_seg, ret["ids_slice"] = slice_segments(z_q, ids, self.segment_size), ids
        ret["wav_out"] = self.decoder(z_seg, g=spk).squeeze(1)

    def _sample_and_decode(self, frame_mask, spk, mu_p, logs_p, noise, ret, mask_decoder=False):
        """synthesis side: reparameterised prior sample, flow inverse, full-length decode (visinger.py:105-110).
        mask_decoder (not in the reference, which synthesises one utterance at a time): decode a padded batch so that every
        item's samples equal its standalone synthesis (Generator.forward x_mask)."""
        eps = torch.randn_like(mu_p) if noise is None else noise
        z_p = (mu_p + eps * torch.exp(logs_p)) * frame_mask
        z_q = self.flow(z_p, frame_mask, g=spk, reverse=True) * frame_mask
        if mask_decoder:
            ret["wav_out"] = self.decoder(z_q * frame_mask, g=spk, x_mask=frame_mask).squeeze(1)
        else:
            ret["wav_out"] = self.decoder(z_q * frame_mask, g=spk).squeeze(1)

    def forward(self, text_tokens, pitch_tokens, dur_tokens, mel2ph, spk_embed=None, spk_id=None, f0=None, uv=None,
                mel=None, infer=False, noise=None, noise_q=None, u_slice=None, mask_decoder=False, **kwargs):
        ret = {}
        frame_mask, spk, mu_p, logs_p = self._prior(text_tokens, pitch_tokens, dur_tokens, mel2ph, spk_embed, spk_id, f0, uv, ret)
        if infer:
            self._sample_and_decode(frame_mask, spk, mu_p, logs_p, noise, ret, mask_decoder=mask_decoder)
        else:
            self._posterior_branch(mel, frame_mask, spk, mu_p, logs_p, noise_q, u_slice, ret)
        return ret

    def speaker_embedding(self, spk_embed=None, spk_id=None):
        """sum of the enabled speaker conditions, [B, 1, gin] (visinger.py:114-120)"""
        parts = []
        if self.hparams["use_spk_embed"]:
            parts.append(self.spk_embed_proj(spk_embed)[:, None, :])
        if self.hparams["use_spk_id"]:
            parts.append(self.spk_id_proj(spk_id)[:, None, :])
        return sum(parts) if parts else 0

    def forward_pitch(self, pitch_inp, f0, uv, spk_emb, tgt_nonpadding, ret):
        """pitch predictor + the f0 condition of the frame prior (visinger.py:122-135): predicted f0 / voicing when no ground
        truth is given; the gradient into the prior input is scaled by `predictor_grad`."""
        scale = self.hparams["predictor_grad"]
        if scale != 1:
            stopped = pitch_inp.detach()
            pitch_inp = stopped + scale * (pitch_inp - stopped)
        pred = ret["f0_pred"] = self.pitch_predictor(pitch_inp, tgt_nonpadding, spk_emb)      # [B, T, 2]
        if f0 is None:
            f0, voiced = pred[:, :, 0], pred[:, :, 1] <= 0
        else:
            voiced = uv == 0
        return (f0 * voiced).unsqueeze(1) * tgt_nonpadding


class MultiPeriodDiscriminator(nn.Module):
    """models/visinger.py:138-158: the scale discriminator followed by the period discriminators (2, 3, 5, 7, 11), each applied
    to the real and to the generated waveform.  Returns (logits_real, logits_generated, fmaps_real, fmaps_generated), one entry
    per discriminator, in that order."""

    PERIODS = (2, 3, 5, 7, 11)

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        self.discriminators = nn.ModuleList(
            [DiscriminatorS(use_spectral_norm=use_spectral_norm)] +
            [DiscriminatorP(period, use_spectral_norm=use_spectral_norm) for period in self.PERIODS])

    def forward(self, y, y_hat):
        if y.shape == y_hat.shape and not y_hat.requires_grad and not y.requires_grad:
            # the discriminator pass (both inputs detached, tasks/visinger.py:80-86): real and generated waveforms as ONE batch of 2B --
            # no op of either discriminator mixes batch items, so every output is bit-identical to the two separate passes, at half the
            # launches and half the weight re-packs of a step that is bound by both
            B = y.shape[0]
            both = [disc(torch.cat([y, y_hat], 0)) for disc in self.discriminators]
            return ([o[:B] for o, _ in both], [o[B:] for o, _ in both],
                    [[f[:B] for f in fm] for _, fm in both], [[f[B:] for f in fm] for _, fm in both])
        real, fake = zip(*((disc(y), disc(y_hat)) for disc in self.discriminators))      # ((logit, fmap), ...) per input
        return [r[0] for r in real], [f[0] for f in fake], [r[1] for r in real], [f[1] for f in fake]


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
