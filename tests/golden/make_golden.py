#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only).

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing from the reference is copied: the reference's modules are instantiated at small sizes with
seeded non-trivial weights, run on seeded inputs, and the (weights, inputs, outputs) triples are stored
as .npz data files.  These pin oracle/ (tests/test_oracle_golden.py) and, through it, the HIP path.

Reference symbols exercised (file:line under /root/reference):
  modules/visinger/encoder.py:130-213   WaveNet + gate
  modules/visinger/encoder.py:76-101    PosteriorEncoder
  modules/visinger/encoder.py:58-73     FramePriorNetwork
  modules/visinger/encoder.py:14-55     TextEncoder
  modules/visinger/flow.py:15-95        ResidualCouplingBlock / Layer / Flip
  modules/visinger/decoder.py:13-137    Generator / ResBlock1 / ResBlock2
  modules/visinger/predictor.py:7-35    PitchPredictor / PhonemePredictor
  modules/rel_transformer.py:24-345     LayerNorm, SinusoidalPositionalEmbedding, MultiHeadAttention,
                                        RelativeEncoder, FFN
  modules/commons/utils.py:86-110       slice_segments, rand_slice_segments, get_padding
  models/commons/align_ops.py:22-26     expand_states
  models/visinger.py:18-112             VISinger (state-dict manifest + tiny infer forward)
  tasks/visinger.py:127-169, tasks/base.py:227-238   the training losses (called unbound)
  utils/audio/align.py:58-104, utils/audio/io.py:8-12   get_note2dur, save_wav
"""
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REF = os.environ.get("VISINGER_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
import importlib.util  # noqa: E402
# torchaudio is what utils/audio/mel_processing.py wraps; it is absent from this image and from /root/reference, so SURVEY.md 8f-2 stays "parity unpinned".
# The day it is importable in the build container, gen_mel_processing() below writes the pin (tests/test_audio_gpu.py picks the fixture up by itself).
HAVE_TORCHAUDIO = importlib.util.find_spec("torchaudio") is not None
for name in ("librosa", "librosa.filters", "pyloudnorm", "webrtcvad", "skimage", "skimage.transform",
             "parselmouth", "pyworld") + (() if HAVE_TORCHAUDIO else ("torchaudio",)):
    sys.modules.setdefault(name, MagicMock())
sys.path.insert(0, REF)

from modules.visinger.encoder import WaveNet, PosteriorEncoder, FramePriorNetwork, TextEncoder  # noqa: E402
from modules.visinger.flow import ResidualCouplingBlock, ResidualCouplingLayer, Flip  # noqa: E402
from modules.visinger.decoder import Generator, ResBlock1, ResBlock2  # noqa: E402
from modules.visinger.predictor import PitchPredictor, PhonemePredictor  # noqa: E402
from modules.rel_transformer import (LayerNorm, SinusoidalPositionalEmbedding, MultiHeadAttention,  # noqa: E402
                                     RelativeEncoder, FFN)
from modules.commons.utils import slice_segments, rand_slice_segments, get_padding  # noqa: E402
from models.commons.align_ops import expand_states  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(4)


def randomize(module, seed, scale=1.0):
    """Seeded non-trivial weights: every parameter ~ N(0, s) with s set per parameter kind so that
    activations stay O(1).  (Zero-initialised `post` convs would make the flow an identity.)"""
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        if name.endswith("weight_g"):
            p.copy_(0.5 + torch.rand(p.shape, generator=g))
        elif name.endswith("gamma"):
            p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
        elif name.endswith("beta") or name.endswith("bias"):
            p.copy_(0.1 * torch.randn(p.shape, generator=g))
        elif "emb_rel" in name:
            p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5))
        else:
            fan = max(1, int(np.prod(p.shape[1:])))
            p.copy_(scale * torch.randn(p.shape, generator=g) / np.sqrt(fan))
    return module


def sd_np(module, prefix="w."):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def ragged_mask(B, T, lens):
    m = torch.zeros(B, 1, T)
    for b, l in enumerate(lens):
        m[b, 0, :l] = 1.0
    return m


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


class CaptureRandn:
    """Record what torch.randn_like / torch.rand return while the reference runs, so the sampled noise can
    be injected into the oracle and the HIP path (CPU and GPU generators differ)."""
    def __enter__(self):
        self.draws, self.uniform = [], []
        self._randn_like, self._rand = torch.randn_like, torch.rand

        def randn_like(*a, **k):
            out = self._randn_like(*a, **k)
            self.draws.append(out.clone().contiguous())
            return out

        def rand(*a, **k):
            out = self._rand(*a, **k)
            self.uniform.append(out.clone())
            return out
        torch.randn_like, torch.rand = randn_like, rand
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.rand = self._randn_like, self._rand
        return False


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ----------------------------------------------------------------------------------------------
def gen_wavenet():
    for tag, (H, k, dr, L, gin, B, T, lens) in {
        "wavenet_g": (16, 5, 1, 3, 8, 2, 37, [37, 29]),
        "wavenet_nog_dil2": (8, 3, 2, 4, 0, 2, 41, [33, 41]),
        "wavenet_1layer": (8, 5, 1, 1, 4, 1, 19, [19]),
    }.items():
        m = randomize(WaveNet(H, k, dr, L, gin_channels=gin).eval(), 11)
        x = rnd(1, B, H, T)
        mask = ragged_mask(B, T, lens)
        g = rnd(2, B, gin, 1) if gin else None
        y = m(x, mask, g=g)
        arrs = dict(sd_np(m), x=x, mask=mask, y=y,
                    cfg=np.array([H, k, dr, L, gin], dtype=np.int64))
        if g is not None:
            arrs["g"] = g
        save(tag, **arrs)


def gen_posterior():
    cin, cout, H, k, dr, L, gin, B, T = 21, 12, 16, 5, 1, 3, 8, 2, 33
    m = randomize(PosteriorEncoder(cin, cout, H, k, dr, L, gin).eval(), 12)
    x = rnd(3, B, cin, T).abs()
    mask = ragged_mask(B, T, [33, 20])
    g = rnd(4, B, gin, 1)
    torch.manual_seed(777)
    with CaptureRandn() as cap:   # the only RNG draw inside forward (encoder.py:97)
        z, mu, logs = m(x, mask, g=g)
    noise = cap.draws[0]
    save("posterior", **sd_np(m), x=x, mask=mask, g=g, noise=noise, z=z, mu=mu, logs=logs,
         cfg=np.array([cin, cout, H, k, dr, L, gin], dtype=np.int64))


def gen_flow():
    C, H, k, dr, L, gin, B, T = 12, 16, 5, 1, 2, 8, 2, 35
    mask = ragged_mask(B, T, [35, 22])
    x = rnd(5, B, C, T)
    g = rnd(6, B, gin, 1)
    for mean_only in (False, True):
        m = randomize(ResidualCouplingLayer(C, H, k, dr, L, gin_channels=gin, mean_only=mean_only).eval(), 13,
                      scale=0.5)
        y, logdet = m(x, mask, g=g, reverse=False)
        xr = m(y, mask, g=g, reverse=True)     # inverse of the forward output
        yi = m(x, mask, g=g, reverse=True)     # inverse applied to x itself
        save(f"coupling_meanonly{int(mean_only)}", **sd_np(m), x=x, mask=mask, g=g, y=y, logdet=logdet,
             x_roundtrip=xr, y_inv=yi, cfg=np.array([C, H, k, dr, L, gin, int(mean_only)], dtype=np.int64))
    xf, ld = Flip()(x, mask, reverse=False)
    save("flip", x=x, y=xf, logdet=ld, y_rev=Flip()(x, reverse=True))
    blk = randomize(ResidualCouplingBlock(C, H, k, dr, L, n_flows=4, gin_channels=gin).eval(), 14, scale=0.5)
    y = blk(x, mask, g=g, reverse=False)
    yi = blk(x, mask, g=g, reverse=True)
    xr = blk(y, mask, g=g, reverse=True)
    save("flow_block", **sd_np(blk), x=x, mask=mask, g=g, y=y, y_inv=yi, x_roundtrip=xr,
         cfg=np.array([C, H, k, dr, L, 4, gin], dtype=np.int64))
    # no-conditioning variant
    blk = randomize(ResidualCouplingBlock(C, H, 3, 2, 3, n_flows=2, gin_channels=0).eval(), 15, scale=0.5)
    y = blk(x, mask, reverse=False)
    yi = blk(x, mask, reverse=True)
    save("flow_block_nog", **sd_np(blk), x=x, mask=mask, y=y, y_inv=yi,
         cfg=np.array([C, H, 3, 2, 3, 2, 0], dtype=np.int64))


def gen_generator():
    # (tag, init_ch, resblock, rb_kernels, rb_dils, rates, up_init, up_kernels, gin, B, T)
    cases = [
        ("generator_hop256_like", 12, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 2, 2], 64, [16, 16, 4, 4], 8, 2, 9),
        ("generator_hop300_like", 12, "1", [3, 7], [[1, 3, 5]] * 2, [5, 3, 2], 32, [11, 7, 4], 8, 2, 11),
        ("generator_rb2_nog", 10, "2", [3, 5], [[1, 3]] * 2, [4, 2], 16, [8, 4], 0, 1, 13),
    ]
    for tag, ic, rb, rk, rd, rates, ui, uk, gin, B, T in cases:
        m = Generator(ic, rb, rk, rd, rates, ui, uk, gin_channels=gin).eval()
        randomize(m, 21, scale=1.0)
        x = rnd(7, B, ic, T)
        g = rnd(8, B, gin, 1) if gin else None
        y = m(x, g=g)
        arrs = dict(sd_np(m), x=x, y=y, rb=np.array(int(rb)), rk=np.array(rk), rd=np.array(rd),
                    rates=np.array(rates), uk=np.array(uk), cfg=np.array([ic, ui, gin], dtype=np.int64))
        if g is not None:
            arrs["g"] = g
        save(tag, **arrs)
    # stand-alone resblocks (with and without mask; decoder.py:91-104,124-133)
    for tag, cls, C, k, d in (("resblock1", ResBlock1, 8, 7, (1, 3, 5)), ("resblock2", ResBlock2, 8, 5, (1, 3))):
        m = randomize(cls(C, k, d).eval(), 22)
        x = rnd(9, 2, C, 45)
        mask = ragged_mask(2, 45, [45, 30])
        save(tag, **sd_np(m), x=x, mask=mask, y=m(x), y_masked=m(x, mask),
             cfg=np.array([C, k] + list(d), dtype=np.int64))


def gen_transformer():
    C, nh, ws, B, T = 16, 2, 4, 2, 23
    mask = ragged_mask(B, T, [23, 14])
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    x = rnd(10, B, C, T)
    ln = randomize(LayerNorm(C).eval(), 31)
    save("layernorm", **sd_np(ln), x=x, y=ln(x))
    mha = randomize(MultiHeadAttention(C, C, nh, window_size=ws).eval(), 32)
    y = mha(x, x, attn_mask)
    save("mha_rel", **sd_np(mha), x=x, attn_mask=attn_mask, mask=mask, y=y, p_attn=mha.attn,
         cfg=np.array([C, nh, ws], dtype=np.int64))
    # short sequence: T <= window (rel_transformer.py:199-212 slice path)
    xs = rnd(11, 1, C, 3)
    ms = torch.ones(1, 1, 3)
    ys = mha(xs, xs, ms.unsqueeze(2) * ms.unsqueeze(-1))
    save("mha_rel_short", **sd_np(mha), x=xs, mask=ms, y=ys, p_attn=mha.attn, cfg=np.array([C, nh, ws]))
    # fully masked rows -> uniform attention, not NaN (rel_transformer.py:167)
    ffn = randomize(FFN(C, C, 24, 9).eval(), 33)
    save("ffn", **sd_np(ffn), x=x, mask=mask, y=ffn(x, mask), cfg=np.array([C, C, 24, 9]))
    for tag, gin, ks, nl in (("rel_encoder_g", 1, 9, 2), ("rel_encoder_nog", None, 3, 3), ("rel_encoder_spk", 8, 5, 1)):
        enc = randomize(RelativeEncoder(C, 24, nh, nl, kernel_size=ks, gin_channels=gin).eval(), 34)
        arrs = dict(sd_np(enc), x=x, mask=mask, cfg=np.array([C, 24, nh, nl, ks, -1 if gin is None else gin]))
        if gin is None:
            arrs["y"] = enc(x, mask)
        else:
            g = rnd(12, B, gin, T if gin == 1 else 1)
            arrs["g"] = g
            arrs["y"] = enc(x, mask, g)
        save(tag, **arrs)


def gen_transformer_options():
    """constructor options of modules/rel_transformer.py that VISinger itself never sets: proximal_bias and block_length of
    MultiHeadAttention (rel_transformer.py:163-170), pre_ln of RelativeEncoder (:284, 301-317), activation='gelu' of FFN (:338-341)"""
    C, nh, ws, B, T = 16, 2, 4, 2, 23
    mask = ragged_mask(B, T, [23, 14])
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    x = rnd(40, B, C, T)
    for tag, kw in (("mha_proximal", dict(proximal_bias=True)), ("mha_block", dict(block_length=5)), ("mha_proximal_block", dict(proximal_bias=True, block_length=3))):
        mha = randomize(MultiHeadAttention(C, C, nh, window_size=ws, **kw).eval(), 41)
        save(tag, **sd_np(mha), x=x, attn_mask=attn_mask, mask=mask, y=mha(x, x, attn_mask),
             cfg=np.array([C, nh, ws, int(bool(kw.get("proximal_bias"))), kw.get("block_length") or -1], dtype=np.int64))
    ffn = randomize(FFN(C, C, 24, 9, activation="gelu").eval(), 42)
    save("ffn_gelu", **sd_np(ffn), x=x, mask=mask, y=ffn(x, mask), cfg=np.array([C, C, 24, 9]))
    for tag, gin in (("rel_encoder_preln", None), ("rel_encoder_preln_g", 8)):
        enc = randomize(RelativeEncoder(C, 24, nh, 2, kernel_size=5, pre_ln=True, gin_channels=gin).eval(), 43)
        arrs = dict(sd_np(enc), x=x, mask=mask, cfg=np.array([C, 24, nh, 2, 5, -1 if gin is None else gin]))
        if gin is None:
            arrs["y"] = enc(x, mask)
        else:
            arrs["g"] = rnd(44, B, gin, 1)
            arrs["y"] = enc(x, mask, arrs["g"])
        save(tag, **arrs)


def gen_wrappers():
    C, F_, nh, B, T = 16, 24, 2, 2, 21
    mask = ragged_mask(B, T, [21, 13])
    x = rnd(13, B, C, T) * mask
    fp = randomize(FramePriorNetwork(C, F_, nh, 2, 5, gin_channels=1, p_dropout=0.0).eval(), 41)
    mu, logs = fp(x, mask, None)
    # NB: with g given the reference transposes it (encoder.py:68-69) -> g must arrive as [B, T, 1]
    gT = rnd(14, B, T, 1)
    mu_g, logs_g = fp(x, mask, gT)
    save("frame_prior", **sd_np(fp), x=x, mask=mask, mu=mu, logs=logs, g_BT1=gT, mu_g=mu_g, logs_g=logs_g,
         cfg=np.array([C, F_, nh, 2, 5, 1]))
    pp = randomize(PitchPredictor(C, F_, nh, 2, 5, 0.0, gin_channels=8, out_dim=2).eval(), 42)
    spk = rnd(15, B, 8, 1)
    save("pitch_predictor", **sd_np(pp), x=x, mask=mask, spk=spk, y=pp(x, mask, spk), cfg=np.array([C, F_, nh, 2, 5, 8, 2]))
    ph = randomize(PhonemePredictor(11, C, F_, nh, 1, 3, 0.0).eval(), 43)
    save("phoneme_predictor", **sd_np(ph), x=x, mask=mask, y=ph(x, mask), cfg=np.array([11, C, F_, nh, 1, 3]))
    # TextEncoder (use_pos_embed True as models/visinger.py:40 instantiates it)
    te = randomize(TextEncoder(13, 9, 7, C, F_, nh, 2, 3, 0.0, True).eval(), 44)
    Tph = 6
    gi = torch.Generator().manual_seed(45)
    text = torch.randint(1, 13, (B, Tph), generator=gi)
    pitch = torch.randint(1, 9, (B, Tph), generator=gi)
    dur = torch.randint(1, 7, (B, Tph), generator=gi)
    text[1, 4:] = 0
    pitch[1, 4:] = 0
    dur[1, 4:] = 0
    mel2ph = torch.zeros(B, T, dtype=torch.long)
    mel2ph[0] = torch.tensor([1] * 3 + [2] * 4 + [3] * 2 + [4] * 5 + [5] * 3 + [6] * 4)
    mel2ph[1, :13] = torch.tensor([1] * 4 + [2] * 3 + [3] * 2 + [4] * 4)
    y = te(text, pitch, dur, mel2ph)
    save("text_encoder", **sd_np(te), text=text, pitch=pitch, dur=dur, mel2ph=mel2ph, y=y,
         cfg=np.array([13, 9, 7, C, F_, nh, 2, 3]))


def gen_integer():
    gi = torch.Generator().manual_seed(51)
    B, Tph, T, H = 3, 7, 29, 5
    h = rnd(16, B, Tph, H)
    mel2ph = torch.randint(0, Tph + 1, (B, T), generator=gi)
    mel2ph[2, 20:] = 0
    save("expand_states", h=h, mel2ph=mel2ph, y=expand_states(h, mel2ph))
    # frames per token (utils/audio/align.py:105-129; caller tasks/base.py:342)
    from utils.audio.align import mel2token_to_dur
    save("mel2token_to_dur", mel2ph=mel2ph, dur=mel2token_to_dur(mel2ph, Tph), dur_clamped=mel2token_to_dur(mel2ph, Tph, max_dur=4),
         dur_1d=mel2token_to_dur(mel2ph[0], Tph), dur_auto=mel2token_to_dur(mel2ph[:, :11]))
    # make_positions / sinusoidal embedding (rel_transformer.py:59-100)
    inp = rnd(17, B, T)
    inp[0, 5] = 0.0
    inp[1, 10:] = 0.0
    inp[2, :3] = 0.0
    pos = SinusoidalPositionalEmbedding.make_positions(inp, 0)
    emb = SinusoidalPositionalEmbedding(12, 0, init_size=16)   # forces the auto-grow path (T=29 > 16)
    y = emb(B, T, inp)
    emb_odd = SinusoidalPositionalEmbedding.get_embedding(9, 7, 0)
    save("positions", x=inp, positions=pos, y=y, table=emb.weights, table_odd=emb_odd)
    # slice_segments / rand_slice_segments (modules/commons/utils.py:86-100)
    x = rnd(18, B, H, T)
    ids = torch.tensor([0, 11, 21])
    torch.manual_seed(1234)
    rs, rids = rand_slice_segments(x, 8)
    torch.manual_seed(1234)
    u = torch.rand([B])
    save("slice_segments", x=x, ids=ids, y=slice_segments(x, ids, 8), rand_u=u, rand_ids=rids, rand_y=rs,
         pads=np.array([get_padding(k, d) for k in (3, 5, 7, 11) for d in (1, 3, 5)]))


def gen_discriminators():
    """modules/discriminator.py:13-75 + MultiPeriodDiscriminator models/visinger.py:138-158 (a13).  The channel widths
    are hard-coded in the reference, so the fixture stores input/outputs and a SEED: the weights are re-created with
    `randomize` on the identical architecture by the test (they would be 40 MB)."""
    from modules.discriminator import DiscriminatorP, DiscriminatorS
    y = rnd(71, 2, 1, 250) * 0.3
    ds = randomize(DiscriminatorS().eval(), 72)
    out, fmap = ds(y)
    arrs = dict(y=y, s_logits=out, s_fmap_last=fmap[-1], s_fmap0=fmap[0], s_fmap3_sum=fmap[3].sum(), s_fmap3_abs=fmap[3].abs().sum())
    for p_ in (2, 3, 11):
        dp = randomize(DiscriminatorP(p_).eval(), 73 + p_)
        out, fmap = dp(y)
        arrs[f"p{p_}_logits"] = out
        arrs[f"p{p_}_fmap0"] = fmap[0]
        arrs[f"p{p_}_fmap_last"] = fmap[-1]
    from models.visinger import MultiPeriodDiscriminator
    mpd = MultiPeriodDiscriminator()
    with open(os.path.join(OUT, "mpd_state_dict_manifest.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in mpd.state_dict().items()}, f, indent=0, sort_keys=True)
    save("discriminators", **arrs)


def gen_model():
    """State-dict manifest of the full-size reference model + a tiny end-to-end infer forward
    (use_pitch_embed=False: the shipped default raises in FramePriorNetwork)."""
    from models.visinger import VISinger
    hp = dict(enc_layers=6, dec_blocks="1", hidden_size=192, use_pos_embed=True, segment_size=32, num_mel_bins=128,
              use_spk_id=True, use_spk_embed=False, num_spk=1, gin_channels=256, ffn_filter_channels=768,
              num_heads=2, ffn_kernel_size=9, p_dropout=0.1, use_pitch_embed=True, pitch_predictor_layers=6,
              use_phoneme_pred=True, phoneme_predictor_layers=2, frame_prior_layers=4, num_linear_bins=1025,
              dec_kernel_size=[3, 7, 11], dec_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[5, 5, 3, 2, 2],
              initial_upsample_channels=512, upsample_kernel_sizes=[11, 11, 7, 4, 4], predictor_grad=1.0)
    m = VISinger(64, 117, 131, hp)
    manifest = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "visinger_state_dict_manifest.json"), "w") as f:
        json.dump({"hparams": hp, "ph_dict_size": 64, "pitch_size": 117, "dur_size": 131,
                   "state_dict": manifest}, f, indent=0, sort_keys=True)
    print("manifest:", len(manifest), "tensors,", sum(int(np.prod(s)) for s in manifest.values()), "params")

    hp_t = dict(hp, enc_layers=2, hidden_size=16, segment_size=4, gin_channels=8, ffn_filter_channels=24,
                ffn_kernel_size=3, use_pitch_embed=False, use_phoneme_pred=True, phoneme_predictor_layers=1,
                frame_prior_layers=2, num_linear_bins=21, dec_kernel_size=[3, 5], dec_dilation_sizes=[[1, 3, 5]] * 2,
                upsample_rates=[4, 2], initial_upsample_channels=32, upsample_kernel_sizes=[8, 4], p_dropout=0.0)
    m = VISinger(13, 9, 7, hp_t).eval()
    randomize(m, 61, scale=0.7)
    B, T, Tph = 2, 21, 6
    gi = torch.Generator().manual_seed(62)
    text = torch.randint(1, 13, (B, Tph), generator=gi)
    pitch = torch.randint(1, 9, (B, Tph), generator=gi)
    dur = torch.randint(1, 7, (B, Tph), generator=gi)
    text[1, 4:] = 0
    pitch[1, 4:] = 0
    dur[1, 4:] = 0
    mel2ph = torch.zeros(B, T, dtype=torch.long)
    mel2ph[0] = torch.tensor([1] * 3 + [2] * 4 + [3] * 2 + [4] * 5 + [5] * 3 + [6] * 4)
    mel2ph[1, :13] = torch.tensor([1] * 4 + [2] * 3 + [3] * 2 + [4] * 4)
    spk_id = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(4321)
    with CaptureRandn() as cap:      # randn_like(mu_p) at models/visinger.py:107
        ret = m(text, pitch, dur, mel2ph, spk_id=spk_id, infer=True)
    noise = cap.draws[0]
    # training-side forward (posterior + flow fwd + segment decode)
    lin = rnd(63, B, T, 21).abs()
    torch.manual_seed(999)
    with CaptureRandn() as cap:      # encoder.py:97 (randn_like) then modules/commons/utils.py:98 (rand)
        ret_t = m(text, pitch, dur, mel2ph, spk_id=spk_id, mel=lin, infer=False)
    noise_q = cap.draws[0]
    u_slice = cap.uniform[0]
    save("visinger_tiny", **sd_np(m), text=text, pitch=pitch, dur=dur, mel2ph=mel2ph, spk_id=spk_id,
         noise=noise, wav_out=ret["wav_out"], lin=lin, noise_q=noise_q, u_slice=u_slice,
         t_wav_out=ret_t["wav_out"], t_z_p=ret_t["z_p"], t_kl=ret_t["kl"], t_ids_slice=ret_t["ids_slice"],
         t_ph_pred=ret_t["ph_pred"])
    with open(os.path.join(OUT, "visinger_tiny_hparams.json"), "w") as f:
        json.dump(hp_t, f, sort_keys=True)

    # ---- gradients of a scalar functional of the training forward, from the reference's own autograd --------------
    torch.set_grad_enabled(True)
    m.train()                                   # p_dropout = 0 in the tiny config: train == eval numerically
    m.zero_grad()
    torch.manual_seed(999)
    with CaptureRandn() as cap:
        out = m(text, pitch, dur, mel2ph, spk_id=spk_id, mel=lin, infer=False)
    c_w, c_p, c_z = rnd(64, *out["wav_out"].shape), rnd(65, *out["ph_pred"].shape), rnd(66, *out["z_p"].shape)
    loss = out["kl"] + 0.1 * (out["wav_out"] * c_w).sum() + 0.01 * (out["ph_pred"] * c_p).sum() + 0.01 * (out["z_p"] * c_z).sum()
    loss.backward()
    pick = ["decoder.conv_pre.weight", "decoder.ups.0.weight_v", "decoder.ups.1.weight_g",
            "decoder.resblocks.0.convs1.0.weight_g", "decoder.resblocks.3.convs2.2.weight_v", "decoder.conv_post.weight",
            "decoder.cond.weight", "flow.flows.0.post.weight", "flow.flows.2.enc.in_layers.1.weight_v",
            "flow.flows.6.pre.bias", "posterior_encoder.enc.res_skip_layers.0.bias", "posterior_encoder.pre.weight",
            "posterior_encoder.enc.cond_layer.weight_g", "frame_prior.encoder.attn_layers.0.emb_rel_k",
            "frame_prior.encoder.attn_layers.1.emb_rel_v", "frame_prior.encoder.norm_layers_1.0.gamma",
            "frame_prior.proj.weight", "text_encoder.ph_emb.weight", "text_encoder.linear.weight",
            "text_encoder.text_encoder.ffn_layers.0.conv_1.weight", "text_encoder.text_encoder.attn_layers.1.conv_q.weight",
            "phoneme_predictor.ph_proj.bias", "phoneme_predictor.phoneme_predictor.attn_layers.0.conv_o.weight",
            "spk_id_proj.weight"]
    named = dict(m.named_parameters())
    grads = {"g." + k: named[k].grad.detach().clone() for k in pick}
    save("visinger_tiny_grads", loss=loss.detach(), c_w=c_w, c_p=c_p, c_z=c_z, noise_q=cap.draws[0], u_slice=cap.uniform[0],
         **grads)
    torch.set_grad_enabled(False)


def gen_model_pitch():
    """The synthesis graph WITH the pitch predictor (use_pitch_embed=True, the shipped default: config/models/visinger.yaml:36) -- the
    graph bench.py times.  The reference's own call chain raises there (forward_pitch returns [B, 1, T],
    FramePriorNetwork.forward transposes it once more, encoder.py:68-69); the ONE change made here is at that call site, on the live
    object: the condition is handed over as [B, T, 1], so the reference's own transpose restores what its Conv1d(1, H, 1) expects.
    Every module forward and VISinger.forward / forward_pitch run unmodified."""
    from models.visinger import VISinger
    hp_t = dict(enc_layers=2, dec_blocks="1", hidden_size=16, use_pos_embed=True, segment_size=4, num_mel_bins=128,
                use_spk_id=True, use_spk_embed=False, num_spk=1, gin_channels=8, ffn_filter_channels=24, num_heads=2,
                ffn_kernel_size=3, p_dropout=0.0, use_pitch_embed=True, pitch_predictor_layers=2, use_phoneme_pred=True,
                phoneme_predictor_layers=1, frame_prior_layers=2, num_linear_bins=21, dec_kernel_size=[3, 5],
                dec_dilation_sizes=[[1, 3, 5]] * 2, upsample_rates=[4, 2], initial_upsample_channels=32,
                upsample_kernel_sizes=[8, 4], predictor_grad=1.0)
    m = VISinger(13, 9, 7, hp_t).eval()
    randomize(m, 71, scale=0.7)
    inner = m.frame_prior.forward
    m.frame_prior.forward = lambda x, x_mask, g=None: inner(x, x_mask, None if g is None else g.transpose(1, 2))
    B, T, Tph = 2, 23, 6
    gi = torch.Generator().manual_seed(72)
    text = torch.randint(1, 13, (B, Tph), generator=gi)
    pitch = torch.randint(1, 9, (B, Tph), generator=gi)
    dur = torch.randint(1, 7, (B, Tph), generator=gi)
    mel2ph = torch.zeros(B, T, dtype=torch.long)
    mel2ph[0] = torch.tensor([1] * 3 + [2] * 4 + [3] * 2 + [4] * 5 + [5] * 3 + [6] * 6)
    mel2ph[1, :17] = torch.tensor([1] * 4 + [2] * 3 + [3] * 2 + [4] * 4 + [5] * 2 + [6] * 2)
    spk_id = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(4322)
    with CaptureRandn() as cap:
        ret = m(text, pitch, dur, mel2ph, spk_id=spk_id, infer=True)
    margin = float(ret["f0_pred"][:, :, 1].abs().min())
    assert margin > 1e-3, margin          # no voicing decision of the fixture sits on the threshold
    save("visinger_tiny_pitch", **sd_np(m), text=text, pitch=pitch, dur=dur, mel2ph=mel2ph, spk_id=spk_id, noise=cap.draws[0],
         wav_out=ret["wav_out"], f0_pred=ret["f0_pred"])
    with open(os.path.join(OUT, "visinger_tiny_pitch_hparams.json"), "w") as f:
        json.dump(hp_t, f, sort_keys=True)



def _import_reference_task():
    """tasks/visinger.py imports the whole preprocessing / dataset stack at module level (and opens ./preprocessor/text/dict/korean.json
    relative to the working directory): stub what the image lacks and import it from the reference's root.  Nothing of it is instantiated:
    the loss METHODS are called unbound on seeded tensors."""
    for name in ("torchaudio.transforms", "tensorboard", "torch.utils.tensorboard", "textgrid", "miditoolkit", "g2pk", "jamo", "resemblyzer",
                 "essentia", "essentia.standard", "pypinyin", "g2p_en", "praatio"):
        sys.modules.setdefault(name, MagicMock())
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import tasks.visinger as tv
        import tasks.base as tb
    finally:
        os.chdir(cwd)
    return tv, tb


def gen_losses():
    """The loss functions of the GAN training step, from the reference's own task classes (tasks/visinger.py:100-107 KL weighting,
    127-145 pitch, 140-145 CTC, 147-169 LSGAN + feature matching; tasks/base.py:227-238 masked mel L1 with the `mel_losses: l1:45.0`
    table of config/models/visinger.yaml:55) called UNBOUND on seeded tensors with the reference's global `hparams` dict set to the YAML
    values (config/models/visinger.yaml:51-64; lambda_uv / lambda_f0 exist in no YAML: SURVEY 3.5 -- set to lambda_pitch as
    visinger_amd/train.py does).  Pins visinger_amd/train.py (VERDICT r3 missing #2).  The mel TRANSFORM (torchaudio) stays unpinned."""
    import yaml
    from types import SimpleNamespace
    tv, tb = _import_reference_task()
    from utils.commons.hparams import hparams
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "models", "visinger.yaml")))
    keys = ("lambda_pitch", "lambda_ctc", "lambda_mel_adv", "lambda_kl", "lambda_fm", "kl_start_steps", "kl_min", "mel_losses")
    hp = {k: cfg[k] for k in keys}
    hp["lambda_uv"] = hp["lambda_f0"] = hp["lambda_pitch"]
    hparams.clear()
    hparams.update(hp)
    g = torch.Generator().manual_seed(77)
    B, T, M, Tph, D = 3, 40, 12, 9, 14
    Task = tv.VISingerTask
    # ---- masked mel L1 (tasks/base.py:227-238) through add_mel_loss with the parsed `mel_losses` table
    mel_out = torch.randn(B, T, M, generator=g)
    mel_tgt = torch.randn(B, T, M, generator=g)
    mel_tgt[1, 30:] = 0.0                       # padded frames of the target: weight 0
    mel_tgt[2, 17] = 0.0
    me = SimpleNamespace(mel_losses={"l1": 45.0})
    me.l1_loss = lambda a, b: tb.SpeechBaseTask.l1_loss(me, a, b)
    losses = {}
    tb.SpeechBaseTask.add_mel_loss(me, mel_out, mel_tgt, losses)
    out = {"mel_out": mel_out, "mel_tgt": mel_tgt, "mel_l1": losses["mel_l1"], "mel_l1_unweighted": tb.SpeechBaseTask.l1_loss(me, mel_out, mel_tgt)}
    # ---- pitch (tasks/visinger.py:127-139)
    mel2ph = torch.randint(1, Tph + 1, (B, T), generator=g)
    mel2ph[1, 33:] = 0
    f0 = torch.rand(B, T, generator=g) * 2 + 4
    uv = (torch.rand(B, T, generator=g) < 0.3).float()
    f0_pred = torch.randn(B, T, 2, generator=g)
    losses = {}
    Task.add_pitch_loss(None, {"f0_pred": f0_pred}, {"f0": f0, "uv": uv, "mel2ph": mel2ph}, losses)
    out.update(p_mel2ph=mel2ph, p_f0=f0, p_uv=uv, p_pred=f0_pred, uv_loss=losses["uv"], f0_loss=losses["f0"])
    # ---- CTC (tasks/visinger.py:140-145)
    ph_pred = torch.log_softmax(torch.randn(B, D, T, generator=g), dim=1)
    text = torch.randint(1, D, (B, Tph), generator=g)
    mel_len = torch.tensor([T, 33, T])
    txt_len = torch.tensor([Tph, 6, 8])
    losses = {}
    Task.add_ctc_loss(None, {"ph_pred": ph_pred}, {"mel_lengths": mel_len, "text_tokens": text, "text_lengths": txt_len}, losses)
    out.update(c_ph_pred=ph_pred, c_text=text, c_mel_len=mel_len, c_txt_len=txt_len, ctc_loss=losses["ctc"])
    # ---- LSGAN + feature matching (tasks/visinger.py:147-169): 6 discriminators' logits, 6 x 3 feature maps
    d_tgt = [torch.randn(B, 5 + i, generator=g) for i in range(6)]
    d_gen = [torch.randn(B, 5 + i, generator=g) for i in range(6)]
    f_tgt = [[torch.randn(B, 4, 7 + j, i + 1, generator=g) for j in range(3)] for i in range(6)]
    f_gen = [[torch.randn(B, 4, 7 + j, i + 1, generator=g) for j in range(3)] for i in range(6)]
    out.update(disc_loss=Task.add_discriminator_loss(None, d_tgt, d_gen), gen_loss=Task.add_generator_loss(None, d_gen),
               fm_loss=Task.add_feature_matching_loss(None, f_tgt, f_gen))
    for i in range(6):
        out[f"d_tgt{i}"], out[f"d_gen{i}"] = d_tgt[i], d_gen[i]
        for j in range(3):
            out[f"f_tgt{i}_{j}"], out[f"f_gen{i}_{j}"] = f_tgt[i][j], f_gen[i][j]
    # ---- KL weighting (tasks/visinger.py:103-107 inside run_model: restated arithmetic of those four lines is NOT callable in isolation;
    # the fixture records the hparams it uses so the test can check the constants train.py carries)
    out["hparams_json"] = np.frombuffer(json.dumps(hp, sort_keys=True).encode(), dtype=np.uint8)
    save("losses", **out)


def gen_align_io():
    """utils/audio/align.py:58-104 get_note2dur (integer frame bookkeeping of a note list) and utils/audio/io.py:8-12 save_wav (int16
    bytes), from the reference's own functions."""
    import io as _io
    import tempfile
    from utils.audio.align import get_note2dur
    from utils.audio.io import save_wav
    from scipy.io import wavfile
    # midi_info rows: (Bar, Pos, Pitch, Duration_midi, start_time, end_time, Tempo, syllable phones, lyric token) -- the function reads
    # [4], [5], [7] and [8]
    r = np.random.default_rng(5)
    cases = {}
    for name, hop, sr, min_sil in (("a", 300, 24000, 0.0), ("b", 256, 22050, 0.05), ("c", 300, 24000, 0.02)):
        t, rows = 0.0, []
        n = 14
        for i in range(n):
            # (the reference asserts that every frame belongs to a note: gaps must be closed by min_sil_duration, the first note starts at 0)
            gap = 0.0 if (i == 0 or min_sil == 0.0) else float(r.choice([0.0, 0.0, 0.004, 0.012]))
            dur = float(r.uniform(0.12, 0.6))
            nph = int(r.choice([1, 2, 3]))
            if i in (4, 5) and name != "a":
                phones, tok = ["|"], "|"          # consecutive rests: merged (align.py:66-67)
                nph = 1
            else:
                phones, tok = [f"p{i}_{j}" for j in range(nph)], f"s{i}"
            start = t + gap
            rows.append([1 + i // 4, i % 4, int(r.integers(40, 80)), int(r.integers(1, 8)), round(start, 4), round(start + dur, 4), 120, phones, tok])
            t = start + dur
        rows_in = json.loads(json.dumps(rows))
        mel2phone, mel2note, duration, ph_list, midi_out = get_note2dur(rows, hop, sr, min_sil_duration=min_sil)
        cases[name] = {"midi_info": rows_in, "hop_size": hop, "sample_rate": sr, "min_sil_duration": min_sil, "mel2phone": mel2phone,
                       "mel2note": mel2note, "duration": duration, "ph_list": ph_list, "midi_info_out": midi_out}
    with open(os.path.join(OUT, "note2dur.json"), "w") as f:
        json.dump(cases, f)
    print("note2dur.json")
    wav32 = (r.standard_normal(4000) * 0.3).astype(np.float32)
    wav32[10], wav32[11] = 0.99999, -1.0
    wav64 = wav32.astype(np.float64) * 1.7
    arrays = {"wav32": wav32, "wav64": wav64}
    with tempfile.TemporaryDirectory() as d:
        for key, wav, norm in (("f32", wav32, False), ("f32_norm", wav32, True), ("f64_norm", wav64, True)):
            path = os.path.join(d, key + ".wav")
            save_wav(wav.copy(), path, 22050, norm=norm)
            sr, pcm = wavfile.read(path)
            arrays["pcm_" + key] = pcm
            arrays["bytes_" + key] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    save("save_wav", **arrays)


def gen_mel_processing():
    """utils/audio/mel_processing.py:15-38 (SpectrogramFixed / MelSpectrogramFixed: torchaudio's transforms, the last frame dropped, log(mel + 1e-3)) with the
    task's parameters (tasks/visinger.py:30-35, datasets/svs/csd/preprocess.yaml: n_fft 2048, win 1200, hop 300, 128 mels, 20-12000 Hz, 24 kHz) and the hop-256
    variant on seeded waveforms -> mel_processing.npz.  Written only where the reference's dependency exists."""
    if not HAVE_TORCHAUDIO:
        print("mel_processing.npz: SKIPPED (torchaudio is not importable here: the mel transform stays parity-unpinned)")
        return
    from utils.audio.mel_processing import MelSpectrogramFixed, SpectrogramFixed
    r = np.random.default_rng(77)
    arrays = {}
    for tag, sr, hop, win in (("hop300", 24000, 300, 1200), ("hop256", 22050, 256, 1024)):
        wav = torch.from_numpy((r.standard_normal((2, hop * 40)) * 0.1).astype(np.float32))
        spec = SpectrogramFixed(n_fft=2048, win_length=win, hop_length=hop, window_fn=torch.hann_window)(wav)
        mel = MelSpectrogramFixed(sample_rate=sr, n_fft=2048, win_length=win, hop_length=hop, f_min=20.0, f_max=12000.0 if sr == 24000 else 11025.0, n_mels=128,
                                  window_fn=torch.hann_window)(wav)
        arrays.update({f"{tag}.wav": wav.numpy(), f"{tag}.spec": spec.numpy(), f"{tag}.mel": mel.numpy(),
                       f"{tag}.params": np.array([sr, 2048, win, hop, 128, 20.0, 12000.0 if sr == 24000 else 11025.0])})
    save("mel_processing", **arrays)


if __name__ == "__main__":
    gen_mel_processing()
    gen_wavenet()
    gen_posterior()
    gen_flow()
    gen_generator()
    gen_transformer()
    gen_transformer_options()
    gen_wrappers()
    gen_integer()
    gen_discriminators()
    gen_model()
    gen_model_pitch()
    gen_losses()
    gen_align_io()
