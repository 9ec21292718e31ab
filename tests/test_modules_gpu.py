"""GPU parity of the nn.Module mirrors (visinger_amd.modules.*) against the golden vectors produced by the
reference itself and against the fp64 oracle.  Every module is built with the reference's constructor arguments,
loaded with the fixture's state_dict (strict=True: the parameter names/shapes ARE the reference's) and run through
the C ABI on the MI355X.  Tolerances: activations 2e-5 abs+rel (fp32 MFMA vs the reference's fp32 CPU result),
waveform 1e-4 abs, integer paths bit-exact, mean_only log-det exactly 0."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def cu(a):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.cuda()


def load(module, w):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return module.cuda().eval()


def close(got, ref, atol=2e-5, rtol=2e-5):
    got = got.detach().cpu().double().numpy()
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f"max err {err.max():.3e} (ref max {np.abs(ref).max():.3e})"


@pytest.mark.parametrize("tag", ["wavenet_g", "wavenet_nog_dil2", "wavenet_1layer"])
def test_wavenet(tag):
    from visinger_amd.modules.visinger.encoder import WaveNet
    w, a = load_golden(tag)
    H, k, dr, Ln, gin = (int(v) for v in a["cfg"])
    m = load(WaveNet(H, k, dr, Ln, gin_channels=gin), w)
    x = cu(a["x"])
    x0 = x.clone()
    with torch.no_grad():
        y = m(x, cu(a["mask"]), g=cu(a["g"]) if "g" in a else None)
    close(y, a["y"])
    assert torch.equal(x, x0), "input must not be mutated"


def test_posterior_encoder():
    from visinger_amd.modules.visinger.encoder import PosteriorEncoder
    w, a = load_golden("posterior")
    cin, cout, H, k, dr, Ln, gin = (int(v) for v in a["cfg"])
    m = load(PosteriorEncoder(cin, cout, H, k, dr, Ln, gin), w)
    with torch.no_grad():
        z, mu, logs = m(cu(a["x"]), cu(a["mask"]), g=cu(a["g"]), noise=cu(a["noise"]))
    close(mu, a["mu"])
    close(logs, a["logs"])
    close(z, a["z"])


@pytest.mark.parametrize("mean_only", [0, 1])
def test_coupling_layer(mean_only):
    from visinger_amd.modules.visinger.flow import ResidualCouplingLayer
    w, a = load_golden(f"coupling_meanonly{mean_only}")
    C, H, k, dr, Ln, gin, mo = (int(v) for v in a["cfg"])
    m = load(ResidualCouplingLayer(C, H, k, dr, Ln, gin_channels=gin, mean_only=bool(mo)), w)
    x, mask, g = cu(a["x"]), cu(a["mask"]), cu(a["g"])
    with torch.no_grad():
        y, logdet = m(x, mask, g=g, reverse=False)
        yi = m(x, mask, g=g, reverse=True)
        xr = m(y, mask, g=g, reverse=True)
    close(y, a["y"])
    close(yi, a["y_inv"])
    close(xr, a["x_roundtrip"])
    if mo:
        assert (logdet == 0).all()                       # exactly zero (flow.py:73-75,80)
    else:
        ld = logdet.cpu().double().numpy()
        assert np.abs(ld - a["logdet"]).max() <= 1e-4 * np.abs(a["logdet"]).max()   # north-star: 1e-4 relative


@pytest.mark.parametrize("tag", ["flow_block", "flow_block_nog"])
def test_flow_block(tag):
    from visinger_amd.modules.visinger.flow import ResidualCouplingBlock
    w, a = load_golden(tag)
    C, H, k, dr, Ln, nf, gin = (int(v) for v in a["cfg"])
    m = load(ResidualCouplingBlock(C, H, k, dr, Ln, n_flows=nf, gin_channels=gin), w)
    x, mask = cu(a["x"]), cu(a["mask"])
    g = cu(a["g"]) if "g" in a else None
    with torch.no_grad():
        y = m(x, mask, g=g, reverse=False)
        yi = m(x, mask, g=g, reverse=True)
        xr = m(y, mask, g=g, reverse=True)
    close(y, a["y"])
    close(yi, a["y_inv"])
    # encode -> decode round trip restores x on valid frames
    close(xr * mask, a["x"] * a["mask"], atol=5e-5)


def test_flip():
    from visinger_amd.modules.visinger.flow import Flip
    _, a = load_golden("flip")
    y, ld = Flip()(cu(a["x"]), reverse=False)
    assert np.array_equal(y.cpu().numpy(), a["y"]) and (ld == 0).all()
    assert np.array_equal(Flip()(cu(a["x"]), reverse=True).cpu().numpy(), a["y_rev"])


@pytest.mark.parametrize("tag", ["generator_hop256_like", "generator_hop300_like", "generator_rb2_nog"])
def test_generator(tag):
    from visinger_amd.modules.visinger.decoder import Generator
    w, a = load_golden(tag)
    ic, ui, gin = (int(v) for v in a["cfg"])
    m = load(Generator(ic, str(int(a["rb"])), a["rk"].tolist(), a["rd"].tolist(), a["rates"].tolist(), ui,
                       a["uk"].tolist(), gin_channels=gin), w)
    with torch.no_grad():
        y = m(cu(a["x"]), g=cu(a["g"]) if "g" in a else None)
    close(y, a["y"], atol=1e-4, rtol=0)        # waveform: 1e-4 abs (tanh-bounded)


def test_resblocks():
    from visinger_amd.modules.visinger.decoder import ResBlock1, ResBlock2
    for tag, cls in (("resblock1", ResBlock1), ("resblock2", ResBlock2)):
        w, a = load_golden(tag)
        C, k, *d = (int(v) for v in a["cfg"])
        m = load(cls(C, k, tuple(d)), w)
        with torch.no_grad():
            close(m(cu(a["x"])), a["y"])
            close(m(cu(a["x"]), cu(a["mask"])), a["y_masked"])


def test_layernorm():
    from visinger_amd.modules.rel_transformer import LayerNorm
    w, a = load_golden("layernorm")
    m = load(LayerNorm(a["x"].shape[1]), w)
    with torch.no_grad():
        close(m(cu(a["x"])), a["y"])


@pytest.mark.parametrize("tag", ["mha_rel", "mha_rel_short"])
def test_mha(tag):
    from visinger_amd.modules.rel_transformer import MultiHeadAttention
    w, a = load_golden(tag)
    C, nh, ws = (int(v) for v in a["cfg"])
    m = load(MultiHeadAttention(C, C, nh, window_size=ws), w)
    x, mask = cu(a["x"]), cu(a["mask"])
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    with torch.no_grad():
        close(m(x, x, attn_mask), a["y"])              # the reference's 4-D mask argument
        close(m(x, x, frame_mask=mask), a["y"])
        assert m.attn is None                          # the streaming kernel keeps no [T, T] tensor ...
        m.store_attn = True                            # ... unless asked: the reference's `self.attn` (rel_transformer.py:143, 171), pinned to ITS probabilities
        close(m(x, x, frame_mask=mask), a["y"])
        assert m.attn.shape == a["p_attn"].shape
        close(m.attn, a["p_attn"])


@pytest.mark.parametrize("tag", ["mha_proximal", "mha_block", "mha_proximal_block"])
def test_mha_options_visinger_never_sets(tag):
    """proximal_bias / block_length (reference modules/rel_transformer.py:163-170, 245-256) against the reference's golden vectors (round 5: they used to raise)"""
    from visinger_amd.modules.rel_transformer import MultiHeadAttention
    w, a = load_golden(tag)
    C, nh, ws, prox, bl = (int(v) for v in a["cfg"])
    m = load(MultiHeadAttention(C, C, nh, window_size=ws, proximal_bias=bool(prox), block_length=None if bl < 0 else bl), w)
    x, mask = cu(a["x"]), cu(a["mask"])
    with torch.no_grad():
        close(m(x, x, mask.unsqueeze(2) * mask.unsqueeze(-1)), a["y"])
        close(m(x, x, frame_mask=mask), a["y"])


def test_ffn_gelu():
    """FFN(activation="gelu"), rel_transformer.py:338-341"""
    from visinger_amd.modules.rel_transformer import FFN
    w, a = load_golden("ffn_gelu")
    cin, cout, fc, ks = (int(v) for v in a["cfg"])
    m = load(FFN(cin, cout, fc, ks, activation="gelu"), w)
    with torch.no_grad():
        close(m(cu(a["x"]), cu(a["mask"])), a["y"])


@pytest.mark.parametrize("tag", ["rel_encoder_preln", "rel_encoder_preln_g"])
def test_rel_encoder_pre_ln(tag):
    """RelativeEncoder(pre_ln=True), rel_transformer.py:284, 301-317"""
    from visinger_amd.modules.rel_transformer import RelativeEncoder
    w, a = load_golden(tag)
    C, F, nh, nl, ks, gin = (int(v) for v in a["cfg"])
    m = load(RelativeEncoder(C, F, nh, nl, kernel_size=ks, pre_ln=True, gin_channels=None if gin < 0 else gin), w)
    with torch.no_grad():
        y = m(cu(a["x"]), cu(a["mask"]), cu(a["g"]) if "g" in a else None)
    close(y, a["y"], atol=5e-5)


def test_ffn():
    from visinger_amd.modules.rel_transformer import FFN
    w, a = load_golden("ffn")
    cin, cout, fc, ks = (int(v) for v in a["cfg"])
    m = load(FFN(cin, cout, fc, ks), w)
    with torch.no_grad():
        close(m(cu(a["x"]), cu(a["mask"])), a["y"])


@pytest.mark.parametrize("tag", ["rel_encoder_g", "rel_encoder_nog", "rel_encoder_spk"])
def test_rel_encoder(tag):
    from visinger_amd.modules.rel_transformer import RelativeEncoder
    w, a = load_golden(tag)
    C, F, nh, nl, ks, gin = (int(v) for v in a["cfg"])
    m = load(RelativeEncoder(C, F, nh, nl, kernel_size=ks, gin_channels=None if gin < 0 else gin), w)
    with torch.no_grad():
        y = m(cu(a["x"]), cu(a["mask"]), cu(a["g"]) if "g" in a else None)
    close(y, a["y"], atol=5e-5)


def test_wrappers():
    from visinger_amd.modules.visinger.encoder import FramePriorNetwork
    from visinger_amd.modules.visinger.predictor import PitchPredictor, PhonemePredictor
    w, a = load_golden("frame_prior")
    C, F, nh, nl, ks, gin = (int(v) for v in a["cfg"])
    m = load(FramePriorNetwork(C, F, nh, nl, ks, gin_channels=gin, p_dropout=0.0), w)
    with torch.no_grad():
        mu, logs = m(cu(a["x"]), cu(a["mask"]), None)
        close(mu, a["mu"], atol=5e-5)
        close(logs, a["logs"], atol=5e-5)
        mu, logs = m(cu(a["x"]), cu(a["mask"]), cu(a["g_BT1"]))
        close(mu, a["mu_g"], atol=5e-5)
        close(logs, a["logs_g"], atol=5e-5)
    w, a = load_golden("pitch_predictor")
    C, F, nh, nl, ks, gin, od = (int(v) for v in a["cfg"])
    m = load(PitchPredictor(C, F, nh, nl, ks, 0.0, gin_channels=gin, out_dim=od), w)
    with torch.no_grad():
        close(m(cu(a["x"]), cu(a["mask"]), cu(a["spk"])), a["y"], atol=5e-5)
    w, a = load_golden("phoneme_predictor")
    D, C, F, nh, nl, ks = (int(v) for v in a["cfg"])
    m = load(PhonemePredictor(D, C, F, nh, nl, ks, 0.0), w)
    with torch.no_grad():
        close(m(cu(a["x"]), cu(a["mask"])), a["y"], atol=5e-5)


def test_text_encoder():
    from visinger_amd.modules.visinger.encoder import TextEncoder
    w, a = load_golden("text_encoder")
    nph, npi, ndu, C, F, nh, nl, ks = (int(v) for v in a["cfg"])
    m = load(TextEncoder(nph, npi, ndu, C, F, nh, nl, ks, 0.0, True), w)
    with torch.no_grad():
        y = m(cu(a["text"]), cu(a["pitch"]), cu(a["dur"]), cu(a["mel2ph"]))
    close(y, a["y"], atol=5e-5)


def test_integer_paths_bit_exact():
    from visinger_amd.models.commons.align_ops import expand_states
    from visinger_amd.modules.rel_transformer import SinusoidalPositionalEmbedding
    from visinger_amd.modules.commons.utils import slice_segments, rand_slice_segments
    _, a = load_golden("expand_states")
    assert np.array_equal(expand_states(cu(a["h"]), cu(a["mel2ph"])).cpu().numpy(), a["y"])
    _, a = load_golden("positions")
    pos = SinusoidalPositionalEmbedding.make_positions(cu(a["x"]), 0)
    assert pos.dtype == torch.int64 and np.array_equal(pos.cpu().numpy(), a["positions"])
    emb = SinusoidalPositionalEmbedding(12, 0, init_size=16).cuda()
    assert np.array_equal(emb(a["x"].shape[0], a["x"].shape[1], cu(a["x"])).cpu().numpy(), a["y"])
    _, a = load_golden("slice_segments")
    assert np.array_equal(slice_segments(cu(a["x"]), cu(a["ids"]), 8).cpu().numpy(), a["y"])
    torch.manual_seed(1234)
    ys, ids = rand_slice_segments(cu(a["x"]), 8)
    assert np.array_equal(ids.cpu().numpy(), a["rand_ids"]) and np.array_equal(ys.cpu().numpy(), a["rand_y"])
    from visinger_amd.models.commons.align_ops import mel2token_to_dur
    _, a = load_golden("mel2token_to_dur")
    m2p, Tph = cu(a["mel2ph"]), a["dur"].shape[1]
    d = mel2token_to_dur(m2p, Tph)
    assert d.dtype == torch.int64 and np.array_equal(d.cpu().numpy(), a["dur"])
    assert np.array_equal(mel2token_to_dur(m2p, Tph, max_dur=4).cpu().numpy(), a["dur_clamped"])
    assert np.array_equal(mel2token_to_dur(m2p[0], Tph).cpu().numpy(), a["dur_1d"])
    assert np.array_equal(mel2token_to_dur(m2p[:, :11]).cpu().numpy(), a["dur_auto"])


def test_fused_path_refuses_autograd():
    """The fused inference kernels have no backward: reaching them with autograd on in training mode is an error
    (module.forward dispatches that case to visinger_amd.autograd instead)."""
    from visinger_amd.modules.hipconv import HipConv1d
    m = HipConv1d(8, 8, 3, padding=1).cuda().train()
    with pytest.raises(NotImplementedError):
        m.run(torch.zeros(1, 8, 16, device="cuda"))


def test_index_ops_random_large(oracle):
    """vs_expand_states / vs_make_positions / vs_slice_segments at the north-star size vs the numpy oracle: bit-exact."""
    from visinger_amd.ops import expand_states, make_positions, slice_segments
    r = np.random.default_rng(5)
    B, Tp, T, H = 32, 128, 1024, 192
    h = r.standard_normal((B, Tp, H)).astype(np.float32)
    m2p = r.integers(0, Tp + 1, (B, T)).astype(np.int64)
    m2p[3, 700:] = 0
    m2p[0, :] = 0                                                     # an all-padding item
    ref = oracle.expand_states(h, m2p)
    assert np.array_equal(expand_states(cu(h), cu(m2p)).cpu().numpy(), ref)
    got_cf = expand_states(cu(np.ascontiguousarray(h.transpose(0, 2, 1))), cu(m2p), h_channels_first=True, out_channels_first=True)
    assert np.array_equal(got_cf.cpu().numpy(), ref.transpose(0, 2, 1))
    x = r.standard_normal((B, T)).astype(np.float32)
    x[r.random((B, T)) < 0.3] = 0.0
    x[1, :] = 0.0
    assert np.array_equal(make_positions(cu(x), 0).cpu().numpy(), oracle.make_positions(x, 0))
    z = r.standard_normal((B, 192, T)).astype(np.float32)
    ids = r.integers(0, T - 32 + 1, (B,)).astype(np.int64)
    ids[0], ids[1] = 0, T - 32
    assert np.array_equal(slice_segments(cu(z), cu(ids), 32).cpu().numpy(), oracle.slice_segments(z, ids, 32))
    from visinger_amd.ops import mel2token_to_dur
    assert np.array_equal(mel2token_to_dur(cu(m2p), Tp).cpu().numpy(), oracle.mel2token_to_dur(m2p, Tp))
    assert int(mel2token_to_dur(cu(m2p), Tp).sum()) == int((m2p > 0).sum())          # every non-padding frame counted once


def test_discriminators_gpu():
    """a13 on the HIP kernels (MFMA engine + grouped VALU kernels, reference state-dict layout) vs the reference golden outputs."""
    from test_oracle_golden import _disc_weights
    from visinger_amd.modules.discriminator import DiscriminatorP, DiscriminatorS
    _, a = load_golden("discriminators")
    m, _ = _disc_weights(DiscriminatorS, 72)
    with torch.no_grad():
        lt, ft = m.cuda()(cu(a["y"]))
    close(lt, a["s_logits"], atol=1e-4)
    close(ft[0], a["s_fmap0"], atol=1e-4)
    for p in (2, 3, 11):
        m, _ = _disc_weights(DiscriminatorP, 73 + p, p)
        with torch.no_grad():
            lt, ft = m.cuda()(cu(a["y"]))
        close(lt, a[f"p{p}_logits"], atol=1e-4)
        close(ft[-1], a[f"p{p}_fmap_last"], atol=1e-4)


@pytest.mark.parametrize("math", ["6", "0"])
def test_modules_random_sweep(monkeypatch, math, vs_option):
    """tools/module_fuzz.py: 60 random (hyper-parameters, batch, length, ragged mask) cases over WaveNet, coupling layer /
    block (both directions, log-det), generator (random upsampling stacks, ResBlock1/2), relative encoder and posterior
    encoder, against the fp64 oracle -- on the default split-bf16 engine and on the fp32 MFMA / F(2,3) engine."""
    vs_option("VS_CONV_MATH", int(math))
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import module_fuzz
    monkeypatch.setattr(sys, "argv", ["module_fuzz.py", "60", "7"])
    module_fuzz.main()


def _torch_reference_discriminator(m, x):
    """the reference's forward (modules/discriminator.py:28-47, 64-75) on the same parameter holders, PyTorch ops"""
    import torch.nn.functional as F
    fmap = []
    if hasattr(m, "period"):
        b, c, t = x.shape
        if t % m.period:
            x = F.pad(x, (0, m.period - t % m.period), "reflect")
        x = x.view(b, c, -1, m.period)
    for conv in m.convs:
        x = F.leaky_relu(conv(x), 0.1)
        fmap.append(x)
    x = m.conv_post(x)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


@pytest.mark.parametrize("which", ["S", "P2", "P3", "P5", "P11"])
def test_discriminator_gradients_match_pytorch_autograd(which):
    """a13 on the HIP kernels: logits, every feature map, the input gradient and every parameter gradient against PyTorch
    autograd on the identical parameter holders (LSGAN + feature-matching shaped objective, waveform length not a multiple of
    the period)."""
    from visinger_amd.modules.discriminator import DiscriminatorP, DiscriminatorS
    torch.manual_seed(17)
    m = (DiscriminatorS() if which == "S" else DiscriminatorP(int(which[1:]))).cuda()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() > 1 and p_.shape[1:].numel() > 1:
                p_.copy_(torch.randn_like(p_) / p_.shape[1:].numel() ** 0.5)
    x = (0.5 * torch.randn(2, 1, 1237, device="cuda")).requires_grad_(True)
    target = [None]

    def objective(fn):
        logits, fmap = fn(m, x)
        loss = ((1 - logits) ** 2).mean() + sum(f.abs().mean() for f in fmap)
        grads = torch.autograd.grad(loss, [x] + list(m.parameters()))
        return logits.detach(), [f.detach() for f in fmap], [g.detach() for g in grads]

    lo_h, fm_h, gr_h = objective(lambda mod, inp: mod(inp))
    lo_r, fm_r, gr_r = objective(_torch_reference_discriminator)

    def rel(a, b_):
        return float((a - b_).abs().max()) / (1e-6 + float(b_.abs().max()))

    assert lo_h.shape == lo_r.shape and rel(lo_h, lo_r) <= 2e-5
    for a, b_ in zip(fm_h, fm_r):
        assert a.shape == b_.shape and rel(a, b_) <= 2e-5
    names = ["x"] + [n for n, _ in m.named_parameters()]
    for n, a, b_ in zip(names, gr_h, gr_r):
        assert a.shape == b_.shape and rel(a, b_) <= 2e-4, (n, rel(a, b_))
