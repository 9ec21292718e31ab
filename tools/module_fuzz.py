#!/usr/bin/env python3
"""Randomised module-level parity sweep: the MI355X-native modules at random hyper-parameters (channel counts that are not
multiples of the tile sizes, 1-4 layers, with / without conditioning, ragged masks, tiny and odd lengths) against the fp64
oracle.    python tools/module_fuzz.py [n_cases] [seed]        (GPU only; the oracle is the checker)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import visinger_oracle as orc  # noqa: E402


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rand_sd(module, rng, scale=1.0):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias") or n.endswith("beta"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("gamma"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                fan = max(1, int(np.prod(p.shape[1:])))
                p.copy_(scale * torch.randn(p.shape, generator=g) / np.sqrt(fan))
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def err(got, ref, scaled=False):
    g = got.detach().cpu().double().numpy()
    assert g.shape == np.shape(ref), (g.shape, np.shape(ref))
    assert np.isfinite(g).all()
    if not g.size:
        return 0.0
    d = np.abs(g - np.asarray(ref, np.float64))
    return float((d / (1.0 + np.abs(ref))).max() if scaled else d.max())


def ragged_mask(rng, B, T):
    lens = rng.integers(1, T + 1, B)
    lens[rng.integers(B)] = T
    return (np.arange(T)[None] < lens[:, None]).astype(np.float32)[:, None]


def case_wavenet(rng):
    from visinger_amd.modules.visinger.encoder import WaveNet
    H, k, L = int(rng.choice([16, 48, 80, 192])), int(rng.choice([3, 5])), int(rng.integers(1, 5))
    gin = int(rng.choice([0, 8, 256]))
    B, T = int(rng.integers(1, 4)), int(rng.choice([1, 3, 17, 64, 200, 257]))
    m = WaveNet(H, k, 1, L, gin_channels=gin)
    sd = rand_sd(m, rng)
    m = m.cuda().eval()
    x = rng.standard_normal((B, H, T)).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    g = rng.standard_normal((B, gin, 1)).astype(np.float32) if gin else None
    ref = orc.wavenet(sd, x, mask, g, hidden_channels=H, kernel_size=k, dilation_rate=1, n_layers=L)
    with torch.no_grad():
        y = m(cu(x), cu(mask), g=None if g is None else cu(g))
    return f"wavenet H{H} k{k} L{L} gin{gin} B{B} T{T}", err(y, ref), 5e-5


def case_flow(rng):
    from visinger_amd.modules.visinger.flow import ResidualCouplingBlock, ResidualCouplingLayer
    C, H = int(rng.choice([2, 16, 62, 64, 192])), int(rng.choice([16, 48, 192]))
    k, L, gin = int(rng.choice([3, 5])), int(rng.integers(1, 4)), int(rng.choice([0, 8]))
    B, T = int(rng.integers(1, 4)), int(rng.choice([1, 5, 33, 128, 300]))
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    g = rng.standard_normal((B, gin, 1)).astype(np.float32) if gin else None
    gg = None if g is None else cu(g)
    if rng.random() < 0.5:
        nf = int(rng.integers(1, 5))
        m = ResidualCouplingBlock(C, H, k, 1, L, n_flows=nf, gin_channels=gin)
        sd = rand_sd(m, rng, 0.5)
        m = m.cuda().eval()
        kw = dict(channels=C, hidden_channels=H, kernel_size=k, dilation_rate=1, n_layers=L, n_flows=nf)
        rev = bool(rng.random() < 0.5)
        ref = orc.flow_block(sd, x, mask, g, rev, **kw)
        with torch.no_grad():
            y = m(cu(x), cu(mask), g=gg, reverse=rev)
        return f"flow block C{C} H{H} k{k} L{L} nf{nf} gin{gin} rev{int(rev)} B{B} T{T}", err(y, ref), 5e-5
    mo = bool(rng.random() < 0.5)
    m = ResidualCouplingLayer(C, H, k, 1, L, gin_channels=gin, mean_only=mo)
    sd = rand_sd(m, rng, 0.5)
    m = m.cuda().eval()
    kw = dict(channels=C, hidden_channels=H, kernel_size=k, dilation_rate=1, n_layers=L, mean_only=mo)
    ref_y, ref_ld = orc.coupling_layer(sd, x, mask, g, False, **kw)
    ref_inv = orc.coupling_layer(sd, x, mask, g, True, **kw)
    with torch.no_grad():
        y, ld = m(cu(x), cu(mask), g=gg, reverse=False)
        yi = m(cu(x), cu(mask), g=gg, reverse=True)
    e = max(err(y, ref_y), err(yi, ref_inv))
    lde = float(np.abs(ld.cpu().double().numpy() - ref_ld).max() / max(1.0, np.abs(ref_ld).max()))
    if mo:
        assert bool((ld == 0).all())
    return f"coupling C{C} H{H} k{k} L{L} gin{gin} mean_only{int(mo)} B{B} T{T}", max(e, lde), 5e-5


def case_generator(rng):
    from visinger_amd.modules.visinger.decoder import Generator
    ic, ui = int(rng.choice([16, 48, 192])), int(rng.choice([32, 64, 96, 128]))
    rates, kers = [], []
    for _ in range(int(rng.integers(1, 4))):
        u = int(rng.choice([2, 3, 4, 5, 8]))
        kk = u + 2 * int(rng.integers(0, 3))
        if kk // u > 3:
            kk = u
        rates.append(u)
        kers.append(kk)
    while ui // (2 ** len(rates)) < 4:
        rates.pop(); kers.pop()
    rb = str(rng.choice(["1", "2"]))
    rk = [int(v) for v in rng.choice([3, 5, 7, 11], size=int(rng.integers(1, 4)), replace=False)]
    rd = [[1, 3, 5] if rb == "1" else [1, 3]] * len(rk)
    gin = int(rng.choice([0, 8]))
    B, T = int(rng.integers(1, 3)), int(rng.choice([1, 2, 7, 20, 33]))
    m = Generator(ic, rb, rk, rd, rates, ui, kers, gin_channels=gin)
    sd = rand_sd(m, rng)
    m = m.cuda().eval()
    x = rng.standard_normal((B, ic, T)).astype(np.float32)
    g = rng.standard_normal((B, gin, 1)).astype(np.float32) if gin else None
    ref = orc.generator(sd, x, g, resblock=rb, resblock_kernel_sizes=rk, resblock_dilation_sizes=rd, upsample_rates=rates,
                        upsample_kernel_sizes=kers)
    with torch.no_grad():
        y = m(cu(x), g=None if g is None else cu(g))
    return f"generator ic{ic} ui{ui} rates{rates} kers{kers} rb{rb} rk{rk} gin{gin} B{B} T{T}", err(y, ref), 1e-4


def case_encoder(rng):
    from visinger_amd.modules.rel_transformer import RelativeEncoder
    nh = int(rng.choice([1, 2, 4]))
    C = nh * int(rng.choice([8, 24, 33, 96, 128]))
    F_, L, ks = int(rng.choice([32, 100, 256])), int(rng.integers(1, 3)), int(rng.choice([1, 3, 9]))
    ws = int(rng.choice([1, 4, 7]))
    gin = int(rng.choice([0, 1, 8]))
    B, T = int(rng.integers(1, 4)), int(rng.choice([1, 2, 9, 31, 32, 33, 100, 260]))
    m = RelativeEncoder(C, F_, nh, L, kernel_size=ks, window_size=ws, gin_channels=gin if gin else None)
    sd = rand_sd(m, rng)
    m = m.cuda().eval()
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    g = rng.standard_normal((B, gin, T if rng.random() < 0.5 else 1)).astype(np.float32) if gin else None
    ref = orc.rel_encoder(sd, x, mask, g, n_heads=nh, n_layers=L, kernel_size=ks, window_size=ws)
    with torch.no_grad():
        y = m(cu(x), cu(mask), g=None if g is None else cu(g))
    return f"rel_encoder C{C} nh{nh} F{F_} L{L} ks{ks} ws{ws} gin{gin} g_T{None if g is None else g.shape[2]} B{B} T{T}", err(y, ref), 1e-4


def case_posterior(rng):
    from visinger_amd.modules.visinger.encoder import PosteriorEncoder
    cin, H = int(rng.choice([21, 80, 513])), int(rng.choice([16, 48, 192]))
    L, gin = int(rng.integers(1, 4)), int(rng.choice([0, 8]))
    B, T = int(rng.integers(1, 3)), int(rng.choice([1, 6, 50, 129]))
    m = PosteriorEncoder(cin, H, H, 5, 1, L, gin_channels=gin)
    sd = rand_sd(m, rng)
    m = m.cuda().eval()
    x = rng.standard_normal((B, cin, T)).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    g = rng.standard_normal((B, gin, 1)).astype(np.float32) if gin else None
    noise = rng.standard_normal((B, H, T)).astype(np.float32)
    z, mu, logs = orc.posterior_encoder(sd, x, mask, g, noise, out_channels=H, hidden_channels=H, kernel_size=5, dilation_rate=1,
                                        n_layers=L)
    with torch.no_grad():
        zz, mm, ll = m(cu(x), cu(mask), g=None if g is None else cu(g), noise=cu(noise))
    # (z = mu + noise * exp(logs) with random weights reaches |z| ~ 1e2: error scaled by 1 + |ref|)
    return f"posterior cin{cin} H{H} L{L} gin{gin} B{B} T{T}", max(err(zz, z, True), err(mm, mu, True), err(ll, logs, True)), 5e-5


def case_attention(rng):
    from visinger_amd.modules.rel_transformer import MultiHeadAttention
    nh = int(rng.choice([1, 2, 3, 4]))
    dk = int(rng.choice([4, 16, 31, 32, 64, 96, 128, 160]))
    C = nh * dk
    ws = None if rng.random() < 0.25 else int(rng.choice([1, 2, 4, 7]))
    share = bool(rng.random() < 0.6)
    B, T = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 100, 129, 300]))
    m = MultiHeadAttention(C, C, nh, window_size=ws, heads_share=share)
    sd = rand_sd(m, rng)
    m = m.cuda().eval()
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    if rng.random() < 0.3:
        mask = None
        ref = orc.mha_rel(sd, x, x, None, n_heads=nh, window_size=ws)
    else:
        mask = ragged_mask(rng, B, T)
        ref = orc.mha_rel(sd, x, x, mask, n_heads=nh, window_size=ws)
    with torch.no_grad():
        xx = cu(x)
        y = m(xx, xx, frame_mask=None if mask is None else cu(mask[:, 0]))
    return f"attention nh{nh} dk{dk} ws{ws} share{int(share)} mask{int(mask is not None)} B{B} T{T}", err(y, ref, True), 5e-5


def case_index_ops(rng):
    from visinger_amd.ops import expand_states, make_positions, mel2token_to_dur, slice_segments
    B, Tp, T, H = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.choice([1, 7, 64, 65, 300])), int(rng.choice([1, 5, 192]))
    h = rng.standard_normal((B, Tp, H)).astype(np.float32)
    m2p = rng.integers(0, Tp + 1, (B, T)).astype(np.int64)
    ok = np.array_equal(expand_states(cu(h), cu(m2p)).cpu().numpy(), orc.expand_states(h, m2p))
    x = rng.standard_normal((B, T)).astype(np.float32)
    x[rng.random((B, T)) < 0.4] = 0.0
    ok &= np.array_equal(make_positions(cu(x), 0).cpu().numpy(), orc.make_positions(x, 0))
    ok &= np.array_equal(mel2token_to_dur(cu(m2p), Tp).cpu().numpy(), orc.mel2token_to_dur(m2p, Tp))
    seg = int(rng.integers(1, T + 1))
    z = rng.standard_normal((B, H, T)).astype(np.float32)
    ids = rng.integers(0, T - seg + 1, (B,)).astype(np.int64)
    ok &= np.array_equal(slice_segments(cu(z), cu(ids), seg).cpu().numpy(), orc.slice_segments(z, ids, seg))
    return f"index ops B{B} Tp{Tp} T{T} H{H} seg{seg}", 0.0 if ok else 1.0, 0.5


CASES = [case_wavenet, case_flow, case_generator, case_encoder, case_posterior, case_attention, case_index_ops]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = {}
    for i in range(n):
        fn = CASES[i % len(CASES)]
        desc, e, tol = fn(rng)
        if e > worst.get(fn.__name__, (0, ""))[0]:
            worst[fn.__name__] = (e, desc)
        if e > tol:
            print("FAIL", desc, "err", e, "tol", tol)
            sys.exit(1)
    for k, (e, d) in worst.items():
        print(f"{k:16s} worst abs err {e:.2e}  ({d})")
    print(f"OK {n} cases")


if __name__ == "__main__":
    main()
