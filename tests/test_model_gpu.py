"""Model-level GPU parity (the caller of the hot path, reference models/visinger.py:71-112) against the reference's
golden outputs, plus full-size (hidden 192) parity against the fp64 oracle and size-independent properties."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from visinger_amd import _lib as L

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def maxerr(got, ref):
    return float(np.abs(got.detach().cpu().double().numpy() - np.asarray(ref, np.float64)).max())


@pytest.fixture(scope="module")
def tiny():
    from visinger_amd.models.visinger import VISinger
    w, a = load_golden("visinger_tiny")
    hp = json.load(open(os.path.join(GOLDEN, "visinger_tiny_hparams.json")))
    m = VISinger(13, 9, 7, hp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return m.cuda().eval(), a, hp, w


def test_visinger_infer_matches_reference_golden(tiny):
    m, a, hp, _ = tiny
    with torch.no_grad():
        ret = m(cu(a["text"]), cu(a["pitch"]), cu(a["dur"]), cu(a["mel2ph"]), spk_id=cu(a["spk_id"]), infer=True,
                noise=cu(a["noise"]))
    assert ret["wav_out"].shape == a["wav_out"].shape
    assert maxerr(ret["wav_out"], a["wav_out"]) <= 1e-4          # waveform: 1e-4 abs (north star)


def test_visinger_training_forward_matches_reference_golden(tiny):
    m, a, hp, _ = tiny
    with torch.no_grad():
        ret = m(cu(a["text"]), cu(a["pitch"]), cu(a["dur"]), cu(a["mel2ph"]), spk_id=cu(a["spk_id"]), mel=cu(a["lin"]),
                infer=False, noise_q=cu(a["noise_q"]), u_slice=torch.from_numpy(a["u_slice"]))
    assert np.array_equal(ret["ids_slice"].cpu().numpy(), a["t_ids_slice"])      # integer indexing: bit-exact
    assert maxerr(ret["z_p"], a["t_z_p"]) <= 5e-5
    assert abs(float(ret["kl"]) - float(a["t_kl"])) <= 1e-4 * max(1.0, abs(float(a["t_kl"])))
    assert maxerr(ret["ph_pred"], a["t_ph_pred"]) <= 1e-4
    assert maxerr(ret["wav_out"], a["t_wav_out"]) <= 1e-4


def _rand_sd(module, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                fan = max(1, int(np.prod(p.shape[1:])))
                p.copy_(scale * torch.randn(p.shape, generator=g) / np.sqrt(fan))
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def test_full_size_coupling_logdet_1e4_relative(oracle):
    """hidden 192, gin 256, 4 WaveNet layers, mean_only=False: flow log-det within 1e-4 relative of the fp64 oracle."""
    from visinger_amd.modules.visinger.flow import ResidualCouplingLayer
    B, T = 2, 200
    m = ResidualCouplingLayer(192, 192, 5, 1, 4, gin_channels=256, mean_only=False)
    sd = _rand_sd(m, 5, scale=0.5)
    m = m.cuda().eval()
    r = np.random.default_rng(3)
    x = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    mask = np.ones((B, 1, T), np.float32)
    mask[1, :, 150:] = 0
    ref_y, ref_ld = oracle.coupling_layer(sd, x, mask, g, False, channels=192, hidden_channels=192, kernel_size=5,
                                          dilation_rate=1, n_layers=4, mean_only=False)
    with torch.no_grad():
        y, ld = m(cu(x), cu(mask), g=cu(g), reverse=False)
        xr = m(y, cu(mask), g=cu(g), reverse=True)
    assert maxerr(y, ref_y) <= 5e-5
    rel = np.abs(ld.cpu().double().numpy() - ref_ld) / np.abs(ref_ld)
    assert rel.max() <= 1e-4, rel
    assert maxerr(xr[:, 96:] , x[:, 96:] * mask) <= 5e-5            # decode(encode(x)) == x on valid frames


@pytest.mark.parametrize("math", [0, 3, 6])
def test_affine_coupling_logdet_is_run_to_run_bit_identical(oracle, math, vs_option):
    """the log-det of the affine coupling (mean_only=False) is a wavefront-shuffle sum per tile stored in the tile's own
    slot plus a fixed-order pass over the slots (logdet_reduce_kernel) -- no atomics: the same bits every run, on every engine, at a
    length that spans many workgroups (T = 5000: 20+ column tiles x 3 channel pairs per item), ragged mask; and still within 1e-4
    relative of the fp64 oracle."""
    from visinger_amd.modules.visinger.flow import ResidualCouplingLayer
    vs_option("VS_CONV_MATH", math)
    B, T = 3, 5000
    m = ResidualCouplingLayer(192, 192, 5, 1, 4, gin_channels=256, mean_only=False)
    sd = _rand_sd(m, 5, scale=0.5)
    m = m.cuda().eval()
    r = np.random.default_rng(8)
    x = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    mask = np.ones((B, 1, T), np.float32)
    mask[1, :, 3333:] = 0
    mask[2, :, 17:] = 0
    xs, ms, gs = cu(x), cu(mask), cu(g)
    with torch.no_grad():
        runs = [m(xs, ms, g=gs, reverse=False) for _ in range(6)]
    for y, ld in runs[1:]:
        assert torch.equal(ld, runs[0][1]) and torch.equal(y, runs[0][0])
    _, ref_ld = oracle.coupling_layer(sd, x, mask, g, False, channels=192, hidden_channels=192, kernel_size=5, dilation_rate=1, n_layers=4,
                                      mean_only=False)
    rel = np.abs(runs[0][1].cpu().double().numpy() - ref_ld) / np.abs(ref_ld)
    assert rel.max() <= 1e-4, rel


def test_full_size_flow_inverse_and_generator_vs_oracle(oracle):
    """BASELINE config-2 path (flow inverse + HiFi-GAN, hop 256) at hidden 192 on a short clip vs the fp64 oracle."""
    from visinger_amd.modules.visinger.flow import ResidualCouplingBlock
    from visinger_amd.modules.visinger.decoder import Generator
    B, T = 1, 24
    flow = ResidualCouplingBlock(192, 192, 5, 1, 4, gin_channels=256)
    sdf = _rand_sd(flow, 6, scale=0.5)
    gen = Generator(192, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 2, 2], 512, [16, 16, 4, 4], gin_channels=256)
    sdg = _rand_sd(gen, 7)
    flow, gen = flow.cuda().eval(), gen.cuda().eval()
    r = np.random.default_rng(4)
    z = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    mask = np.ones((B, 1, T), np.float32)
    mask[0, :, 20:] = 0
    zq_ref = oracle.flow_block(sdf, z, mask, g, True, channels=192, hidden_channels=192, kernel_size=5, dilation_rate=1,
                               n_layers=4)
    wav_ref = oracle.generator(sdg, zq_ref * mask, g, resblock="1", resblock_kernel_sizes=[3, 7, 11],
                               resblock_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[8, 8, 2, 2],
                               upsample_kernel_sizes=[16, 16, 4, 4])
    with torch.no_grad():
        zq = flow(cu(z), cu(mask), g=cu(g), reverse=True)
        wav = gen(zq * cu(mask), g=cu(g))
    assert maxerr(zq, zq_ref) <= 5e-5
    assert wav.shape == (B, 1, T * 256)
    assert maxerr(wav, wav_ref) <= 1e-4


def test_full_size_generator_reference_config_hop300(oracle):
    """The reference's own generator configuration (config/models/visinger.yaml:23-28: hop 300 = [5,5,3,2,2], kernels
    [11,11,7,4,4], 512 initial channels) at full width vs the fp64 oracle.  Its stride-3 stage has 6 row tiles (192 virtual
    rows): the workgroup that carries the two padding tiles used to skip the chunk barriers (found by tools/conv_fuzz.py)."""
    from visinger_amd.modules.visinger.decoder import Generator
    B, T = 2, 9
    gen = Generator(192, "1", [3, 7, 11], [[1, 3, 5]] * 3, [5, 5, 3, 2, 2], 512, [11, 11, 7, 4, 4], gin_channels=256)
    sdg = _rand_sd(gen, 11)
    gen = gen.cuda().eval()
    r = np.random.default_rng(300)
    z = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    wav_ref = oracle.generator(sdg, z, g, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                               upsample_rates=[5, 5, 3, 2, 2], upsample_kernel_sizes=[11, 11, 7, 4, 4])
    with torch.no_grad():
        wav = gen(cu(z), g=cu(g))
    assert wav.shape == (B, 1, T * 300)
    assert maxerr(wav, wav_ref) <= 1e-4


def test_full_size_generator_error_by_arithmetic(oracle, capsys):
    """Whole hop-256 generator (72 convs deep) at full width vs the fp64 oracle under each arithmetic of the conv engine: the
    split-bf16 x6 default stays within the error of the fp32 MFMA / F(2,3) kernels, bf16 operands do not (what `dtype` in the bench
    line stands for)."""
    from visinger_amd.modules.hipconv import set_conv_math
    from visinger_amd.modules.visinger.decoder import Generator
    B, T = 1, 24
    gen = Generator(192, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 2, 2], 512, [16, 16, 4, 4], gin_channels=256)
    sdg = _rand_sd(gen, 21)
    gen = gen.cuda().eval()
    r = np.random.default_rng(256)
    z = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    wav_ref = oracle.generator(sdg, z, g, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                               upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4])
    err = {}
    for name, math in (("split3", L.MATH_SPLIT3), ("split6", L.MATH_SPLIT6), ("f32", L.MATH_F32), ("bf16", L.MATH_BF16)):
        set_conv_math(gen, math)
        with torch.no_grad():
            wav = gen(cu(z), g=cu(g))
        e = wav.double().cpu().numpy() - wav_ref
        err[name] = (float(np.sqrt((e ** 2).mean())), float(np.abs(e).max()))
    set_conv_math(gen, None)
    with capsys.disabled():
        print("\n   generator waveform error vs fp64 (rms, max): " + "  ".join(f"{k}: {v[0]:.2e}, {v[1]:.2e}" for k, v in err.items()))
    assert err["split6"][1] <= 1e-4 and err["f32"][1] <= 1e-4 and err["split3"][1] <= 1e-4
    assert err["split6"][0] <= 1.5 * err["f32"][0] + 1e-8
    assert err["split3"][0] <= 1.5 * err["f32"][0] + 1e-8          # the default arithmetic (split-f16 x3): within the fp32 MFMA engine's error
    assert err["bf16"][0] > 10 * err["split6"][0]


def test_north_star_shape_properties():
    """B=32, T_mel=1024 (the BASELINE.json size): size-independent checks -- flow encode->decode round trip,
    batch-item independence (what makes the utterance shard exact), masked frames stay zero, finite waveform."""
    from visinger_amd.modules.visinger.flow import ResidualCouplingBlock
    torch.manual_seed(0)
    B, T = 32, 1024
    flow = ResidualCouplingBlock(192, 192, 5, 1, 4, gin_channels=256)
    _rand_sd(flow, 8, scale=0.5)
    flow = flow.cuda().eval()
    x = torch.randn(B, 192, T, device="cuda")
    g = torch.randn(B, 256, 1, device="cuda")
    lens = torch.randint(T // 2, T + 1, (B,), device="cuda")
    mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()[:, None]
    with torch.no_grad():
        y = flow(x, mask, g=g, reverse=False)
        xr = flow(y, mask, g=g, reverse=True)
        y7 = flow(x[7:9].contiguous(), mask[7:9].contiguous(), g=g[7:9].contiguous(), reverse=False)
    assert torch.isfinite(y).all()
    # x0 halves pass through unmasked, x1 halves are masked by every coupling (flow.py:78,83)
    assert float(((xr - x) * mask).abs().max()) <= 2e-4
    assert torch.equal(y[7:9], y7), "batch items must be independent (utterance sharding is exact)"


def test_stream_rotation_gives_the_same_waveforms_as_one_stream(tiny):
    """synth.synthesize(streams=2) -- consecutive batches on alternating HIP streams, the next batch's transformers under this batch's generator -- returns
    bit for bit what one stream returns: the batches share only read-only state.  Many small batches of different shapes, twice (the second pass reuses
    every cached workspace from the other stream's pool)."""
    from visinger_amd import synth
    m, a, hp, _ = tiny
    hop = int(np.prod(hp["upsample_rates"]))
    items = []
    for rep in range(9):
        for b in range(2):
            n = int((a["mel2ph"][b] > 0).sum()) - (rep % 4)
            nph = int((a["text"][b] > 0).sum())
            mel2ph = a["mel2ph"][b][:n]
            items.append(dict(text_tokens=a["text"][b][:nph], pitch_tokens=a["pitch"][b][:nph], dur_tokens=a["dur"][b][:nph], mel2ph=mel2ph))
    budget = 2 * max(len(it["mel2ph"]) for it in items)                       # two or three items a batch: about eight batches
    ref = synth.synthesize(m, items, hop, max_frames_per_batch=budget, generator=torch.Generator(device="cuda").manual_seed(5), streams=1)
    for _ in range(2):
        got = synth.synthesize(m, items, hop, max_frames_per_batch=budget, generator=torch.Generator(device="cuda").manual_seed(5), streams=2)
        assert all(np.array_equal(x, y) for x, y in zip(ref, got))
    assert len(synth.bucket_by_length([len(it["mel2ph"]) for it in items], budget)) >= 6


def test_inference_under_autocast_runs_the_bf16_operand_arithmetic(tiny):
    """the reference's `amp: true` (config/models/base_config.yaml:5, utils/commons/trainer.py:325: autocast around the forward): inside torch.autocast("cuda") the
    inference modules compute with bf16 operands and fp32 accumulation -- bit for bit what set_conv_math(model, L.MATH_BF16) gives wherever no aten matmul sits in
    between (the generator), and the model returns to its own arithmetic afterwards; the training path refuses an autocast region."""
    from visinger_amd import _lib as L
    from visinger_amd.modules.hipconv import set_conv_math
    m, a, hp, _ = tiny
    cu = lambda k: torch.from_numpy(a[k]).cuda()
    g = torch.Generator(device="cuda").manual_seed(9)
    z = torch.randn(2, m.hidden_size, 24, device="cuda", generator=g)
    spk = m.speaker_embedding(None, cu("spk_id")).transpose(1, 2).contiguous()
    with torch.no_grad():
        plain = m.decoder(z, g=spk)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            amp = m.decoder(z, g=spk)
            whole = m(cu("text"), cu("pitch"), cu("dur"), cu("mel2ph"), spk_id=cu("spk_id"), infer=True, noise=cu("noise"))["wav_out"]
        again = m.decoder(z, g=spk)
        set_conv_math(m.decoder, L.MATH_BF16)
        try:
            bf = m.decoder(z, g=spk)
        finally:
            set_conv_math(m.decoder, None)
        last = m.decoder(z, g=spk)
    assert amp.dtype == torch.float32 and torch.equal(amp, bf) and not torch.equal(amp, plain)
    assert torch.equal(again, plain) and torch.equal(last, plain)                    # back on the arithmetic the handles were created with
    assert float((amp - plain).abs().max()) <= 5e-2 * float(plain.abs().max()) + 1e-3
    assert whole.dtype == torch.float32 and bool(torch.isfinite(whole).all())
    m.train()
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with pytest.raises(NotImplementedError, match="autocast"):
                m.decoder(z, g=spk)
    finally:
        m.eval()


def test_batched_synthesis_driver_matches_single_items(tiny):
    """Length-bucketed batching is exact: an utterance synthesised inside a padded batch equals the same utterance
    synthesised alone (batch items are independent, padding is masked) -- what makes the utterance shard exact."""
    from visinger_amd import synth
    m, a, hp, _ = tiny
    hop = int(np.prod(hp["upsample_rates"]))
    items = []
    for b in range(2):
        n = int((a["mel2ph"][b] > 0).sum())
        nph = int((a["text"][b] > 0).sum())
        items.append(dict(text_tokens=a["text"][b][:nph], pitch_tokens=a["pitch"][b][:nph], dur_tokens=a["dur"][b][:nph],
                          mel2ph=a["mel2ph"][b][:n]))
    assert synth.bucket_by_length([5, 9, 2, 9], 18) == [[1, 3], [0, 2]]
    g = torch.Generator(device="cuda").manual_seed(0)
    both = synth.synthesize(m, items, hop, generator=g)
    assert [len(w) for w in both] == [int((a["mel2ph"][b] > 0).sum()) * hop for b in range(2)]
    # same noise for the longer item when run alone (it is first in its bucket, so it consumed the first draws)
    g = torch.Generator(device="cuda").manual_seed(0)
    T0 = len(items[0]["mel2ph"])
    noise = torch.randn((2, m.hidden_size, T0), device="cuda", generator=g)[:1]
    batch = synth.collate(items[:1], "cuda")
    with torch.no_grad():
        alone = m(batch["text_tokens"], batch["pitch_tokens"], batch["dur_tokens"], batch["mel2ph"], spk_id=batch["spk_id"],
                  infer=True, noise=noise)["wav_out"][0].cpu().numpy()
    assert np.abs(alone - both[0]).max() <= 1e-6
    # A SHORTER item in a batch of padded frames: its waveform -- tail included -- must equal its one-at-a-time synthesis
    # (tasks/visinger.py:244-263) on the same noise; the generator runs with the frame mask at every stage for that.  (Same token
    # sequence, fewer frames: the reference's TextEncoder views its positional table by the PADDED token length, encoder.py:52-54,
    # so only items of equal token count can agree with their standalone run at all -- synth.synthesize(equal_tokens=True).)
    T1 = T0 - 5
    short = dict(items[0], mel2ph=items[0]["mel2ph"][:T1])
    pair = [items[0], short]
    g = torch.Generator(device="cuda").manual_seed(0)
    both2 = synth.synthesize(m, pair, hop, generator=g, equal_tokens=True)
    assert len(both2[1]) == T1 * hop and np.abs(both2[0] - both[0]).max() <= 1e-6
    g = torch.Generator(device="cuda").manual_seed(0)
    nz = torch.randn((2, m.hidden_size, T0), device="cuda", generator=g)
    batch1 = synth.collate([short], "cuda")
    with torch.no_grad():
        alone1 = m(batch1["text_tokens"], batch1["pitch_tokens"], batch1["dur_tokens"], batch1["mel2ph"], spk_id=batch1["spk_id"],
                   infer=True, noise=nz[1:2, :, :T1].contiguous())["wav_out"][0].cpu().numpy()
        # and WITHOUT the mask the padded batch differs in the tail (what the unmasked generator of a padded batch gives)
        pb = synth.collate(pair, "cuda")
        unmasked = m(pb["text_tokens"], pb["pitch_tokens"], pb["dur_tokens"], pb["mel2ph"], spk_id=pb["spk_id"], infer=True,
                     noise=nz)["wav_out"][1, :T1 * hop].cpu().numpy()
    assert alone1.shape == both2[1].shape and np.abs(alone1 - both2[1]).max() <= 2e-6
    assert np.abs(unmasked - alone1).max() > 1e-5      # the leak the mask removes (conv_pre's bias + speaker condition in the padding)
    # graph-replayed batches (synth.GraphedStep): first call captures each batch shape, the second replays; same waveforms
    graphs = {}
    for _ in range(2):
        g = torch.Generator(device="cuda").manual_seed(0)
        replayed = synth.synthesize(m, pair, hop, generator=g, equal_tokens=True, graphs=graphs)
        assert all(np.array_equal(a_, b_) for a_, b_ in zip(replayed, both2))
    assert len(graphs) == 1
    pcm = synth.to_int16(both[0])
    assert pcm.dtype == np.int16 and np.abs(pcm).max() == 32767


def test_synthesis_step_is_graph_capturable(tiny):
    """The whole synthesis step (every ctypes launch goes to torch's current stream, no allocation through hipMalloc and no
    host synchronisation after warm-up) can be captured into a HIP graph and replayed bit-identically -- what a serving loop
    with fixed shapes would do (tools/graph_capture_check.py times it at the benchmark size)."""
    model, a, _, _ = tiny
    args = [cu(a[k]) for k in ("text", "pitch", "dur", "mel2ph")]
    spk, noise = cu(a["spk_id"]), cu(a["noise"])

    def step():
        with torch.no_grad():
            return model(*args, spk_id=spk, infer=True, noise=noise)["wav_out"]

    ref = step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_synthesis_is_run_to_run_deterministic(tiny):
    """No atomics and no data-dependent scheduling on the synthesis path: the same inputs give the same bits, run after run
    (the flow's forward log-det has no atomics either: test_affine_coupling_logdet_is_run_to_run_bit_identical)."""
    model, a, _, _ = tiny
    args = [cu(a[k]) for k in ("text", "pitch", "dur", "mel2ph")]
    spk, noise = cu(a["spk_id"]), cu(a["noise"])
    with torch.no_grad():
        runs = [model(*args, spk_id=spk, infer=True, noise=noise)["wav_out"].clone() for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])


@pytest.mark.parametrize("hop", [256, 300])
def test_full_size_model_end_to_end_vs_oracle(oracle, hop):
    """The whole synthesis graph at the production width (hidden 192, 6 + 4 transformer layers, 4 couplings, 512-channel
    generator; hop-256 benchmark variant and the reference's hop 300), ragged batch of two short clips, against the fp64 oracle
    end to end: prior statistics, latent after the flow inverse, waveform."""
    from visinger_amd.models.visinger import REFERENCE_HPARAMS, VISinger, hop256_hparams
    hp = dict(hop256_hparams() if hop == 256 else REFERENCE_HPARAMS, use_pitch_embed=False)      # the oracle's runnable configuration
    torch.manual_seed(11)
    m = VISinger(40, 50, 60, hp)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for name, p_ in m.named_parameters():
            if name.endswith("weight_g"):
                p_.copy_(0.5 + torch.rand(p_.shape, generator=g))
            elif ".post." in name:                       # zero-initialised in the reference: make the flow non-trivial
                p_.copy_(0.05 * torch.randn(p_.shape, generator=g))
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    B, Tph, T = 2, 5, 14
    text = torch.randint(4, 40, (B, Tph), generator=g)
    pitch = torch.randint(1, 50, (B, Tph), generator=g)
    dur = torch.randint(4, 60, (B, Tph), generator=g)
    mel2ph = torch.tensor([[1, 1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5], [1, 1, 2, 2, 2, 3, 4, 4, 5, 0, 0, 0, 0, 0]])
    spk = torch.zeros(B, dtype=torch.long)
    noise = torch.randn(B, 192, T, generator=g)
    ref = oracle.visinger_infer(sd, hp, text.numpy(), pitch.numpy(), dur.numpy(), mel2ph.numpy(), spk.numpy(), noise.numpy(),
                                return_all=True)
    with torch.no_grad():
        wav = m(text.cuda(), pitch.cuda(), dur.cuda(), mel2ph.cuda(), spk_id=spk.cuda(), infer=True, noise=noise.cuda())["wav_out"]
    assert wav.shape == (B, T * hop)
    assert maxerr(wav, ref["wav_out"]) <= 1e-4


@pytest.mark.parametrize("rates,kernels", [([8, 8, 2, 2], [16, 16, 4, 4]), ([5, 5, 3, 2, 2], [11, 11, 7, 4, 4])])
def test_generator_with_bf16_resident_activations(oracle, capsys, rates, kernels):
    """BASELINE config 5 ("bf16 activations"): the generator with every tensor between conv_pre and conv_post held as bf16
    (hipconv.set_activation_storage + L.MATH_BF16) against the fp64 oracle and against the same arithmetic on fp32 tensors: the extra
    error of the bf16 residual stream stays within a small multiple of the arithmetic's own, and the waveform stays within the stated
    bf16 tolerance of the config-5 test (rms 2e-2 of the signal rms)."""
    from visinger_amd.modules.hipconv import set_activation_storage, set_conv_math
    from visinger_amd.modules.visinger.decoder import Generator
    B, T = 2, 24
    # (hop 300, the reference's own generator: it ends in a 16-channel stage, which has no bf16 instance -- the tensors go back to fp32 at
    #  the 128 -> 64 transposed conv)
    gen = Generator(192, "1", [3, 7, 11], [[1, 3, 5]] * 3, rates, 512, kernels, gin_channels=256)
    sdg = _rand_sd(gen, 21)
    gen = gen.cuda().eval()
    r = np.random.default_rng(257)
    z = r.standard_normal((B, 192, T)).astype(np.float32)
    g = r.standard_normal((B, 256, 1)).astype(np.float32)
    wav_ref = oracle.generator(sdg, z, g, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                               upsample_rates=rates, upsample_kernel_sizes=kernels)
    set_activation_storage(gen, torch.bfloat16)
    with pytest.raises(L.VisingerHipError):            # not with the fp32-class arithmetic
        with torch.no_grad():
            gen(cu(z), g=cu(g))
    set_conv_math(gen, L.MATH_BF16)
    xm = torch.ones(B, 1, T, device="cuda")
    xm[1, :, 17:] = 0                                      # a padded item: the masked (unfused) convs of the narrow stages on bf16 tensors
    with torch.no_grad():
        wav_b = gen(cu(z), g=cu(g))
        wav_bm = gen(cu(z), g=cu(g), x_mask=xm)
        set_activation_storage(gen, None)
        wav_f = gen(cu(z), g=cu(g))
        wav_fm = gen(cu(z), g=cu(g), x_mask=xm)
    set_conv_math(gen, None)
    hop = wav_b.shape[-1] // T
    assert bool(torch.isfinite(wav_bm).all()) and float(wav_bm[1, ..., 17 * hop:].abs().max()) == 0.0
    assert float((wav_bm - wav_fm).pow(2).mean().sqrt()) <= 1e-2 * float(wav_fm.pow(2).mean().sqrt())
    assert wav_b.dtype == torch.float32 and wav_b.shape == wav_f.shape and bool(torch.isfinite(wav_b).all())
    rms = float(np.sqrt((wav_ref ** 2).mean()))
    eb = float(np.sqrt(((wav_b.double().cpu().numpy() - wav_ref) ** 2).mean()))
    ef = float(np.sqrt(((wav_f.double().cpu().numpy() - wav_ref) ** 2).mean()))
    with capsys.disabled():
        print(f"\n   bf16 arithmetic, waveform rms error vs fp64 / signal rms: fp32 tensors {ef / rms:.2e}, bf16-resident activations {eb / rms:.2e}")
    assert eb <= 2e-2 * rms and eb <= 4 * ef + 1e-6
