"""A masked input transform must be the identity where the mask is 1, must equal the conv of the pre-masked input bit for bit, and a launch must give the same
bits every time.

Round 5 found that the plain-bf16 instances of the TILE kernel (conv_split_kernel<..., 1>, csrc/conv_split_body.inc) violate this at production sizes -- two
workgroups per CU: with `in_act = VS_IN_MASK` an all-ones mask changed 2-25 % of the outputs by up to 0.7 of their rms, differently from run to run, always in
lanes 48-63 of a staged column group (tools/mask_race_probe.py).  Round 6 root-caused it (tools/mask_race_probe2.py, tools/ubench/pk_opsel_probe.hip, DESIGN.md
4.5): NOT a race in the kernel's protocol but an instruction form -- hipcc's SLP vectoriser had paired `value * mask` into
`v_pk_mul_f32 d, a, m op_sel:[0,1] op_sel_hi:[1,0]`, and on gfx950 a packed-fp32 instruction whose low result reads the HIGH register of its second source returns
wrong low results in lanes 48-63 while the other wave of its SIMD issues MFMAs (standalone: 0.1-0.5 % of such instructions, none with one wave per SIMD, none for
any other op_sel form).  The multiply is an opaque scalar instruction now (conv_common.h: mul_f32_scalar) and csrc/build.py fails the build on any such
instruction in any object.  This file holds BOTH dispatches -- conv_ktap_kernel and, under VS_NO_KTAP=1, the legacy tile kernel -- to the three properties at the
shapes where it showed, plus tap counts / channel counts that only the tile kernel serves (k = 5, k = 4, C_in = 200), in both arithmetics, with all-ones and
with ragged masks (the split-f16 path multiplied IN PLACE: there a dropped lane left the unmasked value behind, which an all-ones mask cannot see)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [   # C_in, C_out, k, B, T
    (512, 1536, 1, 2, 4096),      # fused q | k | v at hidden 512 (two 128 x 256 workgroups per CU)
    (512, 1536, 1, 8, 4096),      # ... at the benched batch
    (512, 2048, 9, 2, 4096),      # FFN conv_1
    (2048, 512, 1, 8, 4096),      # FFN conv_2
    (2048, 512, 1, 8, 512),       # text encoder (T_ph): 32 x 128 tiles
    (512, 2048, 9, 8, 512),
    (192, 768, 9, 32, 1024),      # the headline's width
    (768, 192, 1, 32, 1024),
]


LEGACY_ONLY_SHAPES = [      # no conv_ktap instance: the tile kernel serves these in production too
    (512, 1536, 5, 2, 4096),      # k = 5 on unpaired rows
    (512, 1536, 4, 2, 4096),      # an even tap count
    (200, 1536, 1, 2, 4096),      # C_in not a multiple of 16 (the last chunk is partly padding)
]


@pytest.mark.parametrize("math", [1, 3], ids=["bf16", "split3"])
@pytest.mark.parametrize("noktap", [0, 1], ids=["ktap", "legacy"])
@pytest.mark.parametrize("shape", SHAPES + LEGACY_ONLY_SHAPES, ids=lambda s: "c%d-%d_k%d_B%d_T%d" % s)
def test_masked_launch_equals_the_conv_of_the_premasked_input(shape, math, noktap, vs_option):
    """y(x, IN_MASK, m) == y(x * m, IN_NONE) bit for bit for a ragged 0 / 1 mask (x * 1 = x and x * 0 = 0 exactly, so the two launches stage identical
    values): a lane whose masked value was dropped shows up as the un-masked x in a padded frame, leaking into the valid outputs through the taps.  Four runs,
    other work in between (different co-residency every time)."""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, B, T = shape
    vs_option("VS_CONV_MATH", math)
    vs_option("VS_NO_KTAP", noktap)
    vs_option("VS_NO_SMALL_GRID", 1)
    g = torch.Generator(device="cuda").manual_seed(23 + cin + k)
    op = ConvOp(L.CONV1D, cin, cout, k, 1, k // 2)
    op.set_weights(torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5, None, torch.randn(cout, device="cuda", generator=g) * 0.1)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    lens = torch.randint(T // 3, T, (B,), device="cuda", generator=g)
    lens[0] = T
    mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
    want = op.forward((x * mask[:, None]).contiguous(), in_act=L.IN_NONE).clone()
    for _ in range(4):
        got = op.forward(x, in_act=L.IN_MASK, mask=mask).clone()
        name = op.kernel_instance()
        op.forward(torch.randn_like(x), in_act=L.IN_MASK, mask=mask)
        assert torch.equal(got, want), (name, float((got - want).abs().max()), int((got != want).sum()))
    if noktap:
        assert not name.startswith("conv_ktap_kernel<"), name
    ones = torch.ones_like(mask)
    assert torch.equal(op.forward(x, in_act=L.IN_MASK, mask=ones), op.forward(x, in_act=L.IN_NONE)), name


@pytest.mark.parametrize("math", [1, 3], ids=["bf16", "split3"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "c%d-%d_k%d_B%d_T%d" % s)
def test_all_ones_mask_is_the_identity_and_launches_are_deterministic(shape, math, vs_option):
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, B, T = shape
    vs_option("VS_CONV_MATH", math)
    g = torch.Generator(device="cuda").manual_seed(17 + cin + k)
    op = ConvOp(L.CONV1D, cin, cout, k, 1, k // 2)
    op.set_weights(torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5, None, torch.randn(cout, device="cuda", generator=g) * 0.1)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    mask = torch.ones(B, T, device="cuda")
    plain = op.forward(x, in_act=L.IN_NONE).clone()
    runs = []
    for _ in range(4):
        runs.append(op.forward(x, in_act=L.IN_MASK, mask=mask).clone())
        x2 = torch.randn_like(x)                                   # (other work in between: different co-residency every time)
        op.forward(x2, in_act=L.IN_MASK, mask=mask)
    name = op.kernel_instance()
    if math == 1:
        assert name.startswith("conv_ktap_kernel<"), name          # plain bf16: the production dispatch of these shapes must not be a legacy instance
    for y in runs:
        assert torch.equal(y, runs[0]), (name, float((y - runs[0]).abs().max()))
    assert torch.equal(runs[0], plain), (name, float((runs[0] - plain).abs().max()), int((runs[0] != plain).sum()))


def test_config5_graph_reaches_no_legacy_bf16_instance_with_a_masked_input(vs_option):
    """the whole BASELINE configs[4] synthesis graph (hidden 512, T_mel 4096, plain bf16 + bf16-resident activations), B = 2: every conv launch whose input
    transform carries the frame mask must be a conv_ktap instance"""
    import bench
    from visinger_amd import _lib as L
    from visinger_amd import ops
    from visinger_amd.modules.hipconv import set_activation_storage, set_conv_math
    model, hp = bench.build_model(hidden=512)
    model = model.cuda()
    B, T = 2, 4096
    batch = [t.cuda() for t in bench.synthetic_batch(B, T, T // 8, 64, 77, "cpu", hidden=512)]
    text, pitch, dur, mel2ph, spk, noise = batch
    seen = []
    orig = ops.ConvOp.forward

    def forward(self, x, *a, **kw):
        y = orig(self, x, *a, **kw)
        seen.append((self.kernel_instance(), int(kw.get("in_act", 0)), self.math))
        return y

    set_conv_math(model, L.MATH_BF16)
    set_activation_storage(model, torch.bfloat16)
    ops.ConvOp.forward = forward
    try:
        with torch.no_grad():
            out = model(text, pitch, dur, mel2ph, spk_id=spk, infer=True, noise=noise)
        torch.cuda.synchronize()
    finally:
        ops.ConvOp.forward = orig
        set_activation_storage(model, None)
        set_conv_math(model, None)
    assert bool(torch.isfinite(out["wav_out"]).all())
    masked = [(n, a) for n, a, m in seen if a in (L.IN_MASK, L.IN_LRELU_MASK) and m == L.MATH_BF16]
    assert masked, "the transformers' convs carry the frame mask"
    legacy = sorted({n for n, a in masked if not n.startswith("conv_ktap_kernel<")})
    assert not legacy, legacy
