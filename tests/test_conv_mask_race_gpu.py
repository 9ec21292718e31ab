"""A masked input transform must be the identity where the mask is 1, and a launch must give the same bits every time.

Round 5 found that the plain-bf16 instances of the TILE kernel (conv_split_kernel<..., 1>, csrc/conv_split_body.inc) violate both at production sizes -- B >= 2
items of T = 4096 frames, two workgroups per CU: with `in_act = VS_IN_MASK` an all-ones mask changed 2-25 % of the outputs by up to 0.7 of their rms, differently
from run to run, always in the staged columns that lanes 32-63 of every second column group wrote (tools/mask_race_probe.py).  The BASELINE configs[4] transformer
convs (masked 1 x 1 projections and FFN convs at hidden 512, reference modules/rel_transformer.py:290-345) ran exactly there in round 4; the bf16 tolerances of the
model-level tests did not see it.  conv_ktap_kernel (csrc/conv_ktap.inc) has no such race and takes over every masked plain-bf16 launch of that configuration:
this file holds the PRODUCTION dispatch to the two properties at those shapes, in both arithmetics, and checks that the config-5 graph never reaches a legacy
plain-bf16 instance with a masked input."""
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
