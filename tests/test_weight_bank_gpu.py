"""weight_bank.WeightBank: every weight of a network folded (weight norm) and packed at the top of a training pass in a handful of launches
(vs_weight_norm_multi_fwd / _bwd, vs_conv_set_weights_batch) -- against the per-module path it replaces:
  * the batched weight norm and its backward == torch._weight_norm / its autograd, per tensor (ragged shapes, a tensor without gradient);
  * handles packed by the batch call produce BIT-IDENTICAL conv outputs to handles packed one by one (forward and ADJOINT handles, transposed
    convs, a <= 4-row conv and a handle of another arithmetic inside the same batch);
  * a full GAN training step with the bank == the step without it (VS_NO_WEIGHT_BANK=1): same losses, same updated parameters to fp32
    rounding of the norms, in far fewer launches."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("per_tensor", [False, True])
def test_batched_weight_norm_matches_torch_autograd(per_tensor):
    from visinger_amd.weight_bank import WeightBank
    torch.manual_seed(0)
    net = torch.nn.ModuleList([
        torch.nn.utils.weight_norm(torch.nn.Conv1d(7, 33, 5)),
        torch.nn.utils.weight_norm(torch.nn.Conv1d(192, 384, 1)),
        torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(64, 32, 16, 8)),
        torch.nn.utils.weight_norm(torch.nn.Conv2d(32, 128, (5, 1))),
        torch.nn.utils.weight_norm(torch.nn.Conv1d(1024, 1, 3)),
        torch.nn.Conv1d(4, 4, 3),                                        # (no weight norm: not the bank's business)
    ]).cuda()
    bank = WeightBank(net, per_tensor_backward=per_tensor)      # (True: the per-tensor backward nodes taken under data parallelism)
    bank.refresh()
    ws = [m.__dict__["_w_eff"] for m in list(net)[:5]]
    cs = [torch.randn_like(w) for w in ws]
    loss = sum((w * c).sum() for w, c in list(zip(ws, cs))[:4])          # the fifth tensor takes no gradient
    loss.backward()
    bank.release()
    assert all("_w_eff" not in m.__dict__ for m in net)
    for i, m in enumerate(list(net)[:5]):
        v, g = m.weight_v.detach().clone().requires_grad_(True), m.weight_g.detach().clone().requires_grad_(True)
        ref = torch._weight_norm(v, g, 0)
        assert float((ws[i].detach() - ref.detach()).abs().max()) <= 2e-6 * float(ref.abs().max())
        if i < 4:
            (ref * cs[i]).sum().backward()
            for got, want in ((m.weight_v.grad, v.grad), (m.weight_g.grad, g.grad)):
                assert got.shape == want.shape
                assert float((got - want).abs().max()) <= 1e-5 * max(1e-6, float(want.abs().max())), i
        else:
            assert m.weight_v.grad is None and m.weight_g.grad is None
    # a second pass reuses the table and the buffers
    st = bank._state
    bank.refresh()
    assert bank._state is st
    bank.release()


def test_batch_pack_is_bit_identical_to_single_packs():
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    torch.manual_seed(1)
    specs = [  # (kind, c_in, c_out, k, dil_or_stride, pad, flags, math)
        (L.CONV1D, 192, 384, 5, 1, 2, 0, L.MATH_SPLIT3),
        (L.CONV1D, 384, 192, 5, 1, 2, L.CONV_ADJOINT, L.MATH_SPLIT3),
        (L.CONV1D, 48, 80, 3, 3, 3, 0, L.MATH_SPLIT3),
        (L.CONV_TRANSPOSE1D, 64, 32, 16, 8, 4, 0, L.MATH_SPLIT3),
        (L.CONV1D, 32, 1, 7, 1, 3, 0, L.MATH_SPLIT3),                   # <= 4 rows: the VALU path's plain pack inside the batch call
        (L.CONV1D, 40, 24, 1, 1, 0, 0, L.MATH_SPLIT6),                  # another arithmetic: plain pack inside the batch call
        (L.CONV1D, 1024, 1024, 5, 1, 2, 0, L.MATH_SPLIT3),              # > 1024 blocks of work: the grid-stride cap
    ]
    jobs, singles, xs = [], [], []
    for kind, ci, co, k, d, p, flags, math in specs:
        adj = bool(flags & L.CONV_ADJOINT)
        wshape = (ci, co, k) if (kind == L.CONV_TRANSPOSE1D or adj) else (co, ci, k)
        w = torch.randn(wshape, device="cuda") * (0.02 if ci > 256 else 0.2)
        b = None if adj else torch.randn(co, device="cuda")
        a, s = ConvOp(kind, ci, co, k, d, p, flags), ConvOp(kind, ci, co, k, d, p, flags)
        a.set_math(math), s.set_math(math)
        jobs.append((a, w, b, ("k", len(jobs))))
        s.set_weights_from(w, b, None)
        singles.append(s)
        xs.append(torch.randn(2, ci, 300, device="cuda"))
    ConvOp.set_weights_batch(jobs)
    for (a, _, _, key), s, x in zip(jobs, singles, xs):
        assert a.has_weights_of(key)
        ya, ys = a.forward(x), s.forward(x)
        assert torch.equal(ya, ys), (a.kind, a.c_in, a.c_out, a.k)
    # and again with new weights (the handles' double-buffered weight maxima alternate)
    jobs2 = [(a, w * 3.0, b, ("k2", i)) for i, (a, w, b, _) in enumerate(jobs)]
    ConvOp.set_weights_batch(jobs2)
    for (a, w, b, _), s, x in zip(jobs2, singles, xs):
        s.set_weights_from(w, b, None)
        assert torch.equal(a.forward(x), s.forward(x))
    ConvOp.set_weights_batch([])
    ConvOp.set_weights_batch(jobs2[:1])


def test_training_step_with_bank_equals_step_without(vs_option):
    from visinger_amd.train import VISingerTrainer, synthetic_train_batch
    hp = json.load(open(os.path.join(GOLDEN, "visinger_tiny_hparams.json")))
    hp = dict(hp, use_pitch_embed=True, pitch_predictor_layers=1, segment_size=8, p_dropout=0.0)

    def run(no_bank):
        vs_option("VS_NO_WEIGHT_BANK", 1 if no_bank else 0)
        torch.manual_seed(0)
        tr = VISingerTrainer(64, 117, 131, hp, dict(fft_size=64, win_size=32, num_mel_bins=16, fmin=0.0, fmax=4000.0, sample_rate=8000))
        tr = tr.cuda().train().configure()
        batch = synthetic_train_batch(2, 48, 6, tr.hop, 64, hp["num_linear_bins"], 1, "cuda")
        g = torch.Generator().manual_seed(3)
        batch["noise_q"] = torch.randn(2, hp["hidden_size"], 48, generator=g).cuda()
        batch["u_slice"] = torch.rand(2, generator=g).cuda()
        logs = tr.training_step(batch)
        # gradients of the second step's generator pass, before the optimizer consumes them
        from visinger_amd.autograd import bump_weight_epoch
        bump_weight_epoch()
        tr.backward_pass(batch, 0)
        grads = {n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None}
        tr.zero_grad(set_to_none=True)
        logs2 = tr.training_step(batch)
        return logs, logs2, grads, {n: p.detach().clone() for n, p in tr.named_parameters()}

    l1, l2, g_b, p_b = run(False)
    m1, m2, g_n, p_n = run(True)
    for a, b in ((l1, m1), (l2, m2)):
        assert set(a) == set(b)
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(b[k])), (k, a[k], b[k])
    assert set(g_b) == set(g_n)
    worst = 0.0
    for n in g_b:
        ref = float(g_n[n].abs().max())
        worst = max(worst, float((g_b[n] - g_n[n]).abs().max()) / max(ref, 1e-3))
    assert worst <= 2e-3, worst
    assert all(torch.isfinite(v).all() for v in p_b.values())
