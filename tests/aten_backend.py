"""TEST-ONLY reference backend for the training path: the SAME module graph (visinger_amd.autograd's compositions of the reference's forward()s)
with stock PyTorch-ROCm aten kernels in place of every HIP kernel -- F.conv1d / F.conv_transpose1d for the convs, the torch formulations of the gate,
the channel LayerNorm and the [T, T] attention core -- so that torch autograd gives a reference gradient for every parameter at sizes where the
reference itself cannot run on the GPU box (it cannot travel).  Round 5: this used to be a switch inside the product package
(VS_TRAIN_ATEN in visinger_amd/autograd.py); a stock-aten execution backend does not belong there, so the tests patch it in from here.
Only tests import this file."""
import contextlib

import torch.nn.functional as F


@contextlib.contextmanager
def aten_backend():
    from visinger_amd import _lib as L
    from visinger_amd import autograd as A
    from visinger_amd.modules import discriminator as D

    def conv(m, x, lrelu=False, res=None):
        """reference modules' Conv1d / ConvTranspose1d under weight norm (e.g. modules/visinger/decoder.py:24, 72-87) on aten"""
        w = A.effective_weight(m)
        if lrelu:
            x = F.leaky_relu(x, A.LRELU_SLOPE)
        if m._kind == L.CONV_TRANSPOSE1D:
            y = F.conv_transpose1d(x, w, m.bias, stride=m.stride[0], padding=m.padding[0])
        else:
            y = F.conv1d(x, w, m.bias, padding=m.padding[0], dilation=m.dilation[0])
        return y if res is None else y + res

    def disc_conv1d(holder, x, w, b, stride, pad, groups=1):
        """modules/discriminator.py:28-47, 64-75: the discriminators' strided / grouped convs on aten"""
        return F.conv1d(x.float(), w, b, stride=stride, padding=pad, groups=groups)

    saved = (A.conv, A.disc_conv1d, D.disc_conv1d)
    switches = {n: L.get_option(n) for n in ("VS_NO_TRAIN_FUSED", "VS_NO_TRAIN_ATTN")}
    A.conv, A.disc_conv1d, D.disc_conv1d = conv, disc_conv1d, disc_conv1d
    for n in switches:
        L.set_option(n, 1)          # gate, LayerNorm (+ residual) and the attention core as differentiable torch expressions
    try:
        yield
    finally:
        A.conv, A.disc_conv1d, D.disc_conv1d = saved
        for n, v in switches.items():
            L.set_option(n, v)
