"""nn.Conv1d / nn.ConvTranspose1d whose forward runs on the HIP conv engine (C ABI vs_conv_*).

They ARE torch.nn.Conv1d / ConvTranspose1d subclasses, so construction, default initialisation,
``torch.nn.utils.weight_norm`` (old-style ``weight_g`` / ``weight_v`` parameters), ``remove_weight_norm`` and the
``state_dict`` keys are exactly the reference's; only the arithmetic moves to gfx950.  Parent modules call
``run(...)`` to fuse their elementwise neighbours (activation, mask, residual, gate, coupling) into the conv.

Training mode (autograd enabled) is served by visinger_amd.autograd: HIP forward for the convs, PyTorch-ROCm backward.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from ..ops import ConvOp


def _forward_only_guard(module):
    """The fused inference kernels have no backward: reaching them in training mode with autograd on is a bug in the
    dispatch (every module routes that case to visinger_amd.autograd first)."""
    if torch.is_grad_enabled() and module.training:
        raise NotImplementedError(
            "fused HIP inference path reached in training mode with autograd enabled; use the module's forward() "
            "(which dispatches to visinger_amd.autograd) or call .eval() / torch.no_grad().")


class _HipConvMixin:
    _kind = L.CONV1D

    def _weights(self):
        if hasattr(self, "weight_g"):           # torch.nn.utils.weight_norm applied
            return self.weight_v, self.weight_g
        return self.weight, None

    def _op(self, kind=None, flags=0, bind=True):
        """ConvOp for (kind, flags); created lazily so that modules can be built/loaded without a GPU.
        bind=False returns the handle without (re)packing the module's parameters (the autograd path packs the live
        folded weight itself)."""
        kind = self._kind if kind is None else kind
        ops = self.__dict__.setdefault("_hip_ops", {})
        key = (kind, flags)
        if key not in ops:
            if self._kind == L.CONV_TRANSPOSE1D:
                ops[key] = ConvOp(L.CONV_TRANSPOSE1D, self.in_channels, self.out_channels, self.kernel_size[0],
                                  self.stride[0], self.padding[0], flags)
            else:
                assert self.stride[0] == 1 and self.groups == 1, "HIP conv engine: stride-1 ungrouped convs only"
                ops[key] = ConvOp(kind, self.in_channels, self.out_channels, self.kernel_size[0],
                                  self.dilation[0], self.padding[0], flags)
        op = ops[key]
        math = self.__dict__.get("_hip_math")
        if math is not None and op.math != math:
            op.set_math(math)
        if bind:
            w, g = self._weights()
            op.set_weights(w, g, self.bias)
        return op

    def run(self, x, *, kind=None, flags=0, **kw):
        """Launch the conv with fused options (see visinger_amd.ops.ConvOp.forward)."""
        _forward_only_guard(self)
        return self._op(kind, flags).forward(x, **kw)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            from ..autograd import conv
            return conv(self, x)
        return self.run(x.contiguous().float())

    def __getstate__(self):                      # handles are process-local (forward, backward-data and discriminator caches)
        return drop_process_local_state(self.__dict__.copy())


class HipConv1d(_HipConvMixin, nn.Conv1d):
    _kind = L.CONV1D


class HipConvTranspose1d(_HipConvMixin, nn.ConvTranspose1d):
    _kind = L.CONV_TRANSPOSE1D


def set_conv_math(module, math):
    """Select the arithmetic of every HIP conv under `module` (L.MATH_SPLIT6: fp32-class split-bf16, the default;
    L.MATH_F32: fp32 MFMA / Winograd F(2,3); L.MATH_BF16: bf16 operands, fp32 accumulate -- BASELINE.json's long-form bf16
    configuration).  None restores the library default for handles created afterwards."""
    for m in module.modules():
        if isinstance(m, _HipConvMixin):
            if math is None:
                m.__dict__.pop("_hip_math", None)
            else:
                m.__dict__["_hip_math"] = int(math)
    return module


def set_activation_storage(module, dtype):
    """bf16-RESIDENT activations (torch.bfloat16; None / torch.float32 restores fp32 tensors) for the modules under `module` that
    support them: the generator keeps the activations of its wide stages (>= 128 channels) as bf16 tensors between its convs --
    BASELINE.json's long-form configuration ("bf16 activations"), only together with L.MATH_BF16 (the arithmetic that rounds every
    conv operand to bf16 anyway).  On the bf16 matrix pipe those convs are HBM-bound with fp32 tensors."""
    for m in module.modules():
        if dtype is None or dtype == torch.float32:
            m.__dict__.pop("_hip_storage", None)
        else:
            assert dtype == torch.bfloat16, dtype
            m.__dict__["_hip_storage"] = dtype
    return module


def repack_weights(module):
    """Drop every packed-weight cache under `module`: the next forward re-folds and re-packs from the live parameters.  The
    cache key (data_ptr, in-place version) of visinger_amd.ops.ConvOp.set_weights follows optimizer steps, ``load_state_dict``
    and ``copy_`` on the parameter itself, but NOT edits made through ``p.data`` (EMA swaps, manual re-initialisation: `.data`
    carries its own version counter) -- call this after such an edit."""
    for m in module.modules():
        for key in ("_hip_ops", "_hip_bwd_ops", "_hip_disc_ops"):
            for op in m.__dict__.get(key, {}).values():
                op.invalidate()
        for key in DERIVED_CACHES:      # caches keyed on (data_ptr, _version) of SEVERAL parameters: rebuilt from the live ones
            m.__dict__.pop(key, None)
    return module


# per-module caches derived from more than one parameter (MultiHeadAttention: the fused q | k | v projection of the inference path,
# its training twin): dropped by repack_weights and never pickled
DERIVED_CACHES = ("_hip_qkv_inf", "_hip_qkv")


def drop_process_local_state(state):
    """__getstate__ helper: remove conv handles and derived caches (device copies, ctypes handles) from a module's __dict__ copy"""
    for key in ("_hip_ops", "_hip_bwd_ops", "_hip_disc_ops") + DERIVED_CACHES:
        state.pop(key, None)
    return state


def mask2d(x_mask, B, T):
    """[B,1,T] (or [B,T]) nonpadding mask -> contiguous fp32 [B,T]"""
    if x_mask is None:
        return None
    return x_mask.reshape(B, T).to(torch.float32).contiguous()
