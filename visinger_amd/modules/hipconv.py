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


def apply_math(holder, op):
    """Give handle `op` the arithmetic its module should run NOW: the module's explicit choice (set_conv_math); else, inside a `torch.autocast("cuda")` region,
    bf16 operands with fp32 accumulation (L.MATH_BF16) -- the counterpart of the reference's `amp: true` (config/models/base_config.yaml:5,
    utils/commons/trainer.py:325: its convs and matmuls then run in half precision; here the tensors stay fp32 and only the matrix-product operands are rounded,
    with bfloat16's exponent range); else the arithmetic the handle was created with.  A handle re-packs its planes when its arithmetic changes, so a model that
    alternates between autocast and full precision pays one re-pack per conv per switch.  INFERENCE modules only: the training path (visinger_amd.autograd)
    refuses an autocast region (its Functions exchange fp32 tensors with aten ops that autocast would hand bf16 tensors to)."""
    math = holder.get("_hip_math")
    if math is None:
        base = op.__dict__.get("_base_math")
        if base is None:
            base = op.__dict__["_base_math"] = op.math
        math = L.MATH_BF16 if torch.is_autocast_enabled("cuda") else base
    if op.math != math:
        op.set_math(math)


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
        apply_math(self.__dict__, op)
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

    def _load_from_state_dict(self, *args, **kwargs):
        """nn.Module's loader for this conv's own tensors, then the weight-range check of INTEGRATION.md 4 on what was loaded (round 6): a real checkpoint whose
        rows lie further apart than one power-of-two scale per conv can carry runs this conv on the exact bf16 x3 split without the user having to ask."""
        super()._load_from_state_dict(*args, **kwargs)
        auto_select_math(self)


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
            m.__dict__.pop("_hip_math_auto", None)         # (an explicit choice replaces one the load-time weight-range check made)
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


def weight_row_drop_bits(conv):
    """How far the weakest OUTPUT ROW of a HIP conv sits below the conv's largest weight, in bits: log2(max |w| / min over rows of the row's max |w|) of the
    effective (weight-norm-folded) weight; all-zero rows do not count.  The split-f16 arithmetic (L.MATH_SPLIT3) packs a conv's weights under ONE power-of-two
    scale: a row d bits below the largest weight keeps its own largest element to 22 bits while d <= 18, but an element another c bits below that only to
    40 - d - c bits (absolute floor 2^-25 of the scaled planes) -- INTEGRATION.md 4."""
    w, g = conv._weights()
    w = w.detach().float()
    if g is not None:       # torch.nn.utils.weight_norm, dim 0: w = g * v / ||v|| over every dim but 0
        w = w * (g.detach().float().reshape(-1, 1, 1) / w.flatten(1).norm(dim=1).clamp_min(1e-30).reshape(-1, 1, 1))
    out_dim = 1 if isinstance(conv, nn.ConvTranspose1d) else 0
    row_max = w.abs().amax(dim=[d for d in range(3) if d != out_dim])
    row_max = row_max[row_max > 0]
    if row_max.numel() == 0:
        return 0.0
    return float(torch.log2(row_max.max() / row_max.min()))


def auto_select_math(conv, max_row_drop_bits=10.0):
    """The per-conv form of select_math_by_weight_range, run by every HIP conv when a state dict is loaded into it.  An arithmetic chosen by the user
    (set_conv_math) is left alone; a choice this function made earlier is re-made from the new weights."""
    d = conv.__dict__
    if "_hip_math" in d and not d.get("_hip_math_auto"):
        return
    try:
        default = int(L.get_option("VS_CONV_MATH"))
        drop = weight_row_drop_bits(conv)
    except Exception:                      # (no library on this machine / parameters not materialised: nothing to decide at load time)
        return
    if default not in (-1, L.MATH_SPLIT3):
        return
    if drop > max_row_drop_bits:
        d["_hip_math"], d["_hip_math_auto"] = int(L.MATH_SPLIT6), True
    elif d.pop("_hip_math_auto", False):
        d.pop("_hip_math", None)


def select_math_by_weight_range(module, max_row_drop_bits=10.0, fallback=None):
    """Load-time check for real checkpoints.  `load_state_dict` runs it by itself, conv by conv (auto_select_math); call this after editing parameters in place, or
    with another bound (it reads the weights on the host side of the stream: a synchronisation).
    Every HIP conv under `module` whose weakest output row sits more than `max_row_drop_bits` below its largest weight is switched from the split-f16
    arithmetic to `fallback` (default L.MATH_SPLIT6: three exact bf16 planes, no scale, any dynamic range, twice the matrix work).  With the default bound a
    weight 2^-13 below its ROW's largest still carries 17 significant bits.  Returns [(qualified module name, drop in bits)] of the convs it switched.
    Random-init and weight-normed HiFi-GAN / WaveNet checkpoints sit at 1-6 bits; tests/test_split_robustness_gpu.py drives whole generators with per-channel
    gains over 2^-10 .. 2^10 through it."""
    fallback = L.MATH_SPLIT6 if fallback is None else fallback
    library_default = int(L.get_option("VS_CONV_MATH"))          # (-1: no override -> new handles take the split-f16 arithmetic, csrc/conv_engine.hip vs_conv_create)
    if library_default < 0:
        library_default = L.MATH_SPLIT3
    switched = []
    for name, m in module.named_modules():
        if isinstance(m, _HipConvMixin):
            cur = m.__dict__.get("_hip_math", library_default)
            if cur != L.MATH_SPLIT3:
                continue
            drop = weight_row_drop_bits(m)
            if drop > max_row_drop_bits:
                m.__dict__["_hip_math"] = int(fallback)
                switched.append((name, drop))
    return switched


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
