"""Training-mode (autograd) execution of the hot-path modules -- SURVEY.md 8f-1, first stage.

Forward: every convolution still runs on the HIP conv engine (``HipConvFn``: the folded weight is re-packed each step
because it changes each step); the elementwise neighbours that the inference path fuses into the conv epilogues
(gates, residuals, masks, coupling updates, LayerNorm) and the attention core run as PyTorch-ROCm ops so that
autograd records them.  Backward: the conv grad-input runs on the HIP conv engine again (`conv_backward`: correlation
with the reversed / transposed weight; transposed convs through their de-interleaved phases), the conv grad-weight on
vs_conv_wgrad (csrc/conv_backward.hip), everything else is PyTorch-ROCm autograd -- no MIOpen on the generator side.  The arithmetic restated here follows the same reference lines as the
inference modules; `tests/test_train_gpu.py` checks train-mode forward == eval-mode (fused HIP) forward and the
gradients against the reference's own autograd (golden vectors).
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

from . import _lib as L
from .ops import PROFILER, ConvOp, bias_grad, conv_wgrad, gconv1d_bwd_data, gconv1d_bwd_weight, gconv1d_fwd

LRELU_SLOPE = 0.1


_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Invalidate every packed-weight key at once (the epoch is part of each key).  (data_ptr, _version) follows optimizer steps,
    load_state_dict and copy_ on the parameter, but NOT edits made through ``p.data`` (EMA swaps, ``p.data.clamp_``: `.data` carries
    its own version counter).  VISingerTrainer.training_step bumps the epoch once per step, so such an edit between two steps costs one
    re-pack instead of silently training on stale packed weights; inside a step the cache still serves the frozen network's second use."""
    _WEIGHT_EPOCH[0] += 1


def param_key(holder):
    """Identity and in-place versions of the PARAMETERS the live weight / bias of a conv holder derive from: (weight_v, weight_g) under
    torch.nn.utils.weight_norm, else the plain weight parameter; None when the weight is not a function of parameters alone (spectral
    norm runs a power iteration per forward).  Optimizer steps, load_state_dict and every other in-place write bump the versions; a
    handle whose packed weights carry the same key need not be packed again (ConvOp.has_weights_of)."""
    if L.switch("VS_NO_PACK_CACHE"):
        return None
    sources = holder.__dict__.get("_key_sources")
    if sources is not None:                   # a fused projection (_ConvHolder): the concatenation of several modules' weights
        keys = tuple(param_key(m_) for m_ in sources)
        return None if any(k is None for k in keys) else keys
    if not isinstance(holder, torch.nn.Module):
        return None
    if hasattr(holder, "weight_g") and hasattr(holder, "weight_v"):
        ps = (holder.weight_v, holder.weight_g)
    elif isinstance(holder._parameters.get("weight"), torch.nn.Parameter):
        ps = (holder.weight,)
    else:
        return None
    bias = getattr(holder, "bias", None)
    return tuple((t.data_ptr(), t._version) for t in ps) + (None if bias is None else (bias.data_ptr(), bias._version), _WEIGHT_EPOCH[0])


class HipConvFn(torch.autograd.Function):
    """y = conv(x, w, b) with the forward on the HIP engine and the backward through torch.nn.functional."""

    @staticmethod
    def forward(ctx, x, w, b, module, lrelu=False, res=None):
        """lrelu: y = conv(leaky_relu(x)) with the activation applied on the conv's load path; res: + res in its epilogue (the two neighbours of every conv
        of a HiFi-GAN residual pair, decoder.py:92-101 -- three PyTorch launches per pair forward and their saved outputs)"""
        x = x.contiguous().float()
        op = module._op(bind=False)
        key = param_key(module)
        if not op.has_weights_of(key):
            # the grad-input handle of a stride-1 conv (created by the first backward) takes the same weight: both packs in one pair of launches
            adj = module.__dict__.get("_hip_bwd_ops", {}).get("dxa") if (key is not None and not L.switch("VS_NO_PAIR_PACK")) else None
            if adj is not None and ctx.needs_input_grad[0]:
                op.set_weights_pair(adj, w, b, key)
            else:
                op.set_weights_from(w, b, key)
        if lrelu or res is not None:
            y = op.forward(x, in_act=L.IN_LRELU if lrelu else L.IN_NONE, res=None if res is None else res.contiguous().float())
        else:
            y = op.forward(x)
        ctx.module = module
        ctx.wkey = key                  # the backward handles pack the adjoint of THIS weight: keyed by the forward's key, not by the parameters' state at backward time
        ctx.save_for_backward(x, w, b if b is not None else x.new_empty(0))
        ctx.has_bias = b is not None
        ctx.lrelu, ctx.has_res = bool(lrelu), res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, b = ctx.saved_tensors
        need = ctx.needs_input_grad
        gy = gy.contiguous()
        want_b = bool(need[2] and ctx.has_bias)
        # (lrelu: the conv's input was leaky_relu(x): recomputed for the weight gradient -- one launch, nothing saved -- and its derivative applied to the grad-input)
        xin = F.leaky_relu(x, LRELU_SLOPE) if (ctx.lrelu and need[1]) else x
        gx, gw, gb = conv_backward(ctx.module, xin, w, gy, bool(need[0]), bool(need[1]), key=ctx.wkey, need_b=want_b)
        if ctx.lrelu and gx is not None:
            gx = torch.ops.aten.leaky_relu_backward(gx, x, LRELU_SLOPE, False)
        if want_b and gb is None:
            gb = bias_grad(gy)
        return gx, gw, gb, None, None, (gy if (ctx.has_res and need[5]) else None)


def _bwd_op(module, key, *args):
    """ConvOp handles of the backward-data convs, cached on the module next to the forward ones"""
    ops = module.__dict__.setdefault("_hip_bwd_ops", {})
    if key not in ops:
        ops[key] = ConvOp(*args)
    return ops[key]


def conv_backward(m, x, w, gy, need_x, need_w, key=False, need_b=None):
    """Gradients of y = conv(x, w) for a HipConv1d / HipConvTranspose1d `m` (stride-1 dilated conv, or stride-u transposed
    conv), MIOpen-free:

    * grad-input runs on the HIP conv engine itself: for a conv it is the correlation of gy with the tap-reversed,
      channel-transposed weight (padding d(K-1) - p); for a transposed conv the stride-u gather
      gx[ci, m] = sum_{co,k} w[ci,co,k] gy[co, m u - p + k] becomes a stride-1 conv over the u de-interleaved phases of gy
      stacked as u*C_out input channels with ceil(K/u) taps;
    * grad-weight is the long-thin GEMM gw[co, ci, k] = sum_{b,t} gy[b, co, t] x[b, ci, t + k d - p] on its own MFMA kernel
      (csrc/conv_backward.hip, vs_conv_wgrad); for a transposed conv the same kernel with x in the role of gy over the
      phase-stacked gradient.
    """
    B, Cin, T = x.shape
    K = w.shape[2]
    gx = gw = gb = None
    if m._kind != L.CONV_TRANSPOSE1D:
        Cout, d, p = w.shape[0], m.dilation[0], m.padding[0]
        Tout = gy.shape[2]
        if need_w:
            # (round 5 tried these launches on a second stream beside the grad-input conv: the kernels slow each other down -- device time 81 -> 89 ms a step,
            #  step time 86 -> 89-104 ms -- so they stay in line)
            if need_b:      # the bias gradient from the weight-gradient kernel's pass over gy
                gw, gb = conv_wgrad(gy, x, K, d, p, bias=True)
            else:
                gw = conv_wgrad(gy, x, K, d, p)
        if need_x:
            pb = d * (K - 1) - p
            assert pb >= 0, "conv backward-data: padding larger than the receptive field is not supported"
            # (the handle packs the adjoint -- channel transpose + tap reversal -- of the forward weight by index arithmetic)
            op = _bwd_op(m, "dxa", L.CONV1D, Cout, Cin, K, d, pb, L.CONV_ADJOINT)
            key = param_key(m) if key is False else key
            if not op.has_weights_of(key):
                op.set_weights_from(w, None, key)
            gx = op.forward(gy)
    else:
        Cout, u, p = w.shape[1], m.stride[0], m.padding[0]
        Q = -(-K // u)
        n_out = gy.shape[2]
        Mq = T + Q - 1
        # gyp[n'] = gy[n' - p], zero elsewhere, on n' in [0, Mq u);  G[b, r*Cout + co, j] = gyp[b, co, j u + r]
        right = Mq * u - p - n_out
        gyp = F.pad(gy, (p, max(right, 0)))[:, :, :Mq * u]
        G = gyp.view(B, Cout, Mq, u).permute(0, 3, 1, 2).reshape(B, u * Cout, Mq)
        if need_x:
            op = _bwd_op(m, "dx", L.CONV1D, u * Cout, Cin, Q, 1, 0, 0)
            key = param_key(m) if key is False else key
            if not op.has_weights_of(key):
                wq = F.pad(w.detach(), (0, Q * u - K)).view(Cin, Cout, Q, u).permute(0, 3, 1, 2).reshape(Cin, u * Cout, Q)
                op.set_weights_from(wq, None, key)
            gx = op.forward(G.contiguous())                                             # [B, Cin, Mq - Q + 1 = T]
        if need_w:
            # gw2[ci, j, q] = sum_{b,m} x[b, ci, m] * G[b, j, m + q]: the same kernel with x in the role of the output gradient
            g2 = conv_wgrad(x, G.contiguous(), Q, 1, 0)                                   # [Cin, u*Cout, Q]
            gw = g2.view(Cin, u, Cout, Q).permute(0, 2, 3, 1).reshape(Cin, Cout, Q * u)[:, :, :K].contiguous()
    # (need_b is None: the two-value form of the callers that take the bias gradient elsewhere; gb stays None where this path did not produce it)
    return (gx, gw) if need_b is None else (gx, gw, gb)


# ------------------------------------------------------------------------------------------------------------------
# discriminator convs: dense convs with a stride, and grouped strided convs


def _cached_op(holder, key, *args):
    ops_ = holder.__dict__.setdefault("_hip_disc_ops", {})
    if key not in ops_:
        ops_[key] = ConvOp(*args)
    return ops_[key]


def _phase_geometry(T, K, stride, pad):
    Tout = (T + 2 * pad - K) // stride + 1
    Q = -(-K // stride)
    return Tout, Q, Tout + Q - 1


def _phase_weights(w, stride, Q):
    """w [Cout, C, K] -> [Cout, stride*C, Q] with w2[co, r*C + c, q] = w[co, c, q*stride + r] (zero beyond K)"""
    Cout, C, K = w.shape
    return F.pad(w, (0, Q * stride - K)).view(Cout, C, Q, stride).permute(0, 3, 1, 2).reshape(Cout, stride * C, Q).contiguous()


class StridedConv1dFn(torch.autograd.Function):
    """y = conv1d(x, w, b, stride, padding) (dilation 1, dense) on the HIP conv engine: a stride-s conv is the stride-1 conv of the
    s de-interleaved phases of the padded input stacked as s*C channels with ceil(K/s) taps; its gradients are the engine's
    grad-input conv (the same handle kind with VS_CONV_ADJOINT weights) / vs_conv_wgrad on that form.

    The period discriminators present hundreds of SHORT items (B*p columns of 10-300 positions): the engine tiles time per item
    (256 columns), so the N explicitly padded items are laid end to end as ONE sequence [1, s*C, N*Hq (+ Q-1 zero columns)] -- the
    zero padding between them is already part of each item, outputs at positions that straddle two items are discarded (and enter
    the gradients as zeros).  The layout changes are one launch each (vs_phase_stack from any view of x, vs_phase_items,
    vs_phase_unstack) where pad / view / permute / contiguous / zeros took 3-5.  The result is returned as the [N, Cout, Tout] VIEW of
    the engine's output sequence (the leaky_relu that follows every layer but the last writes it out contiguously).
    `holder` caches the engine handles."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, holder):
        lib = L.require_gpu()
        x, w = x.float(), w.contiguous().float()
        N, C, T = x.shape
        Cout, _, K = w.shape
        Tout, Q, Hq = _phase_geometry(T, K, stride, pad)
        Lf = N * Hq
        XF = torch.empty((1, stride * C, Lf + Q - 1), device=x.device, dtype=torch.float32)
        L.check(lib.vs_phase_stack(ctypes.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), x.stride(2), L.ptr(XF), N, C, T, stride, pad, Hq,
                                   Lf + Q - 1, L.stream_ptr()))
        op = _cached_op(holder, ("fwd", C, Cout, K, stride), L.CONV1D, stride * C, Cout, Q, 1, 0, 0)
        key = param_key(holder)
        if not op.has_weights_of(key):
            wq = _phase_weights(w.detach(), stride, Q) if stride > 1 else w
            adj = holder.__dict__.get("_hip_disc_ops", {}).get(("dxa", C, Cout, K, stride)) if (key is not None and not L.switch("VS_NO_PAIR_PACK")) else None
            if adj is not None and ctx.needs_input_grad[0]:
                op.set_weights_pair(adj, wq, b, key)
            else:
                op.set_weights_from(wq, b, key)
        yF = op.forward(XF)                                                                     # [1, Cout, N*Hq]
        ctx.save_for_backward(XF, w)
        ctx.cfg = (N, C, T, K, stride, pad, Q, Hq, Tout, b is not None, holder)
        ctx.wkey = key          # (the saved `w` is THIS version: the grad-input handle is labelled with it, as HipConvFn does -- ADVICE r4)
        return yF.view(Cout, N, Hq)[:, :, :Tout].permute(1, 0, 2)

    @staticmethod
    def backward(ctx, gy):
        lib = L.require_gpu()
        XF, w = ctx.saved_tensors
        N, C, T, K, stride, pad, Q, Hq, Tout, has_bias, holder = ctx.cfg
        Cout = w.shape[0]
        Lf = N * Hq
        gy = gy.contiguous().float()
        gyF = torch.empty((1, Cout, Lf), device=gy.device, dtype=torch.float32)
        L.check(lib.vs_phase_items(L.ptr(gy), L.ptr(gyF), N, Cout, Tout, Hq, Lf, 1, L.stream_ptr()))
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            if Cout * stride * C >= 256 * 256 or Q > 16:       # (vs_conv_wgrad covers up to 16 taps)
                # wide layers (512 / 1024 channels): per tap a plain [Cout x P] x [P x s*C] GEMM over the folded sequence --
                # library GEMM territory (rocBLAS); vs_conv_wgrad's 32 x 32 tiles re-read both operands once per tile pair
                g2 = torch.stack([gyF[0] @ XF[0][:, q:q + Lf].t() for q in range(Q)], dim=2)
                PROFILER.note("strided-conv wgrad (library GEMM per tap)", 2.0 * Cout * stride * C * Q * Lf)
            elif has_bias and ctx.needs_input_grad[2]:
                g2, gb = conv_wgrad(gyF, XF, Q, 1, 0, bias=True)                                # (the columns of gyF between the items are zeros: its row sums are gy's)
            else:
                g2 = conv_wgrad(gyF, XF, Q, 1, 0)                                               # [Cout, s*C, Q]
            gw = g2.view(Cout, stride, C, Q).permute(0, 2, 3, 1).reshape(Cout, C, Q * stride)[:, :, :K].contiguous() if stride > 1 else g2
        if ctx.needs_input_grad[0]:
            op = _cached_op(holder, ("dxa", C, Cout, K, stride), L.CONV1D, Cout, stride * C, Q, 1, Q - 1, L.CONV_ADJOINT)
            key = ctx.wkey      # (not param_key(holder) now: parameters stepped between forward and backward would label the OLD weight with the new key)
            if not op.has_weights_of(key):      # (the handle packs the adjoint of the phase-stacked forward weight)
                op.set_weights_from(_phase_weights(w.detach(), stride, Q) if stride > 1 else w, None, key)
            gXF = op.forward(gyF)                                                               # [1, s*C, N*Hq + Q - 1]
            gx = torch.empty((N, C, T), device=gy.device, dtype=torch.float32)
            L.check(lib.vs_phase_unstack(L.ptr(gXF), Lf + Q - 1, L.ptr(gx), N, C, T, stride, pad, Hq, L.stream_ptr()))
        if has_bias and ctx.needs_input_grad[2] and gb is None:
            gb = bias_grad(gy)
        return gx, gw, gb, None, None, None


class GroupedConv1dFn(torch.autograd.Function):
    """grouped / strided conv1d on the VALU kernels of csrc/grouped_conv.hip (forward, grad-input, grad-weight)"""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, groups):
        x, w = x.contiguous().float(), w.contiguous().float()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, groups, b is not None)
        return gconv1d_fwd(x, w, None if b is None else b.detach(), stride, pad, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, groups, has_bias = ctx.cfg
        gy = gy.contiguous().float()
        gx = gconv1d_bwd_data(gy, w, x.shape[2], stride, pad, groups) if ctx.needs_input_grad[0] else None
        gw = gconv1d_bwd_weight(gy, x, w.shape[2], stride, pad, groups) if ctx.needs_input_grad[1] else None
        gb = bias_grad(gy) if (has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None, None


def disc_conv1d(holder, x, w, b, stride, pad, groups=1):
    """conv1d of the discriminators: dense -> MFMA engine, grouped -> VALU kernels; differentiable in x, w, b"""
    if groups == 1:
        return StridedConv1dFn.apply(x, w, b, stride, pad, holder)
    return GroupedConv1dFn.apply(x, w, b, stride, pad, groups)


def effective_weight(m):
    """weight, or g * v / ||v|| (torch.nn.utils.weight_norm, dim 0) as a differentiable torch expression"""
    w = m.__dict__.get("_w_eff")          # folded for the whole network at the top of the pass (weight_bank.WeightBank.refresh)
    if w is not None:
        return w
    if hasattr(m, "weight_g"):
        return torch._weight_norm(m.weight_v, m.weight_g, 0)
    return m.weight


def conv(m, x, lrelu=False, res=None):
    """differentiable conv through HipConv1d / HipConvTranspose1d `m`; lrelu / res: y = conv(leaky_relu(x)) + res in the one launch (stride-1 convs)"""
    return HipConvFn.apply(x, effective_weight(m), m.bias, m, lrelu, res)


def training_path(module):
    if module.training and torch.is_grad_enabled():
        if torch.is_autocast_enabled("cuda"):
            raise NotImplementedError(
                "the HIP training path does not run inside a torch.autocast region (the reference's `amp: true`, utils/commons/trainer.py:325, default false): "
                "its autograd Functions exchange fp32 tensors with aten ops that autocast would hand bf16 tensors to; inference modules do support autocast "
                "(bf16 operands, fp32 accumulation: visinger_amd.modules.hipconv.apply_math)")
        return True
    return False


# ------------------------------------------------------------------------------------------------------------------
# module forwards (same reference lines as the inference modules)


class GateFn(torch.autograd.Function):
    """acts = tanh(x_in[:, :H] + g[:, :H]) * sigmoid(x_in[:, H:] + g[:, H:]) (encoder.py:177-180, 206-213): one HIP launch forward
    (vs_gate_fwd), one backward (vs_gate_bwd: both halves of dx_in and the sum over t for dg), instead of ~15 PyTorch kernels.
    g: [B, 2H, 1] (a slice of the cond_layer output: any batch stride) or None."""

    @staticmethod
    def forward(ctx, x_in, g):
        lib = L.require_gpu()
        x_in = x_in.contiguous()
        B, H2, T = x_in.shape
        acts = torch.empty((B, H2 // 2, T), device=x_in.device, dtype=torch.float32)
        gp, gbs = (None, 0) if g is None else (ctypes.c_void_p(g.data_ptr()), g.stride(0))
        L.check(lib.vs_gate_fwd(L.ptr(x_in), gp, gbs, L.ptr(acts), B, H2 // 2, T, L.stream_ptr()))
        ctx.save_for_backward(x_in, g if g is not None else x_in.new_empty(0))
        ctx.has_g = g is not None
        return acts

    @staticmethod
    def backward(ctx, dacts):
        lib = L.require_gpu()
        x_in, g = ctx.saved_tensors
        B, H2, T = x_in.shape
        dacts = dacts.contiguous()
        dx = torch.empty_like(x_in)
        dg = torch.zeros((B, H2, 1), device=x_in.device, dtype=torch.float32) if (ctx.has_g and ctx.needs_input_grad[1]) else None
        gp, gbs = (None, 0) if not ctx.has_g else (ctypes.c_void_p(g.data_ptr()), g.stride(0))
        L.check(lib.vs_gate_bwd(L.ptr(x_in), gp, gbs, L.ptr(dacts), L.ptr(dx), L.ptr(dg), H2, B, H2 // 2, T, L.stream_ptr()))
        return dx, dg


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm_C(a + r) * gamma + beta (rel_transformer.py:33-42 with the residual add of :305, :314 folded in): forward on
    vs_layernorm_c_fwd (the inference kernel), backward on vs_layernorm_c_bwd -- two launches instead of ~30 PyTorch kernels."""

    @staticmethod
    def forward(ctx, a, r, gamma, beta, eps):
        lib = L.require_gpu()
        a = a.contiguous()
        r = None if r is None else r.contiguous()
        B, C, T = a.shape
        y = torch.empty_like(a)
        L.check(lib.vs_layernorm_c_fwd(L.ptr(a), L.ptr(r), L.ptr(gamma.detach().contiguous()), L.ptr(beta.detach().contiguous()), None, 0, 0,
                                       None, L.ptr(y), B, C, T, eps, L.stream_ptr()))
        ctx.save_for_backward(a, r if r is not None else a.new_empty(0), gamma)
        ctx.has_r, ctx.eps = r is not None, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.require_gpu()
        a, r, gamma = ctx.saved_tensors
        B, C, T = a.shape
        dx = torch.empty_like(a)
        dgb = torch.zeros((B, 2, C), device=a.device, dtype=torch.float32)          # per-item partial sums (see vs_layernorm_c_bwd)
        L.check(lib.vs_layernorm_c_bwd(L.ptr(a), L.ptr(r) if ctx.has_r else None, L.ptr(gamma.detach().contiguous()), L.ptr(dy.contiguous()),
                                       L.ptr(dx), ctypes.c_void_p(dgb.data_ptr()),
                                       ctypes.c_void_p(dgb.data_ptr() + 4 * C), B, C, T, ctx.eps, L.stream_ptr()))
        dgb = dgb.sum(0)
        return dx, (dx if ctx.has_r else None), dgb[0], dgb[1], None


class WnStepFn(torch.autograd.Function):
    """x_new = (x + rs[:, :H]) * x_mask, out_new = out_acc + rs[:, H:] (encoder.py:186-193; out_acc None: the first layer): one HIP launch each way
    (vs_wn_step_fwd / vs_wn_step_bwd) where autograd recorded two slices, two adds and a multiply -- 3 launches forward, ~9 backward, per layer."""

    @staticmethod
    def forward(ctx, x, rs, out_acc, mask2):
        x, rs = x.contiguous(), rs.contiguous()
        B, H, T = x.shape
        x_new, out_new = torch.empty_like(x), torch.empty_like(x)
        L.check(L.require_gpu().vs_wn_step_fwd(L.ptr(x), L.ptr(rs), L.ptr(None if out_acc is None else out_acc.contiguous()), L.ptr(mask2), L.ptr(x_new),
                                               L.ptr(out_new), B, H, T, L.stream_ptr()))
        ctx.save_for_backward(mask2)
        ctx.set_materialize_grads(False)
        ctx.has_acc = out_acc is not None
        return x_new, out_new

    @staticmethod
    def backward(ctx, dx_new, dout_new):
        mask2, = ctx.saved_tensors
        if dx_new is None and dout_new is None:
            return None, None, None, None
        ref = dx_new if dx_new is not None else dout_new
        B, H, T = ref.shape
        dx_new = None if dx_new is None else dx_new.contiguous()
        dout_new = None if dout_new is None else dout_new.contiguous()
        d_rs = torch.empty((B, 2 * H, T), device=ref.device, dtype=torch.float32)
        dx = torch.empty((B, H, T), device=ref.device, dtype=torch.float32)
        L.check(L.require_gpu().vs_wn_step_bwd(L.ptr(dx_new), L.ptr(dout_new), L.ptr(mask2), L.ptr(d_rs), L.ptr(dx), B, H, T, L.stream_ptr()))
        return dx, d_rs, (dout_new if ctx.has_acc else None), None


class L1MeanFn(torch.autograd.Function):
    """mean |a - b| with the gradient to `a` only (b: the detached target): vs_l1_mean_fwd / _bwd, one launch each way.  a and b must share one dense
    layout (equal strides, no gaps): the kernels walk the storage."""

    _work = {}

    @staticmethod
    def dense_pair(a, b):
        if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.stride() == b.stride() and a.numel() > 0):
            return False
        order = sorted(range(a.dim()), key=lambda d: -a.stride(d))
        return a.permute(order).is_contiguous()

    @staticmethod
    def forward(ctx, a, b):
        work = L1MeanFn._work.get(a.device)
        if work is None:
            work = L1MeanFn._work[a.device] = torch.zeros(257, device=a.device, dtype=torch.float32)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        L.check(L.require_gpu().vs_l1_mean_fwd(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), L.ptr(work), ctypes.c_void_p(out.data_ptr()),
                                               a.numel(), L.stream_ptr()))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)            # (preserve_format: a's dense strides)
        gout = gout.contiguous().float()
        L.check(L.require_gpu().vs_l1_mean_bwd(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(gout.data_ptr()),
                                               ctypes.c_void_p(da.data_ptr()), a.numel(), L.stream_ptr()))
        return da, None


def l1_mean(a, b):
    """mean |a - b|, differentiable in `a` (b is treated as a constant)"""
    b = b.detach()
    if L1MeanFn.dense_pair(a, b) and not L.switch("VS_NO_TRAIN_FUSED"):
        return L1MeanFn.apply(a, b)
    return torch.mean(torch.abs(b - a))


def wavenet(m, x, x_mask, g=None):
    """encoder.py:167-195"""
    H = m.hidden_channels
    if g is not None:
        g = conv(m.cond_layer, g)
    # (the gate kernel reads the conditioning as ONE column per item, g[b, c, 0]: a time-varying [B, gin, T] condition -- which the
    # reference's WN broadcasts as well -- or a non-fp32 one takes the PyTorch formulation)
    fused = (x.is_cuda and not L.switch("VS_NO_TRAIN_FUSED") and (m.p_dropout == 0 or not m.training) and
             (g is None or (g.shape[2] == 1 and g.dtype == torch.float32)))
    # (one split of the conditioning for all layers -- a single cat backward -- where a slice per layer ran zeros + copy + accumulate each)
    gs = None if g is None else torch.split(g, 2 * H, dim=1)
    step_fused = x.is_cuda and x.dtype == torch.float32 and x_mask.numel() == x.shape[0] * x.shape[2] and not L.switch("VS_NO_TRAIN_FUSED")
    mask2 = x_mask.reshape(x.shape[0], x.shape[2]).float().contiguous() if step_fused else None
    output = None
    for i in range(m.n_layers):
        x_in = conv(m.in_layers[i], x)
        if fused:
            acts = GateFn.apply(x_in, None if gs is None else gs[i])
        else:
            if gs is not None:
                x_in = x_in + gs[i]
            acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
            acts = m.drop(acts)
        rs = conv(m.res_skip_layers[i], acts)
        if i < m.n_layers - 1:
            if step_fused:
                x, output = WnStepFn.apply(x, rs, output, mask2)
            else:
                x = (x + rs[:, :H]) * x_mask
                output = rs[:, H:] if output is None else output + rs[:, H:]
        else:
            output = rs if output is None else output + rs
    return output * x_mask


def coupling_layer(m, x, x_mask, g=None, reverse=False):
    """flow.py:66-85"""
    half = m.half_channels
    x0, x1 = torch.split(x, [half, half], 1)
    h = conv(m.pre, x0) * x_mask
    h = wavenet(m.enc, h, x_mask, g)
    stats = conv(m.post, h) * x_mask
    if not m.mean_only:
        mean, logs = torch.split(stats, [half, half], 1)
    else:
        mean, logs = stats, torch.zeros_like(stats)
    if not reverse:
        x1 = mean + x1 * torch.exp(logs) * x_mask
        return torch.cat([x0, x1], 1), torch.sum(logs, [1, 2])
    x1 = (x1 - mean) * torch.exp(-logs) * x_mask
    return torch.cat([x0, x1], 1)


def flow_block(m, x, x_mask, g=None, reverse=False):
    """flow.py:33-40"""
    if not reverse:
        for f in range(m.n_flows):
            x, _ = coupling_layer(m.flows[2 * f], x, x_mask, g, False)
            x = torch.flip(x, [1])
    else:
        for f in reversed(range(m.n_flows)):
            x = torch.flip(x, [1])
            x = coupling_layer(m.flows[2 * f], x, x_mask, g, True)
    return x


def resblock1(m, x, x_mask=None):
    """decoder.py:91-104"""
    fuse = x_mask is None and x.is_cuda and x.dtype == torch.float32 and not L.switch("VS_NO_TRAIN_FUSED")
    for c1, c2 in zip(m.convs1, m.convs2):
        if fuse:      # both activations on the convs' load paths, the residual add in the second conv's epilogue: two launches where five ran
            x = conv(c2, conv(c1, x, lrelu=True), lrelu=True, res=x)
            continue
        xt = F.leaky_relu(x, LRELU_SLOPE)
        if x_mask is not None:
            xt = xt * x_mask
        xt = conv(c1, xt)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        if x_mask is not None:
            xt = xt * x_mask
        xt = conv(c2, xt)
        x = xt + x
    return x if x_mask is None else x * x_mask


def resblock2(m, x, x_mask=None):
    """decoder.py:124-133"""
    for c in m.convs:
        xt = F.leaky_relu(x, LRELU_SLOPE)
        if x_mask is not None:
            xt = xt * x_mask
        x = conv(c, xt) + x
    return x if x_mask is None else x * x_mask


def generator(m, x, g=None):
    """decoder.py:40-59"""
    x = conv(m.conv_pre, x)
    if g is not None:
        x = x + conv(m.cond, g)
    for i in range(m.num_upsamples):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = conv(m.ups[i], x)
        xs = None
        for j in range(m.num_kernels):
            rb = m.resblocks[i * m.num_kernels + j]
            r = resblock1(rb, x) if hasattr(rb, "convs1") else resblock2(rb, x)
            xs = r if xs is None else xs + r
        x = xs / m.num_kernels
    x = F.leaky_relu(x, LRELU_SLOPE)
    return torch.tanh(conv(m.conv_post, x))


def layer_norm(m, x, r=None):
    """rel_transformer.py:33-42 on x (+ r: the residual add of the encoder layers folded in)"""
    if x.is_cuda and x.dim() == 3 and x.shape[1] <= 1024 and not L.switch("VS_NO_TRAIN_FUSED"):
        return LayerNormFn.apply(x, r, m.gamma, m.beta, m.eps)
    if r is not None:
        x = x + r
    mean = torch.mean(x, 1, keepdim=True)
    var = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + m.eps)
    return x * m.gamma.view(1, -1, 1) + m.beta.view(1, -1, 1)


class AttnCoreFn(torch.autograd.Function):
    """The attention core of rel_transformer.py:148-179 (scores + relative-key bias, -1e4 mask fill, softmax, dropout, P.V + relative-value
    term) on the streaming HIP kernels of csrc/attention_train.hip: forward keeps only the output and the softmax statistics of each query, the
    backward recomputes probabilities tile by tile -- no [B, h, T, T] tensor in either direction (the PyTorch version of this function
    below held five of them per layer).  q / k / v: [B, nh * dk, T]; rel_k / rel_v: [nh_rel, 2w+1, dk] or None; mask [B, T] or None.
    The dropout mask is a hash of (seed, batch * head, query, key); the seed is drawn from torch's CPU generator (no device sync)."""

    @staticmethod
    def forward(ctx, q, k, v, rel_k, rel_v, mask, nh, w, p_drop):
        lib = L.require_gpu()
        packed = k is None                  # q is the [B, 3C, T] output of the fused q|k|v projection: rows [0, C) q, [C, 2C) k, [2C, 3C) v
        if packed:
            qkv = q.contiguous()
            B, C3, T = qkv.shape
            C = C3 // 3
            q, k, v = (ctypes.c_void_p(qkv.data_ptr() + 4 * i * C * T) for i in range(3))
            bs = C3 * T
        else:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            B, C, T = q.shape
            bs = 0
        dk = C // nh
        has_rel = rel_k is not None
        rk = rel_k.detach().contiguous() if has_rel else None
        rv = rel_v.detach().contiguous() if has_rel else None
        mk = None if mask is None else mask.contiguous()
        seed = int(torch.empty((), dtype=torch.int64).random_()) if p_drop > 0 else 0
        dev = qkv.device if packed else q.device
        out = torch.empty((B, C, T), device=dev, dtype=torch.float32)
        lse = torch.empty((2, B, nh, T), device=dev, dtype=torch.float32)        # row maximum, log of the row sum
        qp, kp, vp = (q, k, v) if packed else (L.ptr(q), L.ptr(k), L.ptr(v))
        L.check(lib.vs_relattn_train_fwd(qp, kp, vp, bs, L.ptr(rk), L.ptr(rv), L.ptr(mk), L.ptr(out), 0, L.ptr(lse), B, nh, dk, T,
                                         w if has_rel else -1, rk.shape[0] if has_rel else 1, float(p_drop), seed, L.stream_ptr()))
        PROFILER.note("relattn_train_fwd", 4.0 * B * C * T * T)           # Q K^T and P V over the full [T, T] matrix
        none = out.new_empty(0)
        ctx.save_for_backward(*((qkv, none, none) if packed else (q, k, v)), rk if has_rel else none, rv if has_rel else none,
                              mk if mk is not None else none, out, lse)
        ctx.cfg = (nh, w, float(p_drop), seed, has_rel, mk is not None, packed)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.require_gpu()
        q, k, v, rk, rv, mk, out, lse = ctx.saved_tensors
        nh, w, p_drop, seed, has_rel, has_mask, packed = ctx.cfg
        B, C, T = out.shape
        dk = C // nh
        dout = dout.contiguous().float()
        if packed:
            dqkv = torch.empty_like(q)
            qp, kp, vp = (ctypes.c_void_p(q.data_ptr() + 4 * i * C * T) for i in range(3))
            dqp, dkp, dvp = (ctypes.c_void_p(dqkv.data_ptr() + 4 * i * C * T) for i in range(3))
            bs = 3 * C * T
        else:
            dq, dkk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
            qp, kp, vp, dqp, dkp, dvp, bs = L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(dq), L.ptr(dkk), L.ptr(dv), 0
        R = 2 * w + 1 if has_rel else 0
        nt = -(-T // 32)
        work = torch.empty((B * nh * T * (1 + 2 * R),), device=q.device, dtype=torch.float32)
        pk = torch.empty((B, nh, nt, R, dk), device=q.device, dtype=torch.float32) if has_rel else None
        pv = torch.empty_like(pk) if has_rel else None
        L.check(lib.vs_relattn_train_bwd(qp, kp, vp, bs, L.ptr(rk) if has_rel else None, L.ptr(rv) if has_rel else None,
                                         L.ptr(mk) if has_mask else None, L.ptr(out), L.ptr(dout), 0, L.ptr(lse), dqp, dkp, dvp,
                                         bs, L.ptr(work), L.ptr(pk), L.ptr(pv), B, nh, dk, T, w if has_rel else -1, rk.shape[0] if has_rel else 1,
                                         p_drop, seed, L.stream_ptr()))
        PROFILER.note("relattn_train_bwd", 8.0 * B * C * T * T)           # dV, dP, dQ, dK: four [T, T] x d GEMMs (the recomputed scores not counted)
        drk = drv = None
        if has_rel:
            dims = (0, 1, 2) if rk.shape[0] == 1 else (0, 2)
            drk = pk.sum(dims).view(rk.shape)
            drv = pv.sum(dims).view(rv.shape)
        if packed:
            return dqkv, None, None, drk, drv, None, None, None, None
        return dq, dkk, dv, drk, drv, None, None, None, None


class _ConvHolder:
    """What HipConvFn / conv_backward need from a conv module, for a conv that is not a module of its own: the fused q | k | v projection of
    an attention layer (three 1x1 convs of the same input as ONE launch forward, one grad-input conv and one weight-gradient launch)."""
    _kind = L.CONV1D

    def __init__(self, c_in, c_out, k=1, dilation=1, padding=0):
        self.in_channels, self.out_channels = c_in, c_out
        self.kernel_size, self.dilation, self.padding, self.stride = (k,), (dilation,), (padding,), (1,)

    def _op(self, bind=False):
        ops = self.__dict__.setdefault("_hip_ops", {})
        if "fwd" not in ops:
            ops["fwd"] = ConvOp(L.CONV1D, self.in_channels, self.out_channels, self.kernel_size[0], self.dilation[0], self.padding[0], 0)
        from .modules.hipconv import apply_math
        apply_math(self.__dict__, ops["fwd"])
        return ops["fwd"]


def attention(m, x, frame_mask):
    """rel_transformer.py:138-179: q/k/v/o convs on the HIP engine; the core on the streaming kernels of csrc/attention_train.hip
    (`AttnCoreFn`), or -- heads wider than 128 channels, windows wider than 7, VS_NO_TRAIN_ATTN -- as differentiable torch ops with the
    relative terms laid onto the band through strided views (what the reference's pad / reshape skew implements)."""
    B, C, T = x.shape
    nh, dk, w = m.n_heads, m.k_channels, m.window_size
    plain_core = not getattr(m, "proximal_bias", False) and getattr(m, "block_length", None) is None      # (the streaming kernels carry neither option)
    if x.is_cuda and dk <= 128 and (w is None or w <= 7) and T <= 65535 and plain_core and not m.__dict__.get("store_attn", False) and not L.switch("VS_NO_TRAIN_ATTN"):
        rel_k, rel_v = (m.emb_rel_k, m.emb_rel_v) if w is not None else (None, None)
        pd = m.drop.p if m.training else 0.0
        plain = all(not hasattr(c, "weight_g") and c.bias is not None and c.kernel_size[0] == 1 for c in (m.conv_q, m.conv_k, m.conv_v))
        if plain and not L.switch("VS_NO_FUSED_QKV"):
            # q | k | v as one [3C, C_in] projection (rel_transformer.py:120-122 are three nn.Conv1d(channels, channels, 1) of the same x)
            holder = m.__dict__.get("_hip_qkv")
            if holder is None:
                holder = m.__dict__["_hip_qkv"] = _ConvHolder(m.conv_q.in_channels, 3 * C)
                holder._key_sources = (m.conv_q, m.conv_k, m.conv_v)
            arith = m.conv_q.__dict__.get("_hip_math")
            if arith is not None:
                holder.__dict__["_hip_math"] = arith
            wqkv = torch.cat([m.conv_q.weight, m.conv_k.weight, m.conv_v.weight], 0)
            bqkv = torch.cat([m.conv_q.bias, m.conv_k.bias, m.conv_v.bias], 0)
            out = AttnCoreFn.apply(HipConvFn.apply(x, wqkv, bqkv, holder, False, None), None, None, rel_k, rel_v, frame_mask, nh, w if w is not None else -1, pd)
        else:
            out = AttnCoreFn.apply(conv(m.conv_q, x), conv(m.conv_k, x), conv(m.conv_v, x), rel_k, rel_v, frame_mask, nh,
                                   w if w is not None else -1, pd)
        return conv(m.conv_o, out)
    q = conv(m.conv_q, x).view(B, nh, dk, T).transpose(2, 3)
    k = conv(m.conv_k, x).view(B, nh, dk, T).transpose(2, 3)
    v = conv(m.conv_v, x).view(B, nh, dk, T).transpose(2, 3)
    scale = 1.0 / math.sqrt(dk)
    scores = torch.matmul(q, k.transpose(-2, -1)) * scale
    if w is not None:
        # relative-key logits q_i . emb_rel_k[j - i + w] on the band |j - i| <= w: the [T, 2w+1] table laid onto the diagonals of a
        # zero [T, T + 2w] buffer through a strided view (row stride T + 2w + 1), columns w .. w + T - 1 of which are the bias -- what the
        # reference's pad / reshape skew does (rel_transformer.py:214-243), as two cheap kernels instead of a gather over [T, T] whose
        # backward is a scatter_add over [B, h, T, T] (7 ms of the config-3 step)
        R = 2 * w + 1
        qr = torch.matmul(q, m.emb_rel_k.unsqueeze(0).transpose(-2, -1)) * scale     # [B, nh, T, 2w+1]
        buf = scores.new_zeros((B, nh, T, T + 2 * w))
        buf.as_strided((B, nh, T, R), (buf.stride(0), buf.stride(1), T + 2 * w + 1, 1)).copy_(qr)
        scores = scores + buf[..., w:w + T]
    if getattr(m, "proximal_bias", False):              # rel_transformer.py:163-165, 245-256: -log(1 + |i - j|)
        r = torch.arange(T, dtype=torch.float32, device=x.device)
        scores = scores - torch.log1p(torch.abs(r[None, :] - r[:, None]))[None, None]
    if frame_mask is not None:
        am = frame_mask.view(B, 1, T, 1) * frame_mask.view(B, 1, 1, T)
        scores = scores.masked_fill(am == 0, -1e4)
        if getattr(m, "block_length", None) is not None:    # rel_transformer.py:168-170 (only under a mask, as there)
            bm = torch.ones_like(scores).triu(-m.block_length).tril(m.block_length)
            scores = scores * bm + -1e4 * (1 - bm)
    p = m.drop(F.softmax(scores, dim=-1))
    if m.__dict__.get("store_attn", False):
        m.attn = p                     # (rel_transformer.py:143: `x, self.attn = self.attention(...)` -- the probabilities behind the dropout)
    out = torch.matmul(p, v)
    if w is not None:
        # relative weights p[i, i + r - w] (zero outside the sequence) @ emb_rel_v: the same strided view of the zero-padded probabilities
        pp = F.pad(p, (w, w))
        pw = pp.as_strided((B, nh, T, R), (pp.stride(0), pp.stride(1), T + 2 * w + 1, 1))
        out = out + torch.matmul(pw, m.emb_rel_v.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(B, C, T).float()      # (.float(): inside an autocast region the matmuls above return bf16 tensors)
    return conv(m.conv_o, out)


def ffn(m, x, x_mask):
    """rel_transformer.py:336-345"""
    h = conv(m.conv_1, x * x_mask)
    h = torch.relu(h) if m.activation != "gelu" else h * torch.sigmoid(1.702 * h)
    h = m.dropout(h)
    return conv(m.conv_2, h * x_mask)


def rel_encoder(m, x, x_mask, g=None):
    """rel_transformer.py:290-320: post-LN (VISinger's) and pre-LN (`pre_ln=True`, :301-317: the norm in front of each sub-layer, `last_ln` at the end)"""
    B, C, T = x.shape
    fm = x_mask.reshape(B, T)
    if g is not None:
        g = conv(m.pre_net, g)
    for i in range(m.n_layers):
        if g is not None:
            x = x + g
        x = x * x_mask
        if m.pre_ln:
            x = x + m.drop(attention(m.attn_layers[i], layer_norm(m.norm_layers_1[i], x), fm))
            x = x + m.drop(ffn(m.ffn_layers[i], layer_norm(m.norm_layers_2[i], x), x_mask))
        else:
            y = m.drop(attention(m.attn_layers[i], x, fm))
            x = layer_norm(m.norm_layers_1[i], x, y)
            y = m.drop(ffn(m.ffn_layers[i], x, x_mask))
            x = layer_norm(m.norm_layers_2[i], x, y)
    if m.pre_ln:
        x = layer_norm(m.last_ln, x)
    return x * x_mask
