"""Thin Python objects over the C-ABI operator handles (include/visinger_hip.h).  Plumbing only: tensors in,
device pointers and the current HIP stream out.  Used by visinger_amd.modules.* and by the GPU parity tests."""
import ctypes

import os

import torch

from . import _lib as L


def _off(t, elems):
    """pointer `elems` floats into tensor t"""
    return ctypes.c_void_p(t.data_ptr() + 4 * int(elems))


class _NoEvent:
    """stand-in for a HIP event in counting mode (no per-launch events: a training step has ~6 000 launches)"""

    def record(self):
        pass

    def elapsed_time(self, other):
        return 0.0


class LaunchProfiler:
    """Optional per-launch HIP-event timing of the conv engine (used by bench.py for the roofline line).
    Events are recorded on the stream the kernels are launched on (torch's current stream).
    start(count_only=True): count launches / algorithmic FLOPs / bytes per kernel instance without recording events."""

    def __init__(self):
        self.enabled = False
        self.count_only = False
        self.records = []          # (kernel instance name, algorithmic flops, algorithmic HBM bytes, start event, end event)

    def start(self, count_only=False):
        self.records = []
        self.count_only = count_only
        self.enabled = True

    def stop(self):
        self.enabled = False

    def events(self):
        if self.count_only:
            return _NoEvent(), _NoEvent()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def note(self, name, flops, nbytes=0.0):
        """count work done outside the event-timed launch sites (weight-gradient kernels, library GEMMs of the training path)"""
        if self.enabled:
            e = _NoEvent()
            self.records.append((name, float(flops), float(nbytes), e, e))

    def summary(self):
        """-> {instance: dict(launches, flops, ms)} ; call after torch.cuda.synchronize()."""
        out = {}
        for name, fl, by, e0, e1 in self.records:
            d = out.setdefault(name, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["bytes"] += by
            d["ms"] += e0.elapsed_time(e1)
        return out

    counts = summary


PROFILER = LaunchProfiler()


class ConvOp:
    """vs_conv_t: one nn.Conv1d / nn.ConvTranspose1d site, weights folded + packed on the device."""

    def __init__(self, kind, c_in, c_out, k, dilation_or_stride=1, padding=0, flags=0):
        self.kind, self.c_in, self.c_out, self.k = kind, c_in, c_out, k
        self.dil, self.pad, self.flags = dilation_or_stride, padding, flags
        self._h = None               # vs_conv_t*, created on first use (a handle is process- and device-local)
        self._pending_math = None    # arithmetic to apply when the handle is created (set_math before first use / after a copy)
        self._wkey = None
        self._last_kernel = ""
        if torch.cuda.is_available():
            _ = self.h               # fail early (bad dims, missing library) where a GPU is present

    @property
    def lib(self):
        return L.require_gpu()

    @property
    def h(self):
        if self._h is None:
            h = ctypes.c_void_p()
            L.check(self.lib.vs_conv_create(ctypes.byref(h), self.kind, self.c_in, self.c_out, self.k, self.dil, self.pad, self.flags))
            self._h = h
            if self._pending_math is not None:
                L.check(self.lib.vs_conv_set_math(h, int(self._pending_math), L.stream_ptr()))
        return self._h

    def _args(self):
        return (self.kind, self.c_in, self.c_out, self.k, self.dil, self.pad, self.flags)

    # A handle owns device memory and must be destroyed exactly once: copies (copy.deepcopy of a module for an EMA model,
    # torch.save(module), pickling for a spawned worker) get a FRESH handle with the same geometry and arithmetic and no
    # weights -- the owner re-binds its parameters on the next forward (the packed-weight cache key starts empty).
    def __reduce__(self):
        return (_rebuild_conv_op, (self._args(), self.math if self._h is not None else self._pending_math))

    def __deepcopy__(self, memo):
        return _rebuild_conv_op(self._args(), self.math if self._h is not None else self._pending_math)

    @property
    def math(self):
        """arithmetic of the contraction (L.MATH_F32 / MATH_SPLIT6 / MATH_BF16, include/visinger_hip.h vs_conv_math)"""
        return int(self.lib.vs_conv_get_math(self.h))

    def set_math(self, math):
        self._pending_math = int(math)
        L.check(self.lib.vs_conv_set_math(self.h, int(math), L.stream_ptr()))
        return self

    def invalidate(self):
        """Forget the packed-weight cache key: the next set_weights re-folds and re-packs.  For in-place edits the key cannot
        see (`p.data.copy_()` / `p.data.mul_()`: `.data` has its own version counter)."""
        self._wkey = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib().vs_conv_destroy(h)
            except Exception:
                pass

    def out_len(self, T):
        return int(self.lib.vs_conv_out_len(self.h, T))

    def last_kernel(self):
        """Kernel instance the last launch on this thread went to, as the library reports it (vs_last_kernel_name: the
        dispatch lives in csrc/conv_engine.hip only)."""
        return self.lib.vs_last_kernel_name().decode()

    def kernel_instance(self):
        """Kernel instance this handle's most recent forward() was dispatched to ("" before the first launch)."""
        return self._last_kernel

    def wino_eligible(self):
        """mirrors vs_conv_create: stride-1 'same' conv, odd k >= 3, dilation 1/3/5, whole 32-row tiles -> F(2,3) path
        (taken by vs_conv_forward unless the call splits rows or uses a coupling output mode)"""
        k, d = self.k, self.dil
        return (self.kind == L.CONV1D and k >= 3 and k % 2 == 1 and d in (1, 3, 5) and self.pad == d * (k - 1) // 2 and
                self.c_out % 32 == 0 and 3 * -(-k // 3) * d <= 64)

    def algorithmic_flops(self, B, T):
        """2*MAC of the convolution itself (what torch.utils.flop_counter reports for the reference's op)."""
        if self.kind == L.CONV_TRANSPOSE1D:
            return 2.0 * B * self.c_in * self.c_out * self.k * T
        return 2.0 * B * self.c_in * self.c_out * self.k * self.out_len(T)

    @property
    def rows_out(self):
        return self.c_out // 2 if self.kind == L.CONV1D_PAIRED else self.c_out

    def set_weights(self, w, g=None, bias=None, force=False):
        """w: weight or weight_v; g: weight_g or None; bias or None (contiguous fp32 CUDA tensors).
        Re-packs only when a tensor changed (data_ptr / in-place version) -- valid for PARAMETERS, whose storage persists;
        callers that pass temporaries (the training path's folded weight: a new tensor every step, which the allocator
        happily places at the previous step's address with version 0) must pass force=True."""
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in (w, g, bias))
        if key == self._wkey and not force:
            return
        w = w.detach()
        g = None if g is None else g.detach()
        bias = None if bias is None else bias.detach()
        L.check(self.lib.vs_conv_set_weights(self.h, L.ptr(w.contiguous()), L.ptr(None if g is None else g.contiguous()),
                                             L.ptr(None if bias is None else bias.contiguous()), L.stream_ptr()))
        self._wkey = None if force else key

    def has_weights_of(self, key):
        """True when the packed weights were derived from the parameters identified by `key` (set_weights_from)"""
        return key is not None and key == self._wkey

    def set_weights_from(self, w, bias, key):
        """Training path: pack `w` / `bias`, TEMPORARIES derived from parameters identified by `key` = their (data_ptr, in-place version)
        tuples (autograd.param_key), or None when the derivation is not a function of the parameters alone (spectral norm: a power
        iteration per forward).  Within one optimizer step the same parameters reach a handle several times -- the discriminators see
        the real and the generated batch in the generator pass and both again in the discriminator pass -- and every pack is one or
        two launches + the weight's bytes: the caller asks has_weights_of(key) first and skips deriving `w` altogether."""
        L.check(self.lib.vs_conv_set_weights(self.h, L.ptr(w.detach().contiguous()), None,
                                             L.ptr(None if bias is None else bias.detach().contiguous()), L.stream_ptr()))
        self._wkey = key

    def set_weights_pair(self, adjoint_op, w, bias, key):
        """Training path: pack `w` for this handle AND for the VS_CONV_ADJOINT handle of its grad-input in one pair of launches
        (vs_conv_set_weights_pair); both carry `key` afterwards, so the backward finds its weights in place."""
        L.check(self.lib.vs_conv_set_weights_pair(self.h, adjoint_op.h, L.ptr(w.detach().contiguous()),
                                                  L.ptr(None if bias is None else bias.detach().contiguous()), L.stream_ptr()))
        self._wkey = key
        adjoint_op._wkey = key

    @staticmethod
    def set_weights_batch(jobs):
        """Training path: pack every (op, w, bias, key) of `jobs` -- distinct handles, each with its plain (folded) weight -- in two launches
        for the whole list (vs_conv_set_weights_batch); each handle carries its key afterwards."""
        if not jobs:
            return
        n = len(jobs)
        keep = [(w.detach().contiguous(), None if b is None else b.detach().contiguous()) for _, w, b, _ in jobs]
        vp = ctypes.c_void_p
        hs = (vp * n)(*[op.h for op, _, _, _ in jobs])
        ws = (vp * n)(*[w.data_ptr() for w, _ in keep])
        bs = (vp * n)(*[None if b is None else b.data_ptr() for _, b in keep])
        L.check(jobs[0][0].lib.vs_conv_set_weights_batch(hs, ws, bs, n, L.stream_ptr()))
        for op, _, _, key in jobs:
            op._wkey = key

    def forward(self, x, *, B=None, T=None, x_bs=0, in_act=L.IN_NONE, mask=None, bias_b=None, bias_b_bs=0,
                y=None, y_bs=0, res=None, res_bs=0, acc=None, acc_bs=0, scale=1.0, out_act=L.OUT_NONE, out_mask=False,
                mode=L.MODE_LINEAR, split_row=0, out1=None, pair_mode=L.PAIR_GATE, logdet=None,
                x_ptr=None, y_ptr=None, res_ptr=None, acc_ptr=None, y_dtype=None):
        """Launch.  x: [B, c_in, T] (or pass x_ptr/x_bs/B/T for a channel window of a larger tensor).
        Returns y (allocated [B, rows_out, T_out] when not given; y_dtype = torch.bfloat16 for a bf16-RESIDENT output).
        x / y / res / acc may be bf16 tensors in the plain-bf16 arithmetic (vs_dtype in include/visinger_hip.h); res and acc follow y."""
        if B is None:
            B, _, T = x.shape
        Tout = self.out_len(T)
        if y is None and y_ptr is None:
            rows = split_row if split_row else self.rows_out
            y = torch.empty((B, rows, Tout), device=x.device, dtype=y_dtype or torch.float32)
        io = L.ConvIO()
        if x_ptr is not None:
            io.x = x_ptr
        else:
            io.x, io.x_dtype = L.act_ptr(x)
        io.x_bs, io.B, io.T = x_bs, B, T
        io.in_act = in_act
        io.mask = L.ptr(mask)
        io.bias_b = L.ptr(bias_b)
        io.bias_b_bs = bias_b_bs
        io.split_row = split_row
        o = io.out[0]
        if y_ptr is not None:
            o.y = y_ptr
            o.res = res_ptr if res_ptr is not None else L.ptr(res)
            o.acc = acc_ptr if acc_ptr is not None else L.ptr(acc)
        else:
            o.y, io.y_dtype = L.act_ptr(y)
            for name, t, tp in (("res", res, res_ptr), ("acc", acc, acc_ptr)):
                if tp is not None:
                    setattr(o, name, tp)
                else:
                    q, dt = L.act_ptr(t)
                    if t is not None and dt != io.y_dtype:
                        raise L.VisingerHipError(f"ConvOp.forward: {name} is {t.dtype}, y is {y.dtype}")
                    setattr(o, name, q)
        o.y_bs, o.res_bs, o.acc_bs = y_bs, res_bs, acc_bs
        o.scale, o.out_act, o.out_mask, o.mode = scale, out_act, int(bool(out_mask)), mode
        if out1 is not None:
            o1 = io.out[1]
            o1.y = L.ptr(out1["y"])
            o1.res = L.ptr(out1.get("res"))
            o1.acc = L.ptr(out1.get("acc"))
            o1.y_bs, o1.res_bs, o1.acc_bs = out1.get("y_bs", 0), out1.get("res_bs", 0), out1.get("acc_bs", 0)
            o1.scale, o1.out_act = out1.get("scale", 1.0), out1.get("out_act", L.OUT_NONE)
            o1.out_mask, o1.mode = int(bool(out1.get("out_mask", False))), out1.get("mode", L.MODE_LINEAR)
        io.pair_mode = pair_mode
        io.logdet = L.ptr(logdet)
        if PROFILER.enabled:
            e0, e1 = PROFILER.events()
            e0.record()
            L.check(self.lib.vs_conv_forward(self.h, ctypes.byref(io), L.stream_ptr()))
            e1.record()
            self._last_kernel = self.last_kernel()
            passes = 1 + (res is not None or res_ptr is not None) + (acc is not None or acc_ptr is not None)
            nb = B * ((2.0 if io.x_dtype else 4.0) * self.c_in * T + (2.0 if io.y_dtype else 4.0) * passes * self.rows_out * Tout)   # x once, residual / accumulate inputs once, y once
            PROFILER.records.append((self._last_kernel, self.algorithmic_flops(B, T), nb, e0, e1))
        else:
            L.check(self.lib.vs_conv_forward(self.h, ctypes.byref(io), L.stream_ptr()))
            self._last_kernel = self.last_kernel()
        return y


def _rebuild_conv_op(args, math):
    op = ConvOp.__new__(ConvOp)
    op.kind, op.c_in, op.c_out, op.k, op.dil, op.pad, op.flags = args
    op._h, op._pending_math, op._wkey, op._last_kernel = None, math, None, ""
    return op


def weightnorm_fold(v, g):
    """w = g * v / ||v|| over all dims but 0 (a12)."""
    lib = L.require_gpu()
    v = v.contiguous()
    w = torch.empty_like(v)
    rows = v.shape[0]
    L.check(lib.vs_weightnorm_fold(L.ptr(v), L.ptr(g.contiguous()), L.ptr(w), rows, v.numel() // rows, L.stream_ptr()))
    return w


def rel_attention(qkv, n_heads, rel_k=None, rel_v=None, mask=None, window_size=None, out=None, math=L.MATH_F32, ksplit_auto=True):
    """a6: attention core on a fused [B, 3C, T] q|k|v buffer -> [B, C, T].  math = L.MATH_BF16: both GEMMs on the bf16 matrix
    instruction (the arithmetic of the q / k / v convs around it, BASELINE config 5); otherwise exact fp32."""
    lib = L.require_gpu()
    B, C3, T = qkv.shape
    C = C3 // 3
    if out is None:
        out = torch.empty((B, C, T), device=qkv.device, dtype=torch.float32)
    ws = -1 if window_size is None else int(window_size)
    # launches that do not fill the chip (single utterances: B * heads * ceil(T / 128) workgroups, each walking every key tile) split the
    # keys over several workgroups and merge the partial rows in a second kernel (vs_relattn_fwd_ksplit)
    nwg = B * n_heads * -(-T // 128)
    ksplit = 1
    if ksplit_auto and nwg < 128 and T >= 256 and not L.switch("VS_NO_ATTN_KSPLIT"):
        ksplit = max(1, min(8, 256 // nwg, T // 128))
        if L.switch("VS_ATTN_KSPLIT"):
            ksplit = max(1, min(16, L.switch("VS_ATTN_KSPLIT"), T // 64))
    work = None
    if ksplit > 1:
        R = 0 if rel_k is None else rel_k.shape[1]
        work = torch.empty((B * n_heads * ksplit * (C // n_heads + 2 + R) * T,), device=qkv.device, dtype=torch.float32)
    # plain-bf16 arithmetic on long sequences: scratch for the K / V tile images the library then packs once per launch instead of once
    # per query block (vs_relattn_fwd_work); 0 bytes: not applicable
    kv_bytes = int(lib.vs_relattn_kv_work_bytes(B, n_heads, C // n_heads, T, int(math)))
    kv_work = torch.empty((kv_bytes,), device=qkv.device, dtype=torch.uint8) if kv_bytes else None
    if PROFILER.enabled:
        e0, e1 = PROFILER.events()
        e0.record()
    L.check(lib.vs_relattn_fwd_work(_off(qkv, 0), _off(qkv, C * T), _off(qkv, 2 * C * T), C3 * T,
                                    L.ptr(None if rel_k is None else rel_k.detach().contiguous()),
                                    L.ptr(None if rel_v is None else rel_v.detach().contiguous()), L.ptr(mask), L.ptr(out),
                                    C * T, B, n_heads, C // n_heads, T, ws, 1 if rel_k is None else rel_k.shape[0],
                                    int(math), L.ptr(work), ksplit,
                                    None if kv_work is None else ctypes.c_void_p(kv_work.data_ptr()), kv_bytes, L.stream_ptr()))
    if PROFILER.enabled:
        e1.record()      # algorithmic work: Q K^T and P V over the full [T, T] score matrix = 4 * T * T * k_channels per head
        PROFILER.records.append((lib.vs_last_kernel_name().decode(), 4.0 * B * C * T * T, 4.0 * B * 4 * C * T, e0, e1))
    return out


def layernorm_c(a, gamma, beta, r=None, g=None, mask=None, eps=1e-4, out=None):
    """a7: y = ((LN_C(a + r) * gamma + beta) + g) * mask over [B, C, T]."""
    lib = L.require_gpu()
    B, C, T = a.shape
    if out is None:
        out = torch.empty_like(a)
    g_bs, g_ts = 0, 0
    if g is not None:
        g_ts = 1 if g.shape[-1] == T else 0
        g_bs = C * (T if g_ts else 1)
    L.check(lib.vs_layernorm_c_fwd(L.ptr(a), L.ptr(r), L.ptr(gamma.detach().contiguous()), L.ptr(beta.detach().contiguous()),
                                   L.ptr(g), g_bs, g_ts, L.ptr(mask), L.ptr(out), B, C, T, eps, L.stream_ptr()))
    return out


def _i64ptr(t):
    if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
        raise L.VisingerHipError("expected a contiguous int64 tensor on the GPU")
    return ctypes.c_void_p(t.data_ptr())


def expand_states(h, mel2token, h_channels_first=False, out_channels_first=False):
    """a9: frame expansion by the 1-based mel2token index (0 = padding -> zeros)."""
    lib = L.require_gpu()
    h = h.contiguous().float()
    B = h.shape[0]
    Tp, C = (h.shape[2], h.shape[1]) if h_channels_first else (h.shape[1], h.shape[2])
    T = mel2token.shape[1]
    out = torch.empty((B, C, T) if out_channels_first else (B, T, C), device=h.device, dtype=torch.float32)
    L.check(lib.vs_expand_states(L.ptr(h), _i64ptr(mel2token.contiguous()), L.ptr(out), B, Tp, T, C, int(h_channels_first),
                                 int(out_channels_first), L.stream_ptr()))
    return out


def make_positions(x, padding_idx):
    """a9: cumsum(x != pad) * (x != pad) + pad as int64, x: [B, T] fp32."""
    lib = L.require_gpu()
    x = x.contiguous().float()
    B, T = x.shape
    pos = torch.empty((B, T), device=x.device, dtype=torch.int64)
    L.check(lib.vs_make_positions(L.ptr(x), ctypes.c_void_p(pos.data_ptr()), B, T, int(padding_idx), L.stream_ptr()))
    return pos


def slice_segments(x, ids_str, segment_size):
    """a9: out[b, :, s] = x[b, :, ids_str[b] + s]."""
    lib = L.require_gpu()
    x = x.contiguous().float()
    B, C, T = x.shape
    ids = ids_str.to(device=x.device, dtype=torch.int64).contiguous()
    out = torch.empty((B, C, segment_size), device=x.device, dtype=torch.float32)
    L.check(lib.vs_slice_segments(L.ptr(x), _i64ptr(ids), L.ptr(out), B, C, T, segment_size, L.stream_ptr()))
    return out


def mel2token_to_dur(mel2token, T_txt, max_dur=None):
    """a9 / 8f-3: dur[b, i-1] = number of frames aligned to token i (utils/audio/align.py:105-129), int64, bit-exact."""
    lib = L.require_gpu()
    m = mel2token.to(dtype=torch.int64).contiguous()
    B, T = m.shape
    dur = torch.empty((B, int(T_txt)), device=m.device, dtype=torch.int64)
    L.check(lib.vs_mel2token_to_dur(_i64ptr(m), _i64ptr(dur), B, T, int(T_txt), -1 if max_dur is None else int(max_dur),
                                    L.stream_ptr()))
    return dur


def bias_grad(gy):
    """8f-1: gb[c] = sum_{b,t} gy[b, c, t] (one deterministic launch: vs_bias_grad); gy contiguous fp32 [B, C, T]"""
    lib = L.require_gpu()
    gy = gy.contiguous().float()
    B, C, T = gy.shape
    gb = torch.empty((C,), device=gy.device, dtype=torch.float32)
    L.check(lib.vs_bias_grad(L.ptr(gy), L.ptr(gb), B, C, T, L.stream_ptr()))
    return gb


def conv_wgrad(gy, x, k, dil=1, pad=0, bias=False):
    """8f-1: weight gradient of a stride-1 conv, gw[co, ci, k] = sum_{b,t} gy[b, co, t] * x[b, ci, t + k*dil - pad]
    (vs_conv_wgrad_bias: one partial plane per reduction slice, summed by its second launch).  bias=True: -> (gw, gb) with the conv's bias
    gradient gb[co] = sum_{b,t} gy[b, co, t] from the same pass over gy (one launch less than vs_bias_grad next to it, and gy is read once)."""
    if bias:
        gw = conv_wgrad(gy, x, k, dil, pad, bias=None)
        return gw if isinstance(gw, tuple) else (gw, bias_grad(gy))
    lib = L.require_gpu()
    gy, x = gy.contiguous().float(), x.contiguous().float()
    B, Cout, Tout = gy.shape
    Cin, Tin = x.shape[1], x.shape[2]
    if k == 1 and pad == 0 and Tin == Tout and Cout * Cin >= 64 * 64 and L.switch("VS_WGRAD_GEMM"):
        # A/B switch only (VS_WGRAD_GEMM=1): the library GEMM rounds 1-2 used for the 1x1 weight gradients; since round 3 they run on the
        # split-bf16 weight-gradient kernel (csrc/conv_backward.hip conv_wgrad_split_kernel<1, 1>) like every other tap count
        PROFILER.note("wgrad 1x1 (library GEMM)", 2.0 * B * Cout * Cin * Tout)
        return torch.einsum("bot,bit->oi", gy, x).unsqueeze(2)
    if Tout == 1 and Tin == 1 and k == 1 and pad == 0:
        # a conv over ONE position (the speaker / conditioning projections of a training step): gw = gy^T x, a [Cout x B] . [B x Cin] product with nothing
        # to tile along t -- vs_conv_wgrad's fp32 fallback spent 100-400 us per call on it (tools/wgrad_breakdown.py, round 4); the fp32 library GEMM
        PROFILER.note("wgrad T = 1 (library GEMM)", 2.0 * B * Cout * Cin)
        return (gy[:, :, 0].t() @ x[:, :, 0]).unsqueeze(2)
    if gy.data_ptr() % 16:       # (an offset view that is contiguous: the split kernel loads float4 rows of gy)
        gy = gy.clone()
    planes = lib.vs_conv_wgrad_planes(B, Cout, Cin, Tout, int(k))
    with_bias = bias is None                  # (asked for by the bias=True wrapper above; the special cases before this line return gw alone)
    n = Cout * Cin * k
    part = torch.empty((planes, n + (Cout if with_bias else 0)), device=x.device, dtype=torch.float32)
    out = torch.empty(n + (Cout if with_bias else 0), device=x.device, dtype=torch.float32)
    L.check(lib.vs_conv_wgrad_bias(L.ptr(gy), L.ptr(x), L.ptr(part), L.ptr(out), int(with_bias), B, Cout, Cin, Tout, Tin, int(k), int(dil), int(pad),
                                   L.stream_ptr()))
    PROFILER.note("conv_wgrad (vs_conv_wgrad)", 2.0 * B * Cout * Cin * k * Tout)
    gw = out[:n].view(Cout, Cin, k)
    return (gw, out[n:]) if with_bias else gw


def _gconv_out_len(T, k, stride, pad):
    return (T + 2 * pad - k) // stride + 1


def gconv1d_fwd(x, w, bias, stride, pad, groups):
    """a13: grouped / strided conv1d (scale discriminator), y [B, c_out, T_out]."""
    lib = L.require_gpu()
    x, w = x.contiguous().float(), w.contiguous().float()
    B, Cin, T = x.shape
    Cout, _, k = w.shape
    y = torch.empty((B, Cout, _gconv_out_len(T, k, stride, pad)), device=x.device, dtype=torch.float32)
    L.check(lib.vs_gconv1d_fwd(L.ptr(x), L.ptr(w), L.ptr(None if bias is None else bias.contiguous().float()), L.ptr(y), B, Cin, Cout, T,
                               k, stride, pad, groups, L.stream_ptr()))
    PROFILER.note("gconv_fwd_kernel", 2.0 * B * Cout * (Cin // groups) * k * y.shape[2])
    return y


def gconv1d_bwd_data(gy, w, T, stride, pad, groups):
    lib = L.require_gpu()
    gy, w = gy.contiguous().float(), w.contiguous().float()
    B, Cout, _ = gy.shape
    cig, k = w.shape[1], w.shape[2]
    gx = torch.empty((B, cig * groups, T), device=gy.device, dtype=torch.float32)
    L.check(lib.vs_gconv1d_bwd_data(L.ptr(gy), L.ptr(w), L.ptr(gx), B, cig * groups, Cout, T, k, stride, pad, groups, L.stream_ptr()))
    PROFILER.note("gconv_bwd_data_kernel", 2.0 * B * Cout * cig * k * gy.shape[2])
    return gx


def gconv1d_bwd_weight(gy, x, k, stride, pad, groups):
    lib = L.require_gpu()
    gy, x = gy.contiguous().float(), x.contiguous().float()
    B, Cout, _ = gy.shape
    Cin, T = x.shape[1], x.shape[2]
    planes = torch.empty((B, Cout, Cin // groups, k), device=x.device, dtype=torch.float32)
    L.check(lib.vs_gconv1d_bwd_weight(L.ptr(gy), L.ptr(x), L.ptr(planes), B, Cin, Cout, T, k, stride, pad, groups, L.stream_ptr()))
    PROFILER.note("gconv_bwd_weight_kernel", 2.0 * B * Cout * (Cin // groups) * k * gy.shape[2])
    return planes.sum(0)


def respair_supported(op1, op2, profitable_only=False):
    """a11: can the (conv1, conv2) pair of a resblock run as one fused launch -- and, with profitable_only, did it measure
    faster than the two launches.  fp32 MFMA arithmetic (tools/pair_bench.py with VS_CONV_MATH=0): at 32 channels x1.32-1.47 for k=3, x1.14 for
    k=7, a tie for k=11 (the F(2,3) kernel wins at dilation 1); at 64 channels only the dilated k=3 pairs (x1.06): from k=7 on
    the separate F(2,3) launches do 30 % less matrix work than the fused direct form."""
    if L.switch("VS_NO_RESPAIR") or not op1.lib.vs_respair_supported(op1.h, op2.h):
        return False
    if op1.math != op2.math:
        return False
    if not profitable_only or L.switch("VS_RESPAIR_FORCE"):
        return True
    C, k, d = op1.c_in, op1.k, op1.dil
    if op1.math == L.MATH_BF16:
        # plain bf16 operands: a sixth of the matrix work of the split engine on the same tensor passes -- every 32- / 64-channel
        # conv is far below the HBM ridge as its own launch, the fused pair moves 3 tensor passes instead of 6
        return True
    if op1.math == L.MATH_SPLIT6:
        # csrc/resblock_pair_split.hip against two launches of the split engine (tools/pair_bench.py, B=32 production shapes):
        # 32 channels x1.46-1.61 (k=3), x1.25 (k=7), x1.13 (k=11); 64 channels x1.22 (k=3), a tie at k=7, x0.91 at k=11 (its
        # 64 x 128 tile spends 9 % of both convs on halo columns)
        return C == 32 or k == 3
    return (C == 32 and (k <= 7 or d > 1)) or (C == 64 and k == 3 and d > 1)


def _handle_array(ops):
    return (ctypes.c_void_p * len(ops))(*[op.h for op in ops])


def resblock_supported(ops):
    """a11: can this chain of convs (convs1[0], convs2[0], convs1[1], ...) run as ONE launch of csrc/resblock_f16.hip"""
    if L.switch("VS_NO_RESBLOCK_FUSED") or len(ops) < 2 or len(ops) % 2:
        return False
    return bool(ops[0].lib.vs_resblock_supported(_handle_array(ops), len(ops)))


def resblock_forward(ops, x, y, acc=None, scale=1.0):
    """a11: for each pair (conv1, conv2) of `ops`: x = conv2(lrelu(conv1(lrelu(x)))) + x; y = (x [+ acc]) * scale -- one launch, the
    residual stream in registers between the pairs (csrc/resblock_f16.hip: split-f16 arithmetic on fp32 tensors, or plain bf16 operands on
    bf16-resident tensors)."""
    B, C, T = x.shape
    io = L.ConvIO()
    io.x, io.x_dtype = L.act_ptr(x)
    io.x_bs, io.B, io.T = 0, B, T
    io.in_act = L.IN_LRELU
    o = io.out[0]
    o.y, io.y_dtype = L.act_ptr(y)
    o.acc, adt = L.act_ptr(acc)
    if acc is not None and adt != io.y_dtype:
        raise L.VisingerHipError(f"resblock_forward: acc is {acc.dtype}, y is {y.dtype}")
    o.scale = scale
    lib = ops[0].lib
    if PROFILER.enabled:
        e0, e1 = PROFILER.events()
        e0.record()
        L.check(lib.vs_resblock_forward(_handle_array(ops), len(ops), ctypes.byref(io), L.stream_ptr()))
        e1.record()
        nb = float(x.element_size()) * B * C * T * (2 + (acc is not None))
        PROFILER.records.append((ops[0].last_kernel(), sum(op.algorithmic_flops(B, T) for op in ops), nb, e0, e1))
    else:
        L.check(lib.vs_resblock_forward(_handle_array(ops), len(ops), ctypes.byref(io), L.stream_ptr()))
    name = ops[0].last_kernel()
    for op in ops:
        op._last_kernel = name
    return y


def respair_forward(op1, op2, x, y, res=None, acc=None, scale=1.0):
    """a11: y = conv2(lrelu(conv1(lrelu(x)))) + res [+ acc] [* scale] in one launch (csrc/resblock_pair.hip)."""
    B, C, T = x.shape
    io = L.ConvIO()
    io.x, io.x_dtype = L.act_ptr(x)
    io.x_bs, io.B, io.T = 0, B, T
    io.in_act = L.IN_LRELU
    o = io.out[0]
    o.y, io.y_dtype = L.act_ptr(y)
    for name, t in (("res", res), ("acc", acc)):
        q, dt = L.act_ptr(t)
        if t is not None and dt != io.y_dtype:
            raise L.VisingerHipError(f"respair_forward: {name} is {t.dtype}, y is {y.dtype}")
        setattr(o, name, q)
    o.scale = scale
    if PROFILER.enabled:
        e0, e1 = PROFILER.events()
        e0.record()
        L.check(op1.lib.vs_respair_forward(op1.h, op2.h, ctypes.byref(io), L.stream_ptr()))
        e1.record()
        nb = (2.0 if io.y_dtype else 4.0) * B * C * T * (2 + (res is not None) + (acc is not None))
        PROFILER.records.append((op1.last_kernel(), op1.algorithmic_flops(B, T) + op2.algorithmic_flops(B, T), nb, e0, e1))
    else:
        L.check(op1.lib.vs_respair_forward(op1.h, op2.h, ctypes.byref(io), L.stream_ptr()))
    op1._last_kernel = op2._last_kernel = op1.last_kernel()
    return y
