"""ctypes binding of libvisinger_hip.so (the C ABI declared in include/visinger_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing or no MI355X is
visible, importing callers get a loud RuntimeError instead of a silently different code path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_LIB") or os.path.join(_HERE, "csrc", "libvisinger_hip.so")   # VS_LIB: A/B builds

VS_OK = 0
# enum vs_conv_kind
CONV1D, CONV_TRANSPOSE1D, CONV1D_PAIRED = 0, 1, 2
# enum vs_in_act / vs_out_act / vs_pair_mode / vs_out_mode
IN_NONE, IN_LRELU, IN_MASK, IN_LRELU_MASK = 0, 1, 2, 3
OUT_NONE, OUT_TANH, OUT_RELU = 0, 1, 2
PAIR_GATE, PAIR_COUPLING_FWD, PAIR_COUPLING_INV = 0, 1, 2
MODE_LINEAR, MODE_COUPLING_MEAN_FWD, MODE_COUPLING_MEAN_INV = 0, 1, 2
FLIP_IN, FLIP_OUT, CONV_ADJOINT = 1, 2, 4
# enum vs_conv_math
MATH_F32, MATH_BF16, MATH_SPLIT3, MATH_SPLIT6 = 0, 1, 3, 6
DTYPE_F32, DTYPE_BF16 = 0, 1

EXPECTED_ABI = 7          # include/visinger_hip.h VS_ABI_VERSION this binding was written against

_f32p = ctypes.c_void_p


class ConvOut(ctypes.Structure):
    _fields_ = [("y", _f32p), ("res", _f32p), ("acc", _f32p),
                ("y_bs", ctypes.c_int64), ("res_bs", ctypes.c_int64), ("acc_bs", ctypes.c_int64),
                ("scale", ctypes.c_float), ("out_act", ctypes.c_int), ("out_mask", ctypes.c_int),
                ("mode", ctypes.c_int)]


class ConvIO(ctypes.Structure):
    _fields_ = [("x", _f32p), ("x_bs", ctypes.c_int64), ("B", ctypes.c_int64), ("T", ctypes.c_int64),
                ("in_act", ctypes.c_int), ("x_dtype", ctypes.c_int), ("mask", _f32p), ("bias_b", _f32p), ("bias_b_bs", ctypes.c_int64),
                ("split_row", ctypes.c_int), ("y_dtype", ctypes.c_int), ("out", ConvOut * 2), ("pair_mode", ctypes.c_int),
                ("logdet", _f32p)]


_lib = None


class VisingerHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle of libvisinger_hip.so.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime this library binds to (shared device
    # pointers and streams), so torch is always loaded first.  Loading the .so before torch pulls in a second
    # runtime from /opt/rocm and every hipMalloc then fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise VisingerHipError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m visinger_amd.csrc.build, or "
            f"__graft_entry__.build()).  visinger_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.vs_last_error.restype = ctypes.c_char_p
    L.vs_last_kernel_name.restype = ctypes.c_char_p
    L.vs_abi_version.restype = ctypes.c_int
    got = L.vs_abi_version()
    if got != EXPECTED_ABI:       # a stale or partially rebuilt .so would be called with the wrong argument layout (ConvIO, vs_relattn_fwd)
        raise VisingerHipError(f"{LIB_PATH} exports ABI version {got}, this package binds version {EXPECTED_ABI}: rebuild it "
                               f"(python -m visinger_amd.csrc.build --force)")
    # ... and one built from other SOURCES (same ABI number, different kernels: an object file that survived a checkout) must not run
    # either: the library carries the sha256 of what it was compiled from, recomputed here over the tree it sits in (skipped for a
    # VS_LIB override -- an A/B build of another tree -- and for an installed library without its sources)
    L.vs_source_hash.restype = ctypes.c_char_p
    if not os.environ.get("VS_LIB") and not os.environ.get("VS_SKIP_SOURCE_HASH"):
        from .csrc import build as _build
        if _build.sources():
            built, now = L.vs_source_hash().decode(), _build.source_hash()
            if built != now:
                raise VisingerHipError(f"{LIB_PATH} was built from other sources (library {built[:12]}, tree {now[:12]}): rebuild it "
                                       f"(python -m visinger_amd.csrc.build)")
    L.vs_device_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.vs_set_option.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.vs_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]
    L.vs_reset_option.argtypes = [ctypes.c_char_p]
    L.vs_weightnorm_fold.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.vs_conv_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
    L.vs_conv_destroy.argtypes = [ctypes.c_void_p]
    L.vs_conv_destroy.restype = None
    L.vs_conv_set_weights.argtypes = [ctypes.c_void_p, _f32p, _f32p, _f32p, ctypes.c_void_p]
    L.vs_conv_set_weights_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _f32p, _f32p, ctypes.c_void_p]
    L.vs_conv_set_math.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.vs_conv_get_math.argtypes = [ctypes.c_void_p]
    L.vs_conv_forward.argtypes = [ctypes.c_void_p, ctypes.POINTER(ConvIO), ctypes.c_void_p]
    L.vs_conv_out_len.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.vs_conv_out_len.restype = ctypes.c_int64
    i64, ci, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    L.vs_relattn_fwd.argtypes = [_f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, i64, i64, ci, ci, i64, ci, ci, ci, vp]
    L.vs_relattn_fwd_ksplit.argtypes = [_f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, i64, i64, ci, ci, i64, ci, ci, ci, _f32p, ci, vp]
    L.vs_relattn_fwd_work.argtypes = [_f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, i64, i64, ci, ci, i64, ci, ci, ci, _f32p, ci, vp,
                                      ctypes.c_size_t, vp]
    L.vs_relattn_kv_work_bytes.argtypes = [i64, ci, ci, i64, ci]
    L.vs_relattn_kv_work_bytes.restype = ctypes.c_size_t
    L.vs_layernorm_c_fwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, i64, ci, _f32p, _f32p, i64, i64, i64,
                                     ctypes.c_float, vp]
    L.vs_gate_fwd.argtypes = [_f32p, _f32p, i64, _f32p, i64, i64, i64, vp]
    L.vs_gate_bwd.argtypes = [_f32p, _f32p, i64, _f32p, _f32p, _f32p, i64, i64, i64, i64, vp]
    L.vs_bias_grad.argtypes = [_f32p, _f32p, i64, i64, i64, vp]
    L.vs_conv_set_weights_batch.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ci, vp]
    L.vs_weight_norm_multi_fwd.argtypes = [vp, i64, i64, vp]
    L.vs_wn_step_fwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, i64, i64, i64, vp]
    L.vs_wn_step_bwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, i64, i64, i64, vp]
    L.vs_l1_mean_fwd.argtypes = [_f32p, _f32p, _f32p, _f32p, i64, vp]
    L.vs_l1_mean_bwd.argtypes = [_f32p, _f32p, _f32p, _f32p, i64, vp]
    L.vs_weight_norm_multi_bwd.argtypes = [vp, vp, i64, i64, vp]
    L.vs_layernorm_c_bwd.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, i64, i64, i64, ctypes.c_float, vp]
    u64, cf = ctypes.c_uint64, ctypes.c_float
    L.vs_relattn_train_fwd.argtypes = [_f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, i64, _f32p, i64, ci, ci, i64, ci, ci, cf, u64, vp]
    L.vs_relattn_train_bwd.argtypes = [_f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, _f32p, i64, _f32p, _f32p, _f32p, _f32p, i64,
                                       _f32p, _f32p, _f32p, i64, ci, ci, i64, ci, ci, cf, u64, vp]
    L.vs_phase_stack.argtypes = [_f32p, i64, i64, i64, _f32p, i64, i64, i64, ci, ci, i64, i64, vp]
    L.vs_phase_items.argtypes = [_f32p, _f32p, i64, i64, i64, i64, i64, ci, vp]
    L.vs_phase_unstack.argtypes = [_f32p, i64, _f32p, i64, i64, i64, ci, ci, i64, vp]
    L.vs_spec_power_fwd.argtypes = [_f32p, _f32p, i64, i64, i64, vp]
    L.vs_spec_power_bwd.argtypes = [_f32p, _f32p, _f32p, i64, i64, i64, vp]
    L.vs_expand_states.argtypes = [_f32p, vp, _f32p, i64, i64, i64, i64, ci, ci, vp]
    L.vs_make_positions.argtypes = [_f32p, vp, i64, i64, i64, vp]
    L.vs_slice_segments.argtypes = [_f32p, vp, _f32p, i64, i64, i64, i64, vp]
    L.vs_mel2token_to_dur.argtypes = [vp, vp, i64, i64, i64, i64, vp]
    L.vs_respair_supported.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.vs_respair_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ConvIO), ctypes.c_void_p]
    L.vs_resblock_supported.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    L.vs_resblock_forward.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ConvIO), ctypes.c_void_p]
    L.vs_conv_wgrad.argtypes = [_f32p, _f32p, _f32p, i64, i64, i64, i64, i64, ci, ci, ci, vp]
    L.vs_conv_wgrad_planes.argtypes = [i64, i64, i64, i64, ci]
    L.vs_conv_wgrad_bias.argtypes = [_f32p, _f32p, _f32p, _f32p, ci, i64, i64, i64, i64, i64, ci, ci, ci, vp]
    L.vs_gconv1d_fwd.argtypes = [_f32p, _f32p, _f32p, _f32p, i64, i64, i64, i64, ci, ci, ci, ci, vp]
    L.vs_gconv1d_bwd_data.argtypes = [_f32p, _f32p, _f32p, i64, i64, i64, i64, ci, ci, ci, ci, vp]
    L.vs_gconv1d_bwd_weight.argtypes = [_f32p, _f32p, _f32p, i64, i64, i64, i64, ci, ci, ci, ci, vp]
    _lib = L
    return L


# Switches of the Python layer (A/B and debugging), read from the environment ONCE at import like the library's own
# (vs_set_option): no os.environ lookup on any forward / backward path.  set_option() changes either kind by name.
PY_SWITCHES = {name: (int(os.environ[name]) if os.environ.get(name, "").lstrip("-").isdigit() else int(bool(os.environ.get(name))))
               for name in ("VS_NO_TRAIN_FUSED", "VS_NO_TRAIN_ATTN", "VS_NO_FUSED_QKV", "VS_NO_ATTN_KSPLIT", "VS_ATTN_KSPLIT",
                            "VS_WGRAD_GEMM", "VS_NO_PAIR_PACK", "VS_NO_RESPAIR", "VS_RESPAIR_FORCE", "VS_NO_RESBLOCK_FUSED", "VS_RESBLOCK_PAIRS", "VS_NO_PACK_CACHE", "VS_NO_FUSED_ADAMW", "VS_NO_WEIGHT_BANK")}


def switch(name):
    return PY_SWITCHES[name]


def set_option(name, value):
    """Set a dispatch switch by its (environment-variable) name: a Python-layer switch or one of the library's (vs_set_option)."""
    if name in PY_SWITCHES:
        PY_SWITCHES[name] = int(value)
    else:
        check(lib().vs_set_option(name.encode(), int(value)))


def get_option(name):
    if name in PY_SWITCHES:
        return PY_SWITCHES[name]
    v = ctypes.c_longlong(0)
    check(lib().vs_get_option(name.encode(), ctypes.byref(v)))
    return int(v.value)


class options:
    """with options(VS_NO_WINO=1, VS_NO_KTAP=1): ...  -- switches set for the block, previous values restored after it"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def check(rc):
    if rc != VS_OK:
        raise VisingerHipError(f"libvisinger_hip error {rc}: {lib().vs_last_error().decode()}")


_gpu_ok = False


def require_gpu():
    """Fail loudly when the HIP path cannot run (no GPU / library not built).  (The positive answer is cached: this sits on
    every launch.)"""
    global _gpu_ok
    if _gpu_ok:
        return _lib
    import torch
    L = lib()
    if not torch.cuda.is_available():
        raise VisingerHipError("visinger_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU path.")
    _gpu_ok = True
    return L


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def act_ptr(t):
    """(device pointer, vs_dtype) of a contiguous fp32 or bf16 activation tensor (None -> (NULL, fp32))."""
    if t is None:
        return None, DTYPE_F32
    import torch
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous()):
        raise VisingerHipError(f"expected a contiguous fp32 / bf16 tensor on the GPU, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return ctypes.c_void_p(t.data_ptr()), (DTYPE_BF16 if t.dtype == torch.bfloat16 else DTYPE_F32)


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    import torch
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise VisingerHipError(f"expected a contiguous fp32 tensor on the GPU, got {t.dtype} {t.device} "
                               f"contiguous={t.is_contiguous()}")
    return ctypes.c_void_p(t.data_ptr())
