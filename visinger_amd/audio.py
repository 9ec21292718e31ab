"""On-GPU linear / mel spectrograms: what the reference gets from torchaudio in
``utils/audio/mel_processing.py:15-38`` (MelSpectrogram / Spectrogram with n_fft 2048, win 1200, hop 300, power 2,
centre + reflect padding, HTK mel scale, ``log(x + 1e-3)``, last frame dropped), as matrix work on the HIP conv engine:

* the framed, windowed DFT is ONE strided conv of the reflect-padded waveform: frame t, bin f is
  ``sum_n hann[n] x_pad[t hop + o + n] (cos, -sin)(2 pi f (n + o) / n_fft)`` with ``o = (n_fft - win) / 2`` -- a conv with 2F output
  rows (F real, F imaginary), ``win`` taps and stride ``hop``.  Its ``hop`` de-interleaved phases stacked as channels make it a
  stride-1 conv with ``ceil(win / hop)`` taps (4 for 1200 / 300) and ``hop`` input channels: the shape the MFMA engine is built for
  (`vs_conv_forward`, split-bf16 x6 arithmetic: fp32-class).  160 GFLOP for a B = 32, T = 1024 batch -- under a millisecond of matrix
  pipe, where the FFT's advantage in FLOPs does not matter and no transform library is linked;
* ``power = re^2 + im^2``: `vs_spec_power_fwd` / `vs_spec_power_bwd` (csrc/audio_ops.hip);
* the mel projection is a 1x1 conv with the HTK filterbank on the same engine.

Both transforms are differentiable (the mel loss of the training step is taken on GENERATED audio, tasks/base.py:232-238): the
backward of the DFT is the engine's conv with the transposed, tap-reversed basis.

PARITY UNPINNED w.r.t. torchaudio: it is a third-party dependency that is neither vendored under the reference tree nor installed
here, and the reference ships no test vector for it; this follows torchaudio 0.11's documented defaults (hann window,
center=True, pad_mode='reflect', normalized=False, onesided, mel_scale='htk', norm=None) and is tested against the fp64 framed-DFT
oracle (tests/test_audio_gpu.py) and against ``torch.stft`` (`stft_spectrogram` below: a cross-check, not on the product path).
No CPU fallback: CPU tensors are refused."""
import math

import torch
import torch.nn.functional as F

from . import _lib as L
from .ops import ConvOp


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(mel_scale='htk', norm=None) -> [n_freqs, n_mels]"""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def dft_basis(n_fft, win_length):
    """[2F, win]: rows f < F the real basis hann[n] cos(2 pi f (n + o) / n_fft), rows F + f the imaginary one -hann[n] sin(...), on
    the support of the centred window (o = (n_fft - win) / 2; torch.stft pads the window to n_fft on both sides).  Angles are
    reduced exactly in integers, evaluated in fp64 and rounded once to fp32."""
    Fq = n_fft // 2 + 1
    o = (n_fft - win_length) // 2
    n = torch.arange(win_length, dtype=torch.int64)
    f = torch.arange(Fq, dtype=torch.int64)
    ang = ((f[:, None] * (n[None, :] + o)) % n_fft).double() * (2.0 * math.pi / n_fft)
    win = torch.hann_window(win_length, periodic=True, dtype=torch.float64)
    return torch.cat([torch.cos(ang) * win, -torch.sin(ang) * win], 0).float()


class _Plan:
    """Engine handles and packed bases of one (n_fft, win, hop) transform on one device."""

    def __init__(self, n_fft, win, hop, device):
        self.n_fft, self.win, self.hop, self.F = n_fft, win, hop, n_fft // 2 + 1
        self.Q = -(-win // hop)                                            # taps of the phase-stacked conv
        w = dft_basis(n_fft, win)                                          # [2F, win]
        w2 = F.pad(w, (0, self.Q * hop - win)).view(2 * self.F, self.Q, hop).permute(0, 2, 1).contiguous()   # [2F, hop, Q]: w2[r, p, q] = w[r, q hop + p]
        self.w_fwd = w2.to(device)
        self.w_bwd = w2.flip(2).transpose(0, 1).contiguous().to(device)   # [hop, 2F, Q]: grad-input conv (full correlation)
        # the transform is fp32-class whatever arithmetic the model's convs were switched to (VS_CONV_MATH / set_math)
        self.fwd = ConvOp(L.CONV1D, hop, 2 * self.F, self.Q, 1, 0).set_math(L.MATH_SPLIT6)
        self.bwd = ConvOp(L.CONV1D, 2 * self.F, hop, self.Q, 1, self.Q - 1).set_math(L.MATH_SPLIT6)
        self.fwd.set_weights(self.w_fwd, None, None)
        self.bwd.set_weights(self.w_bwd, None, None)
        self.mel = {}

    def mel_ops(self, key, fb, device):
        if key not in self.mel:
            n_mels = fb.shape[1]
            wf = fb.t().contiguous().unsqueeze(2).to(device)               # [n_mels, F, 1]
            wb = fb.contiguous().unsqueeze(2).to(device)                   # [F, n_mels, 1]
            f = ConvOp(L.CONV1D, self.F, n_mels, 1, 1, 0).set_math(L.MATH_SPLIT6)
            b = ConvOp(L.CONV1D, n_mels, self.F, 1, 1, 0).set_math(L.MATH_SPLIT6)
            f.set_weights(wf, None, None)
            b.set_weights(wb, None, None)
            self.mel[key] = (f, b, wf, wb)
        return self.mel[key][:2]


_PLANS = {}


def _plan(n_fft, win, hop, device):
    key = (n_fft, win, hop, device.index)
    if key not in _PLANS:
        _PLANS[key] = _Plan(n_fft, win, hop, device)
    return _PLANS[key]


class _FramedDftFn(torch.autograd.Function):
    """x [B, Lx] (the padded waveform from the first sample of frame 0's window support) -> y [B, 2F, T] on the conv engine."""

    @staticmethod
    def forward(ctx, x, plan, T):
        B, Lx = x.shape
        hop, Q = plan.hop, plan.Q
        Hq = T + Q - 1
        need = Hq * hop
        xp = F.pad(x, (0, need - Lx)) if Lx < need else x[:, :need]
        X = xp.reshape(B, Hq, hop).transpose(1, 2).contiguous()             # [B, hop, Hq]: channel p holds the phase-p samples
        if T >= 96:
            y = plan.fwd.forward(X)                                         # [B, 2F, T]
        else:
            # short items (the 32-frame segments of the training step): the engine tiles time per item, so the items are laid end to
            # end as one sequence; the Q - 1 outputs that straddle two items are discarded
            XF = X.transpose(0, 1).reshape(1, hop, B * Hq).contiguous()
            yF = plan.fwd.forward(XF)                                       # [1, 2F, B Hq - Q + 1]
            y = F.pad(yF, (0, Q - 1)).view(2 * plan.F, B, Hq)[:, :, :T].transpose(0, 1).contiguous()
        ctx.plan, ctx.cfg = plan, (B, Lx, T, Hq)
        return y

    @staticmethod
    def backward(ctx, gy):
        plan = ctx.plan
        B, Lx, T, Hq = ctx.cfg
        hop, Q = plan.hop, plan.Q
        gy = gy.contiguous().float()
        if T >= 96:
            gX = plan.bwd.forward(gy)                                       # [B, hop, T + Q - 1 = Hq]
        else:
            gyF = torch.zeros((2 * plan.F, B, Hq), device=gy.device, dtype=torch.float32)
            gyF[:, :, :T] = gy.transpose(0, 1)
            gXF = plan.bwd.forward(gyF.view(1, 2 * plan.F, B * Hq))[:, :, :B * Hq]
            gX = gXF.reshape(hop, B, Hq).transpose(0, 1)
        gx = gX.transpose(1, 2).reshape(B, Hq * hop)
        gx = F.pad(gx, (0, Lx - Hq * hop)) if Lx > Hq * hop else gx[:, :Lx]
        return gx.contiguous(), None, None


class _PowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        lib = L.require_gpu()
        B, F2, T = y.shape
        p = torch.empty((B, F2 // 2, T), device=y.device, dtype=torch.float32)
        L.check(lib.vs_spec_power_fwd(L.ptr(y), L.ptr(p), B, F2 // 2, T, L.stream_ptr()))
        ctx.save_for_backward(y)
        return p

    @staticmethod
    def backward(ctx, dp):
        lib = L.require_gpu()
        (y,) = ctx.saved_tensors
        B, F2, T = y.shape
        dy = torch.empty_like(y)
        L.check(lib.vs_spec_power_bwd(L.ptr(y), L.ptr(dp.contiguous().float()), L.ptr(dy), B, F2 // 2, T, L.stream_ptr()))
        return dy


class _FixedConvFn(torch.autograd.Function):
    """y = conv(x, W) for a constant W held by two engine handles (forward, and grad-input with the transposed weights)"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return fwd.forward(x.contiguous())

    @staticmethod
    def backward(ctx, gy):
        return ctx.bwd.forward(gy.contiguous().float()), None, None


def _power_bft(wav, n_fft, win_length, hop_length):
    """wav [B, L] (CUDA) -> (power spectrogram [B, F, T] with T = L // hop frames -- the last of torch.stft's 1 + L // hop frames is the
    one the reference drops -- , plan)"""
    if not wav.is_cuda:
        raise RuntimeError("visinger_amd.audio: the spectrograms run on the MI355X (HIP conv engine); there is no CPU path")
    L.require_gpu()
    wav = wav.float()
    plan = _plan(n_fft, win_length, hop_length, wav.device)
    T = wav.shape[1] // hop_length
    xp = F.pad(wav.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    o = (n_fft - win_length) // 2
    y = _FramedDftFn.apply(xp[:, o:].contiguous(), plan, T)
    return _PowerFn.apply(y), plan


def linear_spectrogram(wav, n_fft=2048, win_length=1200, hop_length=300, power=2.0):
    """wav [B, L] -> [B, T, n_fft/2+1] (power spectrogram, last frame dropped as the reference does)."""
    p, _ = _power_bft(wav, n_fft, win_length, hop_length)
    if power != 2.0:
        p = p.pow(power / 2.0)
    return p.transpose(1, 2)


def mel_spectrogram(wav, sample_rate=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=128, f_min=20.0,
                    f_max=12000.0, eps=1e-3):
    """wav [B, L] -> log-mel [B, T, n_mels]"""
    p, plan = _power_bft(wav, n_fft, win_length, hop_length)
    key = (n_mels, float(f_min), float(f_max), sample_rate)
    if key not in plan.mel:
        plan.mel_ops(key, mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate), wav.device)
    fwd, bwd = plan.mel[key][:2]
    return torch.log(_FixedConvFn.apply(p, fwd, bwd) + eps).transpose(1, 2)


# ------------------------------------------------------------------------------------------------------------------
# cross-check (tests only): the same transforms on torch.stft


def stft_spectrogram(wav, n_fft=2048, win_length=1200, hop_length=300, power=2.0):
    """`linear_spectrogram` restated on ``torch.stft`` (rocFFT on the device, pocketfft on the host): an independent statement of the
    same definition, used by the tests next to the fp64 oracle.  Not called by the package."""
    window = torch.hann_window(win_length, device=wav.device, dtype=wav.dtype)
    spec = torch.stft(wav, n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return spec.abs().pow(power)[..., :-1].transpose(1, 2)


def stft_mel_spectrogram(wav, sample_rate=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=128, f_min=20.0,
                         f_max=12000.0, eps=1e-3):
    lin = stft_spectrogram(wav, n_fft, win_length, hop_length)
    fb = mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate).to(device=wav.device, dtype=wav.dtype)
    return torch.log(lin @ fb + eps)
