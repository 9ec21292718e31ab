"""MI355X-native mirror of the reference's modules/visinger/decoder.py:13-137 -- the HiFi-GAN generator with
multi-receptive-field resblocks (Generator, ResBlock1, ResBlock2).  This is where 90-97 % of the synthesis FLOPs
are.

Fusion plan (no elementwise kernel touches HBM):
  * every leaky_relu is applied while the consuming conv stages its input into LDS;
  * `x = xt + x` is the residual input of the second conv of each pair;
  * the MRF sum `xs += resblock(x)` and the final `/ num_kernels` ride on the epilogue of the LAST conv of each
    resblock (accumulate input + scale);
  * `conv_pre(x) + cond(g)` : cond(g) is a per-item bias of conv_pre;  leaky_relu + conv_post + tanh is one launch;
  * ConvTranspose1d runs as a polyphase conv (no zero-stuffing, no multiplies by structural zeros).
"""
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm, remove_weight_norm

from ... import _lib as L
from ... import autograd
from ...ops import resblock_forward, resblock_supported, respair_forward, respair_supported
from ..commons.utils import init_weights, get_padding
from ..hipconv import HipConv1d, HipConvTranspose1d, mask2d, _forward_only_guard

LRELU_SLOPE = 0.1


def _wn(conv):
    """old-style weight norm (weight_g / weight_v parameters: the reference checkpoints' layout)"""
    return weight_norm(conv)


def _same_convs(channels, kernel_size, dilations):
    """weight-normed channels -> channels convs with length-preserving padding, N(0, 0.01)-initialised (decoder.py:72-87)"""
    convs = nn.ModuleList(_wn(HipConv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)))
                          for d in dilations)
    convs.apply(init_weights)
    return convs


def _strip(convs):
    for conv in convs:
        remove_weight_norm(conv)


class Generator(nn.Module):
    """decoder.py:13-65"""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        block_cls = {"1": ResBlock1}.get(resblock, ResBlock2)
        widths = [upsample_initial_channel >> i for i in range(self.num_upsamples + 1)]      # channel count halves per stage
        self.conv_pre = HipConv1d(initial_channel, widths[0], 7, 1, padding=3)
        # stage i: weight-normed transposed conv widths[i] -> widths[i+1] (stride u, "same * u" padding), then one resblock per
        # (kernel size, dilation tuple), all on widths[i+1] channels, registered flat: resblocks[i * num_kernels + j]
        self.ups = nn.ModuleList(_wn(HipConvTranspose1d(widths[i], widths[i + 1], k, u, padding=(k - u) // 2))
                                 for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)))
        self.resblocks = nn.ModuleList(block_cls(widths[i + 1], k, d) for i in range(self.num_upsamples)
                                       for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes))
        self.conv_post = HipConv1d(widths[-1], 1, 7, 1, padding=3, bias=False)
        self.ups.apply(init_weights)
        if gin_channels != 0:
            self.cond = HipConv1d(gin_channels, widths[0], 1)

    def forward(self, x, g=None, x_mask=None):
        """decoder.py:40-59.  `x_mask` ([B, 1, T] frame mask; not a reference argument) makes a PADDED batch exact: the reference
        decodes one utterance at a time (tasks/visinger.py:244-263), so an item's last frames see zero padding beyond its own
        end, whereas in a padded batch the unmasked generator lets conv_pre's bias + speaker condition in the padding leak back
        through the receptive field.  With the mask every conv reads zeros beyond the item's length at its stage's resolution,
        i.e. each item's valid samples equal its standalone decode."""
        if autograd.training_path(self):
            return autograd.generator(self, x, g)
        _forward_only_guard(self)
        x = x.contiguous().float()
        B, _, T = x.shape
        # (the caller passes x_mask only for batches that really hold padding: no device-side check here, it would be a host
        # synchronisation in the middle of a capturable step)
        mask = mask2d(x_mask, B, T) if x_mask is not None else None
        cb = None
        if g is not None:
            cb = self.cond.run(g.contiguous().float())          # [B, C0, 1] -> per-item bias of conv_pre
        # bf16-RESIDENT activations (hipconv.set_activation_storage, BASELINE config 5): every tensor between conv_pre and conv_post is
        # bf16 -- in the plain-bf16 arithmetic the narrow stages' fused pairs are HBM-bound and the wide convs' epilogues stream residual
        # and output; the waveform (conv_post's output) is fp32.  Everything downstream allocates with empty_like(x), so the element type
        # follows x.  Stages narrower than 32 channels (the reference's hop-300 generator ends at 16) have no bf16 instance: the tensors
        # go back to fp32 at the last transposed conv before them that runs on 128-row tiles (the only shape with a bf16 -> fp32 instance).
        bf = self.__dict__.get("_hip_storage") == torch.bfloat16
        if bf and self.conv_pre._op(bind=False).math != L.MATH_BF16:
            raise L.VisingerHipError("bf16 activation storage needs the plain-bf16 arithmetic: set_conv_math(model, L.MATH_BF16) first")
        n_bf = self.num_upsamples if bf else 0          # stages [0, n_bf) hold bf16 tensors
        if bf:
            chans = [u.out_channels for u in self.ups]
            narrow = next((i for i, c in enumerate(chans) if c < 32), None)
            if narrow is not None:
                # (whole 128-row blocks: with 192 virtual rows -- 128 -> 64 at stride 3 -- the engine takes 64-row tiles)
                n_bf = max((i for i in range(narrow) if (self.ups[i].stride[0] * chans[i]) % 128 == 0), default=0)

        def store(stage):                               # element type of the tensors of `stage` (-1: conv_pre's output)
            return torch.bfloat16 if (bf and stage < n_bf) else None

        x = self.conv_pre.run(x, bias_b=cb, in_act=L.IN_NONE if mask is None else L.IN_MASK, mask=mask, y_dtype=store(-1))
        nk = self.num_kernels
        act = L.IN_LRELU if mask is None else L.IN_LRELU_MASK
        for i in range(self.num_upsamples):
            x = self.ups[i].run(x, in_act=act, mask=mask, y_dtype=store(i))
            if mask is not None:                                # the frame mask at this stage's resolution
                mask = mask.repeat_interleave(x.shape[2] // mask.shape[1], dim=1).contiguous()
            xs = torch.empty_like(x)
            for j in range(nk):
                self.resblocks[i * nk + j]._run_fused(x, xs, first=(j == 0), scale=(1.0 / nk if j == nk - 1 else 1.0), mask=mask)
            x = xs
        return self.conv_post.run(x, in_act=act, mask=mask, out_act=L.OUT_TANH, out_mask=mask is not None)

    def remove_weight_norm(self):
        _strip(self.ups)
        for block in self.resblocks:
            block.remove_weight_norm()


# (channels, k) whose whole block is one launch of resblock_bf16_kernel; the others keep the per-pair / per-conv launches.  `tools/resblock_bench.py bf16`, B = 8,
# ms per block (whole block / one pair per launch / respair_split_kernel of round 2 / conv by conv):
#   128 ch: k=3 0.55 / 0.69 / 0.83 / 0.86   k=7 1.31 / 1.31 / 1.18 / 1.30   k=11 2.52 / 2.14 / 1.52 / 1.68
#   64 ch:  k=3 0.32 / 0.46 / 0.43 / 0.81   k=7 0.73 / 0.74 / 0.73 / 1.12   k=11 1.37 / 1.17 / 0.98 / 1.31
#   32 ch:  k=3 0.23 / 0.32 / 0.33 / 0.61   k=7 0.39 / 0.47 / 0.45 / 0.78   k=11 0.56 / 0.59 / 0.55 / 0.88
# (ties go to the whole block: its residual stream stays in fp32 registers, rms error 3.0e-3 against 3.7e-3 of the bf16-resident chain)
BF16_WHOLE_BLOCK = {(128, 3), (64, 3), (64, 5), (64, 7), (32, 3), (32, 5), (32, 7), (32, 9), (32, 11)}


class ResBlock1(torch.nn.Module):
    """decoder.py:68-110"""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        # pair j: convs1[j] dilated by dilation[j], convs2[j] undilated; all weight-normed, "same" padding
        self.convs1 = _same_convs(channels, kernel_size, dilation)
        self.convs2 = _same_convs(channels, kernel_size, [1] * len(dilation))

    def _run_fused(self, x, out, first=True, scale=1.0, mask=None):
        """out = ((out if not first else 0) + resblock(x)) * scale   [* mask];  x is left untouched."""
        n = len(self.convs1)
        act = L.IN_LRELU if mask is None else L.IN_LRELU_MASK
        if mask is None and x.dtype in (torch.float32, torch.bfloat16):
            # split-f16 arithmetic, 32 / 64 channels: the whole block (or groups of its pairs) as one launch each, the residual stream in
            # registers between the pairs (csrc/resblock_f16.hip); a tile recomputes its halo, which grows with k: see _fused_groups
            groups = self._fused_groups(x.dtype)
            if groups is not None:
                cur = x
                for gi, ops in enumerate(groups):
                    last = gi == len(groups) - 1
                    dst = out if last else torch.empty_like(x)
                    resblock_forward(ops, cur, dst, acc=None if (first or not last) else out, scale=scale if last else 1.0)
                    cur = dst
                return out
        cur = x
        tmp = torch.empty_like(x)
        pp = [torch.empty_like(x) if n > 1 else None, torch.empty_like(x) if n > 2 else None]
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            last = i == n - 1
            dst = out if last else pp[i % 2]
            if mask is None:
                # narrow stages (32 / 64 channels): the whole pair in one launch, the intermediate never leaves the CU
                op1, op2 = c1._op(), c2._op()
                if respair_supported(op1, op2, profitable_only=True):
                    respair_forward(op1, op2, cur, dst, res=cur, acc=None if (first or not last) else out,
                                    scale=scale if last else 1.0)
                    cur = dst
                    continue
            c1.run(cur, in_act=act, mask=mask, y=tmp)
            if not last:
                c2.run(tmp, in_act=act, mask=mask, res=cur, y=dst)
                cur = dst
            else:
                c2.run(tmp, in_act=act, mask=mask, res=cur, acc=None if first else out, y=out, scale=scale,
                       out_mask=mask is not None)
        return out

    def _fused_groups(self, dtype=torch.float32):
        """the conv chain cut into launches of csrc/resblock_f16.hip, or None when it does not apply (other arithmetic / width).
        A launch over p pairs recomputes H = sum of its pads columns at each end of its tile: k = 3 -> 12, k = 7 -> 36, k = 11 -> 60 for a
        whole block: the fusion pays where a conv is short of matrix work (k = 3, 32 channels), not where the halo costs more than the saved
        tensor passes."""
        ops = [c._op() for pair in zip(self.convs1, self.convs2) for c in pair]
        if ops[0].c_in not in (32, 64, 128):
            return None
        # fp32 tensors on the split-f16 arithmetic, or bf16-RESIDENT tensors on plain bf16 operands (resblock_bf16_kernel)
        if (ops[0].math, dtype) not in ((L.MATH_SPLIT3, torch.float32), (L.MATH_BF16, torch.bfloat16)):
            return None
        C, k, n = ops[0].c_in, ops[0].k, len(self.convs1)
        per = L.switch("VS_RESBLOCK_PAIRS")                                     # pairs per launch (A/B switch); 0: by measurement
        if not per:
            # tools/resblock_bench.py, B = 32 production shapes (ms: whole block / pair by pair / conv by conv):
            #   32 ch:  k=3 1.6 / 2.1 / 3.6   k=7 2.6 / 3.0 / 4.5   k=11 4.0 / 4.1 / 5.4      -> whole block
            #   64 ch:  k=3 2.3 / 2.8 / 4.3   k=7 5.3 / 4.9 / 6.0   k=11 10.4 / 7.3 / 7.9     -> whole, pairs, pairs
            #   128 ch: k=3 4.0 / 4.4 / 5.2   k=7 10.3 / 9.0 / 8.6  k=11 21 / 14 / 11.9       -> whole block at k = 3 only (8 waves, 256 columns)
            if dtype == torch.bfloat16:
                per = n if (C, k) in BF16_WHOLE_BLOCK else 0
            elif C == 32:
                per = n
            elif C == 64:
                per = n if k <= 5 else 1
            else:
                per = n if k <= 3 else 0
            if not per:
                return None
        groups = [ops[2 * i:2 * (i + per)] for i in range(0, n, per)]
        return groups if all(resblock_supported(g) for g in groups) else None

    def forward(self, x, x_mask=None):
        if autograd.training_path(self):
            return autograd.resblock1(self, x, x_mask)
        _forward_only_guard(self)
        x = x.contiguous().float()
        B, _, T = x.shape
        return self._run_fused(x, torch.empty_like(x), mask=mask2d(x_mask, B, T))

    def remove_weight_norm(self):
        _strip(self.convs1)
        _strip(self.convs2)


class ResBlock2(nn.Module):
    """decoder.py:113-137"""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = _same_convs(channels, kernel_size, dilation)

    def _run_fused(self, x, out, first=True, scale=1.0, mask=None):
        n = len(self.convs)
        act = L.IN_LRELU if mask is None else L.IN_LRELU_MASK
        cur = x
        for i, c in enumerate(self.convs):
            if i < n - 1:
                nxt = torch.empty_like(x)
                c.run(cur, in_act=act, mask=mask, res=cur, y=nxt)
                cur = nxt
            else:
                c.run(cur, in_act=act, mask=mask, res=cur, acc=None if first else out, y=out, scale=scale,
                      out_mask=mask is not None)
        return out

    def forward(self, x, x_mask=None):
        if autograd.training_path(self):
            return autograd.resblock2(self, x, x_mask)
        _forward_only_guard(self)
        x = x.contiguous().float()
        B, _, T = x.shape
        return self._run_fused(x, torch.empty_like(x), mask=mask2d(x_mask, B, T))

    def remove_weight_norm(self):
        _strip(self.convs)
