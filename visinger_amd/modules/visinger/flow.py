"""MI355X-native mirror of the reference's modules/visinger/flow.py:15-95 -- ResidualCouplingBlock,
ResidualCouplingLayer, Flip (the spline flows at flow.py:98-358 are dead code in the reference and are not built).

Data layout: the block keeps ONE physical [B, C, T] latent in HBM for all couplings.  `Flip` never moves data:
a coupling at odd position reads its x0 half from the upper physical rows through column-reversed `pre` weights
and updates the lower physical rows through row-reversed `post` weights (VS_CONV_FLIP_IN / VS_CONV_FLIP_OUT);
torch.split / torch.cat / torch.flip of the reference become pointer offsets.  The affine update
(x1 = m + x1*exp(logs)*mask, its inverse) and the log-det reduction run in the epilogue of the `post` conv.
"""
import torch
import torch.nn as nn

from ... import _lib as L
from ... import autograd
from ...ops import _off
from ..hipconv import HipConv1d, mask2d, _forward_only_guard
from .encoder import WaveNet


class ResidualCouplingBlock(nn.Module):
    """flow.py:15-44"""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.n_flows = n_flows
        self.gin_channels = gin_channels

        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                    gin_channels=gin_channels, mean_only=True))
            self.flows.append(Flip())

    def forward(self, x, x_mask, g=None, reverse=False):
        if autograd.training_path(self):
            return autograd.flow_block(self, x, x_mask, g, reverse)
        _forward_only_guard(self)
        B, C, T = x.shape
        xp = x.contiguous().float().clone()        # physical latent, updated in place by the couplings
        m2 = mask2d(x_mask, B, T)
        g = None if g is None else g.contiguous().float()
        # Coupling f sees f flips before it in the forward order and n_flows - f in the reverse order.  With an
        # even n_flows (the model's 4) both have the parity of f; an odd n_flows needs one physical flip (after the
        # forward pass / before the reverse pass), after which the parity is again that of f.
        order = range(self.n_flows) if not reverse else reversed(range(self.n_flows))
        if reverse and self.n_flows % 2 == 1:
            xp = torch.flip(xp, [1])
        for f in order:
            self.flows[2 * f]._apply_inplace(xp, m2, g, reverse, flipped=bool(f % 2), logdet=None)
        if not reverse and self.n_flows % 2 == 1:
            xp = torch.flip(xp, [1])
        return xp

    def remove_weight_norm(self):
        for i in range(self.n_flows):
            self.flows[i * 2].remove_weight_norm()


class ResidualCouplingLayer(nn.Module):
    """flow.py:47-85"""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                 p_dropout=0, gin_channels=0, mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only

        self.pre = HipConv1d(self.half_channels, hidden_channels, 1)
        self.enc = WaveNet(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout,
                           gin_channels=gin_channels)
        self.post = HipConv1d(hidden_channels, self.half_channels * (2 - mean_only), 1)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def _apply_inplace(self, xp, m2, g, reverse, flipped, logdet):
        """One coupling on the physical latent xp [B, C, T] (in place on the x1 half)."""
        B, C, T = xp.shape
        half = self.half_channels
        x0_row, x1_row = (half, 0) if flipped else (0, half)
        h = self.pre.run(None, flags=L.FLIP_IN if flipped else 0, B=B, T=T, x_ptr=_off(xp, x0_row * T), x_bs=C * T,
                         mask=m2, out_mask=True,
                         y=torch.empty((B, self.hidden_channels, T), device=xp.device, dtype=torch.float32))
        h = self.enc(h, m2, g=g)
        x1p = _off(xp, x1_row * T)
        flags = L.FLIP_OUT if flipped else 0
        if self.mean_only:
            self.post.run(h, flags=flags, mask=m2, y_ptr=x1p, res_ptr=x1p, y_bs=C * T, res_bs=C * T,
                          mode=L.MODE_COUPLING_MEAN_INV if reverse else L.MODE_COUPLING_MEAN_FWD)
        else:
            self.post.run(h, kind=L.CONV1D_PAIRED, flags=flags, mask=m2, y_ptr=x1p, res_ptr=x1p, y_bs=C * T,
                          res_bs=C * T, pair_mode=L.PAIR_COUPLING_INV if reverse else L.PAIR_COUPLING_FWD,
                          logdet=logdet)

    def forward(self, x, x_mask, g=None, reverse=False):
        if autograd.training_path(self):
            return autograd.coupling_layer(self, x, x_mask, g, reverse)
        _forward_only_guard(self)
        B, C, T = x.shape
        xp = x.contiguous().float().clone()
        m2 = mask2d(x_mask, B, T)
        g = None if g is None else g.contiguous().float()
        if not reverse:
            # mean_only: logs == 0 -> log-det is exactly 0 (flow.py:73-75,80)
            logdet = torch.zeros(B, device=x.device, dtype=torch.float32)
            self._apply_inplace(xp, m2, g, False, False, None if self.mean_only else logdet)
            return xp, logdet
        self._apply_inplace(xp, m2, g, True, False, None)
        return xp

    def remove_weight_norm(self):
        self.enc.remove_weight_norm()


class Flip(nn.Module):
    """flow.py:88-95 (stand-alone use; inside ResidualCouplingBlock the flip is folded into the weights)"""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = torch.flip(x, [1])
        if not reverse:
            logdet = torch.zeros(x.size(0)).to(dtype=x.dtype, device=x.device)
            return x, logdet
        else:
            return x
