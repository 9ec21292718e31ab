"""Every weight of a network derived and packed at the top of a training pass, in a handful of launches.

The reference re-derives each conv's weight inside the module's forward -- ``torch.nn.utils.weight_norm``'s pre-forward hook computes
``g * v / ||v||`` per module (modules/visinger/*.py and modules/discriminator.py wrap every conv in it) -- and autograd runs its
backward per module.  On the HIP engine each derived weight is then packed into MFMA fragment order twice (the conv and the
adjoint handle of its grad-input).  Per training step that was 256 + 182 weight-norm launches and ~670 pack launches of 4-13 us
(tools/train_op_census.py, round 5): a fifth of the step's launches for work that does not depend on any activation.

``WeightBank(net).refresh()`` does all of it before the forward starts:

* one launch folds every weight-normed tensor of the network (vs_weight_norm_multi_fwd) into a persistent flat buffer; the folded
  weights are the outputs of ONE autograd node whose backward (vs_weight_norm_multi_bwd, one launch) runs when all the weight
  gradients of the pass have arrived -- or, under data parallelism, of one light node per tensor (torch's own backward each), so
  that DistributedDataParallel's bucketed all-reduce keeps overlapping the backward;
* two launches pack every conv handle whose packed weights are stale (vs_conv_set_weights_batch).

The modules then find their weight in ``module.__dict__["_w_eff"]`` (autograd.effective_weight, discriminator._live_params) and their
handles already carrying the pass's key; ``release()`` removes the entries again, so a forward outside a bank-managed pass (eval, tests
that call modules directly) takes the per-module path as before.  ``VS_NO_WEIGHT_BANK=1`` is the A/B switch (train.VISingerTrainer).
"""
import torch

from . import _lib as L
from .ops import ConvOp


class _BatchedWeightNorm(torch.autograd.Function):
    """(v_0, g_0, v_1, g_1, ...) -> (w_0, w_1, ...), w_i = g_i * v_i / ||v_i|| over dim 0 (torch._weight_norm per tensor)"""

    @staticmethod
    def forward(ctx, bank, *params):
        st = bank._state
        ctx.bank, ctx.gen = bank, st["gen"]
        ctx.set_materialize_grads(False)
        L.check(L.require_gpu().vs_weight_norm_multi_fwd(st["table"].data_ptr(), st["n"], st["total_rows"], L.stream_ptr()))
        wbuf = st["wbuf"]
        return tuple(wbuf[o:o + nel].view(shape) for o, nel, shape in st["wviews"])

    @staticmethod
    def backward(ctx, *gws):
        st = ctx.bank._state
        assert st["gen"] == ctx.gen, "WeightBank: the bank was refreshed between this pass's forward and its backward"
        dev = st["wbuf"].device
        gbuf = torch.empty(st["total_numel"] + st["total_rows"], device=dev, dtype=torch.float32)
        rows, keep, grads = [], [], [None]
        go = st["total_numel"]
        for (o, nel, shape), r, gw in zip(st["wviews"], st["rows"], gws):
            if gw is None:
                rows += [0, 0, 0, 0]
                grads += [None, None]
            else:
                gw = gw.contiguous().float()
                keep.append(gw)
                gv, gg = gbuf[o:o + nel].view(shape), gbuf[go:go + r].view((r,) + (1,) * (len(shape) - 1))
                rows += [gw.data_ptr(), gv.data_ptr(), gg.data_ptr(), 0]
                grads += [gv, gg]
            go += r
        if keep:
            gtab = torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)      # (pinned: the host allocator keeps the block until the copy has run)
            L.check(L.require_gpu().vs_weight_norm_multi_bwd(st["table"].data_ptr(), gtab.data_ptr(), st["n"], st["total_rows"], L.stream_ptr()))
        return tuple(grads)


class _FoldedSlot(torch.autograd.Function):
    """One tensor of a batch that vs_weight_norm_multi_fwd has already folded: the forward hands out its view of the buffer, the backward is torch's own
    per-tensor weight-norm backward.  Used under data parallelism: every weight-normed gradient then leaves the graph when its conv's weight gradient arrives,
    and DistributedDataParallel overlaps the bucketed all-reduce with the rest of the backward -- the single batched node delivers all of them at the very
    end (one launch instead of ~180, but nothing left to overlap with)."""

    @staticmethod
    def forward(ctx, v, g, w_view, norm_view, bank):
        ctx.save_for_backward(v, g, norm_view)
        ctx.bank, ctx.gen = bank, bank._state["gen"]
        return w_view.view(w_view.shape)

    @staticmethod
    def backward(ctx, gw):
        # (refresh() rewrites wbuf / nbuf through raw pointers -- no autograd version bump -- so a second refresh between a forward and its backward
        #  would hand this node the NEXT pass's norms: refuse, as the batched node does)
        assert ctx.bank._state["gen"] == ctx.gen, "WeightBank: the bank was refreshed between this pass's forward and its backward"
        v, g, norm = ctx.saved_tensors
        gv, gg = torch.ops.aten._weight_norm_interface_backward(gw.contiguous(), v, g, norm.view(g.shape), 0)
        return gv, gg, None, None, None


def _data_parallel():
    d = torch.distributed
    return d.is_available() and d.is_initialized() and d.get_world_size() > 1


class WeightBank:
    def __init__(self, net, per_tensor_backward=None):
        self.net = net
        self.per_tensor_backward = per_tensor_backward      # None: per-tensor backward nodes under data parallelism, the batched node otherwise
        self._state = None
        self._gen = 0
        self._live = []

    # -- the weight-normed tensors of the network and their table ---------------------------------------------------------------
    def _scan(self):
        st = self._state
        if st is not None:      # (the module walk is done once: a refresh re-checks only that the known tensors still sit where the table says -- a network
            #                      whose MODULES change needs a new bank)
            ok = all(isinstance(m._parameters.get("weight_v"), torch.nn.Parameter) for m in st["mods"]) and \
                st["ident"] == tuple((m.weight_v.data_ptr(), m.weight_g.data_ptr(), tuple(m.weight_v.shape)) for m in st["mods"])
            if ok:
                return st
        mods = [m for m in self.net.modules()
                if isinstance(m.__dict__.get("_parameters", {}).get("weight_g"), torch.nn.Parameter) and
                isinstance(m._parameters.get("weight_v"), torch.nn.Parameter)]
        ident = tuple((m.weight_v.data_ptr(), m.weight_g.data_ptr(), tuple(m.weight_v.shape)) for m in mods)
        for m in mods:
            v, g = m.weight_v, m.weight_g
            assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and g.is_contiguous() and g.numel() == v.shape[0], \
                "WeightBank: weight_norm(dim=0) parameters must be contiguous fp32 CUDA tensors"
        dev = mods[0].weight_v.device if mods else None
        wviews, rows, tab = [], [], []
        off = row0 = 0
        for m in mods:
            nel, r = m.weight_v.numel(), m.weight_v.shape[0]
            wviews.append((off, nel, tuple(m.weight_v.shape)))
            rows.append(r)
            off += nel
            row0 += r
        st = {"ident": ident, "mods": mods, "n": len(mods), "wviews": wviews, "rows": rows, "total_numel": off, "total_rows": row0,
              "params": [p for m in mods for p in (m.weight_v, m.weight_g)], "gen": 0}
        if mods:
            st["wbuf"] = torch.empty(off, device=dev, dtype=torch.float32)
            st["nbuf"] = torch.empty(row0, device=dev, dtype=torch.float32)
            row0 = 0
            for m, (o, nel, _), r in zip(mods, wviews, rows):
                tab += [m.weight_v.data_ptr(), m.weight_g.data_ptr(), st["wbuf"].data_ptr() + 4 * o, st["nbuf"].data_ptr() + 4 * row0,
                        r, nel // r, row0, 0]
                row0 += r
            st["table"] = torch.tensor(tab, dtype=torch.int64).to(dev)
        from .modules.hipconv import _HipConvMixin
        st["convs"] = [m for m in self.net.modules() if isinstance(m, _HipConvMixin)]
        self._state = st
        return st

    def refresh(self):
        """Fold every weight-normed weight of the network (one differentiable node) and pack the stale conv handles.  Call at the top of
        a training pass, AFTER requires_grad of the network's parameters is what the pass needs; `release()` when the forward is done."""
        from .autograd import param_key
        st = self._scan()
        self._gen += 1
        st["gen"] = self._gen
        if st["n"]:
            split = _data_parallel() if self.per_tensor_backward is None else self.per_tensor_backward
            if split and any(p.requires_grad for p in st["params"]):
                L.check(L.require_gpu().vs_weight_norm_multi_fwd(st["table"].data_ptr(), st["n"], st["total_rows"], L.stream_ptr()))
                ws, row0 = [], 0
                for m, (o, nel, shape), r in zip(st["mods"], st["wviews"], st["rows"]):
                    ws.append(_FoldedSlot.apply(m.weight_v, m.weight_g, st["wbuf"][o:o + nel].view(shape), st["nbuf"][row0:row0 + r], self))
                    row0 += r
            else:
                ws = _BatchedWeightNorm.apply(self, *st["params"])
            for m, w in zip(st["mods"], ws):
                m.__dict__["_w_eff"] = w
            self._live = st["mods"]
        jobs = []
        for m in st["convs"]:
            op = m.__dict__.get("_hip_ops", {}).get((m._kind, 0))
            if op is None:                 # (first pass: the module's forward creates the handle and packs it)
                continue
            key = param_key(m)
            if key is None:
                continue
            m._op(bind=False)              # (applies a pending set_conv_math to the handle)
            w = m.__dict__.get("_w_eff")
            if w is None:
                if hasattr(m, "weight_g"):
                    continue
                w = m.weight
            if not op.has_weights_of(key):
                jobs.append((op, w, m.bias, key))
            adj = m.__dict__.get("_hip_bwd_ops", {}).get("dxa")
            if adj is not None and not adj.has_weights_of(key):
                jobs.append((adj, w, None, key))
        ConvOp.set_weights_batch(jobs)

    def release(self):
        for m in self._live:
            m.__dict__.pop("_w_eff", None)
        self._live = []
