"""Whole-model fast paths that keep the reference's per-module API intact.

`prequantize_weights(convs)` quantises ALL conv weights of a model with two multi-tensor launches
(alignq_weight_quant_fwd_multi) instead of four launches per tensor, and parks the results in each
`conv.quantize_fn`; the next `weight_quantize_fn.forward(conv.weight)` consumes them.  Autograd goes through one
Function with T inputs, whose backward is again two multi-tensor launches.  Used by TrainStep; calling the
modules without it still works (per-tensor kernels)."""
from __future__ import annotations

import torch

from . import _lib as L


class WeightQuantAllFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, k, formula, *weights):
        lib = L.load()
        ws_ = [L.dense_f32(w, "weight") for w in weights]
        T = len(ws_)
        dev = ws_[0].device
        qs = [torch.empty_like(w) for w in ws_]
        cs = [torch.empty_like(w) for w in ws_]
        ps = [torch.empty_like(w) for w in ws_]
        ms = torch.empty(T, 2, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.alignq_weight_multi_ws_bytes(T), dtype=torch.uint8, device=dev)
        n = L.i64_array([w.numel() for w in ws_])
        L.check(lib.alignq_weight_quant_fwd_multi(T, L.ptr_array(ws_), L.ptr_array(qs), L.ptr_array(cs),
                                                  L.ptr_array(ps), n, L.ptr(ms), int(k), int(formula), L.ptr(scratch),
                                                  L.stream_ptr()), "alignq_weight_quant_fwd_multi")
        ctx.save_for_backward(ms, *ws_)
        ctx.set_materialize_grads(False)     # no zero tensors for the 2T non-differentiable cdf/pdf outputs
        ctx.mark_non_differentiable(*cs, *ps)
        ctx.T = T
        return tuple(qs) + tuple(cs) + tuple(ps)

    @staticmethod
    def backward(ctx, *grads):
        T = ctx.T
        ms, ws_ = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        lib = L.load()
        gs = [torch.zeros_like(w) if g is None else L.like_layout(g, w) for g, w in zip(grads[:T], ws_)]
        dws = [torch.empty_like(w) for w in ws_]
        scratch = torch.empty(lib.alignq_weight_multi_ws_bytes(T), dtype=torch.uint8, device=ws_[0].device)
        L.check(lib.alignq_weight_quant_bwd_multi(T, L.ptr_array(gs), L.ptr_array(list(ws_)), L.ptr(ms),
                                                  L.ptr_array(dws), L.i64_array([w.numel() for w in ws_]),
                                                  L.ptr(scratch), L.stream_ptr()), "alignq_weight_quant_bwd_multi")
        return (None, None) + tuple(dws)


def prequantize_weights(convs):
    """convs: modules with `.weight` and `.quantize_fn` (Conv2d_Q).  All must share w_bit (< 32) and tree."""
    convs = [c for c in convs if c.quantize_fn.w_bit != 32]
    if not convs:
        return
    groups = {}
    for c in convs:
        groups.setdefault((c.quantize_fn.w_bit, c.quantize_fn._formula), []).append(c)
    for (k, formula), cs in groups.items():
        outs = WeightQuantAllFn.apply(k, formula, *[c.weight for c in cs])
        T = len(cs)
        for i, c in enumerate(cs):
            c.quantize_fn._pre = (c.weight, outs[i], outs[T + i], outs[2 * T + i])


# ------------------------------------------------------------------------------------------------------------------
class DeferredLosses:
    """Fast path for a whole-model step: the per-site slab reduction + ADMM loss is launched on a side stream (it is
    not needed by the next layer, only x_q is) and the 21..57 site losses are summed once at the end instead of one
    tiny add kernel per site.  While active, activation modules return the python float 0.0 as their trans_loss and
    park the real loss tensor here; `total()` joins the side stream and returns the sum (differentiable)."""

    def __init__(self, use_side_stream=False):
        # Measured on MI355X / ROCm 7.2 (ResNet-20 step, one HIP graph): forking the 21 reductions onto a side stream
        # costs more in cross-queue graph dependencies than the overlap returns (4.09 vs 3.44 ms per step), so the
        # default keeps everything on one stream and only defers the loss sum.
        self.side = torch.cuda.Stream() if use_side_stream else None
        self.losses = []

    def __enter__(self):
        global _active
        self.losses = []
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None
        return False

    def add(self, loss):
        self.losses.append(loss)

    def total(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        if not self.losses:
            return None
        return torch.stack(self.losses).sum()


_active = None


def active_deferred():
    return _active
