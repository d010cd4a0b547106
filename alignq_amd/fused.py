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


# ------------------------------------------------------------------------------------------------------------------
class BNSiteFn(torch.autograd.Function):
    """act_q(bn(z)) with the batch-norm folded into the ADMM-site kernels (training mode).

    Forward: bn statistics (2 launches) -> site partials reading z with x = a*z + b on load -> slab reduce + loss.
    Backward: prep (S, dalterD, dgamma) -> site backward writing dx and per-tile BN sums -> bn backward apply (dz, dgamma_bn,
    dbeta).  The normalised activation is never written to HBM.  Values: x differs from torch's ((z-mean)*invstd)*gamma+beta
    only by the rounding of one fma (1e-7 relative)."""

    @staticmethod
    def forward(ctx, z, bn_weight, bn_bias, running_mean, running_var, nbt, momentum, bn_eps, alterD, gamma, k, act_range,
                eps, mu, rho, relu):
        z = L.dev_f32(z, "conv output")
        A = L.dev_f32(alterD, "alterD")
        Gm = L.dev_f32(gamma, "gamma")
        B, C, H, W = z.shape
        HW, F = H * W, C * H * W
        dim = A.shape[0]
        lib = L.load()
        dev = z.device
        st = L.stream_ptr()
        ab = torch.empty(2, C, dtype=torch.float32, device=dev)
        save = torch.empty(2, C, dtype=torch.float32, device=dev)
        ws_bn = torch.empty(lib.alignq_bn_ws_bytes(C), dtype=torch.uint8, device=dev)
        L.check(lib.alignq_bn_partial_stats(L.ptr(z), B, C, HW, L.ptr(ws_bn), st), "alignq_bn_partial_stats")
        y = torch.empty_like(z)
        D = torch.empty(B, B, dtype=torch.float32, device=dev)
        stats = torch.empty(4, F, dtype=torch.float32, device=dev)
        scal = torch.empty(4, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
        L.check(lib.alignq_site_partials_bn(L.ptr(z), L.ptr(ws_bn), L.ptr(bn_weight), L.ptr(bn_bias), L.ptr(running_mean),
                                            L.ptr(running_var), L.ptr(nbt), float(momentum), float(bn_eps), L.ptr(ab),
                                            L.ptr(save), C, HW, B, F, int(k), float(act_range), float(eps), int(bool(relu)),
                                            L.ptr(y), L.ptr(stats), L.ptr(ws), st), "alignq_site_partials_bn")
        L.check(lib.alignq_site_reduce_loss(L.ptr(ws), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), dim, float(mu), float(rho),
                                            L.ptr(scal), st), "alignq_site_reduce_loss")
        ctx.save_for_backward(z, ab, save, stats, D, A, Gm, scal, y if relu else None)
        ctx.set_materialize_grads(False)
        ctx.cfg = (float(act_range), float(eps), float(mu), bn_weight is not None, bn_bias is not None)
        ctx.mark_non_differentiable(D)
        return y, scal[0], D

    @staticmethod
    def backward(ctx, g_y, g_loss, _gD):
        z, ab, save, stats, D, A, Gm, scal, y = ctx.saved_tensors
        act_range, eps, mu, has_w, has_b = ctx.cfg
        B, C, H, W = z.shape
        HW, F = H * W, C * H * W
        dim = A.shape[0]
        lib = L.load()
        dev = z.device
        st = L.stream_ptr()
        g_y = None if g_y is None else L.like_layout(g_y, z)
        if g_loss is None:
            g_loss = torch.zeros((), dtype=torch.float32, device=dev)
        g_loss = L.dev_f32(g_loss, "loss grad")
        S = torch.empty(B, B, dtype=torch.float32, device=dev)
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        L.check(lib.alignq_site_prep_fused(L.ptr(D), L.ptr(A), L.ptr(Gm), dim, L.ptr(scal), mu, L.ptr(g_loss), B, F,
                                           L.ptr(S), L.ptr(dA), L.ptr(dG), st), "alignq_site_prep_fused")
        dx = torch.empty_like(z)
        part = torch.empty(lib.alignq_site_bn_part_bytes(F), dtype=torch.uint8, device=dev)
        L.check(lib.alignq_site_bwd_apply_bn(L.ptr(g_y), L.ptr(S), L.ptr(z), L.ptr(ab), L.ptr(save), C, HW, L.ptr(y),
                                             L.ptr(stats), B, F, act_range, eps, L.ptr(dx), L.ptr(part), st),
                "alignq_site_bwd_apply_bn")
        dz = torch.empty_like(z)
        dgam = torch.empty(C, dtype=torch.float32, device=dev) if has_w else None
        dbet = torch.empty(C, dtype=torch.float32, device=dev) if has_b else None
        L.check(lib.alignq_bn_bwd_apply(L.ptr(dx), L.ptr(z), L.ptr(ab), L.ptr(save), L.ptr(part), B, C, HW, L.ptr(dz),
                                        L.ptr(dgam), L.ptr(dbet), st), "alignq_bn_bwd_apply")
        return (dz, dgam, dbet, None, None, None, None, None, dA, dG, None, None, None, None, None, None)


def bn_site_fusable(bn, act, z) -> bool:
    from . import config
    if not (isinstance(bn, torch.nn.BatchNorm2d) and bn.training and bn.track_running_stats and bn.momentum is not None):
        return False
    if not (z.is_cuda and z.dim() == 4 and z.dtype == torch.float32 and z.is_contiguous()):
        return False
    B, C, H, W = z.shape
    a_bit = getattr(act, "a_bit", 32)
    return (64 < B <= L.MAX_BATCH and (H * W) % 64 == 0 and hasattr(act, "opt") and a_bit < 32
            and config.args.method == "ours" and act.opt.alterD.shape[0] >= B)


def bn_site(bn, act, z, eps=0.0, relu=False):
    """out, loss = act(bn(z)) [; out = relu(out) when relu=True] — folded when `bn_site_fusable`, otherwise exactly that
    composition."""
    from . import config
    if not bn_site_fusable(bn, act, z):
        out, loss = act(bn(z))
        return (torch.nn.functional.relu(out) if relu else out), loss
    admm = act.opt
    y, loss, D = BNSiteFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                bn.momentum, bn.eps, admm.alterD, admm.gamma, act.a_bit, config.args.act_range, eps,
                                admm.mu, admm.rho, relu)
    admm.D = D
    deferred = active_deferred()
    if deferred is not None:
        deferred.add(loss)
        return y, 0.0
    return y, loss
