"""Whole-model fast paths that keep the reference's per-module API intact (all opt-in; TrainStep switches them on).

* `prequantize_weights(convs)` / `WeightQuantAllFn`: ALL conv weights of a model quantised with two multi-tensor launches
  (alignq_weight_quant_fwd_multi) instead of four launches per tensor; results are parked in each `conv.quantize_fn` and
  consumed by the next `weight_quantize_fn.forward(conv.weight)`.  The backward is again two multi-tensor launches (and
  first finishes any deferred filter-gradient reductions).
* `DeferredLosses` / `LossSumFn` / `SiteRecord`: every ADMM site's slab reduction + loss in ONE launch after the last layer,
  every site's backward prep in ONE launch.
* `BNSiteFn` / `bn_site`: `act_q(bn(z)) [+ residual] [-> relu]` with the batch-norm folded into the site kernels (NCHW or
  channels-last).
* `DeferredWgrads`: the slab reductions of ops.QConv3x3Fn's filter gradients in ONE launch per backward.
"""
from __future__ import annotations

import ctypes
import os
import threading

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
        # the incoming W_q gradients of ops.QConv3x3Fn may still be partial-sum slabs: this node runs after every
        # convolution's backward, so finish all of them here with one launch
        pending = active_wgrads()
        if pending is not None:
            pending.flush()
        gs = [torch.zeros_like(w) if g is None else L.like_layout(g, w) for g, w in zip(grads[:T], ws_)]
        dws = [torch.empty_like(w) for w in ws_]
        scratch = torch.empty(lib.alignq_weight_multi_ws_bytes(T), dtype=torch.uint8, device=ws_[0].device)
        L.check(lib.alignq_weight_quant_bwd_multi(T, L.ptr_array(gs), L.ptr_array(list(ws_)), L.ptr(ms),
                                                  L.ptr_array(dws), L.i64_array([w.numel() for w in ws_]),
                                                  L.ptr(scratch), L.stream_ptr()), "alignq_weight_quant_bwd_multi")
        return (None, None) + tuple(dws)


def prequantize_weights(convs, pack=False):
    """convs: modules with `.weight` and `.quantize_fn` (Conv2d_Q).  All must share w_bit (< 32) and tree.
    pack: also leave every (<= 8-bit) filter's integer bins as bf16 / f16 bit patterns (ops.pack_filter_bins, one launch per 64
    filters) for the GEMM convolutions of ops.QConvGemmFn."""
    convs = [c for c in convs if c.quantize_fn.w_bit != 32]
    if not convs:
        return
    groups = {}
    for c in convs:
        groups.setdefault((c.quantize_fn.w_bit, c.quantize_fn._formula), []).append(c)
    for (k, formula), cs in groups.items():
        outs = WeightQuantAllFn.apply(k, formula, *[c.weight for c in cs])
        T = len(cs)
        bins = None
        if pack and 1 <= k <= 8:
            from . import ops
            sel = [i for i in range(T) if outs[i].numel() % 4 == 0]
            packed = ops.pack_filter_bins([outs[i].detach() for i in sel], k)
            bins = dict(zip(sel, packed))
        for i, c in enumerate(cs):
            pre = (c.weight, outs[i], outs[T + i], outs[2 * T + i])
            c.quantize_fn._pre = pre + ((bins[i],) if bins is not None and i in bins else ())


# ------------------------------------------------------------------------------------------------------------------
class SiteRecord:
    """Per-site bookkeeping of a batched (deferred) step: the site's forward only launches the partial-slab kernel and
    parks its buffers here; `DeferredLosses.total()` reduces all sites' slabs (+ ADMM loss) in one launch and its backward
    prepares all sites' S / dalterD / dgamma in one launch."""
    __slots__ = ("ws", "D", "A", "Gm", "scal", "B", "F", "dim", "mu", "rho", "S", "dA", "dG", "prepared", "reduced")

    def __init__(self):
        self.prepared = False
        self.reduced = False        # True: a later site's forward launch took this site's reduction along (take_fill)


class LossSumFn(torch.autograd.Function):
    """total = sum_s trans_loss_s over the batched sites.  Forward: alignq_site_reduce_loss_multi (D_s and the loss scalars
    of every site), then one sum.  Backward: alignq_site_prep_fused_multi with the upstream scalar.  The per-site losses
    are inputs, so autograd runs this node's backward before any site's backward."""

    @staticmethod
    def forward(ctx, collector, scal_all, *losses):
        collector.reduce_all()
        ctx.collector = collector
        ctx.recs = list(collector.records)
        return scal_all[:len(losses), 0].sum()

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        st = L.stream_ptr()
        g = L.dev_f32(g, "loss grad")
        groups = {}
        for r in ctx.recs:
            groups.setdefault((r.B, r.dim, r.mu, r.rho), []).append(r)
        for (B, dim, mu, rho), recs in groups.items():
            dev = recs[0].D.device
            for r in recs:
                r.S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, dtype=torch.float32, device=dev)   # fp32 S + bf16 image
                r.dA, r.dG = torch.empty_like(r.A), torch.empty_like(r.Gm)
            L.check(lib.alignq_site_prep_fused_multi(
                len(recs), L.ptr_array([r.D for r in recs]), L.ptr_array([r.A for r in recs]),
                L.ptr_array([r.Gm for r in recs]), L.ptr_array([r.scal for r in recs]), L.ptr(g),
                L.i64_array([r.F for r in recs]), B, dim, mu, L.ptr_array([r.S for r in recs]),
                L.ptr_array([r.dA for r in recs]), L.ptr_array([r.dG for r in recs]), st),
                "alignq_site_prep_fused_multi")
            for r in recs:
                r.prepared = True
        return (None, None) + (g,) * len(ctx.recs)


class DeferredLosses:
    """Fast path for a whole-model step.  While active, activation modules return the python float 0.0 as their
    trans_loss and park the real loss here; `total()` returns the (differentiable) sum.

    batch=True (default): sites with 64 < B <= 128 additionally defer their slab reduction + ADMM loss to `total()`, where
    ONE launch covers all of them (it is off the forward's critical path: only x_q feeds the next layer), and their
    backward shares ONE prep launch.  Other sites keep their own launches and only the loss sum is deferred."""

    _CHUNK = 64

    def __init__(self, use_side_stream=False, batch=True):
        # Measured on MI355X / ROCm 7.2 (ResNet-20 step, one HIP graph): forking the 21 reductions onto a side stream
        # costs more in cross-queue graph dependencies than the overlap returns (4.09 vs 3.44 ms per step), so the
        # default keeps everything on one stream.
        self.side = torch.cuda.Stream() if use_side_stream else None
        self.batch = batch and not use_side_stream
        self.losses = []
        self.records = []
        self._rec_losses = []
        self._scal_all = None

    def __enter__(self):
        global _active
        self.losses, self.records, self._rec_losses, self._scal_all = [], [], [], None
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None
        # the sites' loss tensors were summed inside the context (total / total_with_head); kept any longer they would hold the
        # iteration's autograd graph - and with it every parameter's gradient-accumulation node - until the next iteration
        # (train_step.retained_graph_params)
        self.losses, self._rec_losses = [], []
        return False

    def add(self, loss):
        self.losses.append(loss)

    def new_record(self, B, device):
        """A SiteRecord (with its loss-scalar row) when this site can join the batched launches, else None."""
        if not self.batch or not (64 < B <= L.MAX_BATCH) or len(self.records) >= self._CHUNK:
            return None
        if self._scal_all is None:
            self._scal_all = torch.empty(self._CHUNK, 4, dtype=torch.float32, device=device)
        rec = SiteRecord()
        rec.scal = self._scal_all[len(self.records)]
        self.records.append(rec)
        return rec

    def add_record_loss(self, loss):
        self._rec_losses.append(loss)

    def groups(self):
        g = {}
        for r in self.records:
            g.setdefault((r.B, r.dim, r.mu, r.rho), []).append(r)
        return g

    def take_fill(self, rec, B, F, dim, mu, rho):
        """Earlier sites (same B, dim, mu, rho, not reduced yet) whose slab reduction + ADMM loss the forward launch of the site
        `rec` at (B, F) takes along as its filler role (alignq_site_partials_bn_fill): only the one-tile launches that leave half
        the chip idle have slots (alignq_site_fill_slots); _SITE_FILL caps the items per launch."""
        slots = min(_SITE_FILL, L.load().alignq_site_fill_slots(int(B), int(F)))
        out = []
        if slots > 0:
            for r in self.records:
                if r is rec or r.reduced or getattr(r, "ws", None) is None:
                    continue
                if (r.B, r.dim, r.mu, r.rho) != (B, dim, float(mu), float(rho)):
                    continue
                out.append(r)
                if len(out) == slots:
                    break
        for r in out:
            r.reduced = True
        return out

    def open_groups(self):
        """The batched sites whose slab reduction no forward launch has taken along yet, by (B, dim, mu, rho)."""
        out = {}
        for key, recs in self.groups().items():
            recs = [r for r in recs if not r.reduced]
            if recs:
                out[key] = recs
        return out

    def reduce_all(self):
        """Slab reduction + ADMM loss of every batched site still open (one launch per (B, dim, mu, rho) group)."""
        lib = L.load()
        st = L.stream_ptr()
        for (B, dim, mu, rho), recs in self.open_groups().items():
            L.check(lib.alignq_site_reduce_loss_multi(
                len(recs), L.ptr_array([r.ws for r in recs]), L.ptr_array([r.D for r in recs]),
                L.ptr_array([r.A for r in recs]), L.ptr_array([r.Gm for r in recs]),
                L.ptr_array([r.scal for r in recs]), L.i64_array([r.F for r in recs]), B, dim, mu, rho, st),
                "alignq_site_reduce_loss_multi")

    def can_fuse_head(self):
        """Every site's loss is a batched record (the normal whole-model step): StepLossFn can take both loss roots."""
        return self.side is None and bool(self.records) and not self.losses and len(self._rec_losses) == len(self.records)

    def total_with_head(self, feat, weight, bias, target):
        """(logits, ce, trans_total) through StepLossFn; call instead of total() + HeadCEFn when can_fuse_head()."""
        return StepLossFn.apply(self, self._scal_all, feat, weight, bias, target, *self._rec_losses)

    def total(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        parts = []
        if self.records:
            assert len(self._rec_losses) == len(self.records)
            parts.append(LossSumFn.apply(self, self._scal_all, *self._rec_losses))
        if self.losses:
            parts.append(torch.stack(self.losses).sum())
        if not parts:
            return None
        return parts[0] if len(parts) == 1 else parts[0] + parts[1]


class DeferredWgrads:
    """Collects the partial-sum slabs of ops.QConv3x3Fn's filter gradients during a backward and finishes all of them with
    ONE alignq_conv3x3_wgrad_reduce_multi launch (`flush`).  The consumer of those gradients is the weight quantiser's
    backward: WeightQuantAllFn.backward (all weights, runs after every convolution's backward) flushes before it reads them,
    and so does the per-tensor ops.WeightQuantFn.backward.  Use only around a backward whose W_q gradients have no other
    consumer (TrainStep)."""

    def __init__(self, fresh_grads=False):
        """fresh_grads=True promises that every parameter's .grad is None when the backward starts (zero_grad(set_to_none=True)):
        autograd then adopts an incoming gradient tensor as .grad instead of adding it to an existing one, which lets a LATER
        kernel of the same backward fill gradients autograd already holds (the batch-norm parameter gradients written by
        alignq_conv3x3_nhwc_bwd, see BNSiteFn.backward)."""
        self.items = []
        self.fresh_grads = bool(fresh_grads)

    def __enter__(self):
        global _active_wgrads
        self.items = []
        del _posted_links[:]
        _active_wgrads = self
        return self

    def __exit__(self, exc_type, *exc):
        global _active_wgrads
        _active_wgrads = None
        if exc_type is None:
            flush_bwd_twins()
        else:
            del _bwd_twins.open[:]
        stale = [l for l in _posted_links if l.record is not None]
        del _posted_links[:]
        for l in stale:
            l.record = None
        if stale and exc_type is None:
            # a lazy batch-norm gradient nobody consumed means the producing convolution's backward never ran with it
            # (e.g. its input and weight need no gradient): the BN parameter gradients autograd holds were never written
            raise RuntimeError(f"alignq_amd: {len(stale)} lazy batch-norm gradient record(s) were never consumed by the "
                               "producing convolution's backward; gradients of this backward are incomplete")
        return False

    def add(self, ws, dw, n_slabs, n_elem):
        self.items.append((ws, dw, int(n_slabs), int(n_elem)))

    def take(self, C):
        """Hands the oldest pending reductions to a caller that finishes them inside its own launch (the filler role of
        alignq_conv3x3_nhwc_bwd_fill; C: its channel count); they leave the list.  Measured (NOTES.md 5f): the 16-channel
        launches absorb ~5 MB of slabs for nothing (their one-per-CU filter-gradient role outlasts the data-gradient tiles), the
        32- / 64-channel ones grow by about what the closing reduction saves (_WGRAD_FILL: items per launch by channel count)."""
        most = _WGRAD_FILL.get(C, 0)
        out, self.items = self.items[:most], self.items[most:]
        return out

    def take_site(self, B, F):
        """The same for a site's backward launch (alignq_site_bwd_apply_bn_fill): the narrow sites (F <= 8192) fill half the
        chip, the reductions run beside them for nothing (_WGRAD_FILL_SITE items per launch)."""
        most = min(_WGRAD_FILL_SITE, L.load().alignq_site_bwd_fill_slots(int(B), int(F)))
        out, self.items = self.items[:most], self.items[most:]
        return out

    def flush(self):
        flush_bwd_twins()
        if not self.items:
            return
        T = len(self.items)
        L.check(L.load().alignq_conv3x3_wgrad_reduce_multi(
            T, L.ptr_array([i[0] for i in self.items]), L.ptr_array([i[1] for i in self.items]),
            (ctypes.c_int * T)(*[i[2] for i in self.items]), (ctypes.c_int * T)(*[i[3] for i in self.items]),
            L.stream_ptr()), "alignq_conv3x3_wgrad_reduce_multi")
        self.items = []


_active_wgrads = None
# Filler roles: reductions of EARLIER launches ride in launches that leave part of the chip idle (values by measurement, NOTES.md).
# ALIGNQ_FILL=0 switches all of them off - the one runtime switch of the product path: the PMC passes (tools/make_profiles*.sh)
# need every launch to move its own role's bytes only.
_FILL_ON = os.environ.get("ALIGNQ_FILL", "1") != "0"
_SITE_FILL = 3 if _FILL_ON else 0                        # items per one-tile site forward launch
_WGRAD_FILL_SITE = 2 if _FILL_ON else 0                  # items per narrow site backward launch
_WGRAD_FILL = {16: 2, 32: 0, 64: 0} if _FILL_ON else {}  # items per convolution backward launch, by channel count


def active_wgrads():
    return _active_wgrads


# Lazy batch-norm input gradients: BNSiteFn.backward may hand the gradient g w.r.t. the BN OUTPUT to the convolution that
# produced z instead of the finished dz; ops.QConv3x3Fn.backward forms dz on load (alignq_conv3x3_nhwc_bwd, bn_z ...).
# Producer and consumer are tied together by a LazyLink object created in the convolution's forward (kept on its ctx and
# handed to the BN site with the partial statistics), NOT by the gradient tensor's address: if anything stands between the
# two backward nodes (a tensor hook or retain_grad on z, a second consumer of z whose gradient autograd adds), the tensor
# arriving at the convolution is no longer the posted one and the consumer raises instead of treating it as dz.
# Only used inside a DeferredWgrads context (TrainStep).
class LazyLink:
    __slots__ = ("record",)

    def __init__(self):
        self.record = None


_posted_links = []


def post_lazy_dz(link, g, z, ab, save, ktot, part=None, dgamma=None, dbeta=None):
    """ktot None: the consumer reduces the site backward's per-tile sums `part` itself (alignq_conv3x3_nhwc_bwd) and fills the
    batch-norm parameter gradients dgamma / dbeta, which autograd already holds."""
    link.record = (g, z, ab, save, ktot, part, dgamma, dbeta)
    _posted_links.append(link)


def take_lazy_dz(link, g):
    """The record posted for this convolution's output gradient, or None.  Raises if a record exists but the incoming
    gradient is not the posted tensor (it was altered or summed with another gradient on the way)."""
    flush_bwd_twins()            # (a parked site backward writes what the record below points at)
    if link is None or link.record is None:
        return None
    rec, link.record = link.record, None
    posted = rec[0]
    if g is None or g.data_ptr() != posted.data_ptr() or g.shape != posted.shape or g.stride() != posted.stride():
        raise RuntimeError(
            "alignq_amd: the gradient reaching the convolution is not the lazy batch-norm gradient its BN site posted "
            "(a tensor hook / retain_grad on the convolution output, or a second consumer of it, stands in between). "
            "Remove the hook or build the TrainStep with qconv=False / fuse_bn=False.")
    return rec


_active = None


def active_deferred():
    return _active


class _TwinState(threading.local):
    ctx = None


_twin = _TwinState()


class twin_sites:
    """Inside this context two BN-folded ADMM sites of ONE shape whose single launches leave half the chip idle (F <= 8192 at batch
    128) are launched TOGETHER (alignq_site_partials_bn_twin): the sites behind a transition block's two convolutions, model/
    resnet.py PreActBlock_conv_Q.forward - one node of the step's chain instead of two, bit-identical results.  The first site's
    forward parks its arguments, the second launches both; a site without a partner (or of another shape) is launched on its own,
    at the latest when the context ends.  Only the whole-model deferred step (fused.DeferredLosses) uses it."""

    def __init__(self):
        self.pending = None
        self.bwd = _BwdTwin()

    def __enter__(self):
        self.prev, _twin.ctx = _twin.ctx, self
        return self

    def __exit__(self, *exc):
        _twin.ctx = self.prev
        if exc[0] is None:
            self.flush()
        self.pending = None
        return False

    @staticmethod
    def _single(args, st):
        a = args
        L.check(L.load().alignq_site_partials_bn(a.z, a.bn_part, a.bn_gamma, a.bn_beta, a.running_mean, a.running_var,
                                                 a.num_batches_tracked, a.momentum, a.bn_eps, a.ab, a.save, a.C, a.HW, a.B, a.F, a.k,
                                                 a.act_range, a.eps, a.relu, a.residual, a.nhwc, a.conv_parts, a.xq, a.bins_out,
                                                 a.stats, a.ws, st), "alignq_site_partials_bn")

    def flush(self):
        if self.pending is not None:
            args, _keep, st = self.pending
            self.pending = None
            self._single(args, st)

    def add(self, args, keep, st):
        if self.pending is None:
            self.pending = (args, keep, st)
            return
        pa, _pk, pst = self.pending
        self.pending = None
        rc = L.load().alignq_site_partials_bn_twin(ctypes.byref(pa), ctypes.byref(args), st) if pst == st else L.EUNSUPPORTED
        if rc == L.EUNSUPPORTED:           # another shape, or a site that fills the chip alone: one after the other
            self._single(pa, pst)
            self._single(args, st)
        else:
            L.check(rc, "alignq_site_partials_bn_twin")


def _site_bwd_launch(lib, g_y, S, z, ab, save, C, HW, nhwc, y, ybins, dres, stats, B, F, act_range, eps, dx, part, fill, st):
    nf = len(fill)
    L.check(lib.alignq_site_bwd_apply_bn_fill(
        L.ptr(g_y), L.ptr(S), L.ptr(z), L.ptr(ab), L.ptr(save), C, HW, nhwc, L.ptr(y), L.ptr(ybins),
        ybins.element_size() if ybins is not None else 0, L.ptr(dres) if y is not None else None, L.ptr(stats), B, F,
        act_range, eps, L.ptr(dx), L.ptr(part), nf, L.ptr_array([f[0] for f in fill]) if nf else None,
        L.ptr_array([f[1] for f in fill]) if nf else None, (ctypes.c_int * nf)(*[f[2] for f in fill]) if nf else None,
        (ctypes.c_int * nf)(*[f[3] for f in fill]) if nf else None, st), "alignq_site_bwd_apply_bn_fill")


class _BwdTwin:
    """Where the backward launches of a twin pair meet (created by `twin_sites`, carried by both sites' autograd contexts): the first
    site's backward parks its arguments, the second launches both (alignq_site_bwd_apply_bn_twin).  The only reader of what the
    launch writes - the producing convolution's backward - calls flush_bwd_twins() before it takes the lazy records, so a site whose
    partner never ran is launched on its own there; DeferredWgrads flushes as well when the backward ends."""

    def __init__(self):
        self.pending = None

    @staticmethod
    def _single(a, st):
        L.check(L.load().alignq_site_bwd_apply_bn(a.g, a.S, a.z, a.ab, a.save, a.C, a.HW, a.nhwc, a.y_relu, a.y_bins, a.y_bin_bytes,
                                                  a.dresidual, a.stats, a.B, a.F, a.act_range, a.eps, a.dx, a.dx_part, st),
                "alignq_site_bwd_apply_bn")

    def flush(self):
        if self.pending is not None:
            args, _keep, st = self.pending
            self.pending = None
            if self in _bwd_twins.open:
                _bwd_twins.open.remove(self)
            self._single(args, st)

    def add(self, args, keep, st):
        if self.pending is None:
            self.pending = (args, keep, st)
            _bwd_twins.open.append(self)
            return
        pa, _pk, pst = self.pending
        self.pending = None
        if self in _bwd_twins.open:
            _bwd_twins.open.remove(self)
        rc = L.load().alignq_site_bwd_apply_bn_twin(ctypes.byref(pa), ctypes.byref(args), st) if pst == st else L.EUNSUPPORTED
        if rc == L.EUNSUPPORTED:
            self._single(pa, pst)
            self._single(args, st)
        else:
            L.check(rc, "alignq_site_bwd_apply_bn_twin")


class _BwdTwins(threading.local):
    def __init__(self):
        self.open = []


_bwd_twins = _BwdTwins()


def flush_bwd_twins():
    """Launch every parked site backward of this thread (see _BwdTwin) - called by whoever is about to read or to launch a reader of
    a site backward's outputs."""
    for t in list(_bwd_twins.open):
        t.flush()


# ------------------------------------------------------------------------------------------------------------------
class BNSiteFn(torch.autograd.Function):
    """act_q(bn(z)) with the batch-norm folded into the ADMM-site kernels (training mode).

    Forward: bn statistics (2 launches) -> site partials reading z with x = a*z + b on load -> slab reduce + loss.
    Backward: prep (S, dalterD, dgamma) -> site backward writing dx and per-tile BN sums -> bn backward apply (dz, dgamma_bn,
    dbeta).  The normalised activation is never written to HBM.  Values: x differs from torch's ((z-mean)*invstd)*gamma+beta
    only by the rounding of one fma (1e-7 relative)."""

    @staticmethod
    def forward(ctx, z, bn_weight, bn_bias, running_mean, running_var, nbt, momentum, bn_eps, alterD, gamma, k, act_range,
                eps, mu, rho, relu, rec=None, res=None, conv_part=None, pack=False):
        """pack (N2; only with relu and without a residual): the output is NOT written as fp32; the kernel stores
        the int8 / int16 level index of relu(x_q) instead (1-2 B per element) and the first return value is a data-less HANDLE
        (`packed_handle`) carrying it as `._alignq_bins = (bins, k)` for a consumer that reads indices (ops.QConv3x3Fn:
        forward and filter gradient); this node's own backward takes the ReLU mask from the index as well."""
        z = L.dense_f32(z, "conv output")
        ctx.twin_bwd = None
        nhwc = not z.is_contiguous()             # dense_f32 only lets contiguous or channels-last 4-D tensors through
        if res is not None:
            res = L.dense_f32(res, "residual")
            if res.shape != z.shape or res.stride() != z.stride():
                raise RuntimeError("residual must have the conv output's shape and memory layout")
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
        conv_parts = 0
        if nhwc and conv_part is not None:
            # the convolution that produced z left per-workgroup partial statistics: no pass over z here
            ws_bn, conv_parts = conv_part[0], conv_part[1]
        elif nhwc:
            ws_bn = torch.empty(lib.alignq_bn_nhwc_ws_bytes(C), dtype=torch.uint8, device=dev)
            L.check(lib.alignq_bn_partial_stats_nhwc(L.ptr(z), B, C, HW, L.ptr(ws_bn), st), "alignq_bn_partial_stats_nhwc")
        else:
            ws_bn = torch.empty(lib.alignq_bn_ws_bytes(C), dtype=torch.uint8, device=dev)
            L.check(lib.alignq_bn_partial_stats(L.ptr(z), B, C, HW, L.ptr(ws_bn), st), "alignq_bn_partial_stats")
        bins = None
        if pack:
            from . import ops
            bdt = ops.bin_dtype(k, act_range, L.FORMULA_ADMM)
            if not relu or res is not None or bdt is None or F % 4:
                raise RuntimeError("BNSiteFn(pack=True) needs relu, no residual, k <= 16 and F % 4 == 0")
            bins = torch.empty_strided(z.shape, z.stride(), dtype=bdt, device=dev)
            y = None
        else:
            y = torch.empty_like(z)
        D = torch.empty(B, B, dtype=torch.float32, device=dev)
        stats = torch.empty(4, F, dtype=torch.float32, device=dev)
        scal = rec.scal if rec is not None else torch.empty(4, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
        tw = _twin.ctx
        if (tw is not None and rec is not None and _active is not None and nhwc and res is None
                and lib.alignq_site_fill_slots(B, F) > 0):
            # twin launch (round 6): inside `twin_sites()` two half-chip sites of one shape share ONE launch - the first only parks its
            # arguments (its outputs are allocated; nobody reads them before the partner's forward launches both), no filler role
            args = L.SiteBnArgs(L.ptr(z), L.ptr(ws_bn), L.ptr(bn_weight), L.ptr(bn_bias), L.ptr(running_mean), L.ptr(running_var),
                                L.ptr(nbt), float(momentum), float(bn_eps), L.ptr(ab), L.ptr(save), C, HW, B, F, int(k),
                                float(act_range), float(eps), int(bool(relu)), None, int(nhwc), int(conv_parts), L.ptr(y), L.ptr(bins),
                                L.ptr(stats), L.ptr(ws))
            tw.add(args, (z, ws_bn, bn_weight, bn_bias, running_mean, running_var, nbt, ab, save, y, bins, stats, ws), st)
            ctx.twin_bwd = tw.bwd           # the pair's backward launches meet there again (BNSiteFn.backward)
        else:
            if tw is not None:
                tw.flush()         # (keep the launches in program order: a parked site goes first)
            # filler role: earlier sites' slab reductions ride in this launch when it leaves CUs idle (DeferredLosses.take_fill)
            fill = _active.take_fill(rec, B, F, dim, mu, rho) if (rec is not None and _active is not None) else []
            nf = len(fill)
            L.check(lib.alignq_site_partials_bn_fill(
                L.ptr(z), L.ptr(ws_bn), L.ptr(bn_weight), L.ptr(bn_bias), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt),
                float(momentum), float(bn_eps), L.ptr(ab), L.ptr(save), C, HW, B, F, int(k), float(act_range), float(eps),
                int(bool(relu)), L.ptr(res), int(nhwc), int(conv_parts), L.ptr(y), L.ptr(bins), L.ptr(stats), L.ptr(ws), nf,
                L.ptr_array([r.ws for r in fill]) if nf else None, L.ptr_array([r.D for r in fill]) if nf else None,
                L.ptr_array([r.A for r in fill]) if nf else None, L.ptr_array([r.Gm for r in fill]) if nf else None,
                L.ptr_array([r.scal for r in fill]) if nf else None, L.i64_array([r.F for r in fill]) if nf else None,
                dim, float(mu), float(rho), st), "alignq_site_partials_bn_fill")
        if rec is not None:      # reduced with all other sites in DeferredLosses.total()
            rec.ws, rec.D, rec.A, rec.Gm, rec.B, rec.F, rec.dim = ws, D, A, Gm, B, F, dim
            rec.mu, rec.rho = float(mu), float(rho)
        else:
            L.check(lib.alignq_site_reduce_loss(L.ptr(ws), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), dim, float(mu),
                                                float(rho), L.ptr(scal), st), "alignq_site_reduce_loss")
        ctx.rec = rec
        ctx.from_qconv = int(conv_part[2]) if (nhwc and conv_part is not None and len(conv_part) > 3) else 0
        ctx.link = conv_part[3] if ctx.from_qconv else None
        ctx.bn_params = (bn_weight, bn_bias)
        ctx.save_for_backward(z, ab, save, stats, D, A, Gm, scal, y if (relu and not pack) else None, bins)
        ctx.set_materialize_grads(False)
        ctx.cfg = (float(act_range), float(eps), float(mu), bn_weight is not None, bn_bias is not None, res is not None,
                   int(nhwc))
        ctx.mark_non_differentiable(D)
        if pack:
            y = packed_handle(z.shape, dev)
            L.MB.site_bins = (bins, int(k))
        return y, scal[0], D


    @staticmethod
    def backward(ctx, g_y, g_loss, _gD):
        z, ab, save, stats, D, A, Gm, scal, y, ybins = ctx.saved_tensors
        act_range, eps, mu, has_w, has_b, has_res, nhwc = ctx.cfg
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
        rec = ctx.rec
        if rec is not None and rec.prepared:
            S, dA, dG = rec.S, rec.dA, rec.dG
            rec.S = rec.dA = rec.dG = None       # sole owner: AccumulateGrad takes dA/dG without a copy
        else:
            S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, dtype=torch.float32, device=dev)       # fp32 S + bf16 image
            dA, dG = torch.empty_like(A), torch.empty_like(Gm)
            L.check(lib.alignq_site_prep_fused(L.ptr(D), L.ptr(A), L.ptr(Gm), dim, L.ptr(scal), mu, L.ptr(g_loss), B, F,
                                               L.ptr(S), L.ptr(dA), L.ptr(dG), st), "alignq_site_prep_fused")
        dx = torch.empty_like(z)
        part = torch.empty(lib.alignq_site_bn_part_bytes(F, nhwc), dtype=torch.uint8, device=dev)
        # gradient of the residual: the upstream gradient, ReLU-masked by the kernel when the ReLU was fused
        dres = None
        if has_res and g_y is not None:
            dres = torch.empty_like(z) if y is not None else g_y
        fresh = all(p is None or p.grad is None for p in ctx.bn_params)
        lazy2 = ctx.from_qconv == 2 and active_wgrads() is not None and active_wgrads().fresh_grads and fresh
        if ctx.twin_bwd is not None and lazy2 and g_y is not None:
            # the backward of a twin pair (round 6): nothing else is launched by this node on the lazy path, and the one consumer of dx
            # and of the per-tile sums - the producing convolution's backward - flushes the pair before it reads them (take_lazy_dz)
            args = L.SiteBwdBnArgs(L.ptr(g_y), L.ptr(S), L.ptr(z), L.ptr(ab), L.ptr(save), C, HW, nhwc, L.ptr(y), L.ptr(ybins),
                                   ybins.element_size() if ybins is not None else 0, L.ptr(dres) if y is not None else None,
                                   L.ptr(stats), B, F, act_range, eps, L.ptr(dx), L.ptr(part))
            ctx.twin_bwd.add(args, (g_y, S, z, ab, save, y, ybins, dres, stats, dx, part), st)
        else:
            flush_bwd_twins()
            # filler role: the narrow sites' launches take pending filter-gradient slab reductions along (DeferredWgrads.take_site)
            fill = active_wgrads().take_site(B, F) if active_wgrads() is not None else []
            _site_bwd_launch(lib, g_y, S, z, ab, save, C, HW, nhwc, y, ybins, dres, stats, B, F, act_range, eps, dx, part, fill, st)
        dgam = torch.empty(C, dtype=torch.float32, device=dev) if has_w else None
        dbet = torch.empty(C, dtype=torch.float32, device=dev) if has_b else None
        # the in-kernel form fills dgam / dbet AFTER autograd has adopted them as .grad: only valid while both .grad are
        # really None (the DeferredWgrads(fresh_grads=True) promise, checked here rather than trusted: `fresh` above)
        if lazy2:
            # z is the output of a 3x3 body convolution whose fused backward reduces the per-tile sums itself: nothing is
            # launched here; dgam / dbet are filled by that kernel (later in this backward, before anything reads them).
            # Autograd receives VIEWS: it adopts a gradient as .grad only while nobody else holds the tensor object (it would
            # copy the still unwritten buffer otherwise); the buffers themselves travel with the lazy record.
            post_lazy_dz(ctx.link, dx, z, ab, save, None, part, dgam, dbet)
            return (dx, None if dgam is None else dgam.view_as(dgam), None if dbet is None else dbet.view_as(dbet), None, None,
                    None, None, None, dA, dG, None, None, None, None, None, None, None, dres, None, None)
        if ctx.from_qconv and active_wgrads() is not None:
            # z is the output of one of ops' convolutions and a whole-model backward is running: only the per-channel totals
            # are computed here; the convolution's backward forms dz = a*(g - k0 - zhat*k1) on load (no elementwise pass)
            ktot = torch.empty(2, C, dtype=torch.float32, device=dev)
            L.check(lib.alignq_bn_bwd_totals(L.ptr(part), B, C, HW, L.ptr(ktot), L.ptr(dgam), L.ptr(dbet), st),
                    "alignq_bn_bwd_totals")
            post_lazy_dz(ctx.link, dx, z, ab, save, ktot)
            return (dx, dgam, dbet, None, None, None, None, None, dA, dG, None, None, None, None, None, None, None, dres, None,
                    None)
        dz = torch.empty_like(z)
        L.check(lib.alignq_bn_bwd_apply(L.ptr(dx), L.ptr(z), L.ptr(ab), L.ptr(save), L.ptr(part), B, C, HW, nhwc, L.ptr(dz),
                                        L.ptr(dgam), L.ptr(dbet), st), "alignq_bn_bwd_apply")
        return (dz, dgam, dbet, None, None, None, None, None, dA, dG, None, None, None, None, None, None, None, dres, None, None)


def packed_handle(shape, device):
    """A data-less fp32 tensor of the given shape (one element of storage, all strides 0): what a packed site returns in
    place of its fp32 output.  It carries the autograd edge and the shape; the VALUES live in `._alignq_bins`.  Anything that
    is not an index-reading consumer must go through `materialize` (Conv2d_Q does)."""
    key = (device.type, device.index)
    base = _handle_base.get(key)
    if base is None:                       # one persistent element per device: creating a handle launches nothing
        base = _handle_base[key] = torch.zeros(1, dtype=torch.float32, device=device)
    return base.expand(shape)


_handle_base = {}


class MaterializeFn(torch.autograd.Function):
    """handle + bins -> the fp32 tensor (alignq_bins_dequant; the index is already ReLU-clamped); gradient passes through."""

    @staticmethod
    def forward(ctx, handle, bins, k, act_range):
        from . import ops
        return ops.dequant_bins(bins, k, act_range, L.FORMULA_ADMM)

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


def materialize(t):
    """t itself, or — for a packed handle — the fp32 tensor its level indices stand for."""
    info = getattr(t, "_alignq_bins", None)
    if info is None:
        return t
    from . import config
    return MaterializeFn.apply(t, info[0], info[1], config.args.act_range)


def _is_nhwc(z) -> bool:
    return z.dim() == 4 and not z.is_contiguous() and z.is_contiguous(memory_format=torch.channels_last)


def bn_site_fusable(bn, act, z) -> bool:
    from . import config
    if not (isinstance(bn, torch.nn.BatchNorm2d) and bn.training and bn.track_running_stats and bn.momentum is not None):
        return False
    if not (z.is_cuda and z.dim() == 4 and z.dtype == torch.float32):
        return False
    B, C, H, W = z.shape
    if z.is_contiguous():
        layout_ok = (H * W) % 64 == 0                       # one channel per 64-feature tile
    elif _is_nhwc(z):
        layout_ok = 4 <= C <= 256 and (C & (C - 1)) == 0 and (C * H * W) % 64 == 0     # channel = f mod C
    else:
        return False
    a_bit = getattr(act, "a_bit", 32)
    return (layout_ok and 64 < B <= L.MAX_BATCH and hasattr(act, "opt") and a_bit < 32
            and config.args.method == "ours" and act.opt.alterD.shape[0] >= B)


def bn_site(bn, act, z, eps=0.0, relu=False, residual=None, pack=False):
    """out, loss = act(bn(z)) [; out = out + residual] [; out = relu(out) when relu=True] — folded into the site kernels
    when `bn_site_fusable`, otherwise exactly that composition.  z may be contiguous (NCHW) or channels-last.
    pack=True (N2; needs relu, no residual, channels-last, a_bit <= 8): `out` is a packed handle (see BNSiteFn.forward) whose
    values are int8 / int16 level indices; only hand it to Conv2d_Q (which reads the indices) or through `materialize`."""
    from . import config, ops
    if not bn_site_fusable(bn, act, z) or (residual is not None and not (
            residual.shape == z.shape and residual.stride() == z.stride() and residual.dtype == torch.float32)):
        # not foldable (batch above 128 rows, the exact-global correlation, ...): the batch-norm alone still runs on the folded family's
        # kernels when the tensor is channels-last fp32 in training mode (bn_only; MIOpen's spatial batch-norm took 0.62 ms per layer and
        # step at [1024,16,32,32]: a third of the exact-global step), else the module itself
        out, loss = act(bn_only(bn, z))
        if residual is not None:
            out = out + residual
        return (torch.nn.functional.relu(out) if relu else out), loss
    admm = act.opt
    deferred = active_deferred()
    rec = deferred.new_record(z.shape[0], z.device) if deferred is not None else None
    pack = bool(pack and relu and residual is None and _is_nhwc(z) and act.a_bit <= 8 and (z.shape[1] * z.shape[2] * z.shape[3]) % 4 == 0
                and ops.bin_dtype(act.a_bit, config.args.act_range, L.FORMULA_ADMM) is not None)   # a large act_range has no narrow form
    L.MB.site_bins = None
    y, loss, D = BNSiteFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                bn.momentum, bn.eps, admm.alterD, admm.gamma, act.a_bit, config.args.act_range, eps,
                                admm.mu, admm.rho, relu, rec, residual,
                                getattr(z, "_alignq_bn_part", None) if _is_nhwc(z) else None, pack)
    if pack:
        y._alignq_bins = L.MB.site_bins
        L.MB.site_bins = None
    admm.D = D
    if deferred is not None:
        if rec is not None:
            deferred.add_record_loss(loss)
        else:
            deferred.add(loss)
        return y, 0.0
    return y, loss


# ------------------------------------------------------------------------------------------------------------------
# N1 on the Office path (configuration 5): training-mode batch-norm folded into the PLAIN quantiser + ReLU of the bottleneck
# (`out = relu(act_q1(bn1(conv1(x))))`, dann_office/model/resnet.py:134-143; stem :230-233).  MIOpen produces z; the fold is
# statistics (one read of z) + one elementwise pass (a*z+b -> quantise -> relu) forward and two passes backward: the
# normalised activation never exists in memory.  Any batch; channels-last; C a power of two in [4, 1024].
class BNQuantReluFn(torch.autograd.Function):
    """groups > 1 (the Office step's merged source + target pass, train_step.OfficeTrainStep(dual=True)): z holds `groups`
    equal batch slices that the reference pushes through the module one after the other (dann_office/main.py:351-372): every
    slice gets its own batch statistics, the running statistics are updated slice after slice, and dgamma / dbeta are the sums
    over the slices - one autograd node, so no gradient accumulation kernels."""

    @staticmethod
    def forward(ctx, z, weight, bias, running_mean, running_var, nbt, momentum, bn_eps, k, act_range, formula, relu, groups=1,
                residual=None, conv_part=None, pack=False):
        """residual: added to the quantised value before the ReLU in the same pass (the CDF-only block's `out += shortcut;
        relu`); its gradient - the masked upstream gradient - is a second output of the backward's apply pass."""
        z = L.dense_f32(z, "conv output")
        if residual is not None:
            residual = L.like_layout(L.dense_f32(residual, "residual"), z)
        B, C, H, W = z.shape
        lib = L.load()
        dev = z.device
        Bg = B // groups
        P = Bg * H * W
        ab = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        save = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        # pack (N2 on the Office path; the caller checked: ADMM formula, no residual, int16 index, f16-exact range): the output is NOT
        # written as fp32; the kernel stores the ReLU-clamped level index as int16 and the function returns a data-less handle
        # (packed_handle) that carries it as `._alignq_bins` for consumers that read indices (ops.QConvGemmFn)
        bins = torch.empty(z.shape, dtype=torch.int16, device=dev, memory_format=torch.channels_last) if pack else None
        y = None if pack else torch.empty_like(z)
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=dev)
        # the ReLU mask the backward needs, one bit per element (round 4): the node keeps 1/32 of a tensor instead of reading the
        # fp32 y twice; y itself belongs to whoever consumes it
        mask = torch.empty(lib.alignq_bnq_mask_bytes(P, C, groups), dtype=torch.uint8, device=dev) if relu else None
        # conv_part: (double [groups, parts, C, 2], parts) left by the producing convolution's epilogue (ops.QConvGemmFn): the
        # statistics pass over z is skipped
        cp, cn = (conv_part[0], int(conv_part[1])) if conv_part is not None else (None, 0)
        L.check(lib.alignq_bnq_fwd_parts(L.ptr(z), P, C, groups, L.ptr(weight), L.ptr(bias), L.ptr(running_mean), L.ptr(running_var),
                                         L.ptr(nbt), float(momentum), float(bn_eps), int(k), float(act_range), int(formula),
                                         int(bool(relu)), L.ptr(residual), L.ptr(ab), L.ptr(save), L.ptr(y), L.ptr(mask), L.ptr(ws),
                                         L.ptr(cp), cn, L.ptr(bins), L.stream_ptr()),
                "alignq_bnq_fwd_parts")
        if pack:
            y = packed_handle(z.shape, dev)
            L.MB.bnq_bins = (bins, int(k))
        ctx.save_for_backward(z, (mask if mask is not None else y) if relu else None, ab, save)
        ctx.bitmask = mask is not None
        ctx.has_res = residual is not None
        ctx.cfg = (float(act_range), bool(relu), weight is not None, bias is not None, int(groups))
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, g):
        z, y, ab, save = ctx.saved_tensors
        act_range, relu, has_w, has_b, groups = ctx.cfg
        B, C, H, W = z.shape
        Bg = B // groups
        g = L.like_layout(g, z)
        lib = L.load()
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device) if has_w else None
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device) if has_b else None
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=z.device)
        ym, mk = (None, y) if ctx.bitmask else (y, None)
        # the residual's gradient: g itself without a ReLU behind the sum, else the masked g (written by the apply pass)
        dres = torch.empty_like(z) if (ctx.has_res and relu and ctx.needs_input_grad[13]) else None
        L.check(lib.alignq_bnq_bwd(L.ptr(g), L.ptr(z), L.ptr(ym), L.ptr(mk), L.ptr(ab), L.ptr(save), Bg * H * W, C, groups, act_range,
                                   int(relu), L.ptr(dz), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), L.stream_ptr()),
                "alignq_bnq_bwd")
        if ctx.has_res and not relu and ctx.needs_input_grad[13]:
            dres = g
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, dres, None, None



def _bn_nhwc_ok(bn, z, groups=1) -> bool:
    """groups: the batch slices that are normalised separately (the Office step's source + target halves): the batch must
    divide into them and EACH slice needs its >= 2 values per channel (alignq_bnq_* compute var * n / (n - 1) per slice)."""
    C = z.shape[1] if z.dim() == 4 else 0
    return (bn.training and z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and _is_nhwc(z)
            and 4 <= C <= 2048 and (C & (C - 1)) == 0 and bn.track_running_stats and bn.momentum is not None
            and groups >= 1 and z.shape[0] % groups == 0 and (z.shape[0] // groups) * z.shape[2] * z.shape[3] >= 2)


def _slices(z, groups):
    if z.shape[0] % groups:
        raise ValueError(f"a batch of {z.shape[0]} rows does not split into {groups} equal slices")
    Bg = z.shape[0] // groups
    return [z[i * Bg:(i + 1) * Bg] for i in range(groups)]


def bnq_fusable(bn, act, z, groups=1) -> bool:
    return _bn_nhwc_ok(bn, z, groups) and act.a_bit < 32


class BNAffineFn(torch.autograd.Function):
    """Training-mode nn.BatchNorm2d alone on a channels-last tensor (the Office bottleneck's downsample branch,
    dann_office/model/resnet.py:122-126): alignq_bnq_stats + alignq_bnq_affine forward, alignq_bnq_bwd_dx backward.
    groups: as in BNQuantReluFn."""

    @staticmethod
    def forward(ctx, z, weight, bias, running_mean, running_var, nbt, momentum, bn_eps, groups=1, conv_part=None):
        z = L.dense_f32(z, "conv output")
        B, C, H, W = z.shape
        Bg = B // groups
        lib, dev, P = L.load(), z.device, Bg * H * W
        cp, cn = (conv_part[0], int(conv_part[1])) if conv_part is not None else (None, 0)
        ab = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        save = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        y = torch.empty_like(z)
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=dev)
        st = L.stream_ptr()
        L.check(lib.alignq_bnq_stats_parts(L.ptr(z), P, C, groups, L.ptr(weight), L.ptr(bias), L.ptr(running_mean),
                                           L.ptr(running_var), L.ptr(nbt), float(momentum), float(bn_eps), L.ptr(ab), L.ptr(save),
                                           L.ptr(ws), L.ptr(cp), cn, st), "alignq_bnq_stats_parts")
        L.check(lib.alignq_bnq_affine(L.ptr(z), L.ptr(ab), P, C, groups, L.ptr(y), st), "alignq_bnq_affine")
        ctx.save_for_backward(z, ab, save)
        ctx.has = (weight is not None, bias is not None, int(groups))
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, g):
        z, ab, save = ctx.saved_tensors
        has_w, has_b, groups = ctx.has
        B, C, H, W = z.shape
        Bg = B // groups
        g = L.like_layout(g, z)
        lib = L.load()
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device) if has_w else None
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device) if has_b else None
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=z.device)
        L.check(lib.alignq_bnq_bwd_dx(L.ptr(g), L.ptr(z), L.ptr(ab), L.ptr(save), Bg * H * W, C, groups, L.ptr(dz), L.ptr(dgamma),
                                      L.ptr(dbeta), L.ptr(ws), L.stream_ptr()), "alignq_bnq_bwd_dx")
        return dz, dgamma, dbeta, None, None, None, None, None, None, None


class _ConvGroups(threading.local):
    n = 1


_conv_groups = _ConvGroups()


class conv_groups_scope:
    """How many equal batch slices the tensors of the current traversal hold (resnet_office.ResNet.forward(groups)): a convolution
    that leaves batch-norm partial statistics (Conv2d_Q.emit_bn_stats) sums them per slice.  A scope, per thread (as a
    module global that ResNet.forward only ever set, a Conv2d_Q called OUTSIDE a traversal after a merged one inherited groups = 2
    and alignq_qconv_fwd refused its batch): the count is back to what it was when the traversal ends."""

    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        self.prev, _conv_groups.n = _conv_groups.n, self.groups
        return self

    def __exit__(self, *exc):
        _conv_groups.n = self.prev
        return False


def conv_groups():
    return _conv_groups.n


def conv_partials(z, groups):
    """(partials, parts) a GEMM convolution left on its output for `groups` batch slices (ops.QConvGemmFn.apply_with_stats), or None"""
    rec = getattr(z, "_alignq_bnq_part", None)
    if rec is None or rec[2] != int(groups):
        return None
    return rec[0], rec[1]


def bn_only(bn, z, groups=1):
    """bn(z): the folded-family kernels when the tensor is channels-last fp32 in training mode, else the module itself
    (groups > 1: applied to the batch slices one after the other, as the reference's successive passes do)."""
    if not _bn_nhwc_ok(bn, z, groups):
        if groups == 1:
            return bn(z)
        return torch.cat([bn(zz) for zz in _slices(z, groups)], 0)
    return BNAffineFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                            groups, conv_partials(z, groups))


class Site1Record:
    """One folded small-batch site of a Site1Batch: its buffers between the forward launch and the deferred reduction / preparation."""
    __slots__ = ("ws", "D", "A", "Gm", "scal", "B", "F", "groups", "dim", "mu", "rho", "S", "rA", "rG", "prepared", "g_ptr")


_site1_batch = None


def active_site1():
    return _site1_batch


class Site1LossSumFn(torch.autograd.Function):
    """total = the sum of the sites' loss vectors.  Forward: alignq_site1_groups_reduce_loss_multi (D and the loss scalars of every
    site and slice in one launch), then one concatenation + sum.  Backward: alignq_site1_groups_prep_multi with the upstream
    scalar - the loss vectors are inputs, so autograd runs this node before any site's backward (as LossSumFn for the CIFAR sites)."""

    @staticmethod
    def forward(ctx, batch, *loss_vecs):
        batch.reduce_all()
        ctx.recs = list(batch.records)
        ctx.shapes = [tuple(v.shape) for v in loss_vecs]
        return torch.cat([v.reshape(-1) for v in loss_vecs]).sum()

    @staticmethod
    def backward(ctx, g):
        lib, st = L.load(), L.stream_ptr()
        g = L.dev_f32(g, "loss grad")
        by_key = {}
        for r in ctx.recs:
            by_key.setdefault((r.B, r.groups, r.dim, r.mu, r.rho), []).append(r)
        for (B, groups, dim, mu, rho), recs in by_key.items():
            dev = recs[0].D.device
            n_s = lib.alignq_site_bwd_ws_bytes(B) // 4 * groups
            for r in recs:
                r.S = torch.empty(n_s, dtype=torch.float32, device=dev)
                r.rA, r.rG = torch.empty_like(r.A), torch.empty_like(r.Gm)
            L.check(lib.alignq_site1_groups_prep_multi(
                len(recs), L.ptr_array([r.D for r in recs]), L.ptr_array([r.A for r in recs]), L.ptr_array([r.Gm for r in recs]), dim,
                L.ptr_array([r.scal for r in recs]), mu, L.ptr(g), B, L.i64_array([r.F for r in recs]), groups,
                L.ptr_array([r.S for r in recs]), L.ptr_array([r.rA for r in recs]), L.ptr_array([r.rG for r in recs]), st),
                "alignq_site1_groups_prep_multi")
            for r in recs:
                r.prepared = True
                r.g_ptr = g.data_ptr()        # BNSite1Fn.backward checks that what reaches it IS this scalar, expanded (see there)
        return (None,) + tuple(g.expand(sh) for sh in ctx.shapes)


class Site1Batch:
    """The Office step's merged traversal (resnet_office.ResNet.forward, groups > 1): while active, the folded bottleneck tails
    (BNSite1Fn with a loss vector) launch only their site kernel; `total()` reduces every site's slabs to D + loss in ONE launch
    (off the forward's critical path: only y feeds the next layer) and its backward prepares every site's S / dalterD / dgamma in
    ONE launch - 2 launches per iteration instead of 32.  Bit-identical to the per-site launches."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _site1_batch
        self.records = []
        self._outer = _site1_batch
        _site1_batch = self
        return self

    def __exit__(self, *exc):
        global _site1_batch
        _site1_batch = self._outer
        self.records = []         # (the loss node holds its own list until the backward has run)
        return False

    def add(self, **kw):
        rec = Site1Record()
        for k_, v in kw.items():
            setattr(rec, k_, v)
        rec.prepared = False
        self.records.append(rec)
        return rec

    def reduce_all(self):
        lib, st = L.load(), L.stream_ptr()
        by_key = {}
        for r in self.records:
            by_key.setdefault((r.B, r.groups, r.dim, r.mu, r.rho), []).append(r)
        for (B, groups, dim, mu, rho), recs in by_key.items():
            L.check(lib.alignq_site1_groups_reduce_loss_multi(
                len(recs), L.ptr_array([r.ws for r in recs]), L.i64_array([r.F for r in recs]), B, groups,
                L.ptr_array([r.D for r in recs]), L.ptr_array([r.A for r in recs]), L.ptr_array([r.Gm for r in recs]), dim, mu, rho,
                L.ptr_array([r.scal for r in recs]), st), "alignq_site1_groups_reduce_loss_multi")

    def total(self, loss_vecs):
        """Sum of the tensors in loss_vecs (the sites' loss vectors and whatever else the caller collected)."""
        if not self.records:
            return torch.cat([t.reshape(-1) for t in loss_vecs]).sum()
        return Site1LossSumFn.apply(self, *loss_vecs)


_S1_BN_COLS = True        # False: alignq_site1_groups_bwd + alignq_bnq_bwd_dx (a test's comparison arm; not an environment switch)
_S1_RMASK = True          # False: the bottleneck tail's backward re-reads y for its ReLU mask (tools/ab_office.py's comparison arm)


class BNSite1Fn(torch.autograd.Function):
    """The Office bottleneck's tail with its batch-norm folded in (batches <= 32, channels-last):
        out, loss = act_q3(bn3(z)); out += identity; out = relu(out)          (dann_office/model/resnet.py:146-154)
    forward : alignq_bnq_stats (one read of z) -> alignq_site_partials_res_ab (x = a*z + b on load: quantise, both Grams,
              + residual, ReLU) -> alignq_site_reduce_loss; bn3's output is never written;
    backward: ReLU mask -> alignq_site_prep_fused -> alignq_site_bwd_apply_ab (dx w.r.t. the batch-norm output) ->
              alignq_bnq_bwd_dx in place (dz, dgamma, dbeta).
    groups > 1: z / residual hold `groups` batch slices of <= 32 rows each that the reference sends through the module one
    after the other (source pass, target pass): per-slice statistics and correlation matrices, running statistics updated in
    slice order, the loss is the SUM over the slices, D is the LAST slice's (ADMM.forward overwrites self.D, utils/admm.py:25),
    parameter gradients summed inside the node."""

    @staticmethod
    def forward(ctx, z, weight, bias, running_mean, running_var, nbt, momentum, bn_eps, residual, alterD, gamma, k,
                act_range, eps, mu, rho, groups=1, tok=None, loss_vec=False, conv_part=None):
        ctx.tok = tok             # GradFork's mailbox (see there): a dict shared with whoever forks this node's output
        z = L.dense_f32(z, "conv output")
        A, Gm = L.dev_f32(alterD, "alterD"), L.dev_f32(gamma, "gamma")
        Bt, C, H, W = z.shape
        B = Bt // groups
        F, P = C * H * W, B * H * W
        lib, dev = L.load(), z.device
        if B > A.shape[0]:
            raise RuntimeError(f"batch {B} larger than ADMM dim {A.shape[0]}")
        ab = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        save = torch.empty(groups, 2, C, dtype=torch.float32, device=dev)
        ws_bn = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=dev)
        st = L.stream_ptr()
        if residual is not None:
            residual = L.like_layout(L.dense_f32(residual, "residual"), z)
        y = torch.empty_like(z)
        stats = torch.empty(groups, 4, F, dtype=torch.float32, device=dev)
        D = torch.empty(groups, B, B, dtype=torch.float32, device=dev)
        scal = torch.empty(groups, 4, dtype=torch.float32, device=dev)
        from .ops import _ws
        batch = active_site1() if loss_vec else None
        if batch is not None:      # the slabs wait for the batch's reduction launch: a buffer of this site's own
            ws = torch.empty(lib.alignq_site_ws_bytes(B, F) * groups, dtype=torch.uint8, device=dev)
        else:
            ws = _ws(lib.alignq_site_ws_bytes(B, F) * groups, dev)
        cp, cn = (conv_part[0], int(conv_part[1])) if conv_part is not None else (None, 0)
        L.check(lib.alignq_bnq_stats_parts(L.ptr(z), P, C, groups, L.ptr(weight), L.ptr(bias), L.ptr(running_mean),
                                           L.ptr(running_var), L.ptr(nbt), float(momentum), float(bn_eps), L.ptr(ab), L.ptr(save),
                                           L.ptr(ws_bn), L.ptr(cp), cn, st), "alignq_bnq_stats_parts")
        # every slice in ONE launch per kernel (blockIdx.y = slice; the slices' workspace regions lie back to back)
        # (round 5) the stored output's sign bits for the backward's ReLU mask: 1 bit per element instead of re-reading y (the plain
        # sites have had that since round 4); only with the bounded quantiser form the kernel takes the mask in (2 <= k <= 8 here)
        rmask = None
        if _S1_RMASK and _S1_BN_COLS and residual is not None and 2 <= int(k) <= 8 and abs(float(act_range)) <= 8.0:
            rmask = torch.empty(lib.alignq_site1_mask_bytes(B, F, groups), dtype=torch.uint8, device=dev)
            L.check(lib.alignq_site1_groups_fwd_m(L.ptr(z), L.ptr(ab), C, B, F, groups, int(k), float(act_range), float(eps),
                                                  L.ptr(residual), 1, L.ptr(y), L.ptr(stats), L.ptr(ws), L.ptr(rmask), st),
                    "alignq_site1_groups_fwd_m")
        else:
            L.check(lib.alignq_site1_groups_fwd(L.ptr(z), L.ptr(ab), C, B, F, groups, int(k), float(act_range), float(eps),
                                                L.ptr(residual), 1, L.ptr(y), L.ptr(stats), L.ptr(ws), st), "alignq_site1_groups_fwd")
        ctx.rmask = rmask
        ctx.rec = None
        if batch is not None:
            ctx.rec = batch.add(ws=ws, D=D, A=A, Gm=Gm, scal=scal, B=B, F=F, groups=int(groups), dim=int(A.shape[0]), mu=float(mu),
                                rho=float(rho))
        else:
            L.check(lib.alignq_site1_groups_reduce_loss(L.ptr(ws), B, F, groups, L.ptr(D), L.ptr(A), L.ptr(Gm), A.shape[0], float(mu),
                                                        float(rho), L.ptr(scal), st), "alignq_site1_groups_reduce_loss")
        ctx.save_for_backward(z, y, ab, save, stats, D, A, Gm, scal)
        ctx.cfg = (float(act_range), float(eps), float(mu), weight is not None, bias is not None, residual is not None,
                   int(groups))
        ctx.set_materialize_grads(False)
        Dlast = D[groups - 1]
        if loss_vec:
            # the slices' losses as a VECTOR (a view: no kernel); the caller sums all sites' vectors at once.  One elementwise
            # addition per site fewer on the in-order chain (16 per Office step); the backward takes a gradient per slice.
            loss = scal[:, 0]
        else:
            loss = scal[0, 0]
            for gi in range(1, groups):
                loss = loss + scal[gi, 0]
        ctx.mark_non_differentiable(Dlast, *[t for t in (running_mean, running_var, nbt) if t is not None])
        return y, loss, Dlast

    @staticmethod
    def backward(ctx, g_y, g_loss, _gD):
        z, y, ab, save, stats, D, A, Gm, scal = ctx.saved_tensors
        act_range, eps, mu, has_w, has_b, has_res, groups = ctx.cfg
        Bt, C, H, W = z.shape
        B = Bt // groups
        F, P = C * H * W, B * H * W
        lib, dev = L.load(), z.device
        st = L.stream_ptr()
        g_m = None
        g_y2 = ctx.tok.pop("extra", None) if ctx.tok is not None else None      # the shortcut's addend, left by GradFork.backward
        if g_y2 is not None:
            if g_y is None or g_y2.shape != z.shape:
                g_y = g_y2 if g_y is None else g_y + g_y2                        # (forms without the second pointer: add here)
                g_y2 = None
            else:
                g_y2 = L.like_layout(g_y2, z)
        if g_y is not None:      # the fused ReLU's mask is applied by the site kernel, which also leaves the masked gradient in g_m
            g_y = L.like_layout(g_y, z)
            g_m = torch.empty_like(z) if has_res else None
        rec = ctx.rec
        # Site1LossSumFn.backward has prepared every site of the batch (a second backward over a retained graph finds the record
        # emptied and prepares this site by itself)
        prepared = rec is not None and rec.prepared and rec.S is not None
        if prepared and not (g_loss is not None and g_loss.dim() == 1 and g_loss.stride(0) == 0
                             and g_loss.data_ptr() == getattr(rec, "g_ptr", None)):
            # The record was prepared with the ONE upstream scalar of the summed loss.  That is this site's gradient only if its loss
            # vector reached the total unscaled and through no other path (resnet_office.ResNet.forward); a caller that weights the
            # vector after total(), feeds it to total() twice or through another path as well hands over something else - autograd then
            # delivers a tensor that is not the expanded scalar: prepare this site by itself from the gradient that arrived
            # (a vector left OUT of total() never gets its values at all: they exist only once total() has launched the reduction)
            prepared = False
            rec.S = rec.rA = rec.rG = None
        if prepared:
            # (the record lets go of them: a gradient tensor somebody else still references is CLONED by autograd's accumulation
            # node instead of taken over - 2 copies of 5 us per site)
            S, rA, rG = rec.S, rec.rA, rec.rG
            rec.S = rec.rA = rec.rG = None
        else:
            if g_loss is None:
                g_loss = torch.zeros((), dtype=torch.float32, device=dev)
            if g_loss.dim() == 0:
                g_loss, gs_stride = L.dev_f32(g_loss, "loss grad"), 0
            else:       # loss_vec: one upstream gradient per slice, read where autograd left it (an expanded tensor has stride 0)
                if not (g_loss.is_cuda and g_loss.dtype == torch.float32 and tuple(g_loss.shape) == (groups,)):
                    raise RuntimeError("BNSite1Fn: the gradient of the loss vector must be a float32 device tensor of shape [groups]")
                gs_stride = int(g_loss.stride(0))
            # alignq_site1_groups_prep writes dalterD / dgamma already summed over the slices
            rA, rG = torch.empty_like(A), torch.empty_like(Gm)
            from .ops import _ws
            S = _ws(lib.alignq_site_bwd_ws_bytes(B) * groups, dev)
        dx = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if has_w else None
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if has_b else None
        ws_bn = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=dev)
        # one preparation launch (S per slice; dalterD / dgamma summed over the slices in slice order) and one backward launch
        if not prepared:
            L.check(lib.alignq_site1_groups_prep(L.ptr(D), L.ptr(A), L.ptr(Gm), A.shape[0], L.ptr(scal), mu, L.ptr(g_loss), gs_stride, B,
                                                 F, groups, L.ptr(S), L.ptr(rA), L.ptr(rG), st), "alignq_site1_groups_prep")
        if _S1_BN_COLS:
            # the site kernel leaves the batch-norm backward's sums per feature column; a small reduction, the finalisation and dz
            # (in place) follow in the same entry: no pass of its own over dx and z
            cols = torch.empty(lib.alignq_site1_cols_bytes(F, groups), dtype=torch.uint8, device=dev)
            if ctx.rmask is not None:
                L.check(lib.alignq_site1_groups_bwd_bn_m(L.ptr(g_y), L.ptr(g_y2), L.ptr(None if g_y is None else ctx.rmask), L.ptr(S),
                                                         L.ptr(z), L.ptr(ab), L.ptr(save), C, L.ptr(stats), B, F, groups, act_range, eps,
                                                         L.ptr(dx), L.ptr(g_m), L.ptr(dgamma), L.ptr(dbeta), L.ptr(cols), L.ptr(ws_bn),
                                                         st), "alignq_site1_groups_bwd_bn_m")
            else:
                L.check(lib.alignq_site1_groups_bwd_bn(L.ptr(g_y), L.ptr(g_y2), L.ptr(None if g_y is None else y), L.ptr(S), L.ptr(z),
                                                       L.ptr(ab), L.ptr(save), C, L.ptr(stats), B, F, groups, act_range, eps, L.ptr(dx),
                                                       L.ptr(g_m), L.ptr(dgamma), L.ptr(dbeta), L.ptr(cols), L.ptr(ws_bn), st),
                        "alignq_site1_groups_bwd_bn")
        else:       # (the two-launch form, kept as the comparison arm of tests/test_gpu_round4.py: the sums from a pass over dx and z)
            L.check(lib.alignq_site1_groups_bwd(L.ptr(g_y), L.ptr(g_y2), L.ptr(None if g_y is None else y), L.ptr(S), L.ptr(z),
                                                L.ptr(ab), C, L.ptr(stats), B, F, groups, act_range, eps, L.ptr(dx), L.ptr(g_m), st),
                    "alignq_site1_groups_bwd")
            L.check(lib.alignq_bnq_bwd_dx(L.ptr(dx), L.ptr(z), L.ptr(ab), L.ptr(save), P, C, groups, L.ptr(dx), L.ptr(dgamma),
                                          L.ptr(dbeta), L.ptr(ws_bn), st), "alignq_bnq_bwd_dx")
        return (dx, dgamma, dbeta, None, None, None, None, None, g_m if has_res else None, rA, rG, None, None,
                None, None, None, None, None, None, None)


def bn_site_res_relu(bn, act, z, residual, eps, groups=1, loss_vec=False):
    """(relu(act(bn(z))[0] + residual), loss) for an ADMM site at a batch of at most 32 rows (per group): the folded chain when
    the tensor is channels-last fp32 in training mode (and no deferred-loss context is active), else None (the caller composes
    it).  loss_vec (groups > 1): the loss comes back as the VECTOR of the slices' losses (a view, no kernel) for a caller that sums
    all its sites at once (resnet_office.ResNet.forward), instead of their sum."""
    from . import config
    if not (_bn_nhwc_ok(bn, z, groups) and 2 <= z.shape[0] // groups <= 32 and act.a_bit < 32
            and config.args.method == "ours"
            and active_deferred() is None and residual is not None and residual.shape == z.shape and residual.is_cuda
            and residual.dtype == torch.float32 and z.shape[0] // groups <= act.opt.alterD.shape[0]):
        return None
    admm = act.opt
    tok = {}
    y, loss, D = BNSite1Fn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                                 bn.eps, residual, admm.alterD, admm.gamma, act.a_bit, config.args.act_range, eps, admm.mu, admm.rho,
                                 groups, tok, bool(loss_vec and groups > 1), conv_partials(z, groups))
    if tok is not None:
        y._alignq_site_tok = tok          # read by fork_block_input (the next bottleneck)
    admm.D = D
    return y, loss


class GradFork(torch.autograd.Function):
    """x -> (x, x) for a block input that feeds BOTH the block's first convolution and its shortcut (dann_office/model/
    resnet.py:131-154 without a downsample branch) when x is the output of a folded small-batch site (BNSite1Fn).  Autograd
    would form grad x = g_conv + g_shortcut in an elementwise pass of its own (12 B/element at every such block input); here
    the backward hands g_conv on as "the" gradient and leaves g_shortcut in the producing node's mailbox `tok`; BNSite1Fn.backward
    takes it out and gives both pointers to the site kernel, which reads g + g2 (the same fp32 sum).  The mailbox is a dict
    created by bn_site_res_relu and shared by exactly these two nodes, so nothing can pick up a stale or foreign addend; it is
    only used when x carries that token, i.e. when its producer is known to honour it."""

    @staticmethod
    def forward(ctx, x, tok):
        ctx.tok = tok
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_a, g_b):
        if g_a is None or g_b is None:
            return (g_b if g_a is None else g_a), None
        if "extra" in ctx.tok:
            # a pruned / partial backward on a retained graph left the previous addend behind (its producer never ran): it belongs
            # to that backward, not to this one
            raise RuntimeError("GradFork: the producing site has not consumed the previous shortcut gradient (a partial backward "
                               "through a retained graph?); run whole backwards through the folded bottleneck")
        ctx.tok["extra"] = g_b
        return g_a, None


def fork_block_input(x):
    """(x for the convolution branch, x for the shortcut): the forked pair when x comes from a folded small-batch site."""
    tok = getattr(x, "_alignq_site_tok", None)
    if tok is None or not x.requires_grad or not torch.is_grad_enabled():
        return x, x
    return GradFork.apply(x, tok)


def bn_act_relu(bn, act, z, formula, relu=True, groups=1, residual=None, pack=False):
    """[relu](act(bn(z)) [+ residual]) for a quantiser WITHOUT an ADMM term: one fused chain when `bnq_fusable` (training mode,
    channels-last fp32 CUDA tensor, C = 4 * 2^j <= 2048), else exactly that composition (groups: see BNQuantReluFn).
    pack (N2; folded chain with relu, no residual, ADMM / Office formula, an int16 index that one f16 term holds
    exactly): `out` is a packed handle (BNQuantReluFn.forward) whose values live in `out._alignq_bins = (int16 indices, a_bit)` -
    hand it to a Conv2d_Q only (anything else: fused.materialize)."""
    from . import config
    res_ok = residual is None or (residual.shape == z.shape and residual.is_cuda and residual.dtype == torch.float32)
    if not (bnq_fusable(bn, act, z, groups) and res_ok):
        def one(zz, rr):
            out = act(bn(zz))
            if rr is not None:
                out = out + rr
            return torch.nn.functional.relu(out) if relu else out
        if groups == 1:
            return one(z, residual)
        rs = _slices(residual, groups) if residual is not None else [None] * groups
        return torch.cat([one(zz, rr) for zz, rr in zip(_slices(z, groups), rs)], 0)
    pack = bool(pack and relu and residual is None and formula == L.FORMULA_ADMM and 1 <= act.a_bit <= 14
                and L.load().alignq_bin_bytes(int(act.a_bit), float(config.args.act_range), int(formula)) == 2
                and float(config.args.act_range) * (2 ** int(act.a_bit) - 1) <= 2048.0
                and float(config.args.act_range) == int(config.args.act_range))
    L.MB.bnq_bins = None
    y = BNQuantReluFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                            bn.eps, act.a_bit, config.args.act_range, formula, relu, groups, residual, conv_partials(z, groups), pack)
    if pack:
        y._alignq_bins = L.MB.bnq_bins
        L.MB.bnq_bins = None
    if residual is None:
        tag_levels(y, act.a_bit, config.args.act_range, formula)
    return y


def tag_levels(y, a_bit, act_range, formula):
    """Marks y as a tensor of quantiser outputs idx / n with integer |idx| <= 2048 (ADMM / Office formula: x_q = round(t n) / n,
    |t| <= act_range; model/quantization.py:103-104 Office, :109-110 ADMM tree): Conv2d_Q's GEMM kernels then take the index itself
    as ONE exact f16 operand term (ops.level_count reads the tag; ops.QConvGemmFn).  Only the producing kernels' callers set it."""
    n = float(2 ** int(a_bit) - 1) if 1 <= int(a_bit) <= 16 else 0.0
    if formula == L.FORMULA_ADMM and n > 0 and float(act_range) * n <= 2048.0 and float(act_range) == int(act_range):
        y._alignq_levels = n


# ------------------------------------------------------------------------------------------------------------------
_head_counters = {}


def _head_counter(device, owner=None):
    """The arrival counter of alignq_head_ce_fwd's in-kernel mean: a zeroed word the kernel re-arms.  The last-workgroup
    reduction assumes exclusive ownership of it for the duration of a launch, so two head launches that may be in flight
    together must not share one: the word belongs to `owner` (a step's DeferredLosses: one TrainStep = one in-order chain)
    or, for the stand-alone HeadCEFn, to the (device, stream) it is launched on.  Allocated on the owner's first eager call,
    i.e. outside any graph capture (TrainStep.capture warms up eagerly first)."""
    if owner is not None:
        d = owner.__dict__.setdefault("_head_counters", {})
        key = (device.type, device.index)
    else:
        d = _head_counters
        key = (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream))
    c = d.get(key)
    if c is None:
        # word 0: the head's workgroups; word 1: the sites closed inside alignq_site_reduce_loss_multi_head's launch
        c = d[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return c


_HEAD_ROLE = True        # (tests switch it off to compare with the two launches it replaces)


class StepLossFn(torch.autograd.Function):
    """The two loss roots of a whole-model training step from ONE autograd node: (logits, ce, trans_total) with
    ce = cross_entropy(logit(avgpool(feat)), target) and trans_total = sum of the batched sites' transition losses.
    Forward: alignq_site_reduce_loss_multi_head - the sites' closing slab reductions (as LossSumFn) and the head as two roles of ONE
    launch (round 6), the batch mean and the sum over sites formed by last-arriving workgroups (no reduction launches).
    Backward: alignq_head_ce_bwd_site_prep, the head's backward and every site's S / dalterD / dgamma in one launch.  Two
    launches where HeadCEFn + LossSumFn take seven."""

    @staticmethod
    def forward(ctx, collector, scal_all, feat, weight, bias, target, *losses):
        feat = L.dense_f32(feat, "features")
        if feat.dim() != 4 or feat.is_contiguous():
            raise RuntimeError("StepLossFn needs channels-last 4-D features")
        B, C, H, W = feat.shape
        K = weight.shape[0]
        lib = L.load()
        dev = feat.device
        f32 = dict(dtype=torch.float32, device=dev)
        pooled, logits, probs = torch.empty(B, C, **f32), torch.empty(B, K, **f32), torch.empty(B, K, **f32)
        loss, ce, trans = torch.empty(B, **f32), torch.empty((), **f32), torch.empty((), **f32)
        w = L.dev_f32(weight, "head weight")
        counters = _head_counter(dev, collector)
        open_groups = [(key, recs) for key, recs in collector.open_groups().items()]
        if len(open_groups) == 1 and _HEAD_ROLE:
            # the sites still open and the head in ONE launch (round 6): neither role reads what the other writes
            (sB, dim, mu, rho), recs = open_groups[0]
            L.check(lib.alignq_site_reduce_loss_multi_head(
                len(recs), L.ptr_array([r.ws for r in recs]), L.ptr_array([r.D for r in recs]), L.ptr_array([r.A for r in recs]),
                L.ptr_array([r.Gm for r in recs]), L.ptr_array([r.scal for r in recs]), L.i64_array([r.F for r in recs]), sB, dim, mu,
                rho, L.ptr(feat), L.ptr(w), L.ptr(bias), L.ptr(target), B, H * W, C, K, L.ptr(pooled), L.ptr(logits), L.ptr(probs),
                L.ptr(loss), L.ptr(ce), L.ptr(counters), L.ptr(scal_all), len(losses), L.ptr(trans), L.ptr(counters[1:]),
                L.stream_ptr()), "alignq_site_reduce_loss_multi_head")
        else:
            collector.reduce_all()
            L.check(lib.alignq_head_ce_fwd(L.ptr(feat), L.ptr(w), L.ptr(bias), L.ptr(target), B, H * W, C, K, L.ptr(pooled),
                                           L.ptr(logits), L.ptr(probs), L.ptr(loss), L.ptr(ce), L.ptr(counters),
                                           L.ptr(scal_all), len(losses), L.ptr(trans), L.stream_ptr()), "alignq_head_ce_fwd")
        ctx.save_for_backward(feat, w, target, pooled, probs)
        ctx.has_bias = bias is not None
        ctx.recs = list(collector.records)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return logits, ce, trans

    @staticmethod
    def backward(ctx, _g_logits, g_ce, g_tr):
        feat, w, target, pooled, probs = ctx.saved_tensors
        B, C, H, W = feat.shape
        K = w.shape[0]
        dev = feat.device
        lib = L.load()
        st = L.stream_ptr()
        zero = None
        if g_ce is None or g_tr is None:
            zero = torch.zeros((), dtype=torch.float32, device=dev)
        g_ce = L.dev_f32(zero if g_ce is None else g_ce, "loss grad")
        g_tr = L.dev_f32(zero if g_tr is None else g_tr, "loss grad")
        dfeat, dW = torch.empty_like(feat), torch.empty_like(w)
        db = torch.empty(K, dtype=torch.float32, device=dev) if ctx.has_bias else None
        groups = {}
        for r in ctx.recs:
            groups.setdefault((r.B, r.dim, r.mu, r.rho), []).append(r)
            r.S = torch.empty(lib.alignq_site_bwd_ws_bytes(r.B) // 4, dtype=torch.float32, device=dev)   # fp32 S + bf16 image
            r.dA, r.dG = torch.empty_like(r.A), torch.empty_like(r.Gm)
        head_done = False
        for (sB, dim, mu, rho), recs in groups.items():
            site_args = (len(recs), L.ptr_array([r.D for r in recs]), L.ptr_array([r.A for r in recs]),
                         L.ptr_array([r.Gm for r in recs]), L.ptr_array([r.scal for r in recs]), L.ptr(g_tr),
                         L.i64_array([r.F for r in recs]), sB, dim, mu, L.ptr_array([r.S for r in recs]),
                         L.ptr_array([r.dA for r in recs]), L.ptr_array([r.dG for r in recs]), st)
            if not head_done:
                L.check(lib.alignq_head_ce_bwd_site_prep(L.ptr(g_ce), L.ptr(probs), L.ptr(target), L.ptr(pooled), L.ptr(w), B,
                                                         H * W, C, K, L.ptr(dfeat), L.ptr(dW), L.ptr(db), *site_args),
                        "alignq_head_ce_bwd_site_prep")
                head_done = True
            else:
                L.check(lib.alignq_site_prep_fused_multi(*site_args), "alignq_site_prep_fused_multi")
            for r in recs:
                r.prepared = True
        return (None, None, dfeat, dW, db, None) + (g_tr,) * len(ctx.recs)


class HeadCEFn(torch.autograd.Function):
    """logits = logit(avgpool(feat).flatten(1)); ce = cross_entropy(logits, target) (mean) as one forward and one backward
    launch (alignq_head_ce_fwd / _bwd) for channels-last features.  Returns (logits, ce); only ce is differentiable."""

    @staticmethod
    def forward(ctx, feat, weight, bias, target):
        feat = L.dense_f32(feat, "features")
        if feat.dim() != 4 or feat.is_contiguous():
            raise RuntimeError("HeadCEFn needs channels-last 4-D features")
        B, C, H, W = feat.shape
        K = weight.shape[0]
        lib = L.load()
        dev = feat.device
        pooled = torch.empty(B, C, dtype=torch.float32, device=dev)
        logits = torch.empty(B, K, dtype=torch.float32, device=dev)
        probs = torch.empty(B, K, dtype=torch.float32, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        ce = torch.empty((), dtype=torch.float32, device=dev)
        w = L.dev_f32(weight, "head weight")
        L.check(lib.alignq_head_ce_fwd(L.ptr(feat), L.ptr(w), L.ptr(bias), L.ptr(target), B, H * W, C, K, L.ptr(pooled),
                                       L.ptr(logits), L.ptr(probs), L.ptr(loss), L.ptr(ce), L.ptr(_head_counter(dev)), None, 0,
                                       None, L.stream_ptr()), "alignq_head_ce_fwd")
        ctx.save_for_backward(feat, w, target, pooled, probs)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return logits, ce

    @staticmethod
    def backward(ctx, _g_logits, g_ce):
        feat, w, target, pooled, probs = ctx.saved_tensors
        B, C, H, W = feat.shape
        K = w.shape[0]
        dev = feat.device
        if g_ce is None:
            g_ce = torch.zeros((), dtype=torch.float32, device=dev)
        g_ce = L.dev_f32(g_ce, "loss grad")
        dfeat = torch.empty_like(feat)
        dW = torch.empty_like(w)
        db = torch.empty(K, dtype=torch.float32, device=dev) if ctx.has_bias else None
        L.check(L.load().alignq_head_ce_bwd(L.ptr(g_ce), L.ptr(probs), L.ptr(target), L.ptr(pooled), L.ptr(w), B, H * W, C, K,
                                            L.ptr(dfeat), L.ptr(dW), L.ptr(db), L.stream_ptr()), "alignq_head_ce_bwd")
        return dfeat, dW, db, None


def head_ce_supported(feat, weight, target) -> bool:
    return (feat.is_cuda and feat.dim() == 4 and feat.dtype == torch.float32 and not feat.is_contiguous()
            and feat.is_contiguous(memory_format=torch.channels_last) and feat.shape[1] <= 256 and weight.shape[0] <= 64
            and weight.shape[1] == feat.shape[1] and target.dtype == torch.int64 and target.dim() == 1)

