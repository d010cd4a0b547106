"""torch.autograd.Function wrappers over the C ABI (include/alignq.h).

Each Function allocates its outputs/workspaces with torch (device memory + stream plumbing only) and
enqueues the HIP kernels on torch's current stream, so a whole training step can be captured into a
HIP graph (torch.cuda.CUDAGraph).  No host synchronisation happens here.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L
from . import fused


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------ R1
class UniformQuantizeFn(torch.autograd.Function):
    """uniform_quantize(k) — model/quantization.py:19-38: forward round, backward straight-through."""

    @staticmethod
    def forward(ctx, x, k):
        if k == 32:
            return x
        x = L.dev_f32(x, "input")
        y = torch.empty_like(x)
        if x.numel() == 0:           # (an empty tensor: the reference's elementwise ops return an empty tensor; nothing to launch)
            return y
        L.check(L.load().alignq_uniform_quantize(L.ptr(x), L.ptr(y), x.numel(), int(k), L.stream_ptr()),
                "alignq_uniform_quantize")
        return y

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


# ------------------------------------------------------------------------------------------------ R4 (plain)
class ActQuantFn(torch.autograd.Function):
    """Activation CDF transform + quantise without the correlation branch."""

    @staticmethod
    def forward(ctx, x, k, act_range, formula):
        x = L.dense_f32(x, "activation")
        xq = torch.empty_like(x)
        ctx.empty = x.numel() == 0
        if ctx.empty:                # (as the reference's elementwise chain: empty in, empty out, nothing to launch)
            return xq
        L.check(L.load().alignq_act_quant_fwd(L.ptr(x), L.ptr(xq), None, x.numel(), int(k), float(act_range),
                                              int(formula), L.stream_ptr()), "alignq_act_quant_fwd")
        ctx.save_for_backward(x)
        ctx.act_range = float(act_range)
        return xq

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:
            return torch.empty_like(g), None, None, None
        (x,) = ctx.saved_tensors
        g = L.like_layout(g, x)
        dx = torch.empty_like(x)
        L.check(L.load().alignq_act_quant_bwd(L.ptr(g), L.ptr(x), L.ptr(dx), x.numel(), ctx.act_range,
                                              L.stream_ptr()), "alignq_act_quant_bwd")
        return dx, None, None, None


class ActQuantReluFn(torch.autograd.Function):
    """relu(activation_quantize_fn(x)) — `self.relu(self.act_q1(...))` of the Office bottleneck (dann_office/model/
    resnet.py:137-138, 142-143) — as one launch each way (alignq_act_quant_relu_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, k, act_range, formula):
        x = L.dense_f32(x, "activation")
        y = torch.empty_like(x)
        L.check(L.load().alignq_act_quant_relu_fwd(L.ptr(x), L.ptr(y), x.numel(), int(k), float(act_range), int(formula),
                                                   L.stream_ptr()), "alignq_act_quant_relu_fwd")
        ctx.save_for_backward(x, y)
        ctx.act_range = float(act_range)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = L.like_layout(g, x)
        dx = torch.empty_like(x)
        L.check(L.load().alignq_act_quant_relu_bwd(L.ptr(g), L.ptr(x), L.ptr(y), L.ptr(dx), x.numel(), ctx.act_range,
                                                   L.stream_ptr()), "alignq_act_quant_relu_bwd")
        return dx, None, None, None


# ------------------------------------------------------------------------------------------------ N2: packed bins
_BIN_DTYPES = {(0, 1): torch.int8, (0, 2): torch.int16, (1, 1): torch.uint8, (1, 2): torch.uint16}


def bin_dtype(k, act_range, formula):
    """torch dtype of the stored level index (include/alignq.h, N2): int8 / int16 (ADMM, Office formulas: signed) or
    uint8 / uint16 (CDF-tree formula); None when the arguments have no packed form (k == 32)."""
    nb = L.load().alignq_bin_bytes(int(k), float(act_range), int(formula))
    return _BIN_DTYPES[(int(formula), nb)] if nb else None


def act_quant_pack(x, k, act_range, formula, want_xq=False, relu=False):
    """x -> narrow integer bins (and optionally [relu](x_q) in fp32) in one launch (alignq_act_quant_fwd_packed)."""
    x = L.dense_f32(x, "activation")
    dt = bin_dtype(k, act_range, formula)
    if dt is None:
        raise RuntimeError(f"no packed bin format for k={k}, act_range={act_range}, formula={formula}")
    bins = torch.empty_strided(x.shape, x.stride(), dtype=dt, device=x.device)
    xq = torch.empty_like(x) if want_xq else None
    L.check(L.load().alignq_act_quant_fwd_packed(L.ptr(x), L.ptr(xq), L.ptr(bins), x.numel(), int(k), float(act_range),
                                                 int(formula), int(bool(relu)), L.stream_ptr()), "alignq_act_quant_fwd_packed")
    return (bins, xq) if want_xq else bins


def dequant_bins(bins, k, act_range, formula, relu=False):
    """[relu](value(idx)) in fp32, bit-identical to the fused quantiser's x_q (alignq_bins_dequant)."""
    if not bins.is_cuda:
        raise RuntimeError("alignq_amd: bins must be a CUDA/ROCm tensor (no CPU fallback in the product path)")
    if bins.dtype != bin_dtype(k, act_range, formula):
        raise TypeError(f"bins dtype {bins.dtype} does not match the packed format {bin_dtype(k, act_range, formula)}")
    y = torch.empty_strided(bins.shape, bins.stride(), dtype=torch.float32, device=bins.device)
    L.check(L.load().alignq_bins_dequant(L.ptr(bins), L.ptr(y), bins.numel(), int(k), float(act_range), int(formula),
                                         int(bool(relu)), L.stream_ptr()), "alignq_bins_dequant")
    return y


class ActQuantPackedFn(torch.autograd.Function):
    """[relu](activation_quantize_fn(x)) whose autograd node keeps the 1-2 B level index instead of a 4 B fp32 copy of the
    output for its backward: forward x -> (y fp32, bins), backward reads g, x and the bins (ReLU mask =
    value(idx) > 0).  The second output (bins) is what a consumer that understands the format stores / reads instead of y."""

    @staticmethod
    def forward(ctx, x, k, act_range, formula, relu):
        x = L.dense_f32(x, "activation")
        bins, y = act_quant_pack(x, k, act_range, formula, want_xq=True, relu=relu)
        ctx.save_for_backward(x, bins)
        ctx.cfg = (int(k), float(act_range), int(formula), bool(relu))
        ctx.mark_non_differentiable(bins)
        return y, bins

    @staticmethod
    def backward(ctx, g, _gb):
        x, bins = ctx.saved_tensors
        k, act_range, formula, relu = ctx.cfg
        g = L.like_layout(g, x)
        dx = torch.empty_like(x)
        L.check(L.load().alignq_act_quant_bwd_packed(L.ptr(g), L.ptr(x), L.ptr(bins), L.ptr(dx), x.numel(), k, act_range,
                                                     formula, int(relu), L.stream_ptr()), "alignq_act_quant_bwd_packed")
        return dx, None, None, None, None


def act_quant_bins(x, k, act_range, formula):
    """Parity instrumentation: (x_q, int32 bins) of the activation quantiser."""
    x = L.dev_f32(x, "activation")
    xq = torch.empty_like(x)
    bins = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    L.check(L.load().alignq_act_quant_fwd(L.ptr(x), L.ptr(xq), L.ptr(bins), x.numel(), int(k), float(act_range),
                                          int(formula), L.stream_ptr()), "alignq_act_quant_fwd")
    return xq, bins


# ------------------------------------------------------------------------------------------------ R3
def weight_stats(w):
    w = L.dense_f32(w, "weight")
    lib = L.load()
    ms = torch.empty(2, dtype=torch.float32, device=w.device)
    ws = _ws(lib.alignq_weight_ws_bytes(w.numel()), w.device)
    L.check(lib.alignq_weight_stats(L.ptr(w), w.numel(), L.ptr(ms), L.ptr(ws), L.stream_ptr()), "alignq_weight_stats")
    return ms


def weight_quant_given_stats(w, ms, k, formula, want_aux=True, want_bins=False):
    w = L.dense_f32(w, "weight")
    q = torch.empty_like(w)
    c = torch.empty_like(w) if want_aux else None
    pdf = torch.empty_like(w) if want_aux else None
    bins = torch.empty(w.shape, dtype=torch.int32, device=w.device) if want_bins else None
    L.check(L.load().alignq_weight_quant_fwd(L.ptr(w), L.ptr(ms), L.ptr(q), L.ptr(c), L.ptr(pdf), L.ptr(bins),
                                             w.numel(), int(k), int(formula), L.stream_ptr()),
            "alignq_weight_quant_fwd")
    return q, c, pdf, bins


class CdfFn(torch.autograd.Function):
    """cdf(m, s, quant_src).forward (ADMM tree model/quantization.py:49-59; CDF tree :45-50): (c, pdf) from the weight kernel at the
    given statistics; backward of both outputs w.r.t. the tensor, m and s (alignq_cdf_bwd).  kc = d c / d Phi, scale = what the
    'a' source multiplies the transform by (the kernel writes the 'w' form)."""

    @staticmethod
    def forward(ctx, x, m, s, formula, kc, scale):
        ms = torch.stack([m.detach(), s.detach()])
        _, c, pdf, _ = weight_quant_given_stats(x.detach(), ms, 32, formula, True)
        if scale != 1.0:
            c = c * scale
        ctx.save_for_backward(x, ms)
        ctx.kc = float(kc)
        return c, pdf

    @staticmethod
    def backward(ctx, gc, gp):
        x, ms = ctx.saved_tensors
        lib = L.load()
        need_x = ctx.needs_input_grad[0]
        gc = None if gc is None else L.like_layout(gc, x)
        gp = None if gp is None else L.like_layout(gp, x)
        if gc is None and gp is None:
            return None, None, None, None, None, None
        dx = torch.empty_like(x) if need_x else None
        dms = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = _ws(lib.alignq_weight_ws_bytes(x.numel()), x.device)
        L.check(lib.alignq_cdf_bwd(L.ptr(gc), L.ptr(gp), L.ptr(x), L.ptr(ms), ctx.kc, L.ptr(dx), L.ptr(dms), x.numel(), L.ptr(ws),
                                   L.stream_ptr()), "alignq_cdf_bwd")
        return (dx, dms[0] if ctx.needs_input_grad[1] else None, dms[1] if ctx.needs_input_grad[2] else None, None, None, None)


class WeightQuantFn(torch.autograd.Function):
    """weight_quantize_fn.forward (ADMM tree :71-85; CDF tree :62-78) with the backward through
    mean(W) and std(W).  Returns (W_q, weight_cdf, weight_pdf); only W_q is differentiable."""

    @staticmethod
    def forward(ctx, w, k, formula):
        w = L.dense_f32(w, "weight")
        ms = weight_stats(w)
        q, c, pdf, _ = weight_quant_given_stats(w, ms, k, formula, True)
        ctx.save_for_backward(w, ms)
        ctx.mark_non_differentiable(c, pdf)
        return q, c, pdf

    @staticmethod
    def backward(ctx, g, _gc, _gp):
        w, ms = ctx.saved_tensors
        pending = fused.active_wgrads()
        if pending is not None:      # g may still be partial-sum slabs of a deferred filter gradient
            pending.flush()
        g = L.like_layout(g, w)
        lib = L.load()
        dw = torch.empty_like(w)
        ws = _ws(lib.alignq_weight_ws_bytes(w.numel()), w.device)
        L.check(lib.alignq_weight_quant_bwd(L.ptr(g), L.ptr(w), L.ptr(ms), L.ptr(dw), w.numel(), L.ptr(ws),
                                            L.stream_ptr()), "alignq_weight_quant_bwd")
        return dw, None, None


# ------------------------------------------------------------------------------------------------ R5
def _as_bf(x):
    B = x.shape[0]
    return B, x.numel() // B


def _check_batch(B, what, limit=None):
    """The FUSED site kernels hold the whole batch of a feature tile on chip: 2 <= B <= 128 (include/alignq.h,
    ALIGNQ_MAX_BATCH); corr(x, x) alone runs blocked up to ALIGNQ_MAX_CORR_BATCH = 1024 rows and the module layer composes
    the ADMM site from it above 128 (site_unfused).  The reference accepts any batch; every BASELINE configuration's per-GPU
    batch is 128 or 28.  Raised here with a clear message instead of surfacing as ALIGNQ_EUNSUPPORTED from the C ABI."""
    limit = L.MAX_BATCH if limit is None else limit
    if not (2 <= B <= limit):
        raise RuntimeError(f"alignq_amd: {what} needs a batch of 2..{limit} rows (got {B}); split the batch (data parallel: "
                           "alignq_amd.dp) or use config.args.method != 'ours' for the plain quantiser")


class CorrFn(torch.autograd.Function):
    """corr(x, x) — ADMM tree model/quantization.py:134-137; Office tree :158-161 (eps=1e-5)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = L.dev_f32(x, "corr input")
        B, F = _as_bf(x)
        _check_batch(B, "corr", L.MAX_CORR_BATCH)        # above 128 rows: the blocked Gram of corr_large_kernels.hip
        lib = L.load()
        G = torch.empty(B, B, dtype=torch.float32, device=x.device)
        stats = torch.empty(2, F, dtype=torch.float32, device=x.device)
        ws = _ws(lib.alignq_site_ws_bytes(B, F), x.device)
        L.check(lib.alignq_corr_fwd(L.ptr(x), B, F, float(eps), L.ptr(G), L.ptr(stats), L.ptr(ws), L.stream_ptr()),
                "alignq_corr_fwd")
        ctx.save_for_backward(x, stats)
        ctx.eps = float(eps)
        return G

    @staticmethod
    def backward(ctx, dG):
        x, stats = ctx.saved_tensors
        B, F = _as_bf(x)
        dG = L.dev_f32(dG, "grad")
        dx = torch.empty_like(x)
        lib = L.load()
        ws = _ws(lib.alignq_site_bwd_ws_bytes(B), x.device)
        L.check(lib.alignq_corr_bwd(L.ptr(dG), L.ptr(x), L.ptr(stats), B, F, ctx.eps, L.ptr(dx), L.ptr(ws),
                                    L.stream_ptr()), "alignq_corr_bwd")
        return dx, None


class CorrXYFn(torch.autograd.Function):
    """corr(x, y) with y a different matrix — the general form of model/quantization.py:134-137 (Office :158-161);
    alignq_corr_xy_fwd / _bwd (exact fp32, deterministic)."""

    @staticmethod
    def forward(ctx, x, y, eps):
        x, y = L.dev_f32(x, "corr x"), L.dev_f32(y, "corr y")
        B, F = _as_bf(x)
        _check_batch(B, "corr")
        if _as_bf(y) != (B, F):
            raise RuntimeError(f"corr(x, y): shapes {tuple(x.shape)} and {tuple(y.shape)} do not give the same [B, F]")
        lib = L.load()
        G = torch.empty(B, B, dtype=torch.float32, device=x.device)
        stats = torch.empty(4, F, dtype=torch.float32, device=x.device)
        ws = _ws(lib.alignq_corr_xy_ws_bytes(B, F), x.device)
        L.check(lib.alignq_corr_xy_fwd(L.ptr(x), L.ptr(y), B, F, float(eps), L.ptr(G), L.ptr(stats), L.ptr(ws),
                                       L.stream_ptr()), "alignq_corr_xy_fwd")
        ctx.save_for_backward(x, y, stats)
        ctx.eps = float(eps)
        return G

    @staticmethod
    def backward(ctx, dG):
        x, y, stats = ctx.saved_tensors
        B, F = _as_bf(x)
        dG = L.dev_f32(dG, "grad")
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        L.check(L.load().alignq_corr_xy_bwd(L.ptr(dG), L.ptr(x), L.ptr(y), L.ptr(stats), B, F, ctx.eps, L.ptr(dx), L.ptr(dy),
                                            L.stream_ptr()), "alignq_corr_xy_bwd")
        return dx, dy, None


# ------------------------------------------------------------------------------------------------ R6
class AdmmLossFn(torch.autograd.Function):
    """ADMM.forward — utils/admm.py:24-33; gradients for D, alterD, gamma come from the same launch."""

    @staticmethod
    def forward(ctx, D, alterD, gamma, mu, rho):
        D = L.dev_f32(D, "D")
        A = L.dev_f32(alterD, "alterD")
        Gm = L.dev_f32(gamma, "gamma")
        b, dim = D.shape[0], A.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=D.device)
        dD, dA, dG = torch.empty_like(D), torch.empty_like(A), torch.empty_like(Gm)
        lib = L.load()
        # above 128 rows the loss runs on many workgroups (partial sums in ws) instead of one
        ws = torch.empty(lib.alignq_admm_ws_bytes(dim), dtype=torch.uint8, device=D.device) if dim > 128 else None
        L.check(lib.alignq_admm_loss(L.ptr(D), b, L.ptr(A), L.ptr(Gm), dim, float(mu), float(rho), L.ptr(loss),
                                     L.ptr(dD), L.ptr(dA), L.ptr(dG), L.ptr(ws), L.stream_ptr()), "alignq_admm_loss")
        ctx.save_for_backward(dD, dA, dG)
        return loss

    @staticmethod
    def backward(ctx, g):
        dD, dA, dG = ctx.saved_tensors
        return dD * g, dA * g, dG * g, None, None


# ------------------------------------------------------------------------------------------------ R4+R5+R6 fused
class SiteFn(torch.autograd.Function):
    """One ADMM activation site: x -> (x_q, trans_loss, D).

    activation_quantize_fn.forward, ADMM tree model/quantization.py:102-132 (Office :126-156):
    x_q = quantise(t), D = corr(t,t) - corr(x,x), trans_loss = ADMM(D).  Two launches forward (fused
    quantise + Gram partial slabs; slab reduction with the ADMM-loss epilogue), two launches backward (S/parameter
    gradient prep; fused standardisation-backward + MFMA kernel).  D is returned for ADMM_OPT.step (values only)."""

    @staticmethod
    def forward(ctx, x, alterD, gamma, k, act_range, eps, mu, rho, side=None, bufs=None, rec=None, residual=None,
                relu=False):
        """residual / relu (batches <= 32 only, see site_res_supported): the first output is relu(x_q + residual) — the Office
        bottleneck's `out += identity; out = self.relu(out)` folded into the site forward kernel.
        side: optional torch.cuda.Stream for the slab reduction + loss (it is off the critical path of the
        network's forward: only x_q feeds the next layer); the CALLER must make the consuming stream wait for it.
        bufs: optional dict of persistent per-site buffers (ws, D, scal) — required with `side` so that no
        allocator block is recycled under in-flight side-stream work.
        rec: optional fused.SiteRecord — the slab reduction + loss (and the backward's prep) are then launched for all
        sites at once by fused.DeferredLosses.total(); until then D and the loss hold no value."""
        x = L.dense_f32(x, "activation")
        A = L.dev_f32(alterD, "alterD")
        Gm = L.dev_f32(gamma, "gamma")
        B, F = _as_bf(x)
        _check_batch(B, "an ADMM activation site")
        dim = A.shape[0]
        if B > dim:
            raise RuntimeError(f"batch {B} larger than ADMM dim {dim}")
        lib = L.load()
        dev = x.device
        xq = torch.empty_like(x)
        stats = torch.empty(4, F, dtype=torch.float32, device=dev)
        key = (B, F)
        if bufs is not None and bufs.get("key") == key:
            ws, D, scal = bufs["ws"], bufs["D"], bufs["scal"]
        else:
            D = torch.empty(B, B, dtype=torch.float32, device=dev)
            scal = rec.scal if rec is not None else torch.empty(4, dtype=torch.float32, device=dev)
            ws = _ws(lib.alignq_site_ws_bytes(B, F), dev)
            if bufs is not None:
                bufs.update(key=key, ws=ws, D=D, scal=scal)
        st = L.stream_ptr()
        fold = residual is not None or relu
        if fold:
            if residual is not None:
                residual = L.like_layout(L.dense_f32(residual, "residual"), x)
            L.check(lib.alignq_site_partials_res(L.ptr(x), B, F, int(k), float(act_range), float(eps), L.ptr(residual),
                                                 int(bool(relu)), L.ptr(xq), L.ptr(stats), L.ptr(ws), st),
                    "alignq_site_partials_res")
        else:
            L.check(lib.alignq_site_partials(L.ptr(x), B, F, int(k), float(act_range), float(eps), L.ptr(xq),
                                             L.ptr(stats), L.ptr(ws), st), "alignq_site_partials")
        if side is not None and bufs is not None:
            side.wait_stream(torch.cuda.current_stream())
            st = side.cuda_stream
        if rec is not None:
            rec.ws, rec.D, rec.A, rec.Gm, rec.B, rec.F, rec.dim = ws, D, A, Gm, B, F, dim
            rec.mu, rec.rho = float(mu), float(rho)
        else:
            L.check(lib.alignq_site_reduce_loss(L.ptr(ws), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), dim, float(mu),
                                                float(rho), L.ptr(scal), st), "alignq_site_reduce_loss")
        loss = scal[0]
        ctx.rec = rec
        ctx.save_for_backward(x, stats, D, A, Gm, scal, xq if (fold and relu) else None)
        ctx.set_materialize_grads(False)     # no zero-filled [B,B] gradient for the non-differentiable D
        ctx.cfg = (float(act_range), float(eps), float(mu))
        ctx.fold = (bool(fold), residual is not None)
        ctx.mark_non_differentiable(D)
        return xq, loss, D

    @staticmethod
    def backward(ctx, g_xq, g_loss, _gD):
        x, stats, D, A, Gm, scal, y = ctx.saved_tensors
        act_range, eps, mu = ctx.cfg
        B, F = _as_bf(x)
        dim = A.shape[0]
        g_xq = None if g_xq is None else L.like_layout(g_xq, x)
        if g_loss is None:
            g_loss = torch.zeros((), dtype=torch.float32, device=x.device)
        g_loss = L.dev_f32(g_loss, "loss grad")
        lib = L.load()
        dx = torch.empty_like(x)
        rec = ctx.rec
        fold, has_res = ctx.fold
        dres = None
        if fold and g_xq is not None:
            # the ReLU's backward stays a pass of its own (its result is also the residual's gradient: folding it into the
            # site backward saves no bytes and ran slower); the site backward then sees the masked gradient
            if y is not None:
                g_xq = torch.ops.aten.threshold_backward(g_xq, y, 0.0)
            if has_res and ctx.needs_input_grad[11]:
                dres = g_xq
        if rec is not None and rec.prepared:
            L.check(lib.alignq_site_bwd_apply(L.ptr(g_xq), L.ptr(rec.S), L.ptr(x), L.ptr(stats), B, F, act_range, eps,
                                              L.ptr(dx), L.stream_ptr()), "alignq_site_bwd_apply")
            dA, dG, rec.dA, rec.dG = rec.dA, rec.dG, None, None     # sole owner: AccumulateGrad takes them without a copy
            return dx, dA, dG, None, None, None, None, None, None, None, None, dres, None
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        ws = _ws(lib.alignq_site_bwd_ws_bytes(B), x.device)
        L.check(lib.alignq_site_bwd_fused(L.ptr(g_xq), L.ptr(D), L.ptr(A), L.ptr(Gm), dim, L.ptr(scal), mu,
                                          L.ptr(g_loss), L.ptr(x), L.ptr(stats), B, F, act_range, eps, L.ptr(dx),
                                          L.ptr(dA), L.ptr(dG), L.ptr(ws), L.stream_ptr()), "alignq_site_bwd_fused")
        return dx, dA, dG, None, None, None, None, None, None, None, None, dres, None


class SiteLargeFn(torch.autograd.Function):
    """(x_q, D) of the ADMM site for 128 < B <= ALIGNQ_MAX_CORR_BATCH rows on the blocked Gram (round 4: alignq_site_fwd / _bwd take
    these batches through the PAIR kernels of corr_large_kernels.hip): one sweep leaves x_q and the statistics of x and of the
    pre-round transform t, one Gram launch accumulates D = corr(t, t) - corr(x, x), one backward launch returns
    dx = dD-path + STE-path (model/quantization.py:109-122, corr :134-137; Office :158-161 with eps)."""

    @staticmethod
    def forward(ctx, x, k, act_range, eps):
        x = L.dense_f32(x, "activation")
        B, F = _as_bf(x)
        _check_batch(B, "site", L.MAX_CORR_BATCH)
        lib = L.load()
        xq = torch.empty_like(x)
        D = torch.empty(B, B, dtype=torch.float32, device=x.device)
        stats = torch.empty(4, F, dtype=torch.float32, device=x.device)
        ws = _ws(lib.alignq_site_ws_bytes(B, F), x.device)
        L.check(lib.alignq_site_fwd(L.ptr(x), B, F, int(k), float(act_range), float(eps), L.ptr(xq), L.ptr(D), L.ptr(stats), L.ptr(ws),
                                    L.stream_ptr()), "alignq_site_fwd")
        ctx.save_for_backward(x, stats)
        ctx.cfg = (float(act_range), float(eps))
        ctx.set_materialize_grads(False)
        return xq, D

    @staticmethod
    def backward(ctx, g_xq, g_D):
        x, stats = ctx.saved_tensors
        act_range, eps = ctx.cfg
        B, F = _as_bf(x)
        lib = L.load()
        if g_D is None:
            g_D = torch.zeros(B, B, dtype=torch.float32, device=x.device)
        g_D = L.dev_f32(g_D, "grad of D")
        if g_xq is not None:
            g_xq = L.like_layout(g_xq, x)
        dx = torch.empty_like(x)
        ws = _ws(lib.alignq_site_bwd_ws_bytes(B), x.device)
        L.check(lib.alignq_site_bwd(L.ptr(g_xq), L.ptr(g_D), None, L.ptr(x), L.ptr(stats), B, F, act_range, eps, L.ptr(dx), L.ptr(ws),
                                    L.stream_ptr()), "alignq_site_bwd")
        return dx, None, None, None


class SiteDFn(torch.autograd.Function):
    """D = corr(t, t) - corr(x, x) alone (t = the pre-round transform of x; model/quantization.py:109-122 without the x_q
    output): alignq_site_fwd with xq = NULL, alignq_site_bwd with no upstream x_q gradient, any 2 <= B <= ALIGNQ_MAX_CORR_BATCH.
    The shard kernel of dp.global_site_D (round 4): the exact-global correlation pair from ONE exchange of x."""

    @staticmethod
    def forward(ctx, x, k, act_range, eps):
        x = L.dense_f32(x, "activation")
        B, F = _as_bf(x)
        _check_batch(B, "site", L.MAX_CORR_BATCH)
        lib = L.load()
        D = torch.empty(B, B, dtype=torch.float32, device=x.device)
        stats = torch.empty(4, F, dtype=torch.float32, device=x.device)
        ws = _ws(lib.alignq_site_ws_bytes(B, F), x.device)
        L.check(lib.alignq_site_fwd(L.ptr(x), B, F, int(k), float(act_range), float(eps), None, L.ptr(D), L.ptr(stats), L.ptr(ws),
                                    L.stream_ptr()), "alignq_site_fwd")
        ctx.save_for_backward(x, stats)
        ctx.cfg = (float(act_range), float(eps))
        return D

    @staticmethod
    def backward(ctx, g_D):
        x, stats = ctx.saved_tensors
        act_range, eps = ctx.cfg
        B, F = _as_bf(x)
        lib = L.load()
        g_D = L.dev_f32(g_D, "grad of D")
        dx = torch.empty_like(x)
        ws = _ws(lib.alignq_site_bwd_ws_bytes(B), x.device)
        L.check(lib.alignq_site_bwd(None, L.ptr(g_D), None, L.ptr(x), L.ptr(stats), B, F, act_range, eps, L.ptr(dx), L.ptr(ws),
                                    L.stream_ptr()), "alignq_site_bwd")
        return dx, None, None, None


def site_unfused(x, admm, k, act_range, eps, formula):
    """The ADMM activation site for 128 < B <= ALIGNQ_MAX_CORR_BATCH (the fused site kernels keep all rows of a feature tile on
    chip and stop at 128): (x_q, D) from SiteLargeFn - the pair kernels on the blocked Gram - and loss = ADMM(D); the CDF-only
    formula (no BASELINE site uses it) keeps round 3's composition from the stand-alone kernels.  Same values as the reference's
    lines (model/quantization.py:109-123).  Returns (x_q, loss, D)."""
    if formula == L.FORMULA_ADMM and k < 32 and x.dim() >= 2:
        xq, D = SiteLargeFn.apply(x, k, act_range, float(eps))
        loss = AdmmLossFn.apply(D, admm.alterD, admm.gamma, admm.mu, admm.rho)
        return xq, loss, D
    xq = ActQuantFn.apply(x, k, act_range, formula)
    t = ActQuantFn.apply(x, 32, act_range, formula)
    D = CorrFn.apply(t, float(eps)) - CorrFn.apply(x, float(eps))
    loss = AdmmLossFn.apply(D, admm.alterD, admm.gamma, admm.mu, admm.rho)
    return xq, loss, D


def site_res_supported(x, residual) -> bool:
    """SiteFn(residual=, relu=): the small-batch site kernels (2 <= B <= 32) with a residual of x's shape and layout."""
    if not (x.is_cuda and x.dtype == torch.float32 and 2 <= x.shape[0] <= 32):
        return False
    return residual is None or (residual.shape == x.shape and residual.dtype == torch.float32 and residual.is_cuda
                                and residual.stride() == x.stride())


# ------------------------------------------------------------------------------------------------ R9 (Conv2d_Q's conv)
def qconv3x3_supported(x, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """Shapes alignq_conv3x3_nhwc implements: channels-last fp32, 3x3 / stride 1 / padding 1, C_in == C_out in {16,32,64}
    at width 32 / 16 / 8 (the ResNet-20/56 body), a <= 8-bit quantised filter.  Everything else stays on MIOpen."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8):
        return False
    if tuple(stride) != (1, 1) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and w.dtype == torch.float32):
        return False
    B, C, H, W = x.shape
    if tuple(w.shape) != (C, C, 3, 3) or (C, W) not in ((16, 32), (32, 16), (64, 8)):
        return False
    if H % 8:            # tiles are 8 / 8 / 4-8 whole image rows
        return False
    cl = torch.channels_last
    return (x.is_contiguous(memory_format=cl) and not x.is_contiguous()) and w.is_contiguous(memory_format=cl)


class QConv3x3Fn(torch.autograd.Function):
    """F.conv2d(input, weight_q, None, 1, 1) of Conv2d_Q.forward (model/quantization.py:149-154) on the bf16 matrix cores
    with exact products (integer filter bins x three-way split activations, see csrc/conv_kernels.hip).  The data gradient
    is the same kernel on the flipped / transposed filter; the filter gradient runs on the f32 MFMAs (plain fp32,
    deterministic partial-sum slabs).

    tap=True additionally returns the input itself as a second output (an alias for the block's identity shortcut,
    `shortcut = x`, resnet.py:84-86): its gradient then arrives HERE and is added in the data-gradient kernel's epilogue
    instead of by a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, w, w_bit, tap=False, bn_stats=False, xbins=None, a_bit=0):
        """bn_stats=True: the kernel's epilogue also leaves per-workgroup per-channel {sum y, sum y^2}; they are attached to
        the output as `y._alignq_bn_part = (float tensor [C, parts, 2], parts)` for fused.bn_site, which then skips its own
        statistics pass over y.
        xbins (N2): x is only a HANDLE (fused.packed_handle: shape and autograd edge, no data); the activation is read from its
        int8 / int16 level indices `xbins` (a_bit-bit ADMM-formula quantiser, ReLU already applied), forward and filter gradient."""
        B, C, H, W = x.shape
        lib = L.load()
        dev = w.device
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        part, n_parts = None, 0
        if bn_stats:
            n_parts = lib.alignq_conv3x3_bn_parts(B, H, W, C)
            part = torch.empty(C, n_parts, 2, dtype=torch.float32, device=dev) if n_parts > 0 else None
        xb = xbins.element_size() if xbins is not None else 0
        L.check(lib.alignq_conv3x3_nhwc(None if xbins is not None else L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, C, int(w_bit), 0,
                                        None, L.ptr(part), L.ptr(xbins), xb, int(a_bit), L.stream_ptr()), "alignq_conv3x3_nhwc")
        if xbins is not None:
            ctx.save_for_backward(xbins, w)
        else:
            ctx.save_for_backward(x, w)
        ctx.packed = (xbins is not None, int(a_bit), (B, C, H, W))
        ctx.w_bit = int(w_bit)
        ctx.tap = bool(tap)
        ctx.link = None
        if part is not None:
            # third field: this producer's backward accepts a lazy BN gradient (1), and reduces the site backward's per-tile
            # sums itself when both of its gradients are needed (2: the fused alignq_conv3x3_nhwc_bwd); fourth: the link
            # through which the BN site's backward hands that gradient's record to THIS node's backward
            if x.requires_grad or w.requires_grad:
                ctx.link = fused.LazyLink()
                L.MB.conv3x3 = (part, n_parts, 2 if (x.requires_grad and w.requires_grad) else 1, ctx.link)
            else:       # this node's backward never runs: the BN site must finish its own input gradient
                L.MB.conv3x3 = (part, n_parts)
        if tap:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def apply_with_stats(x, w, w_bit, tap=False, xbins=None, a_bit=0):
        """apply(...) with bn_stats=True; attaches the partial statistics to the returned y (a plain python attribute)."""
        # the partials are created inside forward; fetch them through a one-slot mailbox (autograd hides ctx from callers)
        L.MB.conv3x3 = None
        out = QConv3x3Fn.apply(x, w, w_bit, tap, True, xbins, a_bit)
        y = out[0] if tap else out
        if L.MB.conv3x3 is not None:
            y._alignq_bn_part = L.MB.conv3x3
            L.MB.conv3x3 = None
        return out


    @staticmethod
    def backward(ctx, gy, gtap=None):
        x, w = ctx.saved_tensors
        packed, a_bit, (B, C, H, W) = ctx.packed
        none7 = (None,) * 5
        if gy is None:                     # only the shortcut alias was used downstream
            return (gtap, None) + none7
        lazy = fused.take_lazy_dz(ctx.link, gy)      # (g, z, ab, save, ktot): gy is the gradient w.r.t. the folded BN's OUTPUT
        dev = w.device
        cl = torch.channels_last

        def new_x():                       # an fp32 tensor of x's shape in channels-last memory (x itself may be the index tensor)
            return torch.empty((B, C, H, W), dtype=torch.float32, device=dev, memory_format=cl)
        if not (gy.dim() == 4 and gy.is_contiguous(memory_format=cl)):
            gy = gy.contiguous(memory_format=cl)
        add = None
        if gtap is not None:
            add = gtap if gtap.is_contiguous(memory_format=cl) else gtap.contiguous(memory_format=cl)
        # N2: the x operand of the filter gradient from its level indices
        xp, xbp, xbb = (None, L.ptr(x), x.element_size()) if packed else (L.ptr(x), None, 0)
        dx = dw = None
        lib = L.load()
        pending = fused.active_wgrads()
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and pending is not None:
            # whole-model step: data gradient + filter-gradient partial sums in ONE launch; the slab reduction of all
            # convolutions follows in one launch at the end of the backward (fused.DeferredWgrads.flush)
            dx, dw = new_x(), torch.empty_like(w)
            ws = _ws(lib.alignq_conv3x3_wgrad_ws_bytes(C), dev)
            ns = ctypes.c_int(0)
            bz, bab, bsave, bk, bpart, bdg, bdb = lazy[1:8] if lazy is not None else (None,) * 7
            # filler role: this launch also finishes slab reductions of convolutions whose backward already ran
            fill = pending.take(C)
            nf = len(fill)
            L.check(lib.alignq_conv3x3_nhwc_bwd_fill(
                xp, L.ptr(gy), L.ptr(w), L.ptr(dx), L.ptr(ws), B, H, W, C, ctx.w_bit, ctypes.byref(ns), L.ptr(add), L.ptr(bz),
                L.ptr(bab), L.ptr(bsave), L.ptr(bk), L.ptr(bpart) if bk is None else None, L.ptr(bdg), L.ptr(bdb), xbp, xbb, a_bit,
                nf, L.ptr_array([f[0] for f in fill]) if nf else None, L.ptr_array([f[1] for f in fill]) if nf else None,
                (ctypes.c_int * nf)(*[f[2] for f in fill]) if nf else None,
                (ctypes.c_int * nf)(*[f[3] for f in fill]) if nf else None, L.stream_ptr()), "alignq_conv3x3_nhwc_bwd_fill")
            pending.add(ws, dw, ns.value, 9 * C * C)
            return (dx, dw) + none7
        if lazy is not None:               # not the fused path after all: finish the batch-norm input gradient here
            bz, bab, bsave, bk = _lazy_fields(lazy, True, B, C, H * W, dev)[:4]
            shp = (1, C, 1, 1)
            gy = bab[0].view(shp) * (gy - bk[0].view(shp) - (bz - bsave[0].view(shp)) * bsave[1].view(shp) * bk[1].view(shp))
            gy = gy.contiguous(memory_format=cl)
        if ctx.needs_input_grad[0]:
            dx = new_x()
            L.check(lib.alignq_conv3x3_nhwc(L.ptr(gy), L.ptr(w), L.ptr(dx), B, H, W, C, ctx.w_bit, 1, L.ptr(add), None, None, 0, 0,
                                            L.stream_ptr()), "alignq_conv3x3_nhwc")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)          # channels-last [C,3,3,C] storage like w
            ws = _ws(lib.alignq_conv3x3_wgrad_ws_bytes(C), dev)
            if pending is not None:           # whole-model step: all filter-gradient reductions in one launch at the end
                ns = ctypes.c_int(0)
                L.check(lib.alignq_conv3x3_nhwc_wgrad(xp, L.ptr(gy), None, L.ptr(ws), B, H, W, C, ctypes.byref(ns), xbp, xbb, a_bit,
                                                      L.stream_ptr()), "alignq_conv3x3_nhwc_wgrad")
                pending.add(ws, dw, ns.value, 9 * C * C)
            else:
                L.check(lib.alignq_conv3x3_nhwc_wgrad(xp, L.ptr(gy), L.ptr(dw), L.ptr(ws), B, H, W, C, None, xbp, xbb, a_bit,
                                                      L.stream_ptr()), "alignq_conv3x3_nhwc_wgrad")
        return (dx, dw) + none7


def _lazy_fields(lazy, need_totals_now, B, C, HW, device):
    """(z, ab, save, ktot, part, dgamma, dbeta) of a lazy batch-norm record (fused.post_lazy_dz); when the record carries the
    site backward's per-tile sums instead of finished totals and the caller's kernels cannot publish the parameter gradients
    (need_totals_now), the totals are formed here (alignq_bn_bwd_totals)."""
    if lazy is None:
        return (None,) * 7
    bz, bab, bsave, bk, bpart, bdg, bdb = lazy[1:8]
    if bk is None and need_totals_now:
        bk = torch.empty(2, C, dtype=torch.float32, device=device)
        L.check(L.load().alignq_bn_bwd_totals(L.ptr(bpart), B, C, HW, L.ptr(bk), L.ptr(bdg), L.ptr(bdb), L.stream_ptr()),
                "alignq_bn_bwd_totals")
    return bz, bab, bsave, bk, (bpart if bk is None else None), bdg, bdb


def qconv_gen_supported(x, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """The transition convolutions alignq_conv_gen_nhwc_fwd implements: stride 2, 3x3 (padding 1) or 1x1 (padding 0),
    (C_in, C_out, W_in) in {(16, 32, 32), (32, 64, 16)}, channels-last fp32, <= 8-bit quantised filter."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8) or tuple(stride) != (2, 2) or tuple(dilation) != (1, 1):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and w.dtype == torch.float32):
        return False
    B, CIN, H, W = x.shape
    COUT, ks = w.shape[0], w.shape[2]
    if tuple(w.shape) != (COUT, CIN, ks, ks) or (ks, tuple(padding)) not in ((3, (1, 1)), (1, (0, 0))):
        return False
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and not x.is_contiguous()):
        return False
    if not (w.is_contiguous(memory_format=cl) or (ks == 1 and w.is_contiguous())):
        return False
    return L.load().alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, ks, 2) > 0


class QConvGenFn(torch.autograd.Function):
    """Forward of Conv2d_Q's stride-2 transition convolutions (3x3 and the 1x1 shortcut) on alignq_conv_gen_nhwc_fwd, with the
    batch-norm partial statistics of the output as a by-product; data gradient (alignq_conv_gen_nhwc_dgrad) and filter gradient
    (alignq_conv_gen_nhwc_wgrad) accept the lazy batch-norm form of the incoming gradient.

    tap=True additionally returns the input as a second output (an alias): the transition block feeds it to its OTHER
    convolution (resnet.py:78-86: `skip_conv(x)` and `conv0(x)` read the same x), whose input gradient then arrives here and
    is added in this data-gradient kernel's epilogue instead of by a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, w, w_bit, padding, tap=False):
        B, CIN, H, W = x.shape
        COUT, ks = w.shape[0], w.shape[2]
        lib = L.load()
        y = torch.empty((B, COUT, H // 2, W // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        n_parts = lib.alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, ks, 2)
        part = torch.empty(COUT, n_parts, 2, dtype=torch.float32, device=x.device)
        L.check(lib.alignq_conv_gen_nhwc_fwd(L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, CIN, COUT, ks, 2, int(w_bit), L.ptr(part),
                                             L.stream_ptr()), "alignq_conv_gen_nhwc_fwd")
        ctx.save_for_backward(x, w)
        ctx.w_bit = int(w_bit)
        # 2: the data-gradient kernel reduces the site backward's per-tile sums and publishes the BN parameter gradients
        ctx.link = None
        if x.requires_grad or w.requires_grad:
            ctx.link = fused.LazyLink()
            L.MB.conv3x3 = (part, n_parts, 2 if x.requires_grad else 1, ctx.link)
        else:
            L.MB.conv3x3 = (part, n_parts)
        if tap:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, gtap=None):
        x, w = ctx.saved_tensors
        if gy is None:                     # only the alias was used downstream
            return gtap, None, None, None, None
        add = None if gtap is None else L.like_layout(gtap, x)
        lazy = fused.take_lazy_dz(ctx.link, gy)      # (g, z, ab, save, ktot): gy is the gradient w.r.t. the folded BN's OUTPUT
        gy = gy.contiguous(memory_format=torch.channels_last)
        B, CIN, H, W = x.shape
        COUT, ks = w.shape[0], w.shape[2]
        bz, bab, bsave, bk, bpart, bdg, bdb = _lazy_fields(lazy, not ctx.needs_input_grad[0], B, COUT, (H // 2) * (W // 2),
                                                           x.device)
        lib = L.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(lib.alignq_conv_gen_nhwc_dgrad(L.ptr(gy), L.ptr(w), L.ptr(dx), B, H, W, CIN, COUT, ks, 2, ctx.w_bit,
                                                   L.ptr(add), L.ptr(bz), L.ptr(bab), L.ptr(bsave), L.ptr(bk), L.ptr(bpart),
                                                   L.ptr(bdg), L.ptr(bdb), L.stream_ptr()), "alignq_conv_gen_nhwc_dgrad")
        elif add is not None:
            dx = add
        if ctx.needs_input_grad[1]:        # split-bf16 MFMA, deterministic slabs
            dw = torch.empty_like(w)
            ws = _ws(lib.alignq_conv_gen_wgrad_ws_bytes(CIN, COUT, ks), x.device)
            pending = fused.active_wgrads()
            if pending is not None:
                ns = ctypes.c_int(0)
                L.check(lib.alignq_conv_gen_nhwc_wgrad(L.ptr(x), L.ptr(gy), None, L.ptr(ws), B, H, W, CIN, COUT, ks, 2,
                                                       ctypes.byref(ns), L.ptr(bz), L.ptr(bab), L.ptr(bsave), L.ptr(bk),
                                                       L.ptr(bpart), L.stream_ptr()), "alignq_conv_gen_nhwc_wgrad")
                pending.add(ws, dw, ns.value, ks * ks * CIN * COUT)
            else:
                L.check(lib.alignq_conv_gen_nhwc_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(ws), B, H, W, CIN, COUT, ks, 2,
                                                       None, L.ptr(bz), L.ptr(bab), L.ptr(bsave), L.ptr(bk), L.ptr(bpart),
                                                       L.stream_ptr()), "alignq_conv_gen_nhwc_wgrad")
        return dx, dw, None, None, None

    @staticmethod
    def apply_with_stats(x, w, w_bit, padding, tap=False):
        L.MB.conv3x3 = None
        out = QConvGenFn.apply(x, w, w_bit, padding, tap)
        y = out[0] if tap else out
        if L.MB.conv3x3 is not None:
            y._alignq_bn_part = L.MB.conv3x3
            L.MB.conv3x3 = None
        return out


class QTransitionFn(torch.autograd.Function):
    """(conv0(x), skip_conv(x)) of a transition block — the 3x3 and the 1x1 stride-2 Conv2d_Q reading the same input
    (model/resnet.py PreActBlock_conv_Q.forward) — as ONE launch each way (alignq_transition_nhwc_fwd / _bwd) instead of two
    forward and four backward launches.  Both outputs carry their batch-norm partial statistics and lazy-gradient links like
    QConvGenFn's; values equal the separate launches'."""


    @staticmethod
    def forward(ctx, x, w3, w1, w_bit):
        B, CIN, H, W = x.shape
        COUT = w3.shape[0]
        lib = L.load()
        cl = torch.channels_last
        y3 = torch.empty((B, COUT, H // 2, W // 2), dtype=torch.float32, device=x.device, memory_format=cl)
        y1 = torch.empty((B, COUT, H // 2, W // 2), dtype=torch.float32, device=x.device, memory_format=cl)
        n3 = lib.alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, 3, 2)
        n1 = lib.alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, 1, 2)
        part3 = torch.empty(COUT, n3, 2, dtype=torch.float32, device=x.device)
        part1 = torch.empty(COUT, n1, 2, dtype=torch.float32, device=x.device)
        L.check(lib.alignq_transition_nhwc_fwd(L.ptr(x), L.ptr(w3), L.ptr(w1), L.ptr(y3), L.ptr(y1), B, H, W, CIN, COUT,
                                               int(w_bit), L.ptr(part3), L.ptr(part1), L.stream_ptr()),
                "alignq_transition_nhwc_fwd")
        ctx.save_for_backward(x, w3, w1)
        ctx.w_bit = int(w_bit)
        ctx.link3, ctx.link1 = fused.LazyLink(), fused.LazyLink()
        # mode 2: the data-gradient role reduces the site backward's per-tile sums and publishes the BN parameter gradients
        L.MB.transition = ((part3, n3, 2, ctx.link3), (part1, n1, 2, ctx.link1))
        return y3, y1

    @staticmethod
    def backward(ctx, gy3, gy1):
        x, w3, w1 = ctx.saved_tensors
        lazy3 = fused.take_lazy_dz(ctx.link3, gy3)
        lazy1 = fused.take_lazy_dz(ctx.link1, gy1)
        cl = torch.channels_last
        gy3, gy1 = gy3.contiguous(memory_format=cl), gy1.contiguous(memory_format=cl)
        B, CIN, H, W = x.shape
        COUT = w3.shape[0]
        HWo = (H // 2) * (W // 2)
        f3 = _lazy_fields(lazy3, False, B, COUT, HWo, x.device)
        f1 = _lazy_fields(lazy1, False, B, COUT, HWo, x.device)
        lib = L.load()
        dx = torch.empty_like(x)
        dw3, dw1 = torch.empty_like(w3), torch.empty_like(w1)
        ws3 = _ws(lib.alignq_conv_gen_wgrad_ws_bytes(CIN, COUT, 3), x.device)
        ws1 = _ws(lib.alignq_conv_gen_wgrad_ws_bytes(CIN, COUT, 1), x.device)
        ns3, ns1 = ctypes.c_int(0), ctypes.c_int(0)
        L.check(lib.alignq_transition_nhwc_bwd(
            L.ptr(x), L.ptr(gy3), L.ptr(gy1), L.ptr(w3), L.ptr(w1), L.ptr(dx), L.ptr(ws3), L.ptr(ws1), B, H, W, CIN, COUT,
            ctx.w_bit, ctypes.byref(ns3), ctypes.byref(ns1), None, *[L.ptr(t) for t in f3], *[L.ptr(t) for t in f1],
            L.stream_ptr()), "alignq_transition_nhwc_bwd")
        pending = fused.active_wgrads()
        if pending is not None:
            pending.add(ws3, dw3, ns3.value, 9 * CIN * COUT)
            pending.add(ws1, dw1, ns1.value, CIN * COUT)
        else:
            L.check(lib.alignq_conv3x3_wgrad_reduce_multi(
                2, L.ptr_array([ws3, ws1]), L.ptr_array([dw3, dw1]), (ctypes.c_int * 2)(ns3.value, ns1.value),
                (ctypes.c_int * 2)(9 * CIN * COUT, CIN * COUT), L.stream_ptr()), "alignq_conv3x3_wgrad_reduce_multi")
        return dx, dw3, dw1, None

    @staticmethod
    def apply_with_stats(x, w3, w1, w_bit):
        L.MB.transition = None
        y3, y1 = QTransitionFn.apply(x, w3, w1, w_bit)
        if L.MB.transition is not None:
            y3._alignq_bn_part, y1._alignq_bn_part = L.MB.transition
            L.MB.transition = None
        return y3, y1


def transition_supported(conv3, conv1, x, w3, w1) -> bool:
    """Both convolutions of a transition block on this repository's kernels, gradients needed for x and both filters."""
    if not (getattr(conv3, "use_qconv", False) and getattr(conv1, "use_qconv", False)):
        return False
    if conv3.quantize_fn.w_bit != conv1.quantize_fn.w_bit or not (x.requires_grad and w3.requires_grad and w1.requires_grad):
        return False
    a3 = (x, w3, conv3.stride, conv3.padding, conv3.dilation, conv3.groups, conv3.bias, conv3.quantize_fn.w_bit)
    a1 = (x, w1, conv1.stride, conv1.padding, conv1.dilation, conv1.groups, conv1.bias, conv1.quantize_fn.w_bit)
    return (tuple(w3.shape[2:]) == (3, 3) and tuple(w1.shape[2:]) == (1, 1) and qconv_gen_supported(*a3)
            and qconv_gen_supported(*a1))


def qconv_gemm_supported(x, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """The Conv2d_Q convolutions alignq_qconv_* take (the ResNet-50 / Office-31 shapes of BASELINE config 5): channels-last fp32,
    C_in and C_out multiples of 64, 1x1 (padding 0) or 3x3 (padding 1), stride 1 or 2, <= 8-bit quantised filter."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8) or tuple(dilation) != (1, 1):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_cuda):
        return False
    B, CIN, H, W = x.shape
    COUT, ks = w.shape[0], w.shape[2]
    if w.shape[1] != CIN or w.shape[3] != ks or ks not in (1, 3) or tuple(padding) != ((ks - 1) // 2,) * 2:
        return False
    if stride[0] != stride[1] or stride[0] not in (1, 2):
        return False
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and (ks == 1 or w.is_contiguous(memory_format=cl))):
        return False
    return bool(L.load().alignq_qconv_supported(B, H, W, CIN, COUT, ks, int(stride[0])))


def level_count(t):
    """n_a if `t` is known to hold quantiser outputs idx / n_a with integer |idx| <= 2048 (the tag the folded quantiser chains
    put on their outputs, fused.tag_levels), else 0.0."""
    return float(getattr(t, "_alignq_levels", 0.0) or 0.0)


def pack_filter_bins(weights, w_bit):
    """[(bf16 bins, f16 bins)] of quantised filters (int16 tensors of the filters' shapes and layouts holding the bit patterns):
    alignq_qconv_pack_weights, one launch per 64 filters.  The operands of alignq_qconv_fwd / _dgrad."""
    ws_ = [L.dense_f32(w, "quantised filter") for w in weights]
    bf = [torch.empty_like(w, dtype=torch.int16) for w in ws_]
    hf = [torch.empty_like(w, dtype=torch.int16) for w in ws_]
    L.check(L.load().alignq_qconv_pack_weights(len(ws_), L.ptr_array(ws_), L.i64_array([w.numel() for w in ws_]), int(w_bit),
                                               L.ptr_array(bf), L.ptr_array(hf), L.stream_ptr()), "alignq_qconv_pack_weights")
    return list(zip(bf, hf))


def qconv_gemm_shape_supported(shape, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """qconv_gemm_supported for an input given by its SHAPE only (a packed handle, fused.packed_handle: the values live in int16
    level indices laid out channels-last)."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8) or tuple(dilation) != (1, 1) or len(shape) != 4:
        return False
    if not (w.dtype == torch.float32 and w.is_cuda):
        return False
    B, CIN, H, W = shape
    COUT, ks = w.shape[0], w.shape[2]
    if w.shape[1] != CIN or w.shape[3] != ks or ks not in (1, 3) or tuple(padding) != ((ks - 1) // 2,) * 2:
        return False
    if stride[0] != stride[1] or stride[0] not in (1, 2):
        return False
    if not (ks == 1 or w.is_contiguous(memory_format=torch.channels_last)):
        return False
    return bool(L.load().alignq_qconv_supported(B, H, W, CIN, COUT, ks, int(stride[0])))


class QConvGemmFn(torch.autograd.Function):
    """F.conv2d(input, weight_q, None, stride, padding) of Conv2d_Q.forward (cdf_alignment_admm/dann_office/model/quantization.py:
    164-181) at the ResNet-50 shapes on alignq_qconv_fwd / _dgrad / _wgrad (csrc/qgemm_kernels.hip): exact products on the bf16 /
    f16 matrix cores.  x_levels: see level_count (0.0: a general fp32 input).  bins: (bf16, f16) bit patterns of the filter's
    integer bins from pack_filter_bins (None: packed here, one small launch).  xbins (N2): x is only a HANDLE (fused.packed_handle:
    shape and autograd edge); the activation is read from its int16 level indices `xbins`, forward and filter gradient.  The filter
    gradient's slab reduction is deferred to fused.DeferredWgrads when such a context is active."""

    @staticmethod
    def forward(ctx, x, w, w_bit, stride, x_levels=0.0, groups=1, bn_stats=False, bins=None, xbins=None):
        B, CIN, H, W = x.shape
        COUT, ks = w.shape[0], w.shape[2]
        s = int(stride)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        lib = L.load()
        if bins is None:
            bins = pack_filter_bins([w], w_bit)[0]
        if xbins is not None and not (x_levels and xbins.dtype == torch.int16 and tuple(xbins.shape) == tuple(x.shape)
                                      and xbins.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError("QConvGemmFn: xbins must be the channels-last int16 level indices of a level tensor of x's shape")
        y = torch.empty((B, COUT, Ho, Wo), dtype=torch.float32, device=w.device, memory_format=torch.channels_last)
        part = None
        if bn_stats:
            n_parts = lib.alignq_qconv_bn_parts(B, H, W, CIN, COUT, ks, s, int(groups), float(x_levels))
            part = torch.empty(int(groups), n_parts, COUT, 2, dtype=torch.float64, device=w.device)
            L.MB.gemm = (part, n_parts)
        L.check(lib.alignq_qconv_fwd(L.ptr(xbins if xbins is not None else x), L.ptr(bins[1] if x_levels else bins[0]), L.ptr(y), B, H, W,
                                     CIN, COUT, ks, s, int(w_bit), float(x_levels), 2 if xbins is not None else 0,
                                     int(groups if bn_stats else 1), L.ptr(part), L.stream_ptr()), "alignq_qconv_fwd")
        ctx.save_for_backward(xbins if xbins is not None else x, w, bins[0])
        ctx.cfg = (int(w_bit), s, float(x_levels), ks, xbins is not None, (B, CIN, H, W))
        return y


    @staticmethod
    def apply_with_stats(x, w, w_bit, stride, x_levels=0.0, groups=1, bins=None, xbins=None):
        """apply(...) that also leaves the batch-norm partial statistics of the output on it: y._alignq_bnq_part =
        (double tensor [groups, parts, C_out, 2], parts, groups) for fused.bn_act_relu / bn_only / bn_site_res_relu."""
        L.MB.gemm = None
        y = QConvGemmFn.apply(x, w, w_bit, stride, x_levels, groups, True, bins, xbins)
        if L.MB.gemm is not None:
            y._alignq_bnq_part = L.MB.gemm + (int(groups),)
            L.MB.gemm = None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, wb = ctx.saved_tensors
        w_bit, s, x_levels, ks, packed, (B, CIN, H, W) = ctx.cfg
        COUT = w.shape[0]
        lib = L.load()
        cl = torch.channels_last
        if not gy.is_contiguous(memory_format=cl):
            gy = gy.contiguous(memory_format=cl)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ks == 3 and s != 1 and (H % 2 or W % 2):      # (an odd grid has no parity classes: torch's data gradient for that layer)
                dx = torch.nn.grad.conv2d_input((B, CIN, H, W), w, gy, stride=s, padding=1)
            else:
                dx = torch.empty((B, CIN, H, W), dtype=torch.float32, device=w.device, memory_format=cl)
                nws = lib.alignq_qconv_dgrad_ws_bytes(B, H, W, CIN, COUT, ks, s)        # split-K scratch of the small-M layers
                wsd = _ws(nws, w.device) if nws else None
                L.check(lib.alignq_qconv_dgrad(L.ptr(gy), L.ptr(wb), L.ptr(dx), B, H, W, CIN, COUT, ks, s, w_bit, L.ptr(wsd),
                                               L.stream_ptr()), "alignq_qconv_dgrad")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            ws = _ws(lib.alignq_qconv_wgrad_ws_bytes(B, H, W, CIN, COUT, ks, s), w.device)
            xb = 2 if packed else 0
            pending = fused.active_wgrads()
            if pending is not None:
                ns = ctypes.c_int(0)
                L.check(lib.alignq_qconv_wgrad(L.ptr(x), L.ptr(gy), None, L.ptr(ws), B, H, W, CIN, COUT, ks, s, x_levels, xb,
                                               ctypes.byref(ns), L.stream_ptr()), "alignq_qconv_wgrad")
                pending.add(ws, dw, ns.value, COUT * ks * ks * CIN)
            else:
                L.check(lib.alignq_qconv_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(ws), B, H, W, CIN, COUT, ks, s, x_levels, xb, None,
                                               L.stream_ptr()), "alignq_qconv_wgrad")
        return dx, dw, None, None, None, None, None, None, None


def qconv_stem7_supported(x, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """The Office ResNet-50's stem (dann_office/model/resnet.py:193-195: Conv2d_Q(3, 64, kernel_size=7, stride=2, padding=3)) as
    alignq_qconv_stem7_fwd takes it: channels-last fp32 image that needs no gradient, <= 8-bit quantised filter."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8) or x.requires_grad:
        return False
    if tuple(stride) != (2, 2) or tuple(padding) != (3, 3) or tuple(dilation) != (1, 1):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_cuda):
        return False
    B, C, H, W = x.shape
    cl = torch.channels_last
    return (C == 3 and tuple(w.shape) == (64, 3, 7, 7) and H >= 8 and W >= 8 and x.is_contiguous(memory_format=cl)
            and w.is_contiguous(memory_format=cl) and bool(L.load().alignq_qconv_stem7_bn_parts(B, H, W, 1)))


class QConvStem7Fn(torch.autograd.Function):
    """F.conv2d(image, weight_q, None, 2, 3) of the Office stem on alignq_qconv_stem7_fwd (csrc/qgemm_kernels.hip: the filter's integer
    bins times three exact bf16 terms of the image, gathered straight from global memory); batch-norm statistics of the output in
    the epilogue (bn_stats).  The image needs no gradient; the filter gradient is alignq_qconv_stem7_wgrad's (six leading term pairs,
    deterministic slabs; their reduction is deferred to fused.DeferredWgrads when such a context is active)."""

    @staticmethod
    def forward(ctx, x, w, w_bit, groups=1, bn_stats=False, bins=None):
        B, _, H, W = x.shape
        lib = L.load()
        if bins is None:
            bins = pack_filter_bins([w], w_bit)[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, 64, Ho, Wo), dtype=torch.float32, device=w.device, memory_format=torch.channels_last)
        part = None
        if bn_stats:
            n_parts = lib.alignq_qconv_stem7_bn_parts(B, H, W, int(groups))
            part = torch.empty(int(groups), n_parts, 64, 2, dtype=torch.float64, device=w.device)
            L.MB.gemm = (part, n_parts)
        L.check(lib.alignq_qconv_stem7_fwd(L.ptr(x), L.ptr(bins[0]), L.ptr(y), B, H, W, int(w_bit), int(groups if bn_stats else 1),
                                           L.ptr(part), L.stream_ptr()), "alignq_qconv_stem7_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def apply_with_stats(x, w, w_bit, groups=1, bins=None):
        """apply(...) that also leaves y._alignq_bnq_part (see QConvGemmFn.apply_with_stats)"""
        L.MB.gemm = None
        y = QConvStem7Fn.apply(x, w, w_bit, groups, True, bins)
        if L.MB.gemm is not None:
            y._alignq_bnq_part = L.MB.gemm + (int(groups),)
            L.MB.gemm = None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        dw = None
        if ctx.needs_input_grad[1]:
            if not gy.is_contiguous(memory_format=torch.channels_last):
                gy = gy.contiguous(memory_format=torch.channels_last)
            B, _, H, W = x.shape
            lib = L.load()
            dw = torch.empty_like(w)
            ws = _ws(lib.alignq_qconv_stem7_wgrad_ws_bytes(B, H, W), w.device)
            pending = fused.active_wgrads()
            if pending is not None:         # the slab reduction rides in the step's closing reduction launch
                ns = ctypes.c_int(0)
                L.check(lib.alignq_qconv_stem7_wgrad(L.ptr(x), L.ptr(gy), None, L.ptr(ws), B, H, W, ctypes.byref(ns), L.stream_ptr()),
                        "alignq_qconv_stem7_wgrad")
                pending.add(ws, dw, ns.value, 64 * 147)
            else:
                L.check(lib.alignq_qconv_stem7_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(ws), B, H, W, None, L.stream_ptr()),
                        "alignq_qconv_stem7_wgrad")
        return None, dw, None, None, None, None


def qconv_stem_supported(x, w, stride, padding, dilation, groups, bias, w_bit) -> bool:
    """The stem convolution alignq_conv_stem_nhwc_fwd implements: 3 -> 16 channels, 3x3 / stride 1 / padding 1, width 32,
    channels-last fp32 input that needs no gradient, <= 8-bit quantised filter."""
    if bias is not None or groups != 1 or not (1 <= w_bit <= 8) or x.requires_grad:
        return False
    if tuple(stride) != (1, 1) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and w.dtype == torch.float32):
        return False
    B, C, H, W = x.shape
    cl = torch.channels_last
    return (C == 3 and tuple(w.shape) == (16, 3, 3, 3) and W == 32 and H % 4 == 0 and x.is_contiguous(memory_format=cl)
            and not x.is_contiguous() and w.is_contiguous(memory_format=cl))


class QConvStemFn(torch.autograd.Function):
    """The stem convolution (model/resnet.py: conv0 = Conv2d_Q(3, 16, 3, 1, 1)) on alignq_conv_stem_nhwc_fwd / _wgrad."""

    @staticmethod
    def forward(ctx, x, w, w_bit):
        B, _, H, W = x.shape
        lib = L.load()
        y = torch.empty((B, 16, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        n_parts = lib.alignq_conv_stem_bn_parts(B, H, W)
        part = torch.empty(16, n_parts, 2, dtype=torch.float32, device=x.device)
        L.check(lib.alignq_conv_stem_nhwc_fwd(L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, int(w_bit), L.ptr(part), L.stream_ptr()),
                "alignq_conv_stem_nhwc_fwd")
        ctx.save_for_backward(x, w)
        # 2: the filter-gradient kernel (the stem's only gradient) reduces the site backward's per-tile sums itself
        ctx.link = None
        if w.requires_grad:
            ctx.link = fused.LazyLink()
            L.MB.conv3x3 = (part, n_parts, 2, ctx.link)
        else:
            L.MB.conv3x3 = (part, n_parts)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        lazy = fused.take_lazy_dz(ctx.link, gy)
        gy = gy.contiguous(memory_format=torch.channels_last)
        B, _, H, W = x.shape
        bz, bab, bsave, bk, bpart, bdg, bdb = _lazy_fields(lazy, not ctx.needs_input_grad[1], B, 16, H * W, x.device)
        lib = L.load()
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            ws = _ws(256 * 16 * 27 * 4, x.device)
            pending = fused.active_wgrads()
            if pending is not None:
                ns = ctypes.c_int(0)
                L.check(lib.alignq_conv_stem_nhwc_wgrad(L.ptr(x), L.ptr(gy), None, L.ptr(ws), B, H, W, ctypes.byref(ns),
                                                        L.ptr(bz), L.ptr(bab), L.ptr(bsave), L.ptr(bk), L.ptr(bpart),
                                                        L.ptr(bdg), L.ptr(bdb), L.stream_ptr()), "alignq_conv_stem_nhwc_wgrad")
                pending.add(ws, dw, ns.value, 16 * 27)
            else:
                L.check(lib.alignq_conv_stem_nhwc_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(ws), B, H, W, None, L.ptr(bz),
                                                        L.ptr(bab), L.ptr(bsave), L.ptr(bk), L.ptr(bpart), L.ptr(bdg),
                                                        L.ptr(bdb), L.stream_ptr()), "alignq_conv_stem_nhwc_wgrad")
        return None, dw, None

    @staticmethod
    def apply_with_stats(x, w, w_bit):
        L.MB.conv3x3 = None
        y = QConvStemFn.apply(x, w, w_bit)
        if L.MB.conv3x3 is not None:
            y._alignq_bn_part = L.MB.conv3x3
            L.MB.conv3x3 = None
        return y
