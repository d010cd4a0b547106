"""Host-side mirror of the reference's model/quantization.py for the three source trees.

`make_namespace(tree)` builds the module-level names the reference exports
(`uniform_quantize`, `cdf`, `weight_quantize_fn`, `activation_quantize_fn[2]`, `corr`, `conv2d_Q_fn`)
with the signatures and return arities of that tree:

  tree="admm"   cdf_alignment_admm/resnet-{20,56}-cifar-10/model/quantization.py:19-156
  tree="cdf"    cdf_alignment/*/model/quantization.py:15-122
  tree="office" cdf_alignment_admm/{dann,dsan}_office/model/quantization.py:20-181

Importable drop-ins: alignq_amd.cdf_alignment_admm, alignq_amd.cdf_alignment, alignq_amd.office.
All arithmetic runs in hand-written HIP kernels (alignq_amd/csrc) through ops.py; there is no eager
fallback.  Options the reference reads from its global `args` come from alignq_amd.config.args.
"""
from __future__ import annotations

import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import config, ops

_FORMULA = {"admm": L.FORMULA_ADMM, "office": L.FORMULA_ADMM, "cdf": L.FORMULA_CDF}
_EPS = {"admm": 0.0, "office": 1e-5, "cdf": 0.0}


def uniform_quantize(k):
    """model/quantization.py:19-38 — returns a callable Tensor -> Tensor (round forward, STE backward)."""
    def apply(x):
        return ops.UniformQuantizeFn.apply(x, k)
    return apply


def make_namespace(tree: str) -> types.SimpleNamespace:
    formula, eps = _FORMULA[tree], _EPS[tree]

    class cdf(nn.Module):
        """cdf(m, s, quant_src).forward(tensor) -> (cdf, pdf)  (ADMM tree :41-59; CDF tree :37-50).  Values from the HIP weight
        kernel with the given (m, s); gradients of BOTH outputs w.r.t. `tensor` and - when they are tensors of the autograd graph,
        as in the reference's own use cdf(torch.mean(x), torch.std(x), 'w') (:78) - w.r.t. m and s (ops.CdfFn)."""

        def __init__(self, m, s, quant_src):
            super().__init__()
            self.m, self.s, self.quant_src = m, s, quant_src

        def forward(self, tensor):
            x = L.dev_f32(tensor, "tensor")
            m = torch.as_tensor(self.m, dtype=torch.float32, device=x.device).reshape(())
            s_ = torch.as_tensor(self.s, dtype=torch.float32, device=x.device).reshape(())
            kc = 1.0 if tree == "cdf" else (2.0 * config.args.act_range if self.quant_src == "a" else 2.0)
            scale = config.args.act_range if (tree != "cdf" and self.quant_src == "a") else 1.0
            return ops.CdfFn.apply(x, m, s_, formula, float(kc), float(scale))

    class weight_quantize_fn(nn.Module):
        def __init__(self, w_bit, stage):
            super().__init__()
            self.w_bit = w_bit
            self.stage = stage
            self.uniform_q = uniform_quantize(k=w_bit)
            self._formula = formula
            self._pre = None      # (weight, q, cdf, pdf) parked by fused.prequantize_weights for the next call

        def forward(self, x):
            if self.w_bit == 32:
                if tree != "cdf":
                    self.weight_cdf = x
                    self.weight_q = x
                return x
            pre, self._pre = self._pre, None
            self._bins = None
            if pre is not None and pre[0] is x:
                q, c, pdf = pre[1], pre[2], pre[3]
                if len(pre) > 4:
                    self._bins = (q.data_ptr(), pre[4])          # the filter's packed integer bins (fused.prequantize_weights(pack=True))
            else:
                q, c, pdf = ops.WeightQuantFn.apply(x, self.w_bit, formula)
            if tree != "cdf":   # the CDF tree keeps these as locals (quantization.py:70-72)
                # (weight_q is kept WITHOUT its autograd history: same values for every reader - main.py:326-327 reads cdf / pdf -
                # but a module attribute with history would keep each iteration's graph alive into the next: see
                # train_step.retained_graph_params)
                self.weight_cdf, self.weight_pdf, self.weight_q = c, pdf, q.detach()
            else:
                self._weight_cdf, self._weight_pdf = c, pdf
            return q

        def take_bins(self, weight_q):
            """(bf16, f16) bins of `weight_q` if the last forward left them (else None: the convolution packs them itself)."""
            held = getattr(self, "_bins", None)
            return held[1] if held is not None and held[0] == weight_q.data_ptr() else None

    def _plain_act(x, a_bit, stage):
        if a_bit == 32 and stage != "align":
            return x
        if a_bit == 32 and tree == "cdf":
            # returns the raw cdf (quantization.py:100-101)
            xx = L.dev_f32(x, "activation")
            ms = torch.tensor([0.0, 1.0], dtype=torch.float32, device=xx.device)
            _, c, pdf, _ = ops.weight_quant_given_stats(xx.detach(), ms, 32, L.FORMULA_CDF, True)
            return _AttachGrad.apply(x, c, pdf * 0.5) if x.requires_grad else c
        return ops.ActQuantFn.apply(x, a_bit, config.args.act_range, formula)

    def _plain_act_relu(x, a_bit, stage):
        """relu(_plain_act(x)); one launch each way when the quantiser is active."""
        if (a_bit == 32) or not (x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0):
            return torch.relu(_plain_act(x, a_bit, stage))
        if getattr(config.args, "pack_bins", False) and ops.bin_dtype(a_bit, config.args.act_range, formula) is not None:
            # N2: the node keeps the 1-2 B level index instead of fp32 relu(x_q) (no narrow form: 16 < k < 32 or a large
            # act_range -> the fp32 path below, like the reference)
            return ops.ActQuantPackedFn.apply(x, a_bit, config.args.act_range, formula, True)[0]
        return ops.ActQuantReluFn.apply(x, a_bit, config.args.act_range, formula)

    def _site_act(mod, x):
        a_bit = mod.a_bit
        if a_bit == 32 and mod.stage != "align":
            return x, 0
        gc = getattr(mod, "global_corr", None)        # set per model by dp.attach(..., global_corr=True)
        if gc is None:
            gc = getattr(config.args, "global_corr", None)       # process-wide switch (tests)
        if config.args.method == "ours" and a_bit < 32 and gc is not None:
            # opt-in exact-global-batch correlation (dp.attach(..., global_corr=True)): D is the
            # [B_g, B_g] matrix of the concatenated batch, identical on every rank.  ADMM tree (round 4): x is exchanged ONCE, the
            # feature shard re-forms the transform and leaves D_r from the pair kernels (dp.global_site_D); the CDF-only formula
            # keeps the two-correlation composition
            from . import dp
            grp = None if gc is True else gc
            admm = mod.opt
            r_ = config.args.act_range
            xq = ops.ActQuantFn.apply(x, a_bit, r_, formula)
            if formula == L.FORMULA_ADMM and x.is_cuda:
                D = dp.global_site_D(x, a_bit, r_, eps, grp)
            else:
                t = ops.ActQuantFn.apply(x, 32, r_, formula)        # k == 32 writes the pre-round transform itself
                D = dp.global_corr(t, eps, grp) - dp.global_corr(x, eps, grp)
            return xq, admm(D)
        if config.args.method == "ours" and a_bit < 32 and x.shape[0] > L.MAX_BATCH:
            # above the 128 rows the fused kernels hold on chip: the site composed from the blocked correlation
            admm = mod.opt
            xq, loss, D = ops.site_unfused(x, admm, a_bit, config.args.act_range, eps, formula)
            admm.D = D
            return xq, loss
        if config.args.method == "ours" and a_bit < 32:
            admm = mod.opt
            from . import fused
            deferred = fused.active_deferred()
            if deferred is not None:
                bufs = None
                if deferred.side is not None:        # persistent buffers only matter for side-stream launches
                    if not hasattr(mod, "_site_bufs"):
                        mod._site_bufs = {}
                    bufs = mod._site_bufs
                rec = deferred.new_record(x.shape[0], x.device)
                xq, loss, D = ops.SiteFn.apply(x, admm.alterD, admm.gamma, a_bit, config.args.act_range, eps,
                                               admm.mu, admm.rho, deferred.side, bufs, rec)
                admm.D = D
                (deferred.add_record_loss if rec is not None else deferred.add)(loss)
                return xq, 0.0          # the real loss is summed once by DeferredLosses.total()
            xq, loss, D = ops.SiteFn.apply(x, admm.alterD, admm.gamma, a_bit, config.args.act_range, eps,
                                           admm.mu, admm.rho)
            admm.D = D
            return xq, loss
        return _plain_act(x, a_bit, mod.stage), 0

    def _site_act_res_relu(mod, x, residual):
        """(relu(act(x)[0] + residual), loss): the Office bottleneck's tail.  One launch each way on the small-batch site
        kernels when they apply (and no deferred-loss context is active), else exactly the composition."""
        from . import fused
        if (config.args.method == "ours" and mod.a_bit < 32 and fused.active_deferred() is None
                and ops.site_res_supported(x, residual)):
            admm = mod.opt
            y, loss, D = ops.SiteFn.apply(x, admm.alterD, admm.gamma, mod.a_bit, config.args.act_range, eps, admm.mu,
                                          admm.rho, None, None, None, residual, True)
            admm.D = D
            return y, loss
        out, loss = _site_act(mod, x)
        out = out + residual
        return torch.relu(out), loss

    class _act_plain(nn.Module):
        def __init__(self, a_bit, stage):
            super().__init__()
            self.a_bit, self.stage = a_bit, stage
            self.uniform_q = uniform_quantize(k=a_bit)

        def forward(self, x):
            return _plain_act(x, self.a_bit, self.stage)

        def forward_relu(self, x):
            """relu(self(x)) in one launch each way (not part of the reference's interface: an opt-in for the caller)."""
            return _plain_act_relu(x, self.a_bit, self.stage)

        def forward_bn_relu(self, bn, z, groups=1, pack=False):
            """relu(self(bn(z))) with the training-mode batch-norm folded into the quantiser (SURVEY.md §8f-N1 on the Office
            path; not part of the reference's interface: an opt-in for the caller, alignq_amd.fused.bn_act_relu).  groups:
            z holds that many batch slices which the reference sends through the module one after the other."""
            from . import fused
            return fused.bn_act_relu(bn, self, z, formula, relu=True, groups=groups, pack=pack)

        def forward_packed(self, x, relu=False):
            """([relu](self(x)), bins): the quantised activation both as fp32 and as its narrow integer level index
            (ops.dequant_bins(bins, ...) reproduces the fp32 tensor bit for bit)."""
            return ops.ActQuantPackedFn.apply(x, self.a_bit, config.args.act_range, formula, relu)

    class _act_admm(nn.Module):
        def __init__(self, a_bit, stage, admm):
            super().__init__()
            self.a_bit, self.stage = a_bit, stage
            self.uniform_q = uniform_quantize(k=a_bit)
            self.opt = admm

        def forward(self, x):
            return _site_act(self, x)

        def forward_res_relu(self, x, residual):
            """(relu(self(x)[0] + residual), loss) (not part of the reference's interface: an opt-in for the caller)."""
            return _site_act_res_relu(self, x, residual)

        def forward_bn_res_relu(self, bn, z, residual, groups=1, loss_vec=False):
            """(relu(self(bn(z))[0] + residual), loss) with the training-mode batch-norm folded into the small-batch site
            kernels where that applies (SURVEY.md §8f-N1 on the Office path), else the composition.  groups > 1: the batch
            slices go through one after the other (loss = their sum, ADMM.D = the last slice's, like successive passes)."""
            from . import fused
            out = fused.bn_site_res_relu(bn, self, z, residual, eps, groups, loss_vec)      # (loss_vec: see there)
            if out is not None:
                return out
            if groups == 1:
                return _site_act_res_relu(self, bn(z), residual)
            Bg = z.shape[0] // groups
            outs, loss = [], 0.
            for i in range(groups):
                o, l_ = _site_act_res_relu(self, bn(z[i * Bg:(i + 1) * Bg]), residual[i * Bg:(i + 1) * Bg])
                outs.append(o)
                loss = loss + l_
            return torch.cat(outs, 0), loss

    def corr(x, y):
        """corr(x, y) -> [B,B] (ADMM tree :134-137; Office :158-161).  The reference only ever calls it with y is x: that
        is the SYRK served by the fused MFMA kernels; any other y takes the general exact-fp32 kernels."""
        if y is x or (y.data_ptr() == x.data_ptr() and y.shape == x.shape and y.stride() == x.stride()):
            return ops.CorrFn.apply(x, eps)
        return ops.CorrXYFn.apply(x, y, eps)

    def conv2d_Q_fn(w_bit, stage):
        class Conv2d_Q(nn.Conv2d):
            def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                         bias=True):
                super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
                self.quantize_fn = weight_quantize_fn(w_bit=w_bit, stage=stage)

            def forward(self, input, order=None):
                return self._conv(input, self.quantize_fn(self.weight))

            def _conv(self, input, weight_q):
                packed = getattr(input, "_alignq_bins", None)
                if packed is not None:
                    # N2: the input is a packed handle (fused.bn_site(pack=True)): its values are int8 / int16 level
                    # indices; the 3x3 body convolution reads them directly, anything else gets the dequantised tensor
                    bins, a_bit = packed
                    B_, C_, H_, W_ = input.shape
                    if (getattr(self, "use_qconv", False) and self.bias is None and self.groups == 1 and weight_q.is_cuda
                            and tuple(self.stride) == (1, 1) and tuple(self.padding) == (1, 1) and tuple(self.dilation) == (1, 1)
                            and tuple(weight_q.shape) == (C_, C_, 3, 3) and (C_, W_) in ((16, 32), (32, 16), (64, 8)) and H_ % 8 == 0
                            and 1 <= self.quantize_fn.w_bit <= 8 and weight_q.is_contiguous(memory_format=torch.channels_last)):
                        return ops.QConv3x3Fn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, False, bins, a_bit)
                    if (getattr(self, "use_qconv", False) and bins.dtype == torch.int16 and ops.level_count(input)
                            and ops.qconv_gemm_shape_supported(tuple(input.shape), weight_q, self.stride, self.padding, self.dilation,
                                                               self.groups, self.bias, self.quantize_fn.w_bit)):
                        # the ResNet-50 shapes: the GEMM kernels read the int16 indices directly (forward and filter gradient)
                        fb = self.quantize_fn.take_bins(weight_q)
                        if getattr(self, "emit_bn_stats", False):
                            from . import fused
                            return ops.QConvGemmFn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, self.stride[0],
                                                                    ops.level_count(input), fused.conv_groups(), fb, bins)
                        return ops.QConvGemmFn.apply(input, weight_q, self.quantize_fn.w_bit, self.stride[0], ops.level_count(input),
                                                     1, False, fb, bins)
                    from . import fused
                    input = fused.materialize(input)
                # opt-in (TrainStep(channels_last=True) sets use_qconv): the convolutions on the matrix cores
                # (with the batch-norm statistics of the output as a by-product for fused.bn_site)
                if getattr(self, "use_qconv", False):
                    args = (input, weight_q, self.stride, self.padding, self.dilation, self.groups, self.bias,
                            self.quantize_fn.w_bit)
                    if ops.qconv3x3_supported(*args):
                        return ops.QConv3x3Fn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit)
                    if ops.qconv_gen_supported(*args):
                        return ops.QConvGenFn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, self.padding[0])
                    if ops.qconv_stem_supported(*args):
                        return ops.QConvStemFn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit)
                    if ops.qconv_stem7_supported(*args):     # the Office stem (7x7, stride 2, 3 -> 64 channels)
                        bins = self.quantize_fn.take_bins(weight_q)
                        if getattr(self, "emit_bn_stats", False):
                            from . import fused
                            return ops.QConvStem7Fn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, fused.conv_groups(), bins)
                        return ops.QConvStem7Fn.apply(input, weight_q, self.quantize_fn.w_bit, 1, False, bins)
                    if ops.qconv_gemm_supported(*args):      # the ResNet-50 shapes: exact-product GEMMs (csrc/qgemm_kernels.hip)
                        bins = self.quantize_fn.take_bins(weight_q)
                        if getattr(self, "emit_bn_stats", False):     # the batch-norm behind this convolution takes its statistics
                            from . import fused                        # from the epilogue (fused.conv_partials)
                            return ops.QConvGemmFn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, self.stride[0],
                                                                    ops.level_count(input), fused.conv_groups(), bins)
                        return ops.QConvGemmFn.apply(input, weight_q, self.quantize_fn.w_bit, self.stride[0],
                                                     ops.level_count(input), 1, False, bins)
                return F.conv2d(input, weight_q, self.bias, self.stride, self.padding, self.dilation, self.groups)

            def forward_with_shortcut(self, input):
                """(conv(input), alias of input) for a block that uses its input twice — as the identity shortcut
                (`shortcut = x`) or as the input of the shortcut convolution: when the convolution runs on this repository's
                kernels (and a gradient is needed) the alias is an output of the same autograd node, so the second gradient
                is added inside the data-gradient kernel; otherwise plainly (forward(input), input)."""
                weight_q = self.quantize_fn(self.weight)
                if getattr(self, "use_qconv", False) and input.requires_grad:
                    args = (input, weight_q, self.stride, self.padding, self.dilation, self.groups, self.bias,
                            self.quantize_fn.w_bit)
                    if ops.qconv3x3_supported(*args):
                        return ops.QConv3x3Fn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, True)
                    if ops.qconv_gen_supported(*args):      # transition block: the alias feeds the shortcut convolution
                        return ops.QConvGenFn.apply_with_stats(input, weight_q, self.quantize_fn.w_bit, self.padding[0], True)
                return self._conv(input, weight_q), input

        return Conv2d_Q

    ns = types.SimpleNamespace(uniform_quantize=uniform_quantize, cdf=cdf, weight_quantize_fn=weight_quantize_fn,
                               conv2d_Q_fn=conv2d_Q_fn)
    if tree == "cdf":
        _act_plain.__name__ = _act_plain.__qualname__ = "activation_quantize_fn"
        ns.activation_quantize_fn = _act_plain
    elif tree == "admm":
        _act_admm.__name__ = _act_admm.__qualname__ = "activation_quantize_fn"
        ns.activation_quantize_fn = _act_admm
        ns.corr = corr
    else:
        _act_plain.__name__ = _act_plain.__qualname__ = "activation_quantize_fn"
        _act_admm.__name__ = _act_admm.__qualname__ = "activation_quantize_fn2"
        ns.activation_quantize_fn = _act_plain
        ns.activation_quantize_fn2 = _act_admm
        ns.corr = corr
    return ns


def make_uniform_admm_namespace() -> types.SimpleNamespace:
    """The paper's `use_cdf=False` ablation: cdf_alignment_admm/resnet-20-cifar-10/model/
    quantization_uniform_admm.py — plain uniform quantisers with the ADMM loss still attached.

    weight_quantize_fn.forward (:61-85): W_q = uniform_quantize(k)(W) (no CDF; only `.weight_q` is stored, plus
    `.weight_cdf = x` for w_bit == 32).  activation_quantize_fn.forward (:88-139): x_q = uniform_quantize(k)(x) and, with
    args.method containing 'ours' and a_bit < 32, D = corr(x,x) - corr(x,x) (zero wherever the correlation is finite, NaN
    where a feature has zero batch variance, exactly like the reference) and trans_loss = admm(D).  The two corr terms are
    the same tensor, so the reference's gradient of D w.r.t. x cancels term by term; D is detached here."""
    base = make_namespace("admm")

    class weight_quantize_fn(nn.Module):
        def __init__(self, w_bit, stage):
            super().__init__()
            self.w_bit, self.stage = w_bit, stage
            self.uniform_q = uniform_quantize(k=w_bit)

        def forward(self, x):
            if self.w_bit == 32:
                self.weight_cdf = x
                self.weight_q = x
                return x
            self.weight_q = self.uniform_q(x)
            return self.weight_q

    class activation_quantize_fn(nn.Module):
        def __init__(self, a_bit, stage, admm):
            super().__init__()
            self.a_bit, self.stage = a_bit, stage
            self.uniform_q = uniform_quantize(k=a_bit)
            self.opt = admm

        def forward(self, x):
            if self.a_bit == 32 and self.stage != "align":
                return x, 0
            activation_q = self.uniform_q(x)
            if "ours" in config.args.method and self.a_bit < 32:
                G = ops.CorrFn.apply(x.detach(), 0.0)
                trans_loss = self.opt(G - G)
            else:
                trans_loss = 0
            return (x if self.a_bit == 32 else activation_q), trans_loss

    def conv2d_Q_fn(w_bit, stage):
        class Conv2d_Q(nn.Conv2d):
            def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                         bias=True):
                super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
                self.quantize_fn = weight_quantize_fn(w_bit=w_bit, stage=stage)

            def forward(self, input, order=None):
                weight_q = self.quantize_fn(self.weight)
                return F.conv2d(input, weight_q, self.bias, self.stride, self.padding, self.dilation, self.groups)

        return Conv2d_Q

    return types.SimpleNamespace(uniform_quantize=uniform_quantize, cdf=base.cdf, weight_quantize_fn=weight_quantize_fn,
                                 activation_quantize_fn=activation_quantize_fn, corr=base.corr, conv2d_Q_fn=conv2d_Q_fn)


class _AttachGrad(torch.autograd.Function):
    """y = value (precomputed by a kernel) with dy/dx = jac elementwise."""

    @staticmethod
    def forward(ctx, x, value, jac):
        ctx.save_for_backward(jac)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors
        return g * jac, None, None
