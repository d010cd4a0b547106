"""Harness models: the callers of the hot path for the CIFAR configs (BASELINE.json configs 1-4).

Pre-activation ResNet-20/56 wired like the reference's model/resnet.py (ADMM tree :36-167, CDF tree
:33-137) with the SAME attribute / parameter names (`conv0`, `admm0`, `act_q0`, `layers[i].{admm0,admm1,
admm_skip,act_q0,act_q1,act_skip_q,bn0,conv0,bn1,conv1,skip_conv,skip_bn}`, `bn`, `logit`), so the
reference drivers' gathers (main.py:313-369) and checkpoints (state_dict keys) work unchanged.  Conv/BN/
linear go to MIOpen/rocBLAS through PyTorch-ROCm (not part of the hot path); every weight and activation
quantiser, correlation and ADMM op goes through alignq_amd's HIP kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cdf_alignment as _cdf
from . import cdf_alignment_admm as _admm
from . import config
from .admm import ADMM
from .fused import bn_act_relu, bn_only, bn_site, twin_sites


def _transition_pair(conv3, conv1, x):
    """(conv3(x), conv1(x)) through ops.QTransitionFn when both stride-2 convolutions of a transition block run on this
    repository's kernels (TrainStep(channels_last=True, qconv=True)); None otherwise (the caller takes the separate path)."""
    if not (getattr(conv3, "use_qconv", False) and getattr(conv1, "use_qconv", False) and x.is_cuda):
        return None
    from . import ops
    # (decided on the parameters: the quantised filters have their shape, layout and gradient requirement, and a quantiser
    # call consumes the pre-quantised tensor parked for it)
    if not ops.transition_supported(conv3, conv1, x, conv3.weight, conv1.weight):
        return None
    w3, w1 = conv3.quantize_fn(conv3.weight), conv1.quantize_fn(conv1.weight)
    return ops.QTransitionFn.apply_with_stats(x, w3, w1, conv3.quantize_fn.w_bit)


class PreActBlock_conv_Q(nn.Module):
    """Pre-activation basic block; `tree` selects the ADMM (tuple-returning) or CDF-only activation fn."""

    def __init__(self, stage, wbit, abit, in_planes, out_planes, stride=1, tree="admm"):
        super().__init__()
        self.tree = tree
        self.fuse_bn = False           # set by alignq_amd.resnet.enable_bn_fusion
        ns = _admm if tree == "admm" else _cdf
        Conv2d = ns.conv2d_Q_fn(w_bit=wbit, stage=stage)
        if tree == "admm":
            dim = config.args.train_batch_size if self.training else config.args.eval_batch_size
            self.admm0 = ADMM(dim)
            self.admm1 = ADMM(dim)
            self.act_q0 = ns.activation_quantize_fn(a_bit=abit, stage=stage, admm=self.admm0)
            self.act_q1 = ns.activation_quantize_fn(a_bit=abit, stage=stage, admm=self.admm1)
        else:
            self.act_q0 = ns.activation_quantize_fn(a_bit=abit, stage=stage)
            self.act_q1 = ns.activation_quantize_fn(a_bit=abit, stage=stage)
        self.bn0 = nn.BatchNorm2d(out_planes)
        self.conv0 = Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_planes)
        self.conv1 = Conv2d(out_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.skip_conv = None
        if stride != 1:
            if tree == "admm":
                self.admm_skip = ADMM(dim)
                self.act_skip_q = ns.activation_quantize_fn(a_bit=abit, stage=stage, admm=self.admm_skip)
            else:
                self.act_skip_q = ns.activation_quantize_fn(a_bit=abit, stage=stage)
            self.skip_conv = Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=False)
            self.skip_bn = nn.BatchNorm2d(out_planes)

    def _q(self, fn, x):
        if self.tree == "admm":
            return fn(x)
        return fn(x), 0

    def _bnq(self, bn, fn, z, relu=False, residual=None, pack=False):
        """act(bn(z)) [+ residual] [-> relu]; with fuse_bn the batch-norm, the shortcut add and the ReLU are folded into
        the site kernels (alignq_amd.fused.bn_site).  pack: the output may come back as a packed handle (int8 / int16 level
        indices) - only for a tensor whose SOLE consumer is a Conv2d_Q."""
        if self.tree == "admm" and self.fuse_bn:
            return bn_site(bn, fn, z, relu=relu, residual=residual, pack=pack)
        if self.tree == "cdf" and self.fuse_bn:
            # configuration 1 (cdf_alignment/resnet-20-cifar-10/model/resnet.py:63-79): no correlation term, so the fold is the
            # plain-quantiser family (alignq_bnq_fwd / _bwd, formula 1); it falls back to exactly the composition below when
            # the tensor is not channels-last fp32 in training mode
            return bn_act_relu(bn, fn, z, 1, relu=relu, residual=residual), 0
        # no fold (fuse_bn off: the exact-global correlation, a caller's choice): the batch-norm alone still runs on the folded
        # family's kernels when the tensor is channels-last fp32 in training mode (fused.bn_only), else the module itself
        out, loss = self._q(fn, bn_only(bn, z))
        if residual is not None:
            out += residual
        return (F.relu(out) if relu else out), loss

    def forward(self, x):
        trans_loss = 0.
        if self.skip_conv is not None:
            pair = _transition_pair(self.conv0, self.skip_conv, x)   # both stride-2 convolutions in one launch each way
            if pair is not None:
                z0, zs = pair
            else:
                z0, xa = self.conv0.forward_with_shortcut(x)        # xa = x (skip_conv's input gradient joins conv0's)
                zs = self.skip_conv(xa)
            # the two sites behind the transition's convolutions do not depend on each other: with the fold they share ONE launch
            # (fused.twin_sites: each alone leaves half the chip idle)
            with twin_sites():
                shortcut, loss = self._bnq(self.skip_bn, self.act_skip_q, zs)
                trans_loss += loss
                out, loss = self._bnq(self.bn0, self.act_q0, z0, relu=True,
                                      pack=getattr(self, "pack_bins", False) and getattr(self.conv1, "use_qconv", False))
                trans_loss += loss
        else:
            z0, shortcut = self.conv0.forward_with_shortcut(x)      # shortcut = x (its gradient joins conv0's data gradient)
            # relu(act_q0(bn0(.))) feeds conv1 and nothing else: with pack_bins it travels as its level indices (N2)
            out, loss = self._bnq(self.bn0, self.act_q0, z0, relu=True,
                                  pack=getattr(self, "pack_bins", False) and getattr(self.conv1, "use_qconv", False))
            trans_loss += loss
        out, loss = self._bnq(self.bn1, self.act_q1, self.conv1(out), relu=True, residual=shortcut)   # out += shortcut; relu
        trans_loss += loss
        if self.tree == "admm":
            return out, trans_loss
        return out


class PreActResNet(nn.Module):
    def __init__(self, block, num_units, wbit, abit, stage, num_classes, tree="admm"):
        super().__init__()
        self.tree = tree
        self.fuse_bn = False
        ns = _admm if tree == "admm" else _cdf
        Conv2d = ns.conv2d_Q_fn(w_bit=wbit, stage=stage)
        self.conv0 = Conv2d(3, 16, kernel_size=3, stride=1, padding=1, bias=False)
        if tree == "admm":
            dim = config.args.train_batch_size if self.training else config.args.eval_batch_size
            self.admm0 = ADMM(dim)
            self.act_q0 = ns.activation_quantize_fn(a_bit=abit, stage=stage, admm=self.admm0)
        else:
            self.act_q0 = ns.activation_quantize_fn(a_bit=abit, stage=stage)
        self.layers = nn.ModuleList()
        in_planes = 16
        strides = [1] * num_units[0] + [2] + [1] * (num_units[1] - 1) + [2] + [1] * (num_units[2] - 1)
        channels = [16] * num_units[0] + [32] * num_units[1] + [64] * num_units[2]
        for stride, channel in zip(strides, channels):
            self.layers.append(block(stage, wbit, abit, in_planes, channel, stride, tree=tree))
            in_planes = channel
        self.bn = nn.BatchNorm2d(16)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.logit = nn.Linear(64, num_classes)

    def forward(self, x):
        if self.tree == "admm" and self.fuse_bn:
            out, loss = bn_site(self.bn, self.act_q0, self.conv0(x), relu=True)
            trans_loss = 0. + loss
        elif self.tree == "admm":
            out, loss = self.act_q0(bn_only(self.bn, self.conv0(x)))
            trans_loss = 0. + loss
            out = F.relu(out)
        elif self.fuse_bn:
            out = bn_act_relu(self.bn, self.act_q0, self.conv0(x), 1, relu=True)
        else:
            out = F.relu(self.act_q0(self.bn(self.conv0(x))))
        for layer in self.layers:
            if self.tree == "admm":
                out, loss = layer(out)
                trans_loss += loss
            else:
                out = layer(out)
        if getattr(self, "_features_only", False):     # TrainStep's fused head takes over from here (pool, logit, loss)
            return out, (trans_loss if self.tree == "admm" else None)
        out = self.avgpool(out)
        out = out.view(out.size(0), -1)
        out = self.logit(out)
        if self.tree == "admm":
            return out, trans_loss
        return out


def resnet20_quant(bitW, abitW, stage="second", num_classes=10, tree="admm"):
    return PreActResNet(PreActBlock_conv_Q, [3, 3, 3], bitW, abitW, stage, num_classes, tree=tree)


def resnet56_quant(bitW, abitW, stage="second", num_classes=10, tree="admm"):
    return PreActResNet(PreActBlock_conv_Q, [9, 9, 9], bitW, abitW, stage, num_classes, tree=tree)


def enable_bn_fusion(model, on=True):
    """Fold every batch-norm that feeds an ADMM activation site into the site kernels (training mode, 64 < batch <= 128).
    Results match the unfused composition to fp32 rounding; module/parameter names and state_dict are unchanged."""
    for m in model.modules():
        if hasattr(m, "fuse_bn"):
            m.fuse_bn = bool(on)
    return model
