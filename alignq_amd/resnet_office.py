"""Harness model for BASELINE.json config 5: ResNet-50 + DANN head for Office-31 (224x224 inputs), the caller of the
hot path in the Office tree.  Wiring follows cdf_alignment_admm/dann_office/model/resnet.py: Bottleneck :89-156 (act_q1,
act_q2 = plain CDF quantisers, act_q3 = activation_quantize_fn2 with the block's ADMM), ResNet :159-271 (stem conv7x7 ->
bn -> act_q0 -> relu -> maxpool; the fc layer exists but only `feature` is used), ReverseLayerF :302-313, DANN :316-334.
Attribute and parameter names are the reference's (feature.layerN.M.{conv1,bn1,conv2,bn2,conv3,bn3,admm0,downsample.0,
downsample.1}, class_classifier.c_fc3, domain_classifier.d_fc2) so its checkpoints load.  No ImageNet download: random init
(the reference's `pretrained=True` default needs the network)."""
from __future__ import annotations


import torch
import torch.nn as nn

from . import config
from . import office as Q
from .admm import ADMM


def conv3x3(wbit, stage, cin, cout, stride=1):
    return Q.conv2d_Q_fn(w_bit=wbit, stage=stage)(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(wbit, stage, cin, cout, stride=1):
    return Q.conv2d_Q_fn(w_bit=wbit, stage=stage)(cin, cout, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, wbit, abit, stage, inplanes, planes, stride=1, downsample=None, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.))          # dann_office/model/resnet.py:103 (groups == 1)
        self.conv1 = conv1x1(wbit, stage, inplanes, width)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = conv3x3(wbit, stage, width, width, stride)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = conv1x1(wbit, stage, width, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        dim = config.args.train_batch_size if self.training else config.args.eval_batch_size
        self.admm0 = ADMM(dim)
        self.act_q1 = Q.activation_quantize_fn(a_bit=abit, stage=stage)
        self.act_q2 = Q.activation_quantize_fn(a_bit=abit, stage=stage)
        self.act_q3 = Q.activation_quantize_fn2(a_bit=abit, stage=stage, admm=self.admm0)

    def forward(self, x, groups=1, loss_vec=False):
        """groups > 1 (OfficeTrainStep(dual=True)): x holds the source and the target batch back to back; the convolutions
        (per sample) run once on both, every batch statistic / quantiser site / correlation per slice in pass order.
        loss_vec (groups > 1; ResNet.forward passes it): the trans loss comes back as the VECTOR of the slices' losses (a view, no
        kernel) for a caller that sums all its sites at once; the default is the scalar, like the reference's block."""
        trans_loss = 0.
        identity = x
        if groups > 1:
            # x feeds conv1 AND the shortcut (directly, or through the downsample convolution): the two gradients meet in the
            # producing site's kernel instead of in an elementwise add (fused.GradFork; a no-op unless x comes from a folded site)
            from . import fused
            x, x_short = fused.fork_block_input(x)
            identity = x_short
            pack = getattr(self, "pack_bins", False)        # N2: conv2 / conv3 read int16 level indices instead of fp32 values
            out = self.act_q1.forward_bn_relu(self.bn1, self.conv1(x), groups, pack)
            out = self.act_q2.forward_bn_relu(self.bn2, self.conv2(out), groups, pack)
            if self.downsample is not None:
                identity = fused.bn_only(self.downsample[1], self.downsample[0](x_short), groups)
            out, loss = self.act_q3.forward_bn_res_relu(self.bn3, self.conv3(out), identity, groups, loss_vec=loss_vec)
            return out, loss              # (= 0. + loss without the launch that forms it)
        x_short = x
        if getattr(self, "fuse_bn", False):         # opt-in (OfficeTrainStep): batch-norm + quantiser + ReLU as one chain
            from . import fused
            x, x_short = fused.fork_block_input(x)
            identity = x_short
            pack = getattr(self, "pack_bins", False)
            out = self.act_q1.forward_bn_relu(self.bn1, self.conv1(x), 1, pack)
            out = self.act_q2.forward_bn_relu(self.bn2, self.conv2(out), 1, pack)
        elif getattr(self, "fuse_relu", False):     # opt-in: quantiser + ReLU in one launch each way
            out = self.act_q1.forward_relu(self.bn1(self.conv1(x)))
            out = self.act_q2.forward_relu(self.bn2(self.conv2(out)))
        else:
            out = self.relu(self.act_q1(self.bn1(self.conv1(x))))
            out = self.relu(self.act_q2(self.bn2(self.conv2(out))))
        if getattr(self, "fuse_bn", False):         # bn3 folded into the site kernels, the downsample batch-norm on the same family
            if self.downsample is not None:
                from . import fused
                identity = fused.bn_only(self.downsample[1], self.downsample[0](x_short))
            out, loss = self.act_q3.forward_bn_res_relu(self.bn3, self.conv3(out), identity)
            return out, loss              # (= 0. + loss without the launch that forms it)
        if getattr(self, "fuse_relu", False):       # `out += identity; relu` inside the site kernels
            z = self.bn3(self.conv3(out))
            if self.downsample is not None:
                identity = self.downsample(x)
            out, loss = self.act_q3.forward_res_relu(z, identity)
            return out, trans_loss + loss
        out, loss = self.act_q3(self.bn3(self.conv3(out)))
        trans_loss += loss
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        out = self.relu(out)
        return out, trans_loss


class ResNet(nn.Module):
    def __init__(self, wbit, abit, stage, block, layers, num_classes=1000, width_per_group=64):
        super().__init__()
        self.wbit, self.abit, self.stage = wbit, abit, stage
        self.base_width = width_per_group
        self.act_q0 = Q.activation_quantize_fn(a_bit=abit, stage=stage)
        self.inplanes = 64
        self.conv1 = Q.conv2d_Q_fn(w_bit=wbit, stage=stage)(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.wbit, self.stage, self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.wbit, self.abit, self.stage, self.inplanes, planes, stride, downsample, self.base_width)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.wbit, self.abit, self.stage, self.inplanes, planes, base_width=self.base_width))
        return nn.Sequential(*layers)

    def forward(self, x, groups=1):
        from . import fused
        with fused.conv_groups_scope(groups):        # (the GEMM convolutions' batch-norm statistics epilogue sums per batch slice)
            return self._forward(x, groups)

    def _forward(self, x, groups):
        trans_loss = 0.
        from . import fused
        # staged weight quantisation (OfficeTrainStep.stage_weights, data parallelism): stage i's filters right before stage i
        wq = getattr(self, "_wq_stage", None) or (lambda i: None)
        wq(0)
        if groups > 1:
            q0 = self.act_q0.forward_bn_relu(self.bn1, self.conv1(x), groups)
            x = self.maxpool(q0)
            if hasattr(q0, "_alignq_levels"):        # the maximum of quantiser levels is one of them (fused.tag_levels)
                x._alignq_levels = q0._alignq_levels
            losses = []
            with fused.Site1Batch() as s1:      # the folded tails' reductions / preparations: one launch each for all 16 sites
                for li, layers in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
                    wq(li + 1)
                    for layer in layers:
                        x, loss = layer(x, groups, loss_vec=True)
                        losses.append(loss)
                tens = [l for l in losses if torch.is_tensor(l)]
                total_t = s1.total(tens) if tens else None
            # one stack + one sum instead of 16 scalar additions on the in-order chain (the fast path only: the value may differ
            # from main.py's running sum in the last bit, the gradients - ones - do not)
            # (a site without an ADMM term - abitW == 32, method != 'ours', a deferred-loss context - returns the number 0: those
            # are summed as numbers, as the running sum of the one-pass form does)
            rest = sum(l for l in losses if not torch.is_tensor(l))
            # (the folded sites return the VECTOR of their slices' losses - a view, no kernel: one concatenation + one sum)
            total = total_t + rest if tens else rest
            return torch.flatten(self.avgpool(x), 1), total
        if getattr(self, "fuse_bn", False):
            q0 = self.act_q0.forward_bn_relu(self.bn1, self.conv1(x))
            x = self.maxpool(q0)
            if hasattr(q0, "_alignq_levels"):
                x._alignq_levels = q0._alignq_levels
        elif getattr(self, "fuse_relu", False):
            x = self.maxpool(self.act_q0.forward_relu(self.bn1(self.conv1(x))))
        else:
            x = self.maxpool(self.relu(self.act_q0(self.bn1(self.conv1(x)))))
        for li, layers in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            wq(li + 1)
            for layer in layers:
                x, loss = layer(x)
                trans_loss += loss
        feature = torch.flatten(self.avgpool(x), 1)
        return feature, trans_loss


def resnet50_quant(wbit, abit, stage):
    return ResNet(wbit, abit, stage, Bottleneck, [3, 4, 6, 3])


class ReverseLayerF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg() * ctx.alpha, None


class DANN(nn.Module):
    def __init__(self, arch, wbit, abit, stage, num_classes=31):
        super().__init__()
        self.feature = arch(wbit, abit, stage)
        self.class_classifier = nn.Sequential()
        self.class_classifier.add_module("c_fc3", nn.Linear(2048, num_classes))
        self.domain_classifier = nn.Sequential()
        self.domain_classifier.add_module("d_fc2", nn.Linear(2048, 2))

    def forward(self, input_data, alpha):
        feature, trans_loss = self.feature(input_data)
        feature = feature.view(-1, 2048)
        reverse_feature = ReverseLayerF.apply(feature, alpha)
        return self.class_classifier(feature), self.domain_classifier(reverse_feature), trans_loss

    def forward_dual(self, x_src, x_tgt, alpha):
        """The source and the target pass of one DANN iteration (dann_office/main.py:351-372) in ONE traversal: both batches
        back to back through the per-sample convolutions, every batch statistic / quantiser site / correlation per domain in
        pass order (running statistics: source then target; ADMM.D: the target's), so every parameter has ONE incoming
        gradient.  Returns (class logits of the source batch, domain logits source, domain logits target, trans loss of both
        passes) - the target pass's class logits, which main.py never uses, are not computed."""
        B = x_src.shape[0]
        x = torch.cat([x_src, x_tgt], 0)
        if x_src.dim() == 4 and x_src.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        feature, trans_loss = self.feature(x, groups=2)
        feature = feature.view(-1, 2048)
        dom = self.domain_classifier(ReverseLayerF.apply(feature, alpha))
        return self.class_classifier(feature[:B]), dom[:B], dom[B:], trans_loss


def resnet50_dann(wbit, abit, stage="aligned", **kwargs):
    return DANN(resnet50_quant, wbit, abit, stage)
