"""One training iteration in the reference's order (cdf_alignment_admm/resnet-20-cifar-10/main.py:288-374):
zero_grad x2 -> forward -> CE + trans_loss -> backward -> SGD.step(idx, w_cdf, w_pdf, lam, lam2) ->
ADMM_OPT.step(alterD_idx, gamma_idx, Ds, alterDs, gammas, mus, rhos).

The list gathering of main.py:313-369 is done once (indices) / per step (tensors) without host syncs;
`capture()` records the whole iteration into one HIP graph (static input buffers), which removes the
per-launch host overhead that dominates the small CIFAR shapes."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import config
from .fused import DeferredLosses, DeferredWgrads, HeadCEFn, head_ce_supported, prequantize_weights
from .optimizer import ADMM_OPT, SGD, sgd_admm_step


_PROBE = "alignq_capture_probe"


def retained_graph_params(params):
    """Parameters whose AccumulateGrad node is kept alive by an autograd graph of an EARLIER iteration (a retained loss / output
    tensor, or ADMM.D kept with its graph as utils/admm.py:25 does).  Such a node is bound to the stream that iteration ran on;
    a backward inside a stream capture then makes autograd synchronise that stream with the capturing one, which pulls it into
    the capture: hipStreamEndCapture finds an unjoined stream (ROCm 7.2: a segmentation fault - round 4's crash when a step was
    captured after eager iterations whose outputs were still referenced).  Found by tagging: a node nobody else holds dies when
    the probe's temporary graph goes, and the next lookup creates an untagged one."""
    token = object()
    live = [p for p in params if p.requires_grad]
    # self-check: a parameter created HERE cannot be part of any earlier graph.  If its node survives the probe's
    # temporary graph too, something pins accumulation nodes (hooks registered on the node itself, a torch that caches it): the
    # probe would then call EVERY parameter stale and capture() would blame the caller's tensors for it
    fresh = torch.nn.Parameter(torch.zeros(1))
    for p in live + [fresh]:
        p.expand_as(p).grad_fn.next_functions[0][0].metadata[_PROBE] = token
    stale = []
    for p in live + [fresh]:
        acc = p.expand_as(p).grad_fn.next_functions[0][0]
        if acc.metadata.get(_PROBE) is token:
            stale.append(p)
        acc.metadata.pop(_PROBE, None)
    if any(p is fresh for p in stale):
        raise RuntimeError("retained_graph_params: the probe is unusable in this process - the gradient-accumulation node of a "
                           "parameter created inside the probe outlived its temporary graph (this torch build, or a component "
                           "that registers hooks on the accumulation nodes themselves, keeps them alive), so a retained autograd "
                           "graph cannot be told from a pinned node here; capture() cannot verify its precondition")
    return stale


def release_step_graphs(admms):
    """What the step itself may keep of an earlier iteration's autograd graph: ADMM.D with its history (the exact-global and the
    unfused sites store D as the reference does, utils/admm.py:25; the fused sites store a detached D)."""
    for a in admms:
        if a.D is not None and a.D.grad_fn is not None:
            a.D = a.D.detach()


def assert_no_retained_graph(params, who):
    stale = retained_graph_params(params)
    if stale:
        raise RuntimeError(
            f"{who}: {len(stale)} parameter(s) are still referenced by the autograd graph of an earlier iteration (a loss / output "
            "tensor of an eager step that is still alive).  Their gradient-accumulation nodes are bound to the stream that "
            "iteration ran on; capturing a backward that reuses them pulls that stream into the capture (hipStreamEndCapture "
            "crashes on the unjoined stream).  Drop those tensors (del the step's outputs, or .detach() them) and call capture() "
            "again.")


def _detached(outs):
    """The step's results without their autograd history: the backward already ran inside the step, and a result that kept the
    iteration's graph alive would keep its gradient-accumulation nodes alive too (retained_graph_params)."""
    return tuple(t.detach() if torch.is_tensor(t) else t for t in outs)


class TrainStep:
    def __init__(self, model, lr=0.04, momentum=0.9, weight_decay=1e-4, grad_hook=None, defer_losses=True, fuse_bn=True,
                 channels_last=False, qconv=True, pack_bins=True):
        """channels_last: keep activations and conv weights in torch.channels_last memory (values, parameter names and
        state_dict are unchanged).  MIOpen's NHWC convolution kernels need no layout transposes around the weight-gradient
        igemm (2.22 vs 2.50 ms per ResNet-20 step on MI355X); the quantise / Gram / ADMM kernels are layout-agnostic and the
        BN fold has a channels-last form."""
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
            if qconv:    # Conv2d_Q's 3x3 body convolutions (forward + data gradient) on the matrix cores, exact products
                for m in model.modules():
                    if hasattr(m, "quantize_fn"):
                        m.use_qconv = True
        if channels_last and qconv and fuse_bn and pack_bins:
            # N2: relu(act_q0(bn0(.))) of every block feeds conv1 only -> it is stored as its int8 / int16 level
            # index (no fp32 copy): the site forward writes, the convolution forward / filter gradient and the site backward's
            # ReLU mask read 1-2 B per element instead of 4
            for m in model.modules():
                if hasattr(m, "conv1") and hasattr(m, "act_q0") and hasattr(m, "bn0"):
                    m.pack_bins = True
        self.channels_last = channels_last
        self._wgrads = DeferredWgrads(fresh_grads=True) if (channels_last and qconv and torch.cuda.is_available()) else None
        self.model = model
        if fuse_bn:      # fold BN into the site kernels where shapes allow (training, 64 < batch <= 128); no-op otherwise
            for m in model.modules():
                if hasattr(m, "fuse_bn"):
                    m.fuse_bn = True
        self.defer_losses = defer_losses
        self._deferred = DeferredLosses() if (defer_losses and torch.cuda.is_available()) else None
        named = list(model.named_parameters())
        self.param_t = [(n, p) for n, p in named if "alterD" not in n and "gamma" not in n]
        self.param_admm = [(n, p) for n, p in named if "alterD" in n or "gamma" in n]
        self.optimizer_t = SGD([p for _, p in self.param_t], lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.optimizer_admm = ADMM_OPT([p for _, p in self.param_admm]) if self.param_admm else None
        # main.py:313-317: conv weights except the stem
        self.idx = [j for j, (n, _) in enumerate(self.param_t) if "conv" in n and "weight" in n][1:]
        self.alterD_idx = [j for j, (n, _) in enumerate(self.param_admm) if "alterD" in n]
        self.gamma_idx = [j for j, (n, _) in enumerate(self.param_admm) if "gamma" in n]
        self.convs = [c for layer in model.layers for c in (layer.conv0, layer.conv1, layer.skip_conv) if c is not None]
        self.all_convs = [m for m in model.modules() if hasattr(m, "quantize_fn")]
        self.admms = []
        if self.param_admm:
            self.admms = [model.admm0]
            for layer in model.layers:
                self.admms += [layer.admm0, layer.admm1]
                if layer.skip_conv is not None:
                    self.admms.append(layer.admm_skip)
        self.grad_hook = grad_hook          # e.g. the data-parallel all-reduce (alignq_amd.dp)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph2: Optional[torch.cuda.CUDAGraph] = None
        self._static = None
        self._one = None

    # -------------------------------------------------------------------------------------------
    def _forward_backward(self, x, y, set_to_none=True):
        model = self.model
        self.optimizer_t.zero_grad(set_to_none=set_to_none)
        if self.optimizer_admm is not None:
            self.optimizer_admm.zero_grad(set_to_none=set_to_none)
        if self.channels_last and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)      # no-op for the captured static input
        prequantize_weights(self.all_convs)     # all conv weights in two launches
        fused_head = self.channels_last and hasattr(model, "logit") and hasattr(model, "avgpool")
        model._features_only = fused_head
        ce = None
        try:
            if self._deferred is not None and self.admms:
                with self._deferred as d:
                    out = model(x)
                    logits = out[0] if isinstance(out, tuple) else out
                    if fused_head and d.can_fuse_head() and head_ce_supported(logits, model.logit.weight, y):
                        # both loss roots from one node: no reduction launches, head backward + site preparation together
                        logits, ce, trans_loss = d.total_with_head(logits, model.logit.weight, model.logit.bias, y)
                    else:
                        trans_loss = d.total()          # joins the side stream, one stacked sum
            else:
                out = model(x)
                if isinstance(out, tuple):
                    logits, trans_loss = out
                else:
                    logits, trans_loss = out, None
        finally:
            model._features_only = False
        if ce is not None:
            pass
        elif fused_head and head_ce_supported(logits, model.logit.weight, y):
            # `logits` still holds the pre-pool features: pool + linear + cross-entropy in one launch each way
            logits, ce = HeadCEFn.apply(logits, model.logit.weight, model.logit.bias, y)
        else:
            if fused_head:       # unsupported shape: finish the head the plain way
                feats = logits
                logits = model.logit(model.avgpool(feats).view(feats.size(0), -1))
            ce = F.cross_entropy(logits, y)
        # d(ce + trans_loss) = 1 * d(ce) + 1 * d(trans_loss): two roots with a persistent unit gradient instead of forming the
        # sum (one add and one ones_like fill fewer on the in-order chain; main.py:300-304 only needs the gradients of the sum)
        roots = [ce] if (trans_loss is None or not torch.is_tensor(trans_loss) or not trans_loss.requires_grad) \
            else [ce, trans_loss]
        if self._one is None or self._one.device != ce.device:
            self._one = torch.ones((), dtype=torch.float32, device=ce.device)
        if self._wgrads is not None and set_to_none:
            with self._wgrads as wg:          # all filter-gradient slab reductions in one launch after the backward
                torch.autograd.backward(roots, [self._one] * len(roots))
                wg.flush()
        else:
            torch.autograd.backward(roots, [self._one] * len(roots))
        return logits, ce, trans_loss

    def _optimizer_steps(self):
        if config.args.bitW < 32 and self.admms:
            w_cdf = [c.quantize_fn.weight_cdf for c in self.convs]
            w_pdf = [c.quantize_fn.weight_pdf for c in self.convs]
            a = self.admms
            # the two steps of main.py:330-340 as roles of one launch (disjoint parameters: see optimizer.sgd_admm_step)
            sgd_admm_step(self.optimizer_t, (self.idx, w_cdf, w_pdf, config.args.lam, config.args.lam2), self.optimizer_admm,
                          (self.alterD_idx, self.gamma_idx, [m.D for m in a], [m.alterD for m in a], [m.gamma for m in a],
                           [m.mu for m in a], [m.rho for m in a]))
        else:
            # CDF-only tree (main.py:308 crashes as shipped): plain momentum SGD, no grad rewrite
            self.optimizer_t.step([], [], [], config.args.lam, config.args.lam2)

    def _iteration(self, x, y, set_to_none=True):
        outs = self._forward_backward(x, y, set_to_none)
        if self.grad_hook is not None:
            self.grad_hook(self)
        self._optimizer_steps()
        return _detached(outs)

    def __call__(self, x, y):
        if self._graph is None:
            return self._iteration(x, y)
        sx, sy = self._static[0], self._static[1]
        if x.shape != sx.shape or y.shape != sy.shape:
            # off-shape batch (the reference's loaders have no drop_last: CIFAR's last batch is 80 of 128): a captured
            # graph only fits its static shapes, so this one iteration runs eagerly; the graph's own gradient / D tensors
            # (which p.grad and ADMM.D name between replays) are put back afterwards
            return self._eager_fallback(x, y)
        # a loader that writes its batches straight into `static_inputs()` (the target of its host-to-device copy) hands the
        # same tensors back: nothing to stage
        if x is not sx:
            sx.copy_(x, non_blocking=True)
        if y is not sy:
            sy.copy_(y, non_blocking=True)
        self._graph.replay()
        if self._graph2 is not None:
            # data-parallel: only the collective runs eagerly between the two captured halves (pack / unpack of the flat
            # bucket are captured at the end of the first and the start of the second graph)
            self.grad_hook.reduce() if hasattr(self.grad_hook, "reduce") else self.grad_hook(self)
            self._graph2.replay()
        return self._static[2]

    def static_inputs(self):
        """(x, y) buffers the captured graph reads (x in the step's memory format).  Fill them in place — e.g. as the
        destination of the loader's host-to-device copy — and call the step with these very tensors: the per-step staging
        copies (a layout-converting one for NCHW input) are skipped.  None before `capture`."""
        return None if self._static is None else (self._static[0], self._static[1])

    def _eager_fallback(self, x, y):
        params = [p for _, p in self.param_t + self.param_admm]
        keep_g, keep_D = [p.grad for p in params], [m.D for m in self.admms]
        # a data-parallel hook rebinds its flat bucket to the off-shape layout during this iteration; the graphs keep packing
        # into / unpacking from the bucket that was current at capture, so the eager reduce() between them must see that one
        # again afterwards (otherwise it all-reduces the stale short-batch buffer and the replicas diverge silently)
        hook = self.grad_hook
        keep_hook = hook.snapshot() if hasattr(hook, "snapshot") else None
        try:
            return self._iteration(x, y)
        finally:
            for p, g in zip(params, keep_g):
                p.grad = g
            for m, D in zip(self.admms, keep_D):
                m.D = D
            if hasattr(hook, "restore"):
                hook.restore(keep_hook)

    def _assert_momentum_buffers(self):
        """A momentum buffer created INSIDE a capture is baked into the graph with first=1 (buf = grad on every replay:
        silently momentum-free training) and lives in the graph's private pool.  Capture therefore needs every parameter the
        step updates to own its buffer already, i.e. at least one eager step since the optimizer was created."""
        for group in self.optimizer_t.param_groups:
            if group["momentum"] == 0:
                continue
            for p in group["params"]:
                if p.requires_grad and "momentum_buffer" not in self.optimizer_t.state[p]:
                    raise RuntimeError("TrainStep.capture: a parameter has no momentum buffer yet; run at least one eager "
                                       "iteration (capture(..., warmup>=1)) before capturing the step")

    # -------------------------------------------------------------------------------------------
    def set_lr(self, lr):
        """The reference steps a StepLR scheduler once per epoch (main.py:112,150).  Scalar hyper-parameters are baked into
        a captured graph as kernel arguments, so a captured step is re-captured (no warm-up iterations: the model is not
        touched) whenever the learning rate changes."""
        changed = False
        for g in self.optimizer_t.param_groups:
            changed |= g["lr"] != lr
            g["lr"] = lr
        if changed and self._graph is not None:
            self.capture(self._static[0], self._static[1], warmup=0)
        return self

    def capture(self, x, y, warmup=3):
        """Capture the iteration into HIP graphs.  Runs `warmup` eager iterations first (they DO update
        the model, as real steps) so allocator pools, momentum buffers, pointer tables and MIOpen plans
        exist.  Without a grad_hook the whole iteration is ONE graph; with one (data parallel) it is two
        graphs (forward+backward | optimizer steps) with the all-reduce launched eagerly in between."""
        # refusals first: nothing below may touch the model before them (the warm-up iterations are real steps)
        if getattr(self, "_global_corr_undo", None) is not None or getattr(config.args, "global_corr", None) is not None:
            # The exact-global correlation issues all_to_all / all_reduce from INSIDE the forward and, through
            # dp._FeatureShard.backward, from inside the ONE autograd.backward call of the step: the capture region cannot be cut at
            # them (the phased form of the gradient bucket cuts BETWEEN backward and optimizer).  Capturing the collectives
            # themselves is only defined for a process group whose asynchronous error handling (its watchdog's event polling) was
            # switched off BEFORE init_process_group (TORCH_NCCL_ASYNC_ERROR_HANDLING=0, torch's own rule for whole-network
            # capture with collectives): a property of the process this library cannot change for an existing group - with it on,
            # the round-4 capture never returned.  The mode therefore runs eagerly, by design.
            raise RuntimeError("TrainStep.capture: the exact-global correlation (global_corr) is eager-only: its collectives sit "
                               "inside the forward and inside autograd's backward call, where the capture cannot be split, and "
                               "captured collectives need a process group created with TORCH_NCCL_ASYNC_ERROR_HANDLING=0 "
                               "(DESIGN.md section 6) - call the step eagerly")
        release_step_graphs(self.admms)
        assert_no_retained_graph([p for g in self.optimizer_t.param_groups for p in g["params"]], "TrainStep.capture")
        sx = x.clone(memory_format=torch.channels_last) if (self.channels_last and x.dim() == 4) else x.clone()
        sy = y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._iteration(sx, sy, set_to_none=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._assert_momentum_buffers()
        release_step_graphs(self.admms)          # (an unfused site of the warm-up keeps D with its graph)
        # An initialised process group matters even WITHOUT a hook: its watchdog thread polls HIP events of earlier collectives,
        # which the default (global) capture mode forbids while ANY stream captures; under capture_error_mode="thread_local" HIP
        # calls of other threads neither fail nor invalidate the capture.
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        if dist_on and self.grad_hook is not None:
            # The warm-up iterations issued collectives: a barrier orders every rank behind its peers' warm-up collectives and the
            # device synchronise retires them, so none is in flight when the capture starts.  Only on the all-ranks data-parallel
            # path: a step without a hook (after dp.detach, a rank-local evaluation step) must not meet its peers here.
            torch.distributed.barrier()
            torch.cuda.synchronize()
        cap_mode = dict(capture_error_mode="thread_local") if dist_on else {}
        graph = torch.cuda.CUDAGraph()
        self._graph2 = None
        # inside the capture grads are re-created (set_to_none=True): no zero-fill and no accumulate-add per
        # parameter; the fresh grad tensors live in the graph's private pool at fixed addresses and stay
        # referenced by p.grad, so the captured optimizer kernels (and the eager all-reduce) see them on replay.
        self.optimizer_t.zero_grad(set_to_none=True)
        if self.optimizer_admm is not None:
            self.optimizer_admm.zero_grad(set_to_none=True)
        if self.grad_hook is None:
            with torch.cuda.graph(graph, **cap_mode):
                outs = self._iteration(sx, sy, set_to_none=True)
        else:
            phased = hasattr(self.grad_hook, "reduce")
            with torch.cuda.graph(graph, **cap_mode):
                outs = self._forward_backward(sx, sy, set_to_none=True)
                if phased:
                    self.grad_hook.pack()
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2, pool=graph.pool(), **cap_mode):
                if phased:
                    self.grad_hook.unpack()
                self._optimizer_steps()
            self._graph2 = graph2
        self._graph = graph
        self._static = (sx, sy, _detached(outs))
        return self


class OfficeTrainStep:
    """One DANN iteration of the Office tree (cdf_alignment_admm/dann_office/main.py:343-456): zero_grad x2 -> source pass
    -> target pass -> src class CE + src/tgt domain CE + both trans losses -> backward -> SGD.step(idx, w_cdf, w_pdf, lam,
    lam2) over the three parameter groups (feature incl. alterD/gamma, class head, domain head: main.py:324-328) ->
    ADMM_OPT.step.  Each ADMM.D holds the TARGET pass's D when ADMM_OPT runs (admm.py:25 overwrites) and alterD/gamma are
    SGD-stepped first and then overwritten by the closed form, exactly like the reference."""

    def __init__(self, model, lr=0.04, momentum=0.9, weight_decay=5e-4, alpha=0.5, channels_last=False, fuse_relu=True,
                 grad_hook=None, fuse_bn=True, dual=None, qconv=True, pack_bins=True):
        """grad_hook: the data-parallel all-reduce (alignq_amd.dp.attach_office -> BucketedGradAllReduce): begin() right
        before backward, its buckets' collectives start from autograd hooks while the backward runs, finish() before the
        optimizer steps.
        channels_last: activations and conv weights in torch.channels_last memory (values / names unchanged): MIOpen's NHWC
        kernels run the ResNet-50 step in 30.9 instead of 34.7 ms on MI355X; the quantise / Gram kernels are layout-agnostic.
        fuse_relu: `relu(act_q(.))` of the stem and of each bottleneck's first two sites as one launch each way.
        fuse_bn (channels_last only): additionally the training-mode batch-norm in front of those quantisers is folded into
        them (fused.bn_act_relu: statistics + one elementwise pass; SURVEY.md 8f-N1 on configuration 5).
        dual (default: on with fuse_bn and channels_last): the source and the target pass as ONE traversal (DANN.forward_dual):
        the per-sample convolutions see both batches at once, batch statistics / sites / correlations stay per domain in pass
        order; every parameter then has one incoming gradient (no accumulation kernels) and the weights are quantised once."""
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
        self.channels_last = channels_last
        # qconv (channels_last only): Conv2d_Q's 1x1 / 3x3 convolutions on alignq_qconv_* (exact-product GEMMs on the bf16 / f16
        # matrix cores, csrc/qgemm_kernels.hip) instead of MIOpen's fp32 kernels (the 7x7 stem included: alignq_qconv_stem7_*).  Their filter
        # gradients leave split-K slabs that one reduction launch per 32 filters finishes (fused.DeferredWgrads).
        self.qconv = bool(qconv and channels_last and torch.cuda.is_available())
        for mod in model.modules():
            if hasattr(mod, "quantize_fn"):
                mod.use_qconv = self.qconv
                # every Conv2d_Q of this network is followed by a batch-norm; folded (fuse_bn) it reads the convolution's
                # per-tile statistics instead of making a pass of its own over the output
                mod.emit_bn_stats = bool(self.qconv and fuse_bn and fuse_relu)
        self._wgrads = DeferredWgrads() if self.qconv else None
        # pack_bins (N2; with qconv and fuse_bn): relu(act_q1(bn1(.))) and relu(act_q2(bn2(.))) of every bottleneck feed
        # conv2 / conv3 only - they are stored as int16 level indices (no fp32 copy): the quantiser writes, the convolution's
        # forward and filter gradient read 2 B per element instead of 4
        for mod in model.modules():
            if hasattr(mod, "act_q1") and hasattr(mod, "act_q2") and hasattr(mod, "act_q3"):
                mod.pack_bins = bool(pack_bins and self.qconv and fuse_bn and fuse_relu)
        for mod in model.modules():
            if hasattr(mod, "act_q0") or (hasattr(mod, "act_q1") and hasattr(mod, "act_q2") and hasattr(mod, "act_q3")):
                mod.fuse_relu = bool(fuse_relu)
                mod.fuse_bn = bool(fuse_bn and fuse_relu and channels_last)
        self.dual = bool(fuse_bn and fuse_relu and channels_last) if dual is None else bool(dual)
        self.model, self.alpha = model, alpha
        named = list(model.named_parameters())
        self.param_admm = [(n, p) for n, p in named if "alterD" in n or "gamma" in n]
        self.optimizer_t = SGD([{"params": list(model.feature.parameters())},
                                {"params": list(model.class_classifier.parameters()), "lr": lr},
                                {"params": list(model.domain_classifier.parameters()), "lr": lr}],
                               lr=lr / 10, momentum=momentum, weight_decay=weight_decay)
        self.optimizer_admm = ADMM_OPT([p for _, p in self.param_admm])
        # main.py:405-410 (param_t there lists ALL named parameters, so j indexes feature.parameters())
        self.idx = [j for j, (n, _) in enumerate(named) if ("conv" in n or "downsample.0" in n) and "weight" in n][1:]
        self.alterD_idx = [j for j, (n, _) in enumerate(self.param_admm) if "alterD" in n]
        self.gamma_idx = [j for j, (n, _) in enumerate(self.param_admm) if "gamma" in n]
        f = model.feature
        self.blocks = [b for layer in (f.layer1, f.layer2, f.layer3, f.layer4) for b in layer]
        self.convs = []
        for b in self.blocks:
            for k, conv in enumerate((b.conv1, b.conv2, b.conv3, b.downsample)):
                if conv is not None:
                    self.convs.append(conv[0] if k == 3 else conv)
        self.all_convs = [m for m in model.modules() if hasattr(m, "quantize_fn")]
        self.grad_hook = grad_hook
        self._staged = False
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph2: Optional[torch.cuda.CUDAGraph] = None
        self._static = None

    def stage_weights(self, on=True):
        """Quantise the conv weights per ResNet stage, each stage right before its forward (ResNet.forward calls back), instead of
        all of them before the first layer.  Same per-tensor arithmetic (the multi-tensor kernels treat every filter alone), 5 x 2
        launches each way instead of 2.  For data parallelism: a stage's weight gradients then leave the weight quantiser's
        backward when the backward passes that stage, so their buckets' all-reduces overlap the earlier stages' backward
        (dp.attach_office switches it on)."""
        f = self.model.feature
        if on:
            layers = (f.layer1, f.layer2, f.layer3, f.layer4)
            owner = {id(m): i + 1 for i, layer in enumerate(layers) for m in layer.modules() if hasattr(m, "quantize_fn")}
            stages = [[c for c in self.all_convs if owner.get(id(c), 0) == i] for i in range(len(layers) + 1)]
            f._wq_stage = lambda i: prequantize_weights(stages[i], pack=self.qconv)
        else:
            f._wq_stage = None
        self._staged = bool(on)
        return self

    def _iteration(self, xs, ys, xt, set_to_none=True):
        out = self._forward_backward(xs, ys, xt, set_to_none, overlap=True)
        self._optimizer_steps()
        return _detached(out)

    def _forward_backward(self, xs, ys, xt, set_to_none=True, overlap=False):
        m = self.model
        self.optimizer_t.zero_grad(set_to_none=set_to_none)
        self.optimizer_admm.zero_grad(set_to_none=set_to_none)
        dev = xs.device
        if self.channels_last:
            xs = xs.contiguous(memory_format=torch.channels_last)
            xt = xt.contiguous(memory_format=torch.channels_last)
        # the domain labels (main.py:360-361 builds them every iteration) are constants of the batch sizes: kept, not re-filled
        key = (int(xs.shape[0]), int(xt.shape[0]), dev)
        if getattr(self, "_dom_labels_key", None) != key:
            self._dom_labels = (torch.zeros(xs.shape[0], dtype=torch.long, device=dev),
                                torch.ones(xt.shape[0], dtype=torch.long, device=dev))
            self._dom_labels_key = key
        label_src, label_tgt = self._dom_labels
        if not self._staged:               # (staged: ResNet.forward quantises each stage's weights right before the stage)
            prequantize_weights(self.all_convs, pack=self.qconv)
        if self.dual and xs.shape == xt.shape:
            cls_s, dom_s, dom_t, tl_both = m.forward_dual(xs, xt, alpha=self.alpha)      # (same weights, hence the same W_q,
            tl_s, tl_t = tl_both, 0.0                                                    #  in both of the reference's passes)
        else:
            cls_s, dom_s, tl_s = m(xs, alpha=self.alpha)
            if not self._staged:
                prequantize_weights(self.all_convs, pack=self.qconv)      # the reference quantises every weight once per pass
            _, dom_t, tl_t = m(xt, alpha=self.alpha)
        # (a pass without a loss tensor contributes the NUMBER 0: adding it would be an elementwise launch of its own)
        tl = tl_s if not torch.is_tensor(tl_t) and tl_t == 0 else (tl_t if not torch.is_tensor(tl_s) and tl_s == 0 else tl_s + tl_t)
        loss = F.cross_entropy(cls_s, ys) + F.cross_entropy(dom_s, label_src) + F.cross_entropy(dom_t, label_tgt) + tl
        hook = self.grad_hook if overlap else None
        if hook is not None:
            hook.begin()              # the buckets' all-reduces start from autograd hooks during this backward
        if self._wgrads is not None:
            with self._wgrads as wg:  # (the weight quantiser's backward finishes the slab reductions before it reads them)
                loss.backward()
                wg.flush()
        else:
            loss.backward()
        if hook is not None:
            hook.finish()
        return cls_s, loss, tl

    def _optimizer_steps(self):
        w_cdf = [c.quantize_fn.weight_cdf for c in self.convs]
        w_pdf = [c.quantize_fn.weight_pdf for c in self.convs]
        self.optimizer_t.step(self.idx, w_cdf, w_pdf, config.args.lam, config.args.lam2)
        a = [b.admm0 for b in self.blocks]
        self.optimizer_admm.step(self.alterD_idx, self.gamma_idx, [q.D for q in a], [q.alterD for q in a],
                                 [q.gamma for q in a], [q.mu for q in a], [q.rho for q in a])

    def new_epoch(self, epoch, num_epochs, lr, momentum=0.9, weight_decay=5e-4):
        """dann_office/main.py:321-328: every epoch the reference builds a NEW SGD (so momentum buffers start from scratch)
        with LEARNING_RATE = lr / (1 + 10 (epoch-1) / num_epochs)^0.75 for the two heads and a tenth of it for the feature
        extractor.  A captured step is re-captured (graph kernel arguments hold the learning rates)."""
        rate = lr / (1.0 + 10.0 * (epoch - 1) / num_epochs) ** 0.75
        m = self.model
        self.optimizer_t = SGD([{"params": list(m.feature.parameters())},
                                {"params": list(m.class_classifier.parameters()), "lr": rate},
                                {"params": list(m.domain_classifier.parameters()), "lr": rate}],
                               lr=rate / 10, momentum=momentum, weight_decay=weight_decay)
        if self._graph is not None:
            # the first step of the epoch creates the fresh momentum buffers (buf = grad) and must not be the captured one
            self._graph, self._recapture = None, True
        return rate

    def __call__(self, xs, ys, xt):
        if self._graph is None:
            out = self._iteration(xs, ys, xt)
            if getattr(self, "_recapture", False):
                self._recapture = False
                self.capture(xs, ys, xt, warmup=0)
            return out
        if any(src.shape != dst.shape for dst, src in zip(self._static[:3], (xs, ys, xt))):
            return self._eager_fallback(xs, ys, xt)       # off-shape (short last) batch: one eager iteration
        for dst, src in zip(self._static[:3], (xs, ys, xt)):
            dst.copy_(src, non_blocking=True)
        self._graph.replay()
        if self._graph2 is not None:       # data parallel: the buckets' collectives run eagerly between the two graphs
            self.grad_hook.reduce()
            self._graph2.replay()
        return self._static[3]

    def _eager_fallback(self, xs, ys, xt):
        params = [p for g in self.optimizer_t.param_groups for p in g["params"]]
        admms = [b.admm0 for b in self.blocks]
        keep_g, keep_D = [p.grad for p in params], [q.D for q in admms]
        hook = self.grad_hook
        keep_hook = hook.snapshot() if hasattr(hook, "snapshot") else None      # see TrainStep._eager_fallback
        try:
            return self._iteration(xs, ys, xt)
        finally:
            for p, g in zip(params, keep_g):
                p.grad = g
            for q, D in zip(admms, keep_D):
                q.D = D
            if hasattr(hook, "restore"):
                hook.restore(keep_hook)

    def capture(self, xs, ys, xt, warmup=2):
        admms = [b.admm0 for b in self.blocks]
        release_step_graphs(admms)
        assert_no_retained_graph([p for g in self.optimizer_t.param_groups for p in g["params"]], "OfficeTrainStep.capture")
        fmt = torch.channels_last if self.channels_last else torch.contiguous_format
        sxs, sys_, sxt = xs.clone(memory_format=fmt), ys.clone(), xt.clone(memory_format=fmt)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._iteration(sxs, sys_, sxt, set_to_none=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for group in self.optimizer_t.param_groups:      # see TrainStep._assert_momentum_buffers
            for p in group["params"]:
                if group["momentum"] != 0 and p.grad is not None and "momentum_buffer" not in self.optimizer_t.state[p]:
                    raise RuntimeError("OfficeTrainStep.capture: a parameter has no momentum buffer yet; run at least one "
                                       "eager iteration (capture(..., warmup>=1)) before capturing the step")
        release_step_graphs(admms)
        self.optimizer_t.zero_grad(set_to_none=True)
        self.optimizer_admm.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        self._graph2 = None
        if self.grad_hook is None or not self.grad_hook.active():
            # (an initialised process group's watchdog thread issues HIP calls of its own: thread-local capture mode, see TrainStep.capture)
            pg_on = torch.distributed.is_available() and torch.distributed.is_initialized()
            with torch.cuda.graph(graph, **(dict(capture_error_mode="thread_local") if pg_on else {})):
                outs = self._iteration(sxs, sys_, sxt, set_to_none=True)
        else:
            # data parallel: forward + backward (+ each bucket packed where its last gradient lands) | eager all-reduces, started
            # per bucket by a flag the replayed graph publishes | bucket unpack + optimizer steps
            torch.distributed.barrier()
            torch.cuda.synchronize()
            mode = dict(capture_error_mode="thread_local")
            overlapped = hasattr(self.grad_hook, "capture_begin")
            with torch.cuda.graph(graph, **mode):
                if overlapped:
                    # the hooks pack each bucket and publish its flag where the captured backward completes it; reduce() then
                    # starts bucket i's all-reduce beside the rest of the replayed backward (dp.BucketedGradAllReduce)
                    self.grad_hook.capture_begin()
                    outs = self._forward_backward(sxs, sys_, sxt, set_to_none=True, overlap=False)
                    self.grad_hook.capture_end()
                else:
                    outs = self._forward_backward(sxs, sys_, sxt, set_to_none=True, overlap=False)
                    self.grad_hook.pack()
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2, pool=graph.pool(), **mode):
                self.grad_hook.unpack()
                self._optimizer_steps()
            self._graph2 = graph2
        self._graph = graph
        self._static = (sxs, sys_, sxt, _detached(outs))
        return self
