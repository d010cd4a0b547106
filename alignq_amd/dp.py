"""Data-parallel training of the hot path over the GPUs of one node (one process per GPU,
torch.distributed backend "nccl" == RCCL over xGMI on ROCm).

Semantics: every rank runs the reference semantics at its local batch b (local BN
statistics, local [b,b] correlation matrices, ADMM(dim=b) replicated).  Per step there is ONE collective:
a mean all-reduce of a single flat fp32 bucket holding (1) the gradients of all non-ADMM parameters and
(2) the stacked D matrices of all sites, so that SGD.step and ADMM_OPT.step see identical inputs on every
rank and the replicas stay bit-identical.  ResNet-20: 1.1 MB + 1.4 MB = 2.5 MB -> latency-bound on xGMI,
hence one bucket, not one collective per tensor.  The reference itself has no distributed code (F1)."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


class FlatBucket:
    """A persistent flat buffer with views: tensors are packed / unpacked with one fused copy each way."""

    def __init__(self, shapes: Sequence[torch.Size], device, dtype=torch.float32):
        self.shapes = list(shapes)
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        self.flat = torch.zeros(sum(self.numels), dtype=dtype, device=device)
        self.views, off = [], 0
        for s, n in zip(self.shapes, self.numels):
            self.views.append(self.flat[off:off + n].view(s))
            off += n

    @staticmethod
    def _dense(t):
        return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))

    def matches(self, tensors) -> bool:
        """True when `tensors` are exactly the tensors this bucket was laid out for (count and element counts)."""
        return len(tensors) == len(self.numels) and all(int(t.numel()) == n for t, n in zip(tensors, self.numels))

    def _check(self, tensors):
        # the native copy kernel takes the element counts cached at construction and cannot see the tensors' real sizes:
        # a short last batch (D is [b',b'] instead of [b,b]) or a changed set of non-None gradients must never reach it
        if not self.matches(tensors):
            raise RuntimeError(
                "FlatBucket: tensor list changed since the bucket was laid out "
                f"({len(tensors)} tensors / {sum(int(t.numel()) for t in tensors)} elements now, "
                f"{len(self.numels)} / {sum(self.numels)} at construction); rebuild the bucket")

    def _native(self, tensors, unpack):
        """One HIP launch per 128 tensors (alignq_bucket_copy_multi): dense CUDA fp32 tensors are copied in storage order."""
        from . import _lib as L
        L.check(L.load().alignq_bucket_copy_multi(len(tensors), L.ptr_array(tensors), L.i64_array(self.numels),
                                                  L.ptr(self.flat), int(unpack), L.stream_ptr()), "alignq_bucket_copy_multi")

    def pack(self, tensors: Sequence[torch.Tensor]):
        self._check(tensors)
        if self.flat.is_cuda and all(t.is_cuda and t.dtype == torch.float32 and self._dense(t) for t in tensors):
            return self._native(tensors, False)
        torch._foreach_copy_(self.views, [t.detach() for t in tensors])

    def unpack(self, tensors: Sequence[torch.Tensor]):
        self._check(tensors)
        if self.flat.is_cuda and all(t.is_cuda and t.dtype == torch.float32 and self._dense(t) for t in tensors):
            return self._native(tensors, True)
        torch._foreach_copy_([t.detach() for t in tensors], self.views)


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank `src`'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class GradAndDAllReduce:
    """grad_hook for alignq_amd.train_step.TrainStep (called between backward and the optimizer steps)."""

    def __init__(self, params: List[torch.nn.Parameter], get_Ds: Callable[[], List[torch.Tensor]], group=None,
                 force: bool = False):
        self.params = params
        self.get_Ds = get_Ds
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()      # run the collective even at world size 1 (self-test)
        self.bucket = None
        self._buckets = {}          # layout (tuple of shapes) -> FlatBucket: the short last batch gets its own bucket

    # The hook is three phases so that a captured TrainStep can keep pack / unpack INSIDE its two HIP graphs and leave only
    # the collective itself eager (one RCCL launch between two graph launches per step).
    def active(self) -> bool:
        return self.world > 1 or self.force

    def _tensors(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        Ds = [d for d in self.get_Ds() if d is not None]
        return grads + Ds

    def pack(self):
        if not self.active():
            return
        tensors = self._tensors()
        if self.bucket is None or not self.bucket.matches(tensors):
            # The reference's loaders have no drop_last: the last batch of an epoch is short (D is [b',b']), and which
            # gradients are None may change.  Every rank sees the same shapes (same sampler length), so each layout gets
            # its own persistent bucket; a captured step never comes here with a new layout (static shapes).
            key = tuple(tuple(t.shape) for t in tensors)
            if key not in self._buckets:
                self._buckets[key] = FlatBucket([t.shape for t in tensors], tensors[0].device)
            self.bucket = self._buckets[key]
        self.bucket.pack(tensors)

    def reduce(self):
        if not self.active():
            return
        backend = dist.get_backend(self.group)
        if backend == "nccl":        # RCCL averages in the collective itself: no separate scaling kernel
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.bucket.flat.mul_(1.0 / self.world)

    def unpack(self):
        if not self.active():
            return
        self.bucket.unpack(self._tensors())

    def __call__(self, _step=None):
        self.pack()
        self.reduce()
        self.unpack()


def attach(train_step, group=None, force=False):
    """Wire data parallelism into a TrainStep: broadcast the initial state, install the all-reduce hook."""
    broadcast_module_state(train_step.model, 0, group)
    hook = GradAndDAllReduce([p for _, p in train_step.param_t], lambda: [m.D for m in train_step.admms], group,
                             force=force)
    train_step.grad_hook = hook
    return hook
