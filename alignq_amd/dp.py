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
        """True when `tensors` are exactly the tensors this bucket was laid out for (count and shapes)."""
        return len(tensors) == len(self.shapes) and all(tuple(t.shape) == tuple(s) for t, s in zip(tensors, self.shapes))

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

    # A captured TrainStep bakes the CURRENT bucket's flat buffer into its two graphs (pack at the end of the first, unpack at
    # the start of the second) and calls reduce() eagerly in between.  An eager iteration on another layout (the short last
    # batch) rebinds self.bucket; the step saves / restores the binding around it so that reduce() keeps all-reducing the
    # buffer the graphs pack into.
    def snapshot(self):
        return self.bucket

    def restore(self, state):
        self.bucket = state


def attach(train_step, group=None, force=False, global_corr=False):
    """Wire data parallelism into a TrainStep: broadcast the initial state, install the all-reduce hook.
    global_corr=True (opt-in): every ADMM site computes the correlation pair of the GLOBAL batch
    (global_corr below; the model's ADMM(dim) must have been built with the global batch size, <= 1024: above 128 rows the blocked Gram of corr_large_kernels.hip) instead of the
    per-rank [b,b] matrices; BN fold and deferred site launches are switched off for it (the sites run unfused)."""
    if global_corr:
        # scoped to THIS model's quantiser modules (not the process-global config): other models / steps in the process keep
        # the per-rank semantics; detach() restores what is changed here
        undo = {"fuse_bn": [], "sites": [], "deferred": train_step._deferred}
        for m in train_step.model.modules():
            if hasattr(m, "fuse_bn"):
                undo["fuse_bn"].append((m, m.fuse_bn))
                m.fuse_bn = False
            if hasattr(m, "a_bit") and hasattr(m, "opt"):
                undo["sites"].append(m)
                m.global_corr = True if group is None else group
        train_step._deferred = None
        train_step._global_corr_undo = undo
    broadcast_module_state(train_step.model, 0, group)
    hook = GradAndDAllReduce([p for _, p in train_step.param_t], lambda: [m.D for m in train_step.admms], group,
                             force=force)
    train_step.grad_hook = hook
    return hook


def detach(train_step):
    """Undo attach(): remove the hook and, after global_corr=True, give the sites back their per-rank correlation, the BN
    fold and the deferred launches."""
    train_step.grad_hook = None
    undo = getattr(train_step, "_global_corr_undo", None)
    if undo is not None:
        for m, v in undo["fuse_bn"]:
            m.fuse_bn = v
        for m in undo["sites"]:
            m.global_corr = None
            # the exact-global sites stored D WITH its autograd graph (ADMM.forward, utils/admm.py:25): dropped here, or the last
            # eager iteration's graph - and the gradient-accumulation nodes bound to its stream - would outlive the mode
            # (train_step.retained_graph_params: what crashed a later capture in round 4)
            if getattr(m.opt, "D", None) is not None and m.opt.D.grad_fn is not None:
                m.opt.D = m.opt.D.detach()
        train_step._deferred = undo["deferred"]
        train_step._global_corr_undo = None


class BucketedGradAllReduce:
    """Data parallelism for the Office / DANN step (BASELINE config 5: ResNet-50, ~94 MB of fp32 gradients per step, ring
    time ~1.1 ms per link-bound ring on xGMI): the gradients are all-reduced in >= `min_buckets` flat
    buckets (<= `bucket_bytes` each; a single larger tensor is a bucket of its own), launched FROM AUTOGRAD HOOKS while the backward is still running, so the collectives of
    the deep layers overlap the backward of the shallow ones; the stacked D matrices of the ADMM sites ride in the bucket that
    completes last (the stem's).  Every rank runs the reference semantics at its local batch; after `finish()` every
    p.grad and every ADMM.D holds the mean over ranks, so SGD.step / ADMM_OPT.step keep the replicas bit-identical.

    Order of the buckets = reverse registration order of the parameters (the order the backward produces gradients in).  The
    set of parameters that actually receive gradients is discovered in the first iteration (DANN never uses feature.fc), which
    therefore reduces without overlap.  Interface: begin() before backward, finish() after it; or the phased pack() /
    reduce() / unpack() of GradAndDAllReduce for a captured step (two HIP graphs with the collectives eager in between).

    Captured step WITH overlap (round 6): capture_begin() / capture_end() bracket the captured forward + backward.  The autograd
    hooks then only PACK a completed bucket into its flat buffer and publish a flag behind it (alignq_dp_flag_publish: nodes of
    the graph); after launching that graph, reduce() enqueues on the communication stream, per bucket in completion order,
    alignq_dp_stream_wait_ge(flag_i >= replay number) + the eager all-reduce - bucket i's collective starts when the replayed
    backward has packed it and runs beside the rest of the backward.  (A collective cannot be captured here, DESIGN.md section 6;
    torch refuses external events on ROCm; tools/src/probe_waitvalue.hip is the stand-alone probe of the mechanism.)"""

    def __init__(self, params: List[torch.nn.Parameter], get_Ds: Callable[[], List[torch.Tensor]], group=None,
                 force: bool = False, bucket_bytes: int = 24 << 20, min_buckets: int = 4):
        self.params = [p for p in params if p.requires_grad]
        self.get_Ds = get_Ds
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()
        self.bucket_bytes, self.min_buckets = int(bucket_bytes), int(min_buckets)
        self._live = None            # parameters that receive gradients (discovered), in bucket order
        self._groups = None          # list of lists of parameters
        self._where = {}             # id(p) -> bucket index
        self._layouts = {}           # (bucket index, D shapes) -> FlatBucket
        self._armed = False
        self._pending, self._works, self._buckets_now = [], [], []
        self._comm = None
        self.launched_from_hooks = 0          # diagnostics: buckets whose collective started inside the backward
        self._cap = False                     # inside capture_begin() .. capture_end(): hooks pack + publish, no collective
        self._sync = None                     # device words: [0] replay counter, [1 + i] flag of bucket i
        self._replays = 0                     # host mirror of the replay counter
        self._overlap_ready = False           # the current `_phase` was laid out by a captured backward (flags exist)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def active(self) -> bool:
        return self.world > 1 or self.force

    # ---- layout ---------------------------------------------------------------------------------------------------
    def _discover(self):
        live = [p for p in reversed(self.params) if p.grad is not None]
        total = sum(p.numel() for p in live) * 4
        target = max(1, min(self.bucket_bytes, -(-total // self.min_buckets)))
        # a bucket is closed BEFORE the tensor that would take it past the target (round 6: closing after it let a 9.4 MB filter
        # push a bucket to 30 MiB against bucket_bytes = 24 MiB); only a single tensor larger than the target exceeds it, alone
        groups, cur, size = [], [], 0
        for p in live:
            sz = p.numel() * 4
            if cur and size + sz > target:
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += sz
        if cur:
            groups.append(cur)
        self._live, self._groups = live, groups
        self._where = {id(p): bi for bi, g in enumerate(groups) for p in g}
        self._layouts = {}
        # the replay counter and the buckets' flags of the captured form: allocated HERE, in an eager iteration - inside the capture
        # the allocation would come from the graph's pool and its zero-fill would be a node that resets them on every replay
        dev = live[0].device if live else None
        self._sync = torch.zeros(1 + len(groups), dtype=torch.int32, device=dev) if dev is not None and dev.type == "cuda" else None
        self._replays = 0

    def _tensors(self, bi):
        ts = [p.grad for p in self._groups[bi]]
        if any(t is None for t in ts):
            raise RuntimeError("BucketedGradAllReduce: a parameter that had a gradient in the first iteration has none now; "
                               "call reset() when the set of trained parameters changes")
        if bi == len(self._groups) - 1:
            ts = ts + [d for d in self.get_Ds() if d is not None]
        return ts

    def _bucket(self, bi, tensors):
        key = (bi, tuple(tuple(t.shape) for t in tensors[len(self._groups[bi]):]))     # D shapes: the short last batch
        b = self._layouts.get(key)
        if b is None or not b.matches(tensors):
            b = self._layouts[key] = FlatBucket([t.shape for t in tensors], tensors[0].device)
        return b

    def reset(self):
        self._live = self._groups = None

    # ---- one bucket: pack on the compute stream, all-reduce on the communication stream --------------------------
    def _launch(self, bi, overlap=True):
        tensors = self._tensors(bi)
        b = self._bucket(bi, tensors)
        b.pack(tensors)
        self._buckets_now[bi] = (b, tensors)
        backend = dist.get_backend(self.group)
        if b.flat.is_cuda and overlap:
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)
                self._works[bi] = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True)
        else:
            self._works[bi] = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM,
                                              group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._where.get(id(p))
        if bi is None:
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            if self._cap:
                self._pack_publish(bi)
            else:
                self._launch(bi)
            self.launched_from_hooks += 1

    # ---- captured step with overlap ---------------------------------------------------------------------------------------
    def _pack_publish(self, bi):
        tensors = self._tensors(bi)
        b = self._bucket(bi, tensors)
        b.pack(tensors)
        self._phase.append((b, tensors))
        self._cap_order.append(bi)
        self._cap_done[bi] = True
        if self._sync is not None:
            from . import _lib as L
            base = self._sync.data_ptr()
            L.check(L.load().alignq_dp_flag_publish(base + 4 * (1 + bi), base, L.stream_ptr()), "alignq_dp_flag_publish")

    def capture_begin(self):
        """Inside the capture region, before the forward: the graph's first node bumps the replay counter; the hooks are armed in
        pack-and-publish mode.  Needs the bucket layout of an earlier (warm-up) iteration."""
        if not self.active():
            return
        if self._groups is None:
            raise RuntimeError("BucketedGradAllReduce.capture_begin: no bucket layout yet - run one eager iteration first")
        n = len(self._groups)
        dev = next(p.device for p in self._live)
        self._phase, self._cap_order, self._cap_done = [], [], [False] * n
        self._pending = [len(g) for g in self._groups]
        if dev.type == "cuda":
            from . import _lib as L
            if self._sync is None or self._sync.numel() != 1 + n:
                raise RuntimeError("BucketedGradAllReduce.capture_begin: the flag words must exist before the capture (_discover)")
            # (the counter keeps counting across re-captures; `_replays` mirrors it on the host)
            L.check(L.load().alignq_dp_counter_bump(self._sync.data_ptr(), L.stream_ptr()), "alignq_dp_counter_bump")
        self._cap = self._armed = True

    def capture_end(self):
        """Inside the capture region, after the backward: buckets whose hooks did not all fire are packed here."""
        if not self.active():
            return
        self._armed = self._cap = False
        for bi in range(len(self._groups)):
            if not self._cap_done[bi]:
                self._cap = True
                self._pack_publish(bi)
                self._cap = False
        self._overlap_ready = True

    # ---- eager interface ------------------------------------------------------------------------------------------
    def begin(self):
        """Call after the forward (the D matrices exist), right before backward()."""
        if not self.active():
            return
        self._armed = self._groups is not None
        if self._armed:
            n = len(self._groups)
            self._pending = [len(g) for g in self._groups]
            self._works, self._buckets_now = [None] * n, [None] * n

    def finish(self):
        """Call after backward(): waits for every bucket (the current stream waits, not the host, on RCCL) and scatters the
        means back into p.grad / ADMM.D."""
        if not self.active():
            return
        was_armed, self._armed = self._armed, False
        if self._groups is None:
            self._discover()
        n = len(self._groups)
        if not was_armed:
            self._works, self._buckets_now = [None] * n, [None] * n
        for bi in range(n):
            if self._works[bi] is None:          # first iteration, or a bucket whose hooks did not all fire
                self._launch(bi, overlap=False)
        backend = dist.get_backend(self.group)
        for bi in range(n):
            self._works[bi].wait()
            b, tensors = self._buckets_now[bi]
            if backend != "nccl":
                b.flat.mul_(1.0 / self.world)
            b.unpack(tensors)
        self._works = [None] * n

    def __call__(self, _step=None):
        self.begin()
        self._armed = False
        self.finish()

    # ---- phased interface for a captured step (no hooks inside a graph: the collectives run between two graphs) ----
    def pack(self):
        if not self.active():
            return
        if self._groups is None:
            self._discover()
        self._phase = []
        self._overlap_ready = False
        for bi in range(len(self._groups)):
            tensors = self._tensors(bi)
            b = self._bucket(bi, tensors)
            b.pack(tensors)
            self._phase.append((b, tensors))

    def reduce(self):
        if not self.active():
            return
        backend = dist.get_backend(self.group)
        op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        if self._overlap_ready and self._sync is not None and self._phase and self._phase[0][0].flat.is_cuda:
            # the graph that packs the buckets has just been LAUNCHED (not finished): each collective waits, on the communication
            # stream, for its bucket's flag of THIS replay and then runs beside the rest of the replayed backward
            from . import _lib as L
            lib = L.load()
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            self._replays += 1
            base = self._sync.data_ptr()
            works = []
            with torch.cuda.stream(self._comm):
                for (b, _), bi in zip(self._phase, self._cap_order):
                    L.check(lib.alignq_dp_stream_wait_ge(self._comm.cuda_stream, base + 4 * (1 + bi), self._replays),
                            "alignq_dp_stream_wait_ge")
                    works.append(dist.all_reduce(b.flat, op=op, group=self.group, async_op=True))
            for w in works:
                w.wait()                  # the CURRENT stream waits (behind the graph it has just been given)
            return
        works = [dist.all_reduce(b.flat, op=op, group=self.group, async_op=True) for b, _ in self._phase]
        for w, (b, _) in zip(works, self._phase):
            w.wait()
            if backend != "nccl":
                b.flat.mul_(1.0 / self.world)

    def unpack(self):
        if not self.active():
            return
        for b, tensors in self._phase:
            b.unpack(tensors)

    def snapshot(self):
        return (getattr(self, "_phase", None), getattr(self, "_cap_order", None), self._overlap_ready)

    def restore(self, state):
        self._phase, self._cap_order, self._overlap_ready = state


def attach_office(office_step, group=None, force=False, bucket_bytes=24 << 20, min_buckets=4):
    """Wire data parallelism into an OfficeTrainStep: broadcast the initial state, install the bucketed, overlapped
    all-reduce over every parameter SGD steps (feature extractor incl. alterD / gamma, both heads) + the sites' D."""
    broadcast_module_state(office_step.model, 0, group)
    params = [p for g in office_step.optimizer_t.param_groups for p in g["params"]]
    hook = BucketedGradAllReduce(params, lambda: [b.admm0.D for b in office_step.blocks], group, force=force,
                                 bucket_bytes=bucket_bytes, min_buckets=min_buckets)
    office_step.grad_hook = hook
    # the conv weights' gradients leave the weight quantiser's backward: ONE multi-tensor node for the whole model would hand all
    # 94 MB over at the very end of the backward (nothing left to overlap with); per ResNet stage, each created right before the
    # stage's forward, they leave as the backward passes the stage boundaries
    if hasattr(office_step, "stage_weights"):
        office_step.stage_weights(True)
    return hook


# ---------------------------------------------------------------------------------------------------------------------
# N4 (SURVEY.md §8e "exact-global alternative"): the [B_g, B_g] correlation of the GLOBAL batch instead of per-rank [b, b]
# matrices.  corr standardises every feature over the batch, so a rank needs all B_g samples of the features it contracts:
#   all-to-all  [b, F] (batch-sharded) -> [B_g, F / world] (feature-sharded)
#   local SYRK  G_r = Xh_r Xh_r^T / F_r on the shard (the same split-bf16 MFMA kernels: ops.CorrFn; B_g <= 128)
#   all-reduce  G = sum_r (F_r / F) G_r                                  (exact: features are disjoint)
# and the mirrored backward (dG is identical on every rank; the shard's dX_r goes back through the reverse all-to-all).
# Reference op: model/quantization.py:134-137 applied to the concatenated batch.
class _FeatureShard(torch.autograd.Function):
    """[b, F] on every rank -> [B_g, F / world]: all samples of this rank's feature shard, rows in rank order."""

    @staticmethod
    def forward(ctx, x, group, grad_scale):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        b, F = x.shape
        if F % world:
            raise RuntimeError(f"global corr: feature count {F} is not divisible by the world size {world}")
        Fr = F // world
        ctx.meta = (group, world, rank, b, F, Fr, float(grad_scale))
        if dist.get_backend(group) == "nccl":       # RCCL all-to-all over xGMI: N*4 bytes per rank each way
            send = x.view(b, world, Fr).permute(1, 0, 2).contiguous()          # [dst rank][b][Fr]
            recv = torch.empty_like(send)                                       # [src rank][b][Fr]
            dist.all_to_all_single(recv, send, group=group)
            return recv.view(world * b, Fr)
        # gloo (CPU tests) has no all-to-all: gather every rank's rows and keep this rank's columns
        rows = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(rows, x.contiguous(), group=group)
        return torch.cat(rows, 0)[:, rank * Fr:(rank + 1) * Fr].contiguous()

    @staticmethod
    def backward(ctx, g):
        group, world, rank, b, F, Fr, scale = ctx.meta
        g = g.contiguous()
        if dist.get_backend(group) == "nccl":
            send = g.view(world, b, Fr)                                         # [dst rank (owner of the rows)][b][Fr]
            recv = torch.empty_like(send)                                       # [src rank (owner of the columns)][b][Fr]
            dist.all_to_all_single(recv, send.contiguous(), group=group)
            dx = recv.permute(1, 0, 2).reshape(b, F)
        else:
            shards = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(shards, g, group=group)
            dx = torch.cat([s[rank * b:(rank + 1) * b] for s in shards], 1)     # my rows, every rank's columns
        return dx * scale, None, None


class _SumAcrossRanks(torch.autograd.Function):
    """all-reduce(sum) whose result feeds a loss every rank evaluates identically: d(loss)/d(local term) = d(loss)/d(sum)."""

    @staticmethod
    def forward(ctx, t, group):
        out = t.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def global_corr(x, eps=0.0, group=None, local_corr=None, grad_scale=None):
    """corr(x, x) of the GLOBAL batch (rows = all ranks' samples in rank order), identical on every rank.

    x: this rank's [b, ...] activations.  local_corr(X) -> [B_g, B_g]: the per-shard SYRK (default ops.CorrFn on the HIP
    kernels: fused split-bf16 Gram up to 128 rows, blocked exact-fp32 Gram up to ALIGNQ_MAX_CORR_BATCH = 1024).  grad_scale multiplies the gradient that returns to x;
    default = world size, which makes the usual MEAN all-reduce of the parameter gradients reproduce the gradient of a
    correlation loss that is counted once for the global batch (the per-rank cross-entropy means average correctly as is)."""
    world = dist.get_world_size(group)
    x2 = x.reshape(x.shape[0], -1)
    Bg = x2.shape[0] * world
    if local_corr is None:
        from . import _lib as L
        from . import ops
        if Bg > L.MAX_CORR_BATCH:
            raise RuntimeError(f"global corr: global batch {Bg} exceeds the {L.MAX_CORR_BATCH} rows alignq_corr_fwd takes; "
                               "use the per-rank semantics (SURVEY.md §8e) for larger global batches")
        local_corr = lambda X: ops.CorrFn.apply(X, float(eps))      # noqa: E731
    Xr = _FeatureShard.apply(x2, group, world if grad_scale is None else grad_scale)
    Gr = local_corr(Xr) * (1.0 / world)          # (F_r / F) with equal shards
    return _SumAcrossRanks.apply(Gr, group)


def global_site_D(x, k, act_range, eps=0.0, group=None, local_site=None, grad_scale=None):
    """D = corr(t, t) - corr(x, x) of the GLOBAL batch from ONE exchange of x (round 4; global_corr(t) - global_corr(x) moved
    both tensors: two all-to-alls and two all-reduces each way).  The transform t = r (2 Phi(x) - 1) is elementwise, so the
    feature shard re-forms it from the exchanged x; local_site(X) -> [B_g, B_g] is the shard's pair correlation (default
    ops.SiteDFn: the fused site kernels up to 128 rows, the pair kernels of the blocked Gram up to ALIGNQ_MAX_CORR_BATCH), the
    all-reduce sums the shards' (F_r / F) D_r.  grad_scale as in global_corr.  Reference: model/quantization.py:109-123 applied to
    the concatenated batch."""
    world = dist.get_world_size(group)
    x2 = x.reshape(x.shape[0], -1)
    Bg = x2.shape[0] * world
    if local_site is None:
        from . import _lib as L
        from . import ops
        if Bg > L.MAX_CORR_BATCH:
            raise RuntimeError(f"global corr: global batch {Bg} exceeds the {L.MAX_CORR_BATCH} rows alignq_site_fwd takes; "
                               "use the per-rank semantics (SURVEY.md §8e) for larger global batches")
        local_site = lambda X: ops.SiteDFn.apply(X, int(k), float(act_range), float(eps))      # noqa: E731
    Xr = _FeatureShard.apply(x2, group, world if grad_scale is None else grad_scale)
    Dr = local_site(Xr) * (1.0 / world)          # (F_r / F) with equal shards
    return _SumAcrossRanks.apply(Dr, group)

