"""Data-parallel path on CPU: world_size 2 over gloo.  The collective layer (alignq_amd/dp.py) is device-agnostic
torch.distributed code; the compute in these CPU ranks is the eager-torch oracle (tests may use it), so what is
verified is the N>1 protocol: one flat mean all-reduce of (gradients of non-ADMM parameters + stacked D matrices),
replicas bit-identical after the optimizer steps, and equal to a single process that averages by hand."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    from oracle import torch_ref as R
    cfg = R.Config(tree="admm", bitW=4, abitW=4, train_batch_size=4)
    torch.manual_seed(seed)
    net = R.PreActResNet(cfg, [1, 1, 1], 4, 4).train()
    return R, cfg, net


def _data():
    g = torch.Generator().manual_seed(123)
    return torch.randn(2, 4, 3, 32, 32, generator=g), torch.randint(0, 10, (2, 4), generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        R, cfg, net = _make(seed=rank)              # different init per rank: the broadcast must fix it
        dp.broadcast_module_state(net, 0)
        step = R.TrainStep(net, cfg)
        hook = dp.GradAndDAllReduce([p for _, p in step.param_t], lambda: [m.D for m in net.admm_modules()])
        xs, ys = _data()
        # one iteration in the reference's order with the hook between backward and the optimizer steps
        step.opt_t.zero_grad(); step.opt_admm.zero_grad()
        logits, tl = net(xs[rank])
        (torch.nn.functional.cross_entropy(logits, ys[rank]) + tl).backward()
        local_g0 = step.param_t[0][1].grad.clone()
        for m in net.admm_modules():
            m.D = m.D.detach().clone()
        local_D0 = net.admm_modules()[0].D.clone()
        hook()
        g0_avg = step.param_t[0][1].grad.clone()
        convs = net.quant_convs()
        step.opt_t.step(step.idx, [c.weight_cdf for c in convs], [c.weight_pdf for c in convs], cfg.lam, cfg.lam2)
        mods = net.admm_modules()
        step.opt_admm.step(step.a_idx, step.g_idx, [m.D for m in mods], [m.alterD for m in mods],
                           [m.gamma for m in mods], [m.mu for m in mods], [m.rho for m in mods], bitW=cfg.bitW)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        D0_avg, full_bucket = mods[0].D.clone(), int(hook.bucket.flat.numel())
        # second iteration with a SHORT batch (the reference's loaders have no drop_last: CIFAR's last batch is 80 of 128):
        # D is [3,3] inside ADMM(dim=4), so the flat bucket's layout changes and the hook must switch buckets, not reuse
        # the cached element counts (the native copy kernel would run past D)
        step.opt_t.zero_grad(); step.opt_admm.zero_grad()
        logits, tl = net(xs[rank][:3])
        (torch.nn.functional.cross_entropy(logits, ys[rank][:3]) + tl).backward()
        for m in net.admm_modules():
            m.D = m.D.detach().clone()
        assert tuple(net.admm_modules()[0].D.shape) == (3, 3)
        hook()
        short_bucket = int(hook.bucket.flat.numel())
        step.opt_t.step(step.idx, [c.weight_cdf for c in convs], [c.weight_pdf for c in convs], cfg.lam, cfg.lam2)
        step.opt_admm.step(step.a_idx, step.g_idx, [m.D for m in mods], [m.alterD for m in mods],
                           [m.gamma for m in mods], [m.mu for m in mods], [m.rho for m in mods], bitW=cfg.bitW)
        flat2 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        out[rank] = dict(flat2=flat2.numpy().copy(), short_bucket=short_bucket, n_buckets=len(hook._buckets),
                         flat=flat.numpy().copy(), g0_local=local_g0.numpy(), D0_local=local_D0.numpy(),
                         g0_avg=g0_avg.numpy(), D0_avg=D0_avg.numpy(), bucket=full_bucket)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    # replicas identical after the step (same broadcast init, same averaged grads and D)
    assert np.array_equal(r0["flat"], r1["flat"])
    # the all-reduce is a mean over ranks
    np.testing.assert_allclose(r0["g0_avg"], 0.5 * (r0["g0_local"] + r1["g0_local"]), rtol=0, atol=1e-7)
    np.testing.assert_allclose(r0["D0_avg"], 0.5 * (r0["D0_local"] + r1["D0_local"]), rtol=0, atol=1e-7)
    assert np.array_equal(r0["g0_avg"], r1["g0_avg"]) and np.array_equal(r0["D0_avg"], r1["D0_avg"])
    # ONE bucket: all non-ADMM grads + all D matrices ([1,1,1] net: 9 sites of 4x4)
    R, cfg, net = _make()
    n_t = sum(p.numel() for n, p in net.named_parameters() if "alterD" not in n and "gamma" not in n)
    assert r0["bucket"] == n_t + 9 * 16
    # the short batch got its own bucket (9 sites of 3x3) and the replicas are still bit-identical after it
    assert r0["short_bucket"] == n_t + 9 * 9 and r0["n_buckets"] == 2
    assert np.array_equal(r0["flat2"], r1["flat2"]) and np.isfinite(r0["flat2"]).all()


def test_flat_bucket_roundtrip():
    from alignq_amd.dp import FlatBucket
    ts = [torch.randn(3, 4), torch.randn(7), torch.randn(2, 2, 2)]
    b = FlatBucket([t.shape for t in ts], "cpu")
    b.pack(ts)
    assert torch.equal(b.flat, torch.cat([t.reshape(-1) for t in ts]))
    b.flat.mul_(2.0)
    outs = [torch.empty_like(t) for t in ts]
    b.unpack(outs)
    for o, t in zip(outs, ts):
        assert torch.equal(o, 2 * t)


def test_world_size_one_is_a_noop():
    from alignq_amd.dp import GradAndDAllReduce
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradAndDAllReduce([p], lambda: [])()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_flat_bucket_rejects_a_changed_layout_and_the_hook_switches_buckets():
    """the native pack / unpack kernel copies the element counts cached at construction; a short last
    batch (D [b',b']) or a changed set of non-None gradients must raise in FlatBucket and make the hook lay out a new
    bucket instead of running past the tensors."""
    from alignq_amd.dp import FlatBucket, GradAndDAllReduce
    b = FlatBucket([(4, 4), (3,)], "cpu")
    with pytest.raises(RuntimeError, match="tensor list changed"):
        b.pack([torch.zeros(3, 3), torch.zeros(3)])
    with pytest.raises(RuntimeError, match="tensor list changed"):
        b.unpack([torch.zeros(4, 4)])
    p = torch.nn.Parameter(torch.ones(5))
    p.grad = torch.full((5,), 2.0)
    Ds = [torch.ones(4, 4)]
    hook = GradAndDAllReduce([p], lambda: Ds)
    hook.active = lambda: True
    hook.reduce = lambda: None
    hook()
    first = hook.bucket
    assert first.flat.numel() == 5 + 16
    Ds[0] = torch.full((3, 3), 7.0)            # short batch
    guard = torch.zeros(64)                    # would be overwritten by an unpack past D
    hook()
    assert hook.bucket is not first and hook.bucket.flat.numel() == 5 + 9
    assert torch.equal(Ds[0], torch.full((3, 3), 7.0)) and not guard.any()
    Ds[0] = torch.ones(4, 4)
    hook()
    assert hook.bucket is first                # layouts are cached, not re-allocated every time


# ---------------------------------------------------------------------------------------------------------------------
# Office / DANN step (BASELINE config 5): bucketed all-reduce launched from autograd hooks (alignq_amd.dp.BucketedGradAllReduce)
def _office_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        from oracle import torch_ref as R
        cfg = R.Config(tree="office", bitW=4, abitW=4, train_batch_size=4)
        torch.manual_seed(10 + rank)                   # different init per rank: the broadcast must fix it
        net = R.OfficeDANN(cfg, 4, 4, "aligned", (1, 1, 1, 1), width_per_group=8).train()
        dp.broadcast_module_state(net, 0)
        step = R.OfficeTrainStep(net, cfg, lr=0.004)
        params = [p for g in step.opt_t.param_groups for p in g["params"]]
        blocks = net.feature.blocks()
        hook = dp.BucketedGradAllReduce(params, lambda: [b.admm0.D for b in blocks], bucket_bytes=256 << 10, min_buckets=4)
        step.grad_hook = hook
        g = torch.Generator().manual_seed(100 + rank)  # different data per rank
        rec = {}
        for it in range(2):
            xs, xt = torch.randn(4, 3, 32, 32, generator=g), torch.randn(4, 3, 32, 32, generator=g)
            ys = torch.randint(0, 31, (4,), generator=g)
            # local (pre-reduction) values of one early and one late parameter and of one D, captured through a probe that runs
            # just before finish(): wrap finish
            local = {}
            orig_finish = hook.finish

            def finish(local=local, orig=orig_finish):
                local["g_stem"] = net.feature.conv1.weight.grad.clone()
                local["g_head"] = net.class_classifier.c_fc3.weight.grad.clone()
                local["D0"] = blocks[0].admm0.D.clone()
                local["from_hooks"] = hook.launched_from_hooks
                orig()
            hook.finish = finish
            step(xs, ys, xt)
            hook.finish = orig_finish
            rec[it] = dict(g_stem_local=local["g_stem"].numpy(), g_head_local=local["g_head"].numpy(), D0_local=local["D0"].numpy(),
                           from_hooks=local["from_hooks"], g_stem=None, D0=blocks[0].admm0.D.numpy().copy(),
                           flat=torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy().copy())
        out[rank] = dict(rec=rec, n_buckets=len(hook._groups), n_live=len(hook._live), n_params=len(params),
                         last_has_D=len(hook._tensors(len(hook._groups) - 1)) == len(hook._groups[-1]) + len(blocks),
                         last_is_stem=any(p is net.feature.conv1.weight for p in hook._groups[-1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_office_bucketed_overlapped_allreduce_two_ranks_gloo():
    """SURVEY §8e for config 5: >= 4 buckets in backward order, launched from autograd hooks DURING the backward (second
    iteration; the first discovers which parameters get gradients: DANN never uses feature.fc), D matrices in the bucket that
    completes last; the reduced values are the mean over ranks and the replicas stay bit-identical after SGD + ADMM_OPT."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_office_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["n_buckets"] >= 4 and r0["n_live"] == r0["n_params"] - 2           # feature.fc.{weight,bias} never get grads
    assert r0["last_has_D"] and r0["last_is_stem"]
    for it in range(2):
        a, b = r0["rec"][it], r1["rec"][it]
        assert np.array_equal(a["flat"], b["flat"]) and np.isfinite(a["flat"]).all()      # replicas identical after the step
        assert not np.array_equal(a["g_stem_local"], b["g_stem_local"])                   # the ranks did see different data
        np.testing.assert_allclose(a["D0"], 0.5 * (a["D0_local"] + b["D0_local"]), atol=1e-7)
        assert np.array_equal(a["D0"], b["D0"])
    # iteration 0 reduced without overlap (discovery); in iteration 1 every bucket was launched from an autograd hook, i.e.
    # before finish() ran — the probe reads the counter at the entry of finish()
    assert r0["rec"][0]["from_hooks"] == 0 and r0["rec"][1]["from_hooks"] == r0["n_buckets"]


def _office_captured_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        from oracle import torch_ref as R
        cfg = R.Config(tree="office", bitW=4, abitW=4, train_batch_size=4)
        torch.manual_seed(10 + rank)
        net = R.OfficeDANN(cfg, 4, 4, "aligned", (1, 1, 1, 1), width_per_group=8).train()
        dp.broadcast_module_state(net, 0)
        step = R.OfficeTrainStep(net, cfg, lr=0.004)
        params = [p for g in step.opt_t.param_groups for p in g["params"]]
        blocks = net.feature.blocks()
        cap = 96 << 10
        hook = dp.BucketedGradAllReduce(params, lambda: [b.admm0.D for b in blocks], bucket_bytes=cap, min_buckets=4)

        class CapturedForm:
            """What OfficeTrainStep.capture() + __call__ do around the backward on the data-parallel path: capture_begin() before
            the (captured) forward + backward, capture_end() behind it, then reduce() and unpack() (graph 2)."""
            def __init__(self):
                self.log = {}

            def begin(self):
                hook.capture_begin()

            def finish(self):
                self.log["from_hooks"] = hook.launched_from_hooks
                hook.capture_end()
                self.log["order"] = list(hook._cap_order)
                self.log["local"] = [b.flat.clone() for b, _ in hook._phase]
                self.log["D0_local"] = blocks[0].admm0.D.clone()
                hook.reduce()
                hook.unpack()
        g = torch.Generator().manual_seed(100 + rank)
        data = lambda: (torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 31, (4,), generator=g),      # noqa: E731
                        torch.randn(4, 3, 32, 32, generator=g))
        step.grad_hook = hook                         # iteration 0: eager, discovers the layout
        xs, ys, xt = data()
        step(xs, ys, xt)
        sizes = [sum(p.numel() * 4 for p in grp) for grp in hook._groups]
        biggest = max(p.numel() * 4 for p in hook._live)
        form = CapturedForm()
        step.grad_hook = form
        rec = []
        for it in range(2):
            before = hook.launched_from_hooks
            xs, ys, xt = data()
            step(xs, ys, xt)
            rec.append(dict(packed_from_hooks=form.log["from_hooks"] - before, order=form.log["order"],
                            local=[t.numpy() for t in form.log["local"]],
                            reduced=[b.flat.numpy().copy() for b, _ in hook._phase],
                            D0_local=form.log["D0_local"].numpy(), D0=blocks[0].admm0.D.numpy().copy(),
                            flat=torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy().copy()))
        out[rank] = dict(rec=rec, sizes=sizes, cap=cap, biggest=biggest, n_buckets=len(hook._groups))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_office_captured_form_packs_buckets_from_hooks_and_reduces_them_in_order_two_ranks_gloo():
    """the captured data-parallel Office step's protocol (capture_begin -> hooks pack every bucket where the
    backward completes it -> capture_end -> reduce -> unpack; on the GPU reduce() additionally orders each collective behind its
    bucket's flag of the running replay), world 2 on gloo: every bucket is packed from an autograd hook, in the same completion order on both ranks; the
    reduced buffers are the mean of the two ranks' local ones; replicas stay bit-identical; D = mean.  And the layout honours
    bucket_bytes: no bucket exceeds it unless it is a single larger tensor."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_office_captured_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["n_buckets"] >= 4
    for sz in r0["sizes"]:
        assert sz <= r0["cap"] or sz <= r0["biggest"], (sz, r0["cap"], r0["biggest"])
    for it in range(2):
        a, b = r0["rec"][it], r1["rec"][it]
        # every bucket from a hook; completion order = the order autograd finishes them in (a permutation of the layout order: a
        # downsample branch finishes out of registration order), the SAME on both ranks - the collectives are issued in it
        assert a["packed_from_hooks"] == r0["n_buckets"] and sorted(a["order"]) == list(range(r0["n_buckets"]))
        assert a["order"] == b["order"]
        for la, lb, ra, rb in zip(a["local"], b["local"], a["reduced"], b["reduced"]):
            np.testing.assert_allclose(ra, 0.5 * (la + lb), rtol=0, atol=1e-7)
            assert np.array_equal(ra, rb)
        assert np.array_equal(a["flat"], b["flat"]) and np.isfinite(a["flat"]).all()
        np.testing.assert_allclose(a["D0"], 0.5 * (a["D0_local"] + b["D0_local"]), atol=1e-7)
        assert np.array_equal(a["D0"], b["D0"])


# ---------------------------------------------------------------------------------------------------------------------
# N4: exact-global-batch correlation (alignq_amd.dp.global_corr)
def _gcorr_worker(rank, world, port, out, b=8):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        from oracle import torch_ref as R
        g = torch.Generator().manual_seed(7)
        C, H, W = 6, 4, 4                              # F = 96, divisible by the world size
        X = torch.randn(world * b, C, H, W, generator=g) * 0.9 + 0.2      # the global batch, identical on every rank
        dG = torch.randn(world * b, world * b, generator=g)
        for eps in (0.0, 1e-5):
            x = X[rank * b:(rank + 1) * b].clone().requires_grad_(True)
            G = dp.global_corr(x, eps, None, local_corr=lambda V: R.corr(V, V, eps), grad_scale=1.0)
            (G * dG).sum().backward()
            out[(rank, eps)] = dict(G=G.detach().numpy().copy(), dx=x.grad.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("b", [8, 128])
def test_global_corr_equals_single_process_corr_on_the_concatenated_batch(b):
    """all-to-all (gloo: all-gather) to the feature-sharded layout -> per-shard SYRK -> all-reduce; result and
    the gradient that returns to every rank's samples equal the single-process corr of the concatenated batch (<= 1e-5).
    b = 128 per rank is a GLOBAL batch of 256 (the 128-row cap of round 2 is gone; on the GPU the shard
    SYRK at 256 rows is the blocked Gram, tests/test_gpu_round3.py checks that one against the oracle)."""
    from oracle import torch_ref as R
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gcorr_worker, args=(world, port, out, b), nprocs=world, join=True)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(world * b, 6, 4, 4, generator=g) * 0.9 + 0.2
    dG = torch.randn(world * b, world * b, generator=g)
    for eps in (0.0, 1e-5):
        Xi = X.clone().requires_grad_(True)
        Gref = R.corr(Xi.view(world * b, -1), Xi.view(world * b, -1), eps)
        (Gref * dG).sum().backward()
        for rank in range(world):
            o = out[(rank, eps)]
            np.testing.assert_allclose(o["G"], Gref.detach().numpy(), atol=1e-5)
            np.testing.assert_allclose(o["dx"], Xi.grad[rank * b:(rank + 1) * b].numpy(), atol=1e-5, rtol=1e-4)
        assert np.array_equal(out[(0, eps)]["G"], out[(1, eps)]["G"])          # identical on every rank


def _torch_site_D(V, r, eps):
    """the shard's pair correlation in plain torch (oracle/torch_ref.py: the reference's lines restated)"""
    from oracle import torch_ref as R
    t = r * (2.0 * 0.5 * (1.0 + torch.erf(V / 2.0 ** 0.5)) - 1.0)          # model/quantization.py:49-59 with m = 0, s = 1
    return R.corr(t, t, eps) - R.corr(V, V, eps)


def _gsite_worker(rank, world, port, out, b=8):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        g = torch.Generator().manual_seed(11)
        X = torch.randn(world * b, 6, 4, 4, generator=g) * 0.9 + 0.2
        dD = torch.randn(world * b, world * b, generator=g)
        for eps in (0.0, 1e-5):
            x = X[rank * b:(rank + 1) * b].clone().requires_grad_(True)
            D = dp.global_site_D(x, 4, 2.0, eps, None, local_site=lambda V: _torch_site_D(V, 2.0, eps), grad_scale=1.0)
            (D * dD).sum().backward()
            out[(rank, eps)] = dict(D=D.detach().numpy().copy(), dx=x.grad.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_site_D_from_one_exchange_equals_the_single_process_pair(tmp_path):
    """Round 4: D = corr(t, t) - corr(x, x) of the global batch from ONE exchange of x (dp.global_site_D: the feature shard
    re-forms the elementwise transform): value and the gradient returning to every rank's rows equal the single-process pair on
    the concatenated batch (model/quantization.py:109-123; <= 1e-5), identical on every rank."""
    world, b = 2, 8
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gsite_worker, args=(world, port, out, b), nprocs=world, join=True)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(world * b, 6, 4, 4, generator=g) * 0.9 + 0.2
    dD = torch.randn(world * b, world * b, generator=g)
    for eps in (0.0, 1e-5):
        Xi = X.clone().requires_grad_(True)
        Dref = _torch_site_D(Xi.view(world * b, -1), 2.0, eps)
        (Dref * dD).sum().backward()
        for rank in range(world):
            o = out[(rank, eps)]
            np.testing.assert_allclose(o["D"], Dref.detach().numpy(), atol=1e-5)
            np.testing.assert_allclose(o["dx"], Xi.grad[rank * b:(rank + 1) * b].numpy(), atol=1e-5, rtol=1e-4)
        assert np.array_equal(out[(0, eps)]["D"], out[(1, eps)]["D"])


# ---------------------------------------------------------------------------------------------------------------------
# The phased pack | reduce | unpack interface as a CAPTURED TrainStep drives it (alignq_amd/train_step.py: capture() bakes
# pack() into the first graph and unpack() into the second, __call__ replays them around an eager reduce(), and an off-shape
# last batch runs one eager iteration through _eager_fallback, which saves / restores the hook's bucket binding).
def _phased_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import dp
        g = torch.Generator().manual_seed(7 + rank)
        p = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
        for q in p:
            q.grad = torch.zeros_like(q)
        Ds = [torch.zeros(4, 4), torch.zeros(4, 4)]
        state = {"Ds": Ds}
        hook = dp.GradAndDAllReduce(p, lambda: state["Ds"])

        def fill(tensors):                     # "forward + backward": new local values in the SAME storage (graph replay)
            for t in tensors:
                t.copy_(torch.randn(t.shape, generator=g))

        # ---- capture (train_step.py:capture): pack at the end of graph 1, unpack at the start of graph 2; a replay
        # re-executes exactly these copies on exactly these tensors and this flat buffer
        fill([q.grad for q in p] + Ds)
        hook.pack()
        cap_bucket, cap_tensors = hook.bucket, hook._tensors()
        replay1 = lambda: (fill(cap_tensors), cap_bucket.pack(cap_tensors))          # noqa: E731
        replay2 = lambda: cap_bucket.unpack(cap_tensors)                              # noqa: E731
        hook.reduce(); hook.unpack()

        def captured_step():                   # TrainStep.__call__ with _graph2
            replay1()
            local = [t.clone() for t in cap_tensors]
            hook.reduce()
            replay2()
            return local, [t.clone() for t in cap_tensors]

        rec = {}
        rec["full0"] = captured_step()
        # ---- off-shape last batch: TrainStep._eager_fallback = snapshot, one eager iteration with the hook, restore
        keep_g, keep_D, keep_hook = [q.grad for q in p], state["Ds"], hook.snapshot()
        for q in p:
            q.grad = torch.randn(q.shape, generator=g)
        state["Ds"] = [torch.randn(3, 3, generator=g), torch.randn(3, 3, generator=g)]
        short_local = [q.grad.clone() for q in p] + [d.clone() for d in state["Ds"]]
        hook()
        rec["short"] = (short_local, [q.grad.clone() for q in p] + [d.clone() for d in state["Ds"]])
        rebound = hook.bucket is not cap_bucket
        for q, g0 in zip(p, keep_g):
            q.grad = g0
        state["Ds"] = keep_D
        hook.restore(keep_hook)
        # ---- the next full-shape step replays the graphs again: the eager reduce() must hit the captured bucket
        rec["full1"] = captured_step()
        out[rank] = dict(rec={k: ([t.numpy() for t in a], [t.numpy() for t in b]) for k, (a, b) in rec.items()},
                         rebound=rebound, same_after=hook.bucket is cap_bucket)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_phased_hook_as_a_captured_step_drives_it_with_a_short_batch_between_replays():
    """VERDICT r2 item 8 / ADVICE r2 (high): world 2, gloo.  After the eager short-batch iteration the replayed full-shape
    step must still average what the captured pack wrote (before the fix reduce() all-reduced the short batch's stale
    bucket and graph 2 unpacked un-reduced gradients: replicas diverged silently)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_phased_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["rebound"] and r0["same_after"]          # the fallback did switch buckets, and the binding was put back
    for key in ("full0", "short", "full1"):
        loc0, red0 = r0["rec"][key]
        loc1, red1 = r1["rec"][key]
        for a0, a1, m0, m1 in zip(loc0, loc1, red0, red1):
            np.testing.assert_allclose(m0, 0.5 * (a0 + a1), rtol=0, atol=1e-7)        # mean over the two ranks
            assert np.array_equal(m0, m1)                                             # identical on both


def test_flat_bucket_matches_compares_shapes_not_only_element_counts():
    from alignq_amd.dp import FlatBucket
    b = FlatBucket([(4, 4), (3,)], "cpu")
    assert b.matches([torch.zeros(4, 4), torch.zeros(3)])
    assert not b.matches([torch.zeros(2, 8), torch.zeros(3)])       # same element counts, other tensors


# ---- round 5 (VERDICT r4 item 3 / ADVICE r4): what round 4's crash in hipStreamEndCapture came from, on the CPU -----------------
def test_retained_graph_params_sees_a_graph_kept_alive_by_an_output():
    """train_step.retained_graph_params: an iteration whose loss tensor is still referenced keeps every parameter's
    gradient-accumulation node alive (bound to the stream it ran on); once the tensor is gone the nodes are gone."""
    from alignq_amd.train_step import retained_graph_params
    lin = torch.nn.Linear(4, 3)
    frozen = torch.nn.Parameter(torch.zeros(2), requires_grad=False)
    params = list(lin.parameters()) + [frozen]
    assert retained_graph_params(params) == []
    loss = lin(torch.randn(5, 4)).sum()
    loss.backward()
    stale = retained_graph_params(params)
    assert len(stale) == 2 and all(any(s is p for p in lin.parameters()) for s in stale)
    kept = loss.detach()
    del loss
    assert retained_graph_params(params) == [] and float(kept) == float(kept)
    # idempotent, and no probe tag is left behind on a surviving node
    out = lin(torch.randn(2, 4))
    assert len(retained_graph_params(params)) == 2 and len(retained_graph_params(params)) == 2
    acc = lin.weight.expand_as(lin.weight).grad_fn.next_functions[0][0]
    assert "alignq_capture_probe" not in acc.metadata
    del out


def test_retained_graph_probe_says_so_when_accumulation_nodes_are_pinned(monkeypatch):
    """the probe tells a retained graph from a fresh parameter by whether the tagged gradient-accumulation node dies with
    the probe's temporary graph.  If nodes are kept alive by something else (here: a cache that pins every node it hands out, as a
    component registering hooks on the nodes themselves would), every parameter would look stale and capture() would blame the
    caller's tensors; the probe's own fresh parameter detects that and the error says what is wrong."""
    from alignq_amd import train_step
    lin = torch.nn.Linear(4, 3)
    assert train_step.retained_graph_params(list(lin.parameters())) == []          # healthy: fresh parameters are not stale
    pinned = []
    real = torch.Tensor.expand_as

    def pinning_expand_as(self, other):
        out = real(self, other)
        if out.grad_fn is not None:
            pinned.append(out.grad_fn.next_functions[0][0])       # somebody keeps the accumulation node alive
        return out
    monkeypatch.setattr(torch.Tensor, "expand_as", pinning_expand_as)
    with pytest.raises(RuntimeError, match="probe is unusable"):
        train_step.retained_graph_params(list(lin.parameters()))
    monkeypatch.undo()
    del pinned[:]
    assert train_step.retained_graph_params(list(lin.parameters())) == []


def _attach_detach_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alignq_amd import config, dp
        from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
        from alignq_amd.train_step import TrainStep, release_step_graphs, retained_graph_params
        config.args.bitW = config.args.abitW = 4
        config.args.train_batch_size = 4
        torch.manual_seed(0)
        net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10).train()
        step = TrainStep(net, lr=0.01)

        def snapshot():
            snap = {"step": {k: (id(v) if not isinstance(v, (bool, int, float, str, type(None))) else v)
                             for k, v in vars(step).items() if k not in ("_global_corr_undo",)}}
            for name, m in net.named_modules():
                snap[name] = {k: v for k, v in vars(m).items()
                              if isinstance(v, (bool, int, float, str, type(None))) and not k.startswith("_") and k != "D"}
            return snap
        before = snapshot()
        dp.attach(step, force=True, global_corr=True)
        during = snapshot()
        changed = sorted(n for n in before if before[n] != during.get(n))
        # the eager exact-global iterations leave D WITH its autograd graph on every ADMM module (ADMM.forward, utils/admm.py:25):
        # emulated here (no HIP kernels on the CPU) with a D that hangs on the model's parameters
        for a in step.admms:
            a.D = (a.alterD * 2.0 + net.logit.weight.sum())
        assert len(retained_graph_params(list(net.parameters()))) > 0
        dp.detach(step)
        after = snapshot()
        for a in step.admms:
            assert a.D.grad_fn is None                                  # detach() dropped the graphs with the mode
        assert retained_graph_params(list(net.parameters())) == []
        # everything attach(global_corr=True) changed is back; a site module now carries global_corr = None (getattr default)
        diff = {n: (before[n], after[n]) for n in before if before[n] != after[n]}
        for n, (b, a_) in list(diff.items()):
            extra = {k: v for k, v in a_.items() if k not in b}
            same = all(a_[k] == b[k] for k in b)
            if same and all(k == "global_corr" and v is None for k, v in extra.items()):
                del diff[n]
        # release_step_graphs: what capture() does for a step that was never attached
        step.admms[0].D = step.admms[0].gamma * 3.0
        release_step_graphs(step.admms)
        ok_release = step.admms[0].D.grad_fn is None
        out.put((rank, changed[:6], len(changed), diff, ok_release, getattr(step, "_had_global_corr", None)))
    except BaseException as e:       # the parent must not wait for its queue timeout
        out.put((rank, [], 0, {"worker failed": repr(e)}, False, None))
        raise
    finally:
        dist.destroy_process_group()


def test_attach_global_corr_then_detach_restores_the_step_and_drops_its_graphs():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_attach_detach_worker, args=(0, 1, _free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    _, changed, n_changed, diff, ok_release, had = res
    assert n_changed > 0 and "step" in changed or n_changed > 0          # attach did change state (fuse_bn flags, deferred, hook)
    assert diff == {}, diff                                             # ... and detach restored all of it
    assert ok_release and had is None                                   # no sticky "was global once" flag any more
