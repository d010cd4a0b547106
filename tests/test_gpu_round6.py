"""Round 6: the benchmarked HIP-graph replay is the SAME computation as the oracle-checked eager step.

Every kernel of this repository reduces in a fixed order (per-workgroup partials, last-arriver tickets that fix
the summation order, no float atomics), so from one initial state `N` eager iterations and `capture(warmup=w)` + `N - w` replays must
leave identical bits in every parameter, momentum buffer, batch-norm running statistic, ADMM.D, alterD and gamma.  These tests
assert exactly that on the configurations bench.py times (reference iterations: cdf_alignment_admm/resnet-20-cifar-10/main.py:
288-378 and cdf_alignment_admm/dann_office/main.py:343-456), and print the first differing tensor when it does not hold."""
import numpy as np
import pytest
import torch

from tests.golden.det_init import det_init_

pytestmark = pytest.mark.gpu


def npy(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def full_state(model, step, admms):
    """Everything an iteration leaves behind, by name."""
    st = {}
    for n_, p in model.named_parameters():
        st["param:" + n_] = npy(p)
    for n_, b in model.named_buffers():
        st["buffer:" + n_] = npy(b)
    names = {id(p): n_ for n_, p in model.named_parameters()}
    for p, s in step.optimizer_t.state.items():
        if "momentum_buffer" in s and s["momentum_buffer"] is not None:
            st["momentum:" + names[id(p)]] = npy(s["momentum_buffer"])
    for i, a in enumerate(admms):
        if a.D is not None:
            st["D:%d" % i] = npy(a.D)
    return st


def same_bits(a, b):
    """Equal as fp32 values (+0 == -0: a zero-filled gradient accumulates -0 to +0) with NaNs in the same places."""
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def differing(sa, sb):
    assert set(sa) == set(sb), sorted(set(sa) ^ set(sb))
    out = []
    for key in sa:
        if not same_bits(sa[key], sb[key]):
            d = np.abs(sa[key].astype(np.float64) - sb[key].astype(np.float64))
            out.append((key, int(np.count_nonzero(sa[key] != sb[key])), sa[key].size, float(np.nanmax(d))))
    return out


CIFAR_CASES = [
    # (name, depth, bits, tree)                        BASELINE config
    ("resnet20_8bit_admm", 20, 8, "admm"),           # configs[1]: the headline
    ("resnet20_2bit_admm", 20, 2, "admm"),           # configs[2] per rank
    ("resnet56_4bit_admm", 56, 4, "admm"),           # configs[3] per rank
    ("resnet20_8bit_cdf", 20, 8, "cdf"),             # configs[0]
]


@pytest.mark.parametrize("name,depth,bits,tree", CIFAR_CASES, ids=[c[0] for c in CIFAR_CASES])
def test_graph_replay_equals_eager_bit_for_bit_cifar(dev, name, depth, bits, tree):
    """bench.py's step (TrainStep(channels_last=True, qconv=True, fuse_bn=True, pack_bins=True), batch 128): five eager iterations
    against capture(warmup=3) + two replays from the same initial state and the same batches."""
    from alignq_amd import config
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.train_step import TrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = bits
    config.args.train_batch_size = 128
    try:
        def make():
            torch.manual_seed(7)
            return (resnet20_quant if depth == 20 else resnet56_quant)(bits, bits, tree=tree).to(dev).train()
        g = torch.Generator().manual_seed(13)
        x = torch.randn(128, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (128,), generator=g).to(dev)
        n_total, warm = 5, 3
        m1, m2 = make(), make()
        s1 = TrainStep(m1, channels_last=True, qconv=True, fuse_bn=True)
        s2 = TrainStep(m2, channels_last=True, qconv=True, fuse_bn=True)
        assert bool(s1.admms) == (tree == "admm")
        for _ in range(n_total):
            o1 = s1(x, y)
        s2.capture(x, y, warmup=warm)
        assert s2._graph is not None and s2._graph2 is None          # ONE graph: the form bench.py times at N = 1
        for _ in range(n_total - warm):
            o2 = s2(x, y)
        torch.cuda.synchronize()
        assert torch.isfinite(o1[1]) and torch.isfinite(o2[1])
        bad = differing(full_state(m1, s1, s1.admms), full_state(m2, s2, s2.admms))
        for k_, (a, b) in enumerate(zip(o1, o2)):
            if torch.is_tensor(a) and not same_bits(npy(a), npy(b)):
                bad.append(("output:%d" % k_, -1, a.numel(), float(np.abs(npy(a) - npy(b)).max())))
        assert not bad, "graph replay differs from eager in %d tensors, first: %s" % (len(bad), bad[:6])
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


def test_graph_replay_equals_eager_bit_for_bit_office(dev):
    """BASELINE configs[4] at its real size (28 + 28 images of 224 x 224, OfficeTrainStep(channels_last=True) with its defaults, as
    bench.py --model resnet50_dann builds it): four eager iterations against capture(warmup=2) + two replays.  Every parameter
    behind this repository's kernels bit for bit; the stem's convolution and batch-norm sit behind torch's max-pool backward (atomic
    adds): compared to rounding, as in test_office_iteration_is_reproducible_run_to_run."""
    import alignq_amd.quantization  # noqa: F401
    from alignq_amd import config
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size)
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = config.args.eval_batch_size = 28
    try:
        B = 28
        g = torch.Generator().manual_seed(11)
        xs = torch.randn(B, 3, 224, 224, generator=g).to(dev)
        xt = torch.randn(B, 3, 224, 224, generator=g).to(dev)
        ys = torch.randint(0, 31, (B,), generator=g).to(dev)

        def make():
            return det_init_(resnet50_dann(8, 8)).to(dev).train()
        n_total, warm = 4, 2
        m1, m2 = make(), make()
        s1 = OfficeTrainStep(m1, lr=4e-5, channels_last=True)
        s2 = OfficeTrainStep(m2, lr=4e-5, channels_last=True)
        for _ in range(n_total):
            o1 = s1(xs, ys, xt)
        s2.capture(xs, ys, xt, warmup=warm)
        assert s2._graph is not None and s2._graph2 is None
        for _ in range(n_total - warm):
            o2 = s2(xs, ys, xt)
        torch.cuda.synchronize()
        assert torch.isfinite(o1[1]) and torch.isfinite(o2[1])
        a1, a2 = [b.admm0 for b in s1.blocks], [b.admm0 for b in s2.blocks]
        st1, st2 = full_state(m1, s1, a1), full_state(m2, s2, a2)
        stem = ("feature.conv1.", "feature.bn1.")
        loose = [key for key in st1 if key.split(":", 1)[1].startswith(stem)]
        for key in loose:
            np.testing.assert_allclose(st1[key], st2[key], rtol=1e-5, atol=1e-7 * float(np.abs(st1[key]).max()) + 1e-12, err_msg=key)
            st1.pop(key), st2.pop(key)
        bad = differing(st1, st2)
        assert same_bits(npy(o1[1]), npy(o2[1])) and same_bits(npy(o1[2]), npy(o2[2])), (float(o1[1]), float(o2[1]))
        assert not bad, "graph replay differs from eager in %d tensors, first: %s" % (len(bad), bad[:6])
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size = old


# ------------------------------------------------------------------------------------------------ SURVEY H5 / F9
from tests import oracle_c as O                      # noqa: E402
from tests.conftest import load_golden               # noqa: E402


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _unpack_nan(g, key, shape):
    return np.unpackbits(g[key])[: int(np.prod(shape))].astype(bool).reshape(shape)


@pytest.mark.parametrize("name", ["a", "b"])
def test_constant_column_without_epsilon_gives_the_reference_nan_pattern(dev, name):
    """The CIFAR trees' corr divides by an unguarded std (cdf_alignment_admm/resnet-20-cifar-10/model/quantization.py:134-137): a
    feature that is constant over the batch makes the whole correlation NaN.  Fixture G14 holds the reference's own outputs at
    [128, 4096] ("a": the split-bf16 MFMA site kernels) and [28, 1568] ("b": the wave-autonomous small-batch kernels) with two
    constant columns.  Through the drop-in modules (activation_quantize_fn + ADMM, corr): x_q unaffected and bit-exact, D / loss /
    dx / dalterD / dgamma NaN exactly where the reference's are (everywhere); corr's dx for a finite dG NaN in the constant columns
    only and within 1e-5 of the reference elsewhere; with the Office tree's eps = 1e-5 everything is finite."""
    import alignq_amd.cdf_alignment_admm as NA
    import alignq_amd.office as NO
    from alignq_amd import config, ops
    g = load_golden("g14_constant_column")
    x0 = g[f"x_{name}"].astype(np.float32)
    B, F = x0.shape
    shape = tuple(int(v) for v in g[f"shape_{name}"])
    k, head = int(g["k"]), int(g["head"])
    n = 2 ** k - 1
    cols = [int(c) for c in g["const_cols"]]
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = k, B
    try:
        # ---- corr alone (ops.CorrFn through the namespace's corr)
        x = cu(x0, dev).requires_grad_(True)
        G = NA.corr(x, x)
        G.backward(cu(g[f"dG_{name}"], dev))
        assert np.isnan(npy(G)).all() and _unpack_nan(g, f"G_isnan_{name}", (B, B)).all()
        dx = npy(x.grad)
        want_nan = _unpack_nan(g, f"corr_dx_isnan_{name}", (B, F))
        assert np.array_equal(np.isnan(dx), want_nan), (int(np.isnan(dx).sum()), int(want_nan.sum()))
        ref = g[f"corr_dx_head_{name}"]
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(dx[:, :head][ok], ref[ok], atol=1e-5 * max(1.0, float(np.abs(ref[ok]).max())), rtol=1e-4)
        o_dx = O.corr_bwd(g[f"dG_{name}"], x0, 0.0)
        fin = ~want_nan
        np.testing.assert_allclose(dx[fin], o_dx[fin], atol=1e-5 * max(1.0, float(np.abs(o_dx[fin]).max())), rtol=1e-4)
        # ---- the whole ADMM site
        admm = NA.ADMM(B).to(dev)
        with torch.no_grad():
            admm.alterD.copy_(cu(g[f"alterD0_{name}"], dev))
            admm.gamma.copy_(cu(g[f"gamma0_{name}"], dev))
        act = NA.activation_quantize_fn(k, "second", admm).to(dev)
        xs = cu(x0.reshape(shape), dev).requires_grad_(True)
        gq = (torch.randn(shape, generator=torch.Generator().manual_seed(1)) * 0.01).to(dev)
        xq, loss = act(xs)
        torch.autograd.backward([xq, loss], [gq, torch.ones((), device=dev)])
        oq, oD = O.site_fwd(x0, k, 2.0, 0.0)
        assert np.array_equal(npy(xq).reshape(B, F).view(np.uint32), oq.view(np.uint32))               # x_q: bit for bit
        assert np.array_equal(npy(xq).reshape(B, F)[:, :head], (g[f"bins_head_{name}"].astype(np.float64) / n).astype(np.float32))
        for t, key, shp in ((admm.D, "D", (B, B)), (xs.grad, "dx", (B, F)), (admm.alterD.grad, "dalterD", (B, B)),
                            (admm.gamma.grad, "dgamma", (B, B))):
            got = np.isnan(npy(t).reshape(shp))
            assert np.array_equal(got, _unpack_nan(g, f"{key}_isnan_{name}", shp)), (key, int(got.sum()), got.size)
        assert np.isnan(float(loss.detach())) and np.isnan(g[f"loss_{name}"]) and np.isnan(oD).all()
        # ---- Office tree: std + 1e-5 (dann_office/model/quantization.py:158-161): finite, and equal to the oracle
        xe = cu(x0, dev).requires_grad_(True)
        Ge = NO.corr(xe, xe)
        Ge.backward(cu(g[f"dG_{name}"], dev))
        np.testing.assert_allclose(npy(Ge), O.corr_fwd(x0, 1e-5), atol=1e-5, rtol=0)
        o_dxe = O.corr_bwd(g[f"dG_{name}"], x0, 1e-5)
        np.testing.assert_allclose(npy(xe.grad), o_dxe, atol=1e-5 * max(1.0, float(np.abs(o_dxe).max())), rtol=1e-4)
    finally:
        config.args.abitW, config.args.train_batch_size = old


@pytest.mark.parametrize("B,F", [(192, 1000), (300, 4096)])
def test_constant_column_without_epsilon_above_128_rows(dev, B, F):
    """The same on the blocked Gram (corr_large_kernels.hip, B > 128): G all NaN, dx NaN in the constant columns only, the other
    columns within 1e-5 of the C oracle (which the G14 fixture pins to the reference's pattern at [128, 4096] / [28, 1568])."""
    from alignq_amd import ops
    rng = np.random.default_rng(B)
    x0 = (rng.standard_normal((B, F)) * 0.8 + 0.1).astype(np.float32)
    x0[:, 5], x0[:, 17] = 0.0, 0.75
    dG = rng.standard_normal((B, B)).astype(np.float32)
    x = cu(x0, dev).requires_grad_(True)
    G = ops.CorrFn.apply(x, 0.0)
    G.backward(cu(dG, dev))
    oG, odx = O.corr_fwd(x0, 0.0), O.corr_bwd(dG, x0, 0.0)
    assert np.isnan(oG).all() and np.isnan(npy(G)).all()
    dx = npy(x.grad)
    assert np.array_equal(np.isnan(dx), np.isnan(odx)) and np.isnan(odx).sum() == 2 * B and np.isnan(odx[:, [5, 17]]).all()
    fin = ~np.isnan(odx)
    np.testing.assert_allclose(dx[fin], odx[fin], atol=1e-5 * max(1.0, float(np.abs(odx[fin]).max())), rtol=1e-4)


@pytest.mark.parametrize("nhwc", [False, True])
def test_constant_column_through_the_bn_folded_site(dev, nhwc):
    """The bench path's site (fused.bn_site: batch-norm folded into site_fwd4 / site_bwd4, slab_reduce_multi, ADMM loss; CIFAR tree,
    eps = 0) with a feature position that is constant over the batch in z (hence in x = a z + b): y = relu(x_q + residual) is the
    oracle's up to tie-zone flips (the device's (a, b) differ from the oracle's in the last bit); D, the loss, dz, dgamma, dbeta,
    dalterD, dgamma_admm are NaN like the oracle's (all of them), the residual's gradient (a ReLU mask of the upstream) stays
    finite and bit-exact."""
    import alignq_amd.cdf_alignment_admm as NA
    from alignq_amd import config
    from alignq_amd.fused import bn_site, bn_site_fusable
    g = load_golden("g14_constant_column")
    B, C, H, W = (int(v) for v in g["shape_a"])
    k, r = 8, 2.0
    n = 2 ** k - 1
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = k
    config.args.train_batch_size = B
    try:
        zl = g["x_a"].astype(np.float32) * 1.7 + 0.3                  # [B, F] in MEMORY order (columns 5 and 17 constant)
        rng = np.random.default_rng(8)
        rl = (rng.standard_normal(zl.shape) * 0.7).astype(np.float32)
        gl = (rng.standard_normal(zl.shape) * 0.01).astype(np.float32)
        gam, bet = (rng.random(C) + 0.5).astype(np.float32), (rng.standard_normal(C) * 0.2).astype(np.float32)

        def to_dev(m):
            if nhwc:
                return cu(m.reshape(B, H, W, C), dev).permute(0, 3, 1, 2)       # channels-last strides, logical NCHW
            return cu(m.reshape(B, C, H, W), dev)

        def mem(t):
            t = t.detach()
            return npy(t.permute(0, 2, 3, 1).contiguous() if nhwc else t.contiguous()).reshape(B, -1)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(cu(gam, dev))
            bn.bias.copy_(cu(bet, dev))
        admm = NA.ADMM(B).to(dev)
        A0, G0 = npy(admm.alterD), npy(admm.gamma)
        act = NA.activation_quantize_fn(k, "second", admm)
        z = to_dev(zl).requires_grad_(True)
        res = to_dev(rl).requires_grad_(True)
        assert bn_site_fusable(bn, act, z)
        y, loss = bn_site(bn, act, z, relu=True, residual=res)
        torch.autograd.backward([y, loss], [to_dev(gl), torch.ones((), device=dev)])
        ab_o, save_o, _ = O.bn_fold_ab(zl, C, int(nhwc), gam, bet, 1e-5)
        y_o, D_o, x_o = O.bn_site_fwd(zl, C, int(nhwc), ab_o, k, r, 0.0, rl, True)
        loss_o, dD_o, dA_o, dG_o = O.admm_loss(D_o, A0, G0, 0.2, 0.3)
        dz_o, dgam_o, dbet_o, dres_o, _ = O.bn_site_bwd(gl, dD_o, zl, C, int(nhwc), ab_o, save_o, y_o, r, 0.0)
        assert np.isnan(D_o).all() and np.isnan(loss_o) and np.isnan(dz_o).all() and np.isfinite(dres_o).all()
        flips = np.abs(mem(y) - y_o) * n
        assert np.isfinite(mem(y)).all() and flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 1e-4
        assert np.isnan(npy(admm.D)).all() and np.isnan(float(loss.detach()))
        for got, want, what in ((mem(z.grad), dz_o, "dz"), (npy(bn.weight.grad), dgam_o, "dgamma"), (npy(bn.bias.grad), dbet_o, "dbeta"),
                                (npy(admm.alterD.grad), dA_o, "dalterD"), (npy(admm.gamma.grad), dG_o, "dgamma_admm")):
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).all(), what
        same_mask = (mem(y) > 0) == (y_o > 0)
        assert np.array_equal(mem(res.grad)[same_mask], dres_o[same_mask]) and same_mask.mean() > 1 - 1e-4
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ captured DP with overlap
@pytest.fixture()
def pg_world_one(dev):
    """A one-rank process group on RCCL (backend "nccl"): the collective code paths on the real backend."""
    import os
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        yield None
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield None
    finally:
        dist.destroy_process_group()


def test_captured_office_step_with_overlapped_allreduce_equals_the_plain_step_at_world_one(dev, pg_world_one):
    """dp.attach_office(force=True) + capture(): forward + backward are ONE graph in which the autograd hooks pack every gradient
    bucket where the backward completes it and publish its flag (alignq_dp_flag_publish); each replay's reduce() puts, per bucket,
    alignq_dp_stream_wait_ge + the RCCL all-reduce on the communication stream, so bucket i's collective runs beside the rest of the
    replayed backward; the conv weights are quantised per ResNet stage so that their gradients leave the weight quantiser's backward
    stage by stage.  At world size 1 the mean is the identity and the staged quantiser does the same per-tensor arithmetic: every
    parameter, momentum buffer and ADMM.D is BIT-identical to the plain single-graph step (the stem's two layers, behind torch's
    max-pool backward, to rounding)."""
    import alignq_amd.quantization  # noqa: F401
    from alignq_amd import config, dp
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size)
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = config.args.eval_batch_size = 6
    try:
        g = torch.Generator().manual_seed(4)
        xs = torch.randn(6, 3, 64, 64, generator=g).to(dev)
        xt = torch.randn(6, 3, 64, 64, generator=g).to(dev)
        ys = torch.randint(0, 31, (6,), generator=g).to(dev)

        def make():
            return det_init_(resnet50_dann(8, 8)).to(dev).train()
        m0, m1 = make(), make()
        s0 = OfficeTrainStep(m0, lr=4e-5, channels_last=True)
        s1 = OfficeTrainStep(m1, lr=4e-5, channels_last=True)
        hook = dp.attach_office(s1, force=True, bucket_bytes=24 << 20)
        assert s1._staged and m1.feature._wq_stage is not None
        s0.capture(xs, ys, xt, warmup=2)
        s1.capture(xs, ys, xt, warmup=2)
        assert s0._graph2 is None and s1._graph2 is not None and hook._overlap_ready
        n_b = len(hook._groups)
        assert n_b >= 4 and sorted(hook._cap_order) == list(range(n_b)) and len(hook._phase) == n_b
        cap = 24 << 20
        biggest = max(p.numel() * 4 for p in hook._live)
        assert all(sum(p.numel() * 4 for p in grp) <= max(cap, biggest) for grp in hook._groups)
        for _ in range(3):
            o0 = s0(xs, ys, xt)
            o1 = s1(xs, ys, xt)
        torch.cuda.synchronize()
        assert hook._replays == 3 and int(hook._sync[0]) == 3 and all(int(v) == 3 for v in hook._sync[1:])
        assert torch.isfinite(o0[1]) and same_bits(npy(o0[1]), npy(o1[1])) and same_bits(npy(o0[2]), npy(o1[2]))
        st0 = full_state(m0, s0, [b.admm0 for b in s0.blocks])
        st1 = full_state(m1, s1, [b.admm0 for b in s1.blocks])
        for key in [k_ for k_ in st0 if k_.split(":", 1)[1].startswith(("feature.conv1.", "feature.bn1."))]:
            np.testing.assert_allclose(st0[key], st1[key], rtol=1e-5, atol=1e-7 * float(np.abs(st0[key]).max()) + 1e-12, err_msg=key)
            st0.pop(key), st1.pop(key)
        bad = differing(st0, st1)
        assert not bad, bad[:6]
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size = old


def test_prepared_site_backward_is_only_used_for_the_gradient_it_was_prepared_with(dev):
    """Site1LossSumFn.backward prepares every site of a Site1Batch with the ONE upstream scalar of the summed
    loss.  That is a site's gradient only if its loss vector enters the total exactly once: a vector that enters it TWICE reaches
    BNSite1Fn.backward with twice the scalar (autograd adds the two), and the site must then prepare itself from what actually
    arrived (before the fix it silently used the scalar: gradients of z, alterD, gamma too small by the factor).  (A vector
    weighted BEFORE total() is not a valid use at all: its values only exist once total() has launched the batch's reduction.)
    Reference semantics: plain autograd through trans_loss (cdf_alignment_admm/dann_office/model/resnet.py:146-156)."""
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = 8, 6
    try:
        B, C, H, G = 6, 64, 8, 2
        g = torch.Generator().manual_seed(21)
        z0 = (torch.randn(G * B, C, H, H, generator=g) * 1.3).to(dev).contiguous(memory_format=torch.channels_last)
        r0 = torch.relu(torch.randn(G * B, C, H, H, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
        gy = (torch.randn(G * B, C, H, H, generator=g) * 1e-2).to(dev).contiguous(memory_format=torch.channels_last)
        res = {}
        for arm in ("batched_twice", "per_site_twice", "batched_once"):
            torch.manual_seed(5)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            admm = NO.ADMM(B).to(dev)
            act = NO.activation_quantize_fn2(8, "aligned", admm).to(dev)
            z, r_ = z0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            if arm.startswith("batched"):
                with fused.Site1Batch() as s1:
                    y, lv = fused.bn_site_res_relu(bn, act, z, r_, 1e-5, groups=G, loss_vec=True)
                    total = s1.total([lv, lv] if arm == "batched_twice" else [lv])
            else:
                y, lv = fused.bn_site_res_relu(bn, act, z, r_, 1e-5, groups=G, loss_vec=True)
                total = (lv * 2.0).sum()
            (total + (y * gy).sum()).backward()
            torch.cuda.synchronize()
            res[arm] = dict(total=npy(total), dz=npy(z.grad), dr=npy(r_.grad), dA=npy(admm.alterD.grad), dG=npy(admm.gamma.grad),
                            dw=npy(bn.weight.grad), db=npy(bn.bias.grad))
        a, b, c = res["batched_twice"], res["per_site_twice"], res["batched_once"]
        np.testing.assert_allclose(a["total"], b["total"], rtol=1e-6)
        for key in ("dz", "dr", "dA", "dG", "dw", "db"):
            assert np.isfinite(a[key]).all() and np.array_equal(a[key], b[key]), key
        # and it does matter: the ADMM parameter gradients of the doubled loss are twice those of the single one
        np.testing.assert_allclose(a["dA"], 2.0 * c["dA"], rtol=1e-5, atol=1e-9)
        assert np.abs(c["dA"]).max() > 0
    finally:
        config.args.abitW, config.args.train_batch_size = old


def test_empty_inputs_follow_the_reference_elementwise_ops(dev):
    """Edge case: an empty tensor.  The reference's uniform_quantize (model/quantization.py:19-38) and its plain activation quantiser
    (cdf_alignment/resnet-20-cifar-10/model/quantization.py:81-103: `method != 'ours'` or no ADMM module) are chains of elementwise
    ATen ops - empty in, empty out, gradients empty; the C ABI itself refuses n <= 0 (ALIGNQ_EINVAL), the Python mirror does not launch."""
    import alignq_amd.cdf_alignment as NC
    from alignq_amd import _lib as L
    from alignq_amd import config
    x = torch.empty(0, 16, 4, 4, device=dev, requires_grad=True)
    for k in (1, 2, 8, 32):
        y = NC.uniform_quantize(k)(x)
        assert y.shape == x.shape and y.device == x.device
    old = config.args.abitW
    config.args.abitW = 8
    try:
        act = NC.activation_quantize_fn(8, "second").to(dev)
        y = act(x)
        assert y.shape == x.shape
        y.sum().backward()
        assert x.grad is not None and x.grad.shape == x.shape
    finally:
        config.args.abitW = old
    lib = L.load()
    z = torch.zeros(4, device=dev)
    assert lib.alignq_act_quant_fwd(L.ptr(z), L.ptr(z), None, 0, 8, 2.0, 0, None) < 0          # ALIGNQ_EINVAL: n <= 0


# ------------------------------------------------------------------------------------------------ round 6: twin site launch (headline)
def _site_bn_args(L, lib, dev, z, gamma, beta, k, relu, want_bins, C, HW, B, F):
    f32 = dict(dtype=torch.float32, device=dev)
    part = torch.empty(lib.alignq_bn_nhwc_ws_bytes(C), dtype=torch.uint8, device=dev)
    L.check(lib.alignq_bn_partial_stats_nhwc(L.ptr(z), B, C, HW, L.ptr(part), None), "bn_partial_stats_nhwc")
    t = dict(part=part, ab=torch.empty(2, C, **f32), save=torch.empty(2, C, **f32), rm=torch.zeros(C, **f32), rv=torch.ones(C, **f32),
             nbt=torch.zeros((), dtype=torch.int64, device=dev), stats=torch.empty(4, F, **f32),
             ws=torch.zeros(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev),
             y=None if want_bins else torch.empty_like(z),
             bins=torch.empty_strided(z.shape, z.stride(), dtype=torch.int16, device=dev) if want_bins else None)
    a = L.SiteBnArgs(L.ptr(z), L.ptr(part), L.ptr(gamma), L.ptr(beta), L.ptr(t["rm"]), L.ptr(t["rv"]), L.ptr(t["nbt"]), 0.1, 1e-5,
                     L.ptr(t["ab"]), L.ptr(t["save"]), C, HW, B, F, k, 2.0, 0.0, int(relu), None, 1, 0, L.ptr(t["y"]), L.ptr(t["bins"]),
                     L.ptr(t["stats"]), L.ptr(t["ws"]))
    return a, t


@pytest.mark.parametrize("B,C,H", [(128, 32, 16), (128, 64, 8), (100, 64, 8)])
def test_twin_site_launch_equals_two_launches_bit_for_bit(dev, B, C, H):
    """alignq_site_partials_bn_twin: the two sites behind a transition block's convolutions (cdf_alignment_admm/resnet-56-cifar-10/
    model/resnet.py:81-90: act_skip_q(skip_bn(skip_conv(x))) without ReLU, relu(act_q0(bn0(conv0(x)))) stored as level indices) in ONE
    launch against the two alignq_site_partials_bn launches it replaces: outputs, level indices, column statistics, (a, b), running
    statistics and - after the slab reduction - D and the ADMM loss are bit-identical (same code per workgroup).  And a shape whose
    single launch already fills the chip is refused (ALIGNQ_EUNSUPPORTED = -2): the caller launches the sites one after the other."""
    import ctypes
    from alignq_amd import _lib as L
    lib = L.load()
    k, HW, F = 8, H * H, C * H * H
    g = torch.Generator().manual_seed(B + C)
    mk = lambda: (torch.randn(B, H, H, C, generator=g) * 1.5 + 0.2).to(dev).permute(0, 3, 1, 2)       # channels-last      # noqa: E731
    za, zb = mk(), mk()
    gam = [(torch.rand(C, generator=g) + 0.5).to(dev) for _ in range(2)]
    bet = [(torch.randn(C, generator=g) * 0.2).to(dev) for _ in range(2)]
    A, Gm = (torch.randn(128, 128, generator=g) * 0.05).to(dev), (torch.randn(128, 128, generator=g) * 0.05).to(dev)
    res = {}
    for arm in ("twin", "separate"):
        a, ta = _site_bn_args(L, lib, dev, za, gam[0], bet[0], k, False, False, C, HW, B, F)
        b, tb = _site_bn_args(L, lib, dev, zb, gam[1], bet[1], k, True, True, C, HW, B, F)
        if arm == "twin":
            L.check(lib.alignq_site_partials_bn_twin(ctypes.byref(a), ctypes.byref(b), None), "twin")
        else:
            for s_, t in ((a, ta), (b, tb)):
                L.check(lib.alignq_site_partials_bn(s_.z, s_.bn_part, s_.bn_gamma, s_.bn_beta, s_.running_mean, s_.running_var,
                                                    s_.num_batches_tracked, s_.momentum, s_.bn_eps, s_.ab, s_.save, C, HW, B, F, k, 2.0, 0.0,
                                                    s_.relu, None, 1, 0, s_.xq, s_.bins_out, s_.stats, s_.ws, None), "single")
        out = []
        for t in (ta, tb):
            D, scal = torch.empty(B, B, device=dev), torch.empty(4, device=dev)
            L.check(lib.alignq_site_reduce_loss(L.ptr(t["ws"]), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), 128, 0.2, 0.3, L.ptr(scal), None), "reduce")
            out.append(dict(D=npy(D), loss=npy(scal[:1]), stats=npy(t["stats"]), ab=npy(t["ab"]), save=npy(t["save"]), rm=npy(t["rm"]),
                            rv=npy(t["rv"]), nbt=int(t["nbt"]), y=None if t["y"] is None else npy(t["y"]),
                            bins=None if t["bins"] is None else t["bins"].cpu().numpy()))
        torch.cuda.synchronize()
        res[arm] = out
    for sa, sb in zip(res["twin"], res["separate"]):
        for key in sa:
            if sa[key] is None:
                assert sb[key] is None
            elif isinstance(sa[key], int):
                assert sa[key] == sb[key] == 1
            else:
                assert np.isfinite(sa[key].astype(np.float64)).all() and np.array_equal(sa[key], sb[key]), key
    assert np.abs(res["twin"][0]["D"]).max() > 0 and res["twin"][1]["bins"].min() >= 0            # (the second site's ReLU clamps its indices)
    # a 256-tile site fills the chip on its own: refused
    Bf, Cf, Hf = 128, 16, 32
    zf = (torch.randn(Bf, Hf, Hf, Cf, generator=g)).to(dev).permute(0, 3, 1, 2)
    gf, bf = torch.ones(Cf, device=dev), torch.zeros(Cf, device=dev)
    a, _ta = _site_bn_args(L, lib, dev, zf, gf, bf, k, False, False, Cf, Hf * Hf, Bf, Cf * Hf * Hf)
    b, _tb = _site_bn_args(L, lib, dev, zf.clone(memory_format=torch.preserve_format), gf, bf, k, False, False, Cf, Hf * Hf, Bf, Cf * Hf * Hf)
    assert lib.alignq_site_partials_bn_twin(ctypes.byref(a), ctypes.byref(b), None) == -2


@pytest.mark.parametrize("B,C,H", [(128, 32, 16), (128, 64, 8), (100, 64, 8)])
def test_twin_site_backward_equals_two_launches_bit_for_bit(dev, B, C, H):
    """alignq_site_bwd_apply_bn_twin against the two alignq_site_bwd_apply_bn launches it replaces (the backward of the pair of
    test_twin_site_launch_equals_two_launches_bit_for_bit: one site with an fp32 output and no ReLU, one stored as level indices with
    the ReLU mask taken from them): dx and the per-tile batch-norm sums bit for bit."""
    import ctypes
    from alignq_amd import _lib as L
    lib = L.load()
    k, HW, F = 8, H * H, C * H * H
    g = torch.Generator().manual_seed(B + C + 1)
    mk = lambda sc=1.5: (torch.randn(B, H, H, C, generator=g) * sc + 0.2).to(dev).permute(0, 3, 1, 2)      # noqa: E731
    za, zb = mk(), mk()
    ga, gb = mk(1e-2), mk(1e-2)
    gam = [(torch.rand(C, generator=g) + 0.5).to(dev) for _ in range(2)]
    bet = [(torch.randn(C, generator=g) * 0.2).to(dev) for _ in range(2)]
    a, ta = _site_bn_args(L, lib, dev, za, gam[0], bet[0], k, False, False, C, HW, B, F)
    b, tb = _site_bn_args(L, lib, dev, zb, gam[1], bet[1], k, True, True, C, HW, B, F)
    L.check(lib.alignq_site_partials_bn_twin(ctypes.byref(a), ctypes.byref(b), None), "fwd twin")
    S = [(torch.randn(lib.alignq_site_bwd_ws_bytes(B) // 4, generator=g) * 1e-3).to(dev) for _ in range(2)]
    Dm, A, Gm, scal = torch.empty(B, B, device=dev), torch.zeros(128, 128, device=dev), torch.zeros(128, 128, device=dev), torch.empty(4, device=dev)
    for t, s_ in ((ta, S[0]), (tb, S[1])):       # a consistent S (fp32 + its bf16 image) from the preparation entry point
        L.check(lib.alignq_site_reduce_loss(L.ptr(t["ws"]), B, F, L.ptr(Dm), L.ptr(A), L.ptr(Gm), 128, 0.2, 0.3, L.ptr(scal), None), "reduce")
        one = torch.ones((), device=dev)
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        L.check(lib.alignq_site_prep_fused(L.ptr(Dm), L.ptr(A), L.ptr(Gm), 128, L.ptr(scal), 0.2, L.ptr(one), B, F, L.ptr(s_), L.ptr(dA),
                                           L.ptr(dG), None), "prep")
    res = {}
    for arm in ("twin", "separate"):
        outs = []
        structs = []
        for (t, z, gy, s_, bins) in ((ta, za, ga, S[0], None), (tb, zb, gb, S[1], tb["bins"])):
            dx = torch.full_like(z, float("nan"))
            part = torch.zeros(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
            structs.append(L.SiteBwdBnArgs(L.ptr(gy), L.ptr(s_), L.ptr(z), L.ptr(t["ab"]), L.ptr(t["save"]), C, HW, 1, None, L.ptr(bins),
                                           2 if bins is not None else 0, None, L.ptr(t["stats"]), B, F, 2.0, 0.0, L.ptr(dx), L.ptr(part)))
            outs.append((dx, part))
        if arm == "twin":
            L.check(lib.alignq_site_bwd_apply_bn_twin(ctypes.byref(structs[0]), ctypes.byref(structs[1]), None), "bwd twin")
        else:
            for q in structs:
                L.check(lib.alignq_site_bwd_apply_bn(q.g, q.S, q.z, q.ab, q.save, q.C, q.HW, q.nhwc, q.y_relu, q.y_bins, q.y_bin_bytes,
                                                     q.dresidual, q.stats, q.B, q.F, q.act_range, q.eps, q.dx, q.dx_part, None), "bwd single")
        torch.cuda.synchronize()
        res[arm] = [(npy(dx), part.cpu().numpy()) for dx, part in outs]
    for (dx1, p1), (dx2, p2) in zip(res["twin"], res["separate"]):
        assert np.isfinite(dx1).all() and np.array_equal(dx1, dx2) and np.array_equal(p1, p2)
    assert np.abs(res["twin"][0][0]).max() > 0 and np.abs(res["twin"][1][0]).max() > 0
    # and the same bits behind foreign work (a large GEMM: cold L2, skewed workgroup starts) - two co-resident workgroups of the 512-
    # workgroup F = 8192 launch differed by one ulp in 8 of 20 such launches (round 6; the launcher now keeps them one per CU)
    m = torch.randn(4096, 4096, device=dev)
    for _ in range(12):
        outs = []
        for q in structs:
            dx = torch.full((B, C, H, H), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
            part = torch.zeros(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
            q.dx, q.dx_part = L.ptr(dx), L.ptr(part)
            outs.append((dx, part))
        torch.cuda.synchronize()
        torch.mm(m, m)
        L.check(lib.alignq_site_bwd_apply_bn_twin(ctypes.byref(structs[0]), ctypes.byref(structs[1]), None), "bwd twin")
        torch.cuda.synchronize()
        for (dx, part), (dx2, p2) in zip(outs, res["separate"]):
            assert np.array_equal(npy(dx), dx2) and np.array_equal(part.cpu().numpy(), p2)


def test_eager_iteration_is_bit_identical_with_foreign_work_in_front_of_every_launch(dev):
    """The headline configuration's eager iteration with a 4096^3 GEMM launched in front of EVERY call into the library (cold L2,
    skewed workgroup starts) against the plain eager iteration: the whole state bit for bit (tools/diag_cold_step.py runs the same audit
    for every configuration, profiles/r06_cold_determinism.txt).  Round 6: the backward twin launch with two workgroups per CU failed
    exactly this and nothing else did."""
    from alignq_amd import config, _lib as L
    from alignq_amd.resnet import resnet20_quant
    from alignq_amd.train_step import TrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    real = L.load()
    operand = torch.randn(4096, 4096, device=dev)
    count = [0]

    class Cold:
        def __getattr__(self, name):
            fn = getattr(real, name)
            sig = L.SIGNATURES.get(name)
            if sig is None or not sig[1] or name.endswith(("_bytes", "_slots", "_supported", "_version", "strerror")):
                return fn

            def wrapped(*a):
                count[0] += 1
                torch.mm(operand, operand)
                return fn(*a)
            return wrapped
    try:
        g = torch.Generator().manual_seed(13)
        x = torch.randn(128, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (128,), generator=g).to(dev)

        def run(cold):
            torch.manual_seed(7)
            m = resnet20_quant(8, 8).to(dev).train()
            s = TrainStep(m, channels_last=True, qconv=True, fuse_bn=True)
            L._lib = Cold() if cold else real
            try:
                for _ in range(2):
                    s(x, y)
                torch.cuda.synchronize()
            finally:
                L._lib = real
            return full_state(m, s, s.admms)
        ref, got = run(False), run(True)
        assert count[0] > 150                                     # (every launch of two iterations went through the wrapper)
        bad = differing(ref, got)
        assert not bad, "cold iteration differs in %d tensors, first: %s" % (len(bad), bad[:6])
    finally:
        L._lib = real
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


@pytest.mark.parametrize("depth,bits", [(20, 8), (56, 4)])
def test_head_as_a_role_of_the_closing_reduction_launch_equals_the_two_launches(dev, depth, bits):
    """alignq_site_reduce_loss_multi_head (the sites still open at the end of the forward + the classifier head's forward in ONE launch;
    main.py:300-312) against alignq_site_reduce_loss_multi + alignq_head_ce_fwd: two eager iterations from the same state, everything
    the iterations leave behind and both loss values bit for bit (ResNet-56: 55 sites, more than one argument chunk)."""
    from alignq_amd import config, fused
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.train_step import TrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = bits
    config.args.train_batch_size = 128
    try:
        g = torch.Generator().manual_seed(13)
        x = torch.randn(128, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (128,), generator=g).to(dev)

        def run(role):
            torch.manual_seed(7)
            m = (resnet20_quant if depth == 20 else resnet56_quant)(bits, bits).to(dev).train()
            s = TrainStep(m, channels_last=True, qconv=True, fuse_bn=True)
            fused._HEAD_ROLE = role
            try:
                for _ in range(2):
                    o = s(x, y)
                torch.cuda.synchronize()
            finally:
                fused._HEAD_ROLE = True
            return full_state(m, s, s.admms), [npy(t) for t in o if torch.is_tensor(t)]
        (sa, oa), (sb, ob) = run(True), run(False)
        assert not differing(sa, sb)
        assert len(oa) == len(ob) >= 2 and all(same_bits(a, b) and np.isfinite(a).all() for a, b in zip(oa, ob))
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


@pytest.mark.parametrize("shapes", [[(64, 64, 3, 3)], [(65, 64, 3, 3)], [(16, 3, 3, 3), (64, 64, 3, 3), (32, 16, 1, 1)], [(16, 3, 3, 3), (10, 9, 1, 1)]],
                         ids=["one_launch_max", "two_launches_above_max", "one_launch_mixed", "two_launches_odd_count"])
def test_weight_quantiser_for_small_filters_in_one_launch_equals_the_per_tensor_path(dev, shapes):
    """alignq_weight_quant_fwd_multi quantises filters of at most 36864 elements (n % 4 == 0) with statistics and quantisation in ONE
    launch (mt_weight_fused_kernel, round 6) and keeps its two launches otherwise: both against the per-tensor entry point
    (model/quantization.py weight_quantize_fn.forward: mean / std of the tensor, CDF transform, uniform bins) - W_q and the CDF
    image bit for bit, the saved density to rounding - on either side of the limits."""
    from alignq_amd import ops
    from alignq_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(sum(s[0] for s in shapes))
    ws = [(torch.randn(*s, generator=g) * 0.1 + 0.01).to(dev) for s in shapes]
    T = len(ws)
    q = [torch.empty_like(w) for w in ws]
    cdf = [torch.empty_like(w) for w in ws]
    pdf = [torch.empty_like(w) for w in ws]
    ms = torch.empty(T, 2, device=dev)
    scratch = torch.empty(lib.alignq_weight_multi_ws_bytes(T), dtype=torch.uint8, device=dev)
    L.check(lib.alignq_weight_quant_fwd_multi(T, L.ptr_array(ws), L.ptr_array(q), L.ptr_array(cdf), L.ptr_array(pdf),
                                              L.i64_array([w.numel() for w in ws]), L.ptr(ms), 4, L.FORMULA_ADMM, L.ptr(scratch), None),
            "alignq_weight_quant_fwd_multi")
    torch.cuda.synchronize()
    for i, w in enumerate(ws):
        q2, c2, p2 = ops.WeightQuantFn.apply(w, 4, 0)
        assert np.array_equal(npy(q[i]), npy(q2)) and np.array_equal(npy(cdf[i]), npy(c2))
        np.testing.assert_allclose(npy(pdf[i]), npy(p2), rtol=1e-6)
        np.testing.assert_allclose(npy(ms[i]), [float(w.double().mean()), float(w.double().std())], rtol=1e-6)
