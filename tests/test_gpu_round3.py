"""GPU tests added in round 3 (all through the C ABI via alignq_amd.ops / the Python mirror):
value-level parity at config 5's site sizes; corr / the ADMM site above 128 rows (blocked Gram); NERF32 edge inputs."""
import numpy as np
import pytest
import torch

from tests import oracle_c as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    from alignq_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def npy(t):
    return t.detach().cpu().numpy()


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 2
@pytest.mark.parametrize("F", [100352, 802816])
def test_small_batch_site_by_value_at_config5_sizes(dev, F):
    """The wave-autonomous B <= 32 kernels at the Office tree's own shapes (batch 28, eps 1e-5; dann_office/model/
    quantization.py:112-161, sites of model/resnet.py:131-156): x_q bit for bit, D / loss / dx / dalterD / dgamma within 1e-5
    of the C oracle BY VALUE (round 2 only asserted symmetry / trace / lattice at these sizes).  [28, 802816] is the stem."""
    from alignq_amd import ops
    rng = np.random.default_rng(F % 1000)
    B, k, r, eps = 28, 8, 2.0, 1e-5
    x0 = (rng.standard_normal((B, F)) * 1.1 + 0.15).astype(np.float32)
    A0, G0 = rng.random((B, B), dtype=np.float32), rng.random((B, B), dtype=np.float32)
    gq = (rng.standard_normal((B, F)) * 1e-3).astype(np.float32)
    x = cu(x0, dev).requires_grad_(True)
    A, Gm = cu(A0, dev).requires_grad_(True), cu(G0, dev).requires_grad_(True)
    xq, loss, D = ops.SiteFn.apply(x, A, Gm, k, r, eps, 0.2, 0.3)
    torch.autograd.backward([xq, loss], [cu(gq, dev), torch.ones((), device=dev)])
    oq, oD = O.site_fwd(x0, k, r, eps)
    assert bits_equal(npy(xq), oq)
    np.testing.assert_allclose(npy(D), oD, atol=TOL, rtol=0)
    ol, odD, odA, odG = O.admm_loss(oD, A0, G0, 0.2, 0.3)
    np.testing.assert_allclose(float(loss), ol, atol=TOL)
    odx = O.site_bwd(gq, odD, x0, r, eps)
    np.testing.assert_allclose(npy(x.grad), odx, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(npy(A.grad), odA, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(npy(Gm.grad), odG, atol=1e-7, rtol=1e-4)


@pytest.mark.parametrize("F,k", [(40960, 8), (36864 + 64, 4)])
def test_plain_site_multi_tile_forward_and_looped_backward_vs_oracle(dev, F, k):
    """The plain (no batch-norm fold) B = 128 site above 32768 features: site_fwd4_kernel's multi-tile form (512 threads, two
    workgroups per CU) and site_bwd4_kernel's LOOPED form (full tiles, a whole tile of prefetch distance, statistics through
    LDS; 640 / 577 tiles over 256 workgroups: uneven trip counts) by value against the C oracle - the shapes the roofline
    numbers of `kernels.roofline_shapes.site_128x524288` are measured on are otherwise only timed."""
    from alignq_amd import ops
    rng = np.random.default_rng(F + k)
    B, r, eps = 128, 2.0, 0.0
    x0 = (rng.standard_normal((B, F)) * 1.2 - 0.1).astype(np.float32)
    A0, G0 = (rng.random((B, B), dtype=np.float32) - 0.5) * 0.1, (rng.random((B, B), dtype=np.float32) - 0.5) * 0.1
    gq = (rng.standard_normal((B, F)) * 1e-3).astype(np.float32)
    x = cu(x0, dev).requires_grad_(True)
    A, Gm = cu(A0, dev).requires_grad_(True), cu(G0, dev).requires_grad_(True)
    xq, loss, D = ops.SiteFn.apply(x, A, Gm, k, r, eps, 0.2, 0.3)
    torch.autograd.backward([xq, loss], [cu(gq, dev), torch.ones((), device=dev)])
    oq, oD = O.site_fwd(x0, k, r, eps)
    assert bits_equal(npy(xq), oq)
    np.testing.assert_allclose(npy(D), oD, atol=TOL, rtol=0)
    ol, odD, odA, odG = O.admm_loss(oD, A0, G0, 0.2, 0.3)
    np.testing.assert_allclose(float(loss), ol, atol=TOL)
    odx = O.site_bwd(gq, odD, x0, r, eps)
    np.testing.assert_allclose(npy(x.grad), odx, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(npy(A.grad), odA, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(npy(Gm.grad), odG, atol=1e-7, rtol=1e-4)
    # the gradient of the loss alone (no upstream gradient: the looped kernel reads x in g's place and drops it)
    x2 = cu(x0, dev).requires_grad_(True)
    _, loss2, _ = ops.SiteFn.apply(x2, A.detach(), Gm.detach(), k, r, eps, 0.2, 0.3)
    loss2.backward()
    np.testing.assert_allclose(npy(x2.grad), O.site_bwd(np.zeros_like(gq), odD, x0, r, eps), atol=TOL, rtol=1e-4)


def test_plain_quantiser_by_value_at_the_stem_size(dev):
    """activation_quantize_fn (no ADMM: act_q1 / act_q2 of every bottleneck) at [28, 64, 112, 112]: x_q and the packed level
    indices bit for bit against the oracle over all 22.5 M elements, dx within 1e-5."""
    from alignq_amd import _lib as L
    from alignq_amd import ops
    rng = np.random.default_rng(3)
    n, k, r = 28 * 802816, 8, 2.0
    x0 = (rng.standard_normal(n) * 1.3).astype(np.float32)
    g0 = rng.standard_normal(n).astype(np.float32)
    x = cu(x0, dev).requires_grad_(True)
    q = ops.ActQuantFn.apply(x, k, r, L.FORMULA_ADMM)
    q.backward(cu(g0, dev))
    oq, _, obins = O.act_quant_fwd(x0, k, r, O.FORMULA_ADMM)
    assert bits_equal(npy(q), oq)
    np.testing.assert_allclose(npy(x.grad), O.act_quant_bwd(g0, x0, r), atol=TOL, rtol=1e-4)
    bins = ops.act_quant_pack(x.detach(), k, r, L.FORMULA_ADMM)
    bins = bins[0] if isinstance(bins, (tuple, list)) else bins
    assert np.array_equal(npy(bins).astype(np.int32), obins)


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 3
@pytest.mark.parametrize("B,F,eps", [(192, 1000, 0.0), (256, 4096, 0.0), (512, 2050, 1e-5), (130, 70, 1e-5), (1024, 256, 0.0)])
def test_corr_above_128_rows_vs_oracle(dev, B, F, eps):
    """corr(x, x) on the blocked Gram (corr_large_kernels.hip; reference model/quantization.py:134-137, Office :158-161):
    G and dx for a non-symmetric upstream dG within 1e-5 of the C oracle, ragged F (not a multiple of 4 / 32 / 64) included."""
    from alignq_amd import ops
    rng = np.random.default_rng(B + F)
    x0 = (rng.standard_normal((B, F)) * 0.8 + 0.3).astype(np.float32)
    x0[:, 3] = 0.25 if eps > 0 else x0[:, 3]                 # a constant column: std == 0 (only meaningful with eps)
    dG = rng.standard_normal((B, B)).astype(np.float32)
    x = cu(x0, dev).requires_grad_(True)
    G = ops.CorrFn.apply(x, eps)
    G.backward(cu(dG, dev))
    np.testing.assert_allclose(npy(G), O.corr_fwd(x0, eps), atol=TOL, rtol=0)
    np.testing.assert_allclose(npy(x.grad), O.corr_bwd(dG, x0, eps), atol=TOL, rtol=1e-4)
    assert np.array_equal(npy(G), npy(G).T)                  # the mirrored blocks are copies
    G2 = ops.CorrFn.apply(x.detach(), eps)
    assert bits_equal(npy(G), npy(G2))                       # deterministic run to run


@pytest.mark.parametrize("tree,B,C,H", [("admm", 192, 6, 8), ("office", 256, 6, 8), ("admm", 130, 3, 5), ("office", 1000, 4, 4),
                                        ("admm", 513, 2, 6)])
def test_admm_site_above_128_rows_vs_oracle(dev, tree, B, C, H):
    """activation_quantize_fn with ADMM at a batch the fused kernels do not hold (round 4: ops.SiteLargeFn - alignq_site_fwd /
    alignq_site_bwd on the pair kernels of the blocked Gram - plus the ADMM loss; round 3 composed it from four stand-alone
    passes): x_q bit for bit, D / loss / dx / dalterD / dgamma within 1e-5 of the oracle; ADMM(dim) is sized by the batch like the
    reference does (utils/admm.py:17-27); F not a multiple of 4 (the unaligned loads), 5 and 8 row blocks of 128."""
    import alignq_amd.cdf_alignment_admm as NA
    import alignq_amd.office as NO
    from alignq_amd import config
    ns, eps = (NA, 0.0) if tree == "admm" else (NO, 1e-5)
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = 4, B
    try:
        rng = np.random.default_rng(B)
        x0 = (rng.standard_normal((B, C, H, H)) * 1.2).astype(np.float32)
        gq = (rng.standard_normal((B, C, H, H)) * 1e-2).astype(np.float32)
        admm = ns.ADMM(B).to(dev)
        A0, G0 = npy(admm.alterD), npy(admm.gamma)
        act = (ns.activation_quantize_fn if tree == "admm" else ns.activation_quantize_fn2)(4, "aligned", admm).to(dev)
        x = cu(x0, dev).requires_grad_(True)
        xq, loss = act(x)
        torch.autograd.backward([xq, loss], [cu(gq, dev), torch.ones((), device=dev)])
        F = C * H * H
        oq, oD = O.site_fwd(x0.reshape(B, F), 4, 2.0, eps)
        assert bits_equal(npy(xq).reshape(B, F), oq)
        np.testing.assert_allclose(npy(admm.D), oD, atol=TOL, rtol=0)
        ol, odD, odA, odG = O.admm_loss(oD, A0, G0, 0.2, 0.3)
        np.testing.assert_allclose(float(loss), ol, atol=TOL)
        odx = O.site_bwd(gq.reshape(B, F), odD, x0.reshape(B, F), 2.0, eps)
        np.testing.assert_allclose(npy(x.grad).reshape(B, F), odx, atol=TOL, rtol=1e-4)
        np.testing.assert_allclose(npy(admm.alterD.grad), odA, atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(npy(admm.gamma.grad), odG, atol=1e-7, rtol=1e-4)
    finally:
        config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ NERF32 on the device
def test_nerf32_device_equals_oracle_on_edge_inputs(dev):
    """The table-driven transform on the GPU against its C statement: every node boundary and its fp32 neighbours, the
    clamp, signed zeros, infinities and NaN, through the plain quantiser at k = 32 (which writes the pre-round transform)."""
    from alignq_amd import _lib as L
    from alignq_amd import ops
    edges = (np.arange(0, 120, dtype=np.float64) / 16.0).astype(np.float32)
    xs = np.concatenate([edges, np.nextafter(edges, np.float32(-1)), np.nextafter(edges, np.float32(99)),
                         np.array([0.0, -0.0, np.inf, -np.inf, 5.625, 5.6250005, 1e30, -1e30, 1e-45, -1e-45, 1e-30], np.float32)])
    xs = np.concatenate([xs, -xs]).astype(np.float32)
    xs = np.concatenate([xs, np.zeros((-len(xs)) % 4, np.float32)])
    for formula, of in ((L.FORMULA_ADMM, O.FORMULA_ADMM), (L.FORMULA_CDF, O.FORMULA_CDF)):
        for k in (32, 8, 2):
            got = npy(ops.ActQuantFn.apply(cu(xs, dev), k, 2.0, formula))
            want, _, _ = O.act_quant_fwd(xs, k, 2.0, of)
            assert bits_equal(got, want), (formula, k)
    nan_out = npy(ops.ActQuantFn.apply(cu(np.array([np.nan, 1.0, -np.nan, 0.5], np.float32), dev), 8, 2.0, L.FORMULA_ADMM))
    assert np.isnan(nan_out[0]) and np.isnan(nan_out[2]) and np.isfinite(nan_out[1]) and np.isfinite(nan_out[3])


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 4 (N1, Office)
# (128, 16, 32) / (100, 32, 16) / (3, 4, 5): "small" sites (round 4: the apply kernels finalise the statistics themselves)
@pytest.mark.parametrize("B,C,H,k", [(28, 64, 56, 8), (6, 512, 7, 4), (28, 256, 14, 8), (3, 4, 5, 2), (28, 64, 112, 8), (128, 16, 32, 8),
                                     (100, 32, 16, 4)])
def test_bn_folded_plain_quantiser_vs_oracle_and_torch_batchnorm(dev, B, C, H, k):
    """relu(act_q(bn(z))) of the Office bottleneck's first two sites and of the stem (dann_office/model/resnet.py:134-143,
    :230-233) as the folded chain (fused.bn_act_relu -> alignq_bnq_fwd / _bwd), channels-last, batch 28 shapes included:
    (a, b) within 3e-6 of the double-precision statistics (C oracle, pinned to torch.nn.BatchNorm2d), y BIT-EXACT given (a, b),
    running statistics like torch's, and dz / dgamma / dbeta within 1e-5 of torch-CPU autograd through BatchNorm2d + the
    transform's derivative + the ReLU mask (float64)."""
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    rng = np.random.default_rng(B * C + H)
    r, bn_eps = 2.0, 1e-5
    old = config.args.abitW
    config.args.abitW = k
    try:
        z0 = (rng.standard_normal((B, C, H, H)) * 1.7 + 0.4).astype(np.float32)
        gam = (rng.random(C) + 0.5).astype(np.float32)
        bet = (rng.standard_normal(C) * 0.2).astype(np.float32)
        g0 = rng.standard_normal((B, C, H, H)).astype(np.float32)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(cu(gam, dev)); bn.bias.copy_(cu(bet, dev))
        act = NO.activation_quantize_fn(k, "aligned").to(dev)
        z = cu(z0, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        assert fused.bnq_fusable(bn, act, z)
        y = act.forward_bn_relu(bn, z)
        y.backward(cu(g0, dev).contiguous(memory_format=torch.channels_last))
        # ---- forward: statistics, then bit-exact given the device's (a, b)
        z_mem = np.ascontiguousarray(z0.transpose(0, 2, 3, 1)).reshape(B, -1)          # [B, H*W*C]: channels-last memory order
        ab_o, save_o, var_u = O.bn_fold_ab(z_mem, C, 1, gam, bet, bn_eps)
        # the device's (a, b) are not returned by the module: recover them through the oracle's tolerance on y instead —
        # y must equal relu(quantise(a z + b)) for the oracle's (a, b) except where a 3e-6 difference in (a, b) moves a bin
        x_o = O.bn_apply(z_mem, C, 1, ab_o)
        q_o, t_o, _ = O.act_quant_fwd(x_o, k, r, O.FORMULA_ADMM)
        y_o = np.maximum(q_o, 0.0).reshape(B, H, H, C).transpose(0, 3, 1, 2)
        n = 2 ** k - 1
        frac = t_o.astype(np.float64) * n
        near_tie = (np.abs(frac - np.floor(frac) - 0.5) < 2e-3).reshape(B, H, H, C).transpose(0, 3, 1, 2)
        diff = np.abs(npy(y) - y_o) * n
        assert np.all(diff[~near_tie] == 0), int(np.count_nonzero(diff[~near_tie]))
        assert np.all(diff[near_tie] <= 1.0 + 1e-3)
        # running statistics: torch's own BatchNorm2d on the same input
        ref_bn = torch.nn.BatchNorm2d(C).train()
        with torch.no_grad():
            ref_bn.weight.copy_(torch.from_numpy(gam)); ref_bn.bias.copy_(torch.from_numpy(bet))
        zc = torch.from_numpy(z0).double().requires_grad_(True)
        ref_bn = ref_bn.double()
        xb = ref_bn(zc)
        np.testing.assert_allclose(npy(bn.running_mean), ref_bn.running_mean.float().numpy(), atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(npy(bn.running_var), ref_bn.running_var.float().numpy(), atol=1e-6, rtol=1e-5)
        assert int(bn.num_batches_tracked) == 1
        # ---- backward: d/dx of r*(2 Phi(x) - 1) is r*2*phi(x) (STE through the rounding), masked by the forward's y > 0
        t = r * torch.erf(xb / np.sqrt(2.0))
        mask = torch.from_numpy((npy(y) > 0).astype(np.float64))
        (t * mask * torch.from_numpy(g0).double()).sum().backward()
        np.testing.assert_allclose(npy(z.grad), zc.grad.float().numpy(), atol=TOL, rtol=1e-4)
        np.testing.assert_allclose(npy(bn.weight.grad), ref_bn.weight.grad.float().numpy(), atol=2e-5 * np.sqrt(B * H * H), rtol=1e-4)
        np.testing.assert_allclose(npy(bn.bias.grad), ref_bn.bias.grad.float().numpy(), atol=2e-5 * np.sqrt(B * H * H), rtol=1e-4)
        # ---- and the unfused composition gives the same tensor (same kernels' arithmetic, BN by MIOpen): values only
        bn2 = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn2.weight.copy_(cu(gam, dev)); bn2.bias.copy_(cu(bet, dev))
        y2 = torch.relu(act(bn2(z.detach())))
        assert float((y2 - y.detach()).abs().max()) <= 1.0 / n + 1e-6
    finally:
        config.args.abitW = old


# (6, 4, 3): F = 36, F % 32 != 0 - the lanes beyond F work on the clamped last column and must take ITS channel (round-3 advisor)
@pytest.mark.parametrize("B,C,H,k", [(28, 256, 14, 8), (28, 2048, 7, 8), (6, 64, 8, 4), (28, 256, 56, 8), (6, 4, 3, 8), (5, 16, 3, 4)])
def test_bn_folded_small_batch_admm_site_vs_oracle(dev, B, C, H, k):
    """The Office bottleneck's tail with bn3 folded into the small-batch site kernels (fused.BNSite1Fn: alignq_bnq_stats ->
    alignq_site_partials_res_ab -> alignq_site_reduce_loss; backward alignq_site_bwd_apply_ab -> alignq_bnq_bwd_dx) against
    the C oracle's oq_bn_site_fwd / _bwd (pinned to torch.nn.BatchNorm2d + the golden-pinned site on the CPU), channels-last,
    eps 1e-5, batch 28 at the network's own shapes: y exact outside a near-tie band ((a, b) differ by ~1e-6 from the oracle's),
    D / loss / dz / dgamma / dbeta / dresidual / dalterD / dgamma_admm within 1e-5."""
    import alignq_amd.office as NO
    from alignq_amd import config
    rng = np.random.default_rng(B + C + H)
    r, eps = 2.0, 1e-5
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = k, B
    try:
        z0 = (rng.standard_normal((B, C, H, H)) * 1.4 + 0.3).astype(np.float32)
        res0 = np.maximum(rng.standard_normal((B, C, H, H)), 0).astype(np.float32)
        gy0 = (rng.standard_normal((B, C, H, H)) * 1e-2).astype(np.float32)
        gam = (rng.random(C) + 0.5).astype(np.float32)
        bet = (rng.standard_normal(C) * 0.2).astype(np.float32)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(cu(gam, dev)); bn.bias.copy_(cu(bet, dev))
        admm = NO.ADMM(B).to(dev)
        A0, G0 = npy(admm.alterD), npy(admm.gamma)
        act = NO.activation_quantize_fn2(k, "aligned", admm).to(dev)
        cl = lambda a: cu(a, dev).contiguous(memory_format=torch.channels_last)      # noqa: E731
        z, res = cl(z0).requires_grad_(True), cl(res0).requires_grad_(True)
        from alignq_amd import fused
        assert fused.bn_site_res_relu.__name__ and fused._bn_nhwc_ok(bn, z)
        y, loss = act.forward_bn_res_relu(bn, z, res)
        torch.autograd.backward([y, loss], [cl(gy0), torch.ones((), device=dev)])
        mem = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)).reshape(B, -1)     # noqa: E731
        unmem = lambda a: a.reshape(B, H, H, C).transpose(0, 3, 1, 2)                     # noqa: E731
        zm, rm_, gm = mem(z0), mem(res0), mem(gy0)
        ab_o, save_o, _ = O.bn_fold_ab(zm, C, 1, gam, bet, 1e-5)
        y_o, D_o, x_o = O.bn_site_fwd(zm, C, 1, ab_o, k, r, eps, residual=rm_, relu=True)
        n = 2 ** k - 1
        _, t_o, _ = O.act_quant_fwd(x_o, k, r, O.FORMULA_ADMM)
        frac = t_o.astype(np.float64) * n
        near_tie = np.abs(frac - np.floor(frac) - 0.5) < 2e-3
        diff = np.abs(mem(npy(y)) - y_o) * n
        assert np.all(diff[~near_tie] < 1e-3), int(np.count_nonzero(diff[~near_tie] >= 1e-3))
        assert np.all(diff[near_tie] <= 1.0 + 1e-3)
        np.testing.assert_allclose(npy(admm.D), D_o, atol=TOL, rtol=0)
        ol, odD, odA, odG = O.admm_loss(D_o, A0, G0, 0.2, 0.3)
        np.testing.assert_allclose(float(loss.detach()), ol, atol=TOL)
        dz_o, dg_o, db_o, dres_o, _ = O.bn_site_bwd(gm, odD, zm, C, 1, ab_o, save_o, mem(npy(y)), r, eps)
        np.testing.assert_allclose(mem(npy(z.grad)), dz_o, atol=TOL, rtol=1e-4)
        np.testing.assert_allclose(mem(npy(res.grad)), dres_o, atol=1e-7, rtol=0)
        scale = np.sqrt(B * H * H)
        np.testing.assert_allclose(npy(bn.weight.grad), dg_o, atol=2e-6 * scale, rtol=1e-4)
        np.testing.assert_allclose(npy(bn.bias.grad), db_o, atol=2e-6 * scale, rtol=1e-4)
        np.testing.assert_allclose(npy(admm.alterD.grad), odA, atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(npy(admm.gamma.grad), odG, atol=1e-7, rtol=1e-4)
        assert int(bn.num_batches_tracked) == 1
    finally:
        config.args.abitW, config.args.train_batch_size = old


@pytest.mark.parametrize("B,C,H", [(28, 256, 14), (6, 64, 8), (28, 2048, 7)])
def test_two_batch_slices_in_one_launch_equal_two_passes(dev, B, C, H):
    """The merged source + target traversal of the Office step (groups = 2: alignq_bnq_stats, alignq_site1_groups_fwd /
    _reduce_loss / _bwd, alignq_site_prep_fused_multi, alignq_bnq_bwd_dx with blockIdx.y = batch slice) against the module called
    twice, slice after slice, as dann_office/main.py:296-330 does: outputs, the summed loss, D (the LAST pass's), running
    statistics and every gradient bit for bit - the grouped launches run the same arithmetic on the same data."""
    import alignq_amd.office as NO
    from alignq_amd import config
    rng = np.random.default_rng(B * C + H)
    k = 8
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = k, B
    try:
        cl = lambda a: cu(a, dev).contiguous(memory_format=torch.channels_last)      # noqa: E731
        z0 = (rng.standard_normal((2 * B, C, H, H)) * 1.3 + 0.2).astype(np.float32)
        r0 = np.maximum(rng.standard_normal((2 * B, C, H, H)), 0).astype(np.float32)
        g0 = (rng.standard_normal((2 * B, C, H, H)) * 1e-2).astype(np.float32)
        outs = []
        for merged in (True, False):
            torch.manual_seed(5)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
            admm = NO.ADMM(B).to(dev)
            act = NO.activation_quantize_fn2(k, "aligned", admm).to(dev)
            z, res = cl(z0).requires_grad_(True), cl(r0).requires_grad_(True)
            if merged:
                y, loss = act.forward_bn_res_relu(bn, z, res, groups=2)
            else:
                ya, la = act.forward_bn_res_relu(bn, z[:B], res[:B])
                yb, lb = act.forward_bn_res_relu(bn, z[B:], res[B:])
                y, loss = torch.cat([ya, yb], 0), la + lb
            torch.autograd.backward([y, loss], [cl(g0), torch.ones((), device=dev)])
            outs.append([npy(t) for t in (y, loss.detach(), admm.D, bn.running_mean, bn.running_var, z.grad, res.grad, bn.weight.grad,
                                          bn.bias.grad, admm.alterD.grad, admm.gamma.grad)] + [int(bn.num_batches_tracked)])
        names = "y loss D running_mean running_var dz dres dgamma_bn dbeta_bn dalterD dgamma nbt".split()
        for nm, a, b in zip(names, outs[0], outs[1]):
            if nm in ("loss", "dgamma_bn", "dbeta_bn", "dalterD", "dgamma"):      # sums over the two slices: one rounding apart
                np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-7, err_msg=nm)      # (float + float vs one rounding of the double sum)
            elif nm == "nbt":
                assert a == b == 2
            else:
                assert bits_equal(a, b), nm
    finally:
        config.args.abitW, config.args.train_batch_size = old


def test_bn_alone_on_the_folded_family_vs_torch(dev):
    """fused.bn_only (the downsample branch's batch-norm: alignq_bnq_stats + _affine, backward alignq_bnq_bwd_dx) against
    torch.nn.BatchNorm2d in float64 on the CPU, C = 2048 (the 512-thread instantiation) and C = 256."""
    from alignq_amd import fused
    for B, C, H in ((28, 2048, 7), (28, 256, 56)):
        rng = np.random.default_rng(C)
        z0 = (rng.standard_normal((B, C, H, H)) * 2.0 - 0.7).astype(np.float32)
        g0 = rng.standard_normal((B, C, H, H)).astype(np.float32)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        ref = torch.nn.BatchNorm2d(C).double().train()
        with torch.no_grad():
            w = torch.rand(C) + 0.5; b_ = torch.randn(C) * 0.1
            bn.weight.copy_(w.to(dev)); bn.bias.copy_(b_.to(dev)); ref.weight.copy_(w.double()); ref.bias.copy_(b_.double())
        z = cu(z0, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = fused.bn_only(bn, z)
        y.backward(cu(g0, dev).contiguous(memory_format=torch.channels_last))
        zc = torch.from_numpy(z0).double().requires_grad_(True)
        yr = ref(zc)
        yr.backward(torch.from_numpy(g0).double())
        np.testing.assert_allclose(npy(y), yr.detach().float().numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(npy(z.grad), zc.grad.float().numpy(), atol=2e-5, rtol=1e-4)
        s = np.sqrt(B * H * H)
        np.testing.assert_allclose(npy(bn.weight.grad), ref.weight.grad.float().numpy(), atol=2e-5 * s, rtol=1e-4)
        np.testing.assert_allclose(npy(bn.bias.grad), ref.bias.grad.float().numpy(), atol=2e-5 * s, rtol=1e-4)
        np.testing.assert_allclose(npy(bn.running_var), ref.running_var.float().numpy(), atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 2b (G13)
def test_teacher_forced_office_bottleneck_on_the_hip_paths(dev):
    """G13: what the reference's own Bottleneck (inside the tiny DANN of G10; dann_office/model/resnet.py:131-156) fed to and
    got from its three quantiser sites, its batch-norms and its downsample branch.  Teacher-forced through the HIP paths:
    plain quantiser (bins exact outside the tie zone), quantiser + ReLU, the ADMM site with the eps corr (x_q, D, trans
    loss to 1e-5), and the batch-norm-FOLDED forms of all three plus the downsample batch-norm, which must reproduce the
    reference's tensors from the convolutions' outputs (exact outside a near-tie band; the block's output within one level)."""
    from tests.conftest import load_golden
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    g = load_golden("g13_office_bottleneck_sites")
    k, r = int(g["k"]), float(g["act_range"])
    n = 2 ** k - 1
    B = g["q3/x"].shape[0]
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = k, B
    try:
        def tie_mask(x, band):
            _, t, _ = O.act_quant_fwd(np.ascontiguousarray(x).reshape(-1), k, r, O.FORMULA_ADMM)
            frac = t.astype(np.float64) * n
            return (np.abs(frac - np.floor(frac) - 0.5) < band).reshape(x.shape)

        def check_levels(got, want, tie, what):
            diff = np.abs(got - want) * n
            assert np.all(diff[~tie] < 1e-3), (what, int(np.count_nonzero(diff[~tie] >= 1e-3)))
            assert np.all(diff[tie] <= 1.0 + 1e-3), what
        cl = lambda a: cu(a, dev).contiguous(memory_format=torch.channels_last)      # noqa: E731

        def make_bn(name):
            C = g[f"{name}/weight"].shape[0]
            bn = torch.nn.BatchNorm2d(C, eps=float(g[f"{name}/eps"]), momentum=float(g[f"{name}/momentum"])).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(cu(g[f"{name}/weight"], dev)); bn.bias.copy_(cu(g[f"{name}/bias"], dev))
            return bn
        # ---- the two plain sites
        for q, bnn in (("q1", "bn1"), ("q2", "bn2")):
            act = NO.activation_quantize_fn(k, str(g["stage"])).to(dev)
            x = g[f"{q}/x"]
            check_levels(npy(act(cu(x, dev))), g[f"{q}/xq"], tie_mask(x, 1e-4), q)
            check_levels(npy(act.forward_relu(cl(x))), np.maximum(g[f"{q}/xq"], 0), tie_mask(x, 1e-4), q + " relu")
            bn = make_bn(bnn)
            y = act.forward_bn_relu(bn, cl(g[f"{bnn}/z"]))          # folded: from the convolution's output
            check_levels(npy(y), np.maximum(g[f"{q}/xq"], 0), tie_mask(x, 2e-3), q + " folded")
            np.testing.assert_allclose(npy(bn.running_mean), g[f"{bnn}/running_mean"], atol=1e-6)
            np.testing.assert_allclose(npy(bn.running_var), g[f"{bnn}/running_var"], atol=1e-5, rtol=1e-5)
        # ---- the ADMM site
        admm = NO.ADMM(B).to(dev)
        with torch.no_grad():
            admm.alterD.copy_(cu(g["q3/alterD"], dev)); admm.gamma.copy_(cu(g["q3/gamma"], dev))
        act3 = NO.activation_quantize_fn2(k, str(g["stage"]), admm).to(dev)
        xq, loss = act3(cu(g["q3/x"], dev))
        check_levels(npy(xq), g["q3/xq"], tie_mask(g["q3/x"], 1e-4), "q3")
        np.testing.assert_allclose(npy(admm.D), g["q3/D"], atol=TOL)
        np.testing.assert_allclose(float(loss), float(g["q3/loss"]), atol=TOL)
        # ---- downsample batch-norm alone, then the folded tail: relu(act_q3(bn3(z))[0] + identity)
        bnd = make_bn("bnd")
        idn = fused.bn_only(bnd, cl(g["bnd/z"]))
        np.testing.assert_allclose(npy(idn), g["bnd/out"], atol=3e-5, rtol=1e-5)
        bn3 = make_bn("bn3")
        out, loss2 = act3.forward_bn_res_relu(bn3, cl(g["bn3/z"]), cl(g["bnd/out"]))
        tie3 = tie_mask(g["q3/x"], 2e-3)
        diff = np.abs(npy(out) - g["block/out"]) * n
        assert np.all(diff[~tie3] < 2e-3) and np.all(diff[tie3] <= 1.0 + 2e-3)
        np.testing.assert_allclose(npy(admm.D), g["q3/D"], atol=TOL)
        np.testing.assert_allclose(float(loss2), float(g["q3/loss"]), atol=TOL)
    finally:
        config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 9
def test_head_ticket_is_idempotent(dev):
    """The classifier head's in-kernel batch mean (fused.HeadCEFn: the workgroup whose arrival ticket is last adds the
    per-sample losses and re-arms the counter): 20 launches on the same inputs, on two different streams' counters, give the
    same bits every time and the PyTorch mean cross-entropy."""
    from alignq_amd.fused import HeadCEFn, _head_counter
    torch.manual_seed(2)
    B, C, H, K = 128, 64, 8, 10
    feat = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    lin = torch.nn.Linear(C, K).to(dev)
    y = torch.randint(0, K, (B,), device=dev)
    want = torch.nn.functional.cross_entropy(lin(feat.mean((2, 3))), y)
    outs = []
    side = torch.cuda.Stream()
    for it in range(20):
        if it % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _, ce = HeadCEFn.apply(feat, lin.weight, lin.bias, y)
            torch.cuda.current_stream().wait_stream(side)
        else:
            _, ce = HeadCEFn.apply(feat, lin.weight, lin.bias, y)
        outs.append(float(ce))
    torch.cuda.synchronize()
    assert all(o == outs[0] for o in outs), outs
    np.testing.assert_allclose(outs[0], float(want), rtol=1e-5)
    with torch.cuda.stream(side):
        c_side = _head_counter(dev)
    # (two words since round 6: the head's workgroups, and the sites closed inside alignq_site_reduce_loss_multi_head's launch)
    assert not _head_counter(dev).any() and not c_side.any() and c_side.data_ptr() != _head_counter(dev).data_ptr()


# ------------------------------------------------------------------------------------------------ VERDICT r2 item 6
@pytest.mark.parametrize("b,n_par,n_sites,dim", [(128, 65, 21, 128), (80, 9, 3, 128), (128, 70, 21, 128), (128, 12, 23, 128),
                                                 (200, 9, 3, 256)])
def test_sgd_and_admm_update_in_one_launch_equal_the_two_steps(dev, b, n_par, n_sites, dim):
    """optimizer.sgd_admm_step (alignq_sgd_admm_step_multi: the SGD step and the ADMM update as roles of one launch) leaves the
    bits of SGD.step followed by ADMM_OPT.step in every parameter, momentum buffer, rewritten p.grad, alterD and gamma; b = 80 is
    a short batch in dim 128 (the padded form, utils/optimizer.py:95-103); 70 parameters / 23 sites exceed one argument block
    (the entry point then issues the two launches itself); dim = 256 (round 4): above 128 the call runs the two steps as they
    are (the many-workgroup ADMM update, alignq_admm_update_ws)."""
    from alignq_amd import config
    from alignq_amd.admm import ADMM
    from alignq_amd.optimizer import ADMM_OPT, SGD, sgd_admm_step
    old = config.args.bitW
    config.args.bitW = 4
    try:
        def world():
            g = torch.Generator(device="cpu").manual_seed(5)
            sizes = [432, 36864, 16, 640, 2304, 9216, 10, 18432, 64][:n_par] + [int(s) for s in
                                                                              torch.randint(1, 5000, (max(0, n_par - 9),), generator=g)]
            ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in sizes]
            for p in ps:
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            idx = [0, 1, 4]
            cdfs = [torch.rand(ps[i].shape, generator=g).to(dev) - 0.5 for i in idx]
            pdfs = [torch.rand(ps[i].shape, generator=g).to(dev) for i in idx]
            admms = []
            for _ in range(n_sites):
                m = ADMM(dim).to(dev)
                with torch.no_grad():
                    m.alterD.copy_(torch.rand(dim, dim, generator=g))
                    m.gamma.copy_(torch.rand(dim, dim, generator=g))
                m.alterD.grad = torch.zeros_like(m.alterD)
                m.gamma.grad = torch.zeros_like(m.gamma)
                m.D = (torch.randn(b, b, generator=g) * 0.05).to(dev)
                admms.append(m)
            sgd = SGD(ps, lr=0.1, momentum=0.9, weight_decay=5e-4)
            aps = [p for m in admms for p in (m.alterD, m.gamma)]
            opt = ADMM_OPT(aps)
            sargs = (idx, cdfs, pdfs, 0.7, 1.3)
            aargs = (list(range(0, 2 * n_sites, 2)), list(range(1, 2 * n_sites, 2)), [m.D for m in admms],
                     [m.alterD for m in admms], [m.gamma for m in admms], [m.mu for m in admms], [m.rho for m in admms])
            return ps, admms, sgd, opt, sargs, aargs

        def state(ps, admms, sgd):
            return ([p.detach().clone() for p in ps] + [p.grad.clone() for p in ps] +
                    [sgd.state[p]["momentum_buffer"].clone() for p in ps] +
                    [t.detach().clone() for m in admms for t in (m.alterD, m.gamma)])

        ps, admms, sgd, opt, sargs, aargs = world()
        for it in range(2):                                  # second round: existing momentum buffers
            sgd.step(*sargs)
            opt.step(*aargs)
        want = state(ps, admms, sgd)
        ps, admms, sgd, opt, sargs, aargs = world()
        for it in range(2):
            sgd_admm_step(sgd, sargs, opt, aargs)
        got = state(ps, admms, sgd)
        assert len(want) == len(got)
        for w, g_ in zip(want, got):
            assert torch.equal(w, g_)
    finally:
        config.args.bitW = old


@pytest.mark.parametrize("units,k", [([3, 3, 3], 8), ([2, 1, 2], 4)])
def test_filter_gradient_reduction_as_a_filler_role_leaves_the_same_bits(dev, units, k):
    """alignq_conv3x3_nhwc_bwd_fill / alignq_site_bwd_apply_bn_fill: the slab reductions of earlier convolutions ride in the
    following convolutions' and narrow sites' backward launches (fused.DeferredWgrads.take / take_site); every parameter
    gradient of a full-batch ResNet step must equal, bit for bit, the step that leaves all reductions to the closing
    alignq_conv3x3_wgrad_reduce_multi."""
    from alignq_amd import config, fused
    from tests.test_gpu_bench_path import _run_model_step
    old = fused._WGRAD_FILL, fused._WGRAD_FILL_SITE
    try:
        fused._WGRAD_FILL, fused._WGRAD_FILL_SITE = {}, 0
        a, _ = _run_model_step(dev, units, k, 128, deferred=True)
        for conv_fill, site_fill in (({16: 4, 32: 1, 64: 2}, 0), ({16: 2}, 2), ({}, 4)):
            fused._WGRAD_FILL, fused._WGRAD_FILL_SITE = conv_fill, site_fill
            taken = []
            orig, orig_s = fused.DeferredWgrads.take, fused.DeferredWgrads.take_site

            def spy(self, C):
                out = orig(self, C)
                taken.append(len(out))
                return out

            def spy_s(self, B, F):
                out = orig_s(self, B, F)
                taken.append(len(out))
                return out
            fused.DeferredWgrads.take, fused.DeferredWgrads.take_site = spy, spy_s
            try:
                b, _ = _run_model_step(dev, units, k, 128, deferred=True)
            finally:
                fused.DeferredWgrads.take, fused.DeferredWgrads.take_site = orig, orig_s
            assert sum(taken) >= sum(units[1:]) and max(taken) >= 2     # the filler roles really ran, several items at once
            assert a["grads"].keys() == b["grads"].keys()
            for n_ in a["grads"]:
                assert np.array_equal(a["grads"][n_], b["grads"][n_]), (n_, conv_fill, site_fill)
    finally:
        fused._WGRAD_FILL, fused._WGRAD_FILL_SITE = old
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


@pytest.mark.parametrize("units,k", [([3, 3, 3], 8), ([1, 2, 1], 4)])
def test_site_slab_reduction_as_a_filler_role_leaves_the_same_bits(dev, units, k):
    """alignq_site_partials_bn_fill: the slab reduction + ADMM loss of earlier sites ride in the forward launches of the later,
    narrower sites (fused.DeferredLosses.take_fill).  D of every site, the loss total, the logits and every gradient of a
    full-batch ResNet step must equal, bit for bit, the step that leaves all reductions to alignq_site_reduce_loss_multi."""
    from alignq_amd import config, fused
    from tests.test_gpu_bench_path import _run_model_step
    old = fused._SITE_FILL
    try:
        fused._SITE_FILL = 0
        a, _ = _run_model_step(dev, units, k, 128, deferred=True)
        fused._SITE_FILL = 3
        b, step_b = _run_model_step(dev, units, k, 128, deferred=True)
        n_filled = sum(1 for r in step_b._deferred.records if r.reduced)
        assert n_filled >= len(step_b._deferred.records) - 2 - units[0] * 2       # all but the last site(s) rode along
        assert np.array_equal(a["logits"], b["logits"]) and a["ce"] == b["ce"] and a["tl"] == b["tl"]
        for i, (da, db) in enumerate(zip(a["D"], b["D"])):
            assert np.array_equal(da, db), f"site {i} D"
        assert a["grads"].keys() == b["grads"].keys()
        for n_ in a["grads"]:
            assert np.array_equal(a["grads"][n_], b["grads"][n_]), n_
    finally:
        fused._SITE_FILL = old
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


def test_plain_quantiser_with_streamed_outputs_vs_oracle(dev):
    """From 2^25 elements on the plain quantiser stores its outputs non-temporally (quant_kernels.hip: stream_out); the whole
    33.5 M-element result, ragged tail included, against the C oracle: forward bit for bit, backward to 1e-5."""
    from alignq_amd import ops
    n = (1 << 25) + 5
    rng = np.random.default_rng(77)
    x = (rng.standard_normal(n) * 1.3).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    xt = cu(x, dev).requires_grad_(True)
    y = ops.ActQuantFn.apply(xt, 8, 2.0, 0)
    y.backward(cu(g, dev))
    oq, _, _ = O.act_quant_fwd(x, 8, 2.0, 0)
    assert bits_equal(npy(y), oq)
    np.testing.assert_allclose(npy(xt.grad), O.act_quant_bwd(g, x, 2.0), rtol=1e-5, atol=1e-6)
