"""GPU tests added in round 4 (all through the C ABI via alignq_amd.ops / the Python mirror)."""
import numpy as np
import pytest
import torch

from tests import oracle_c as O  # noqa: F401

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


# ------------------------------------------------------------------------------------------------ ADVICE r3 (medium 2)
@pytest.mark.parametrize("n_groups", [2, 3])
def test_sgd_admm_step_with_several_sgd_groups_on_the_first_step(dev, n_groups):
    """optimizer.sgd_admm_step with MORE than one SGD parameter group (the Office step has three) falls back to SGD.step +
    ADMM_OPT.step.  Round 3 gathered the SGD items first - creating the momentum buffers with torch.empty_like - dropped them
    and gathered again: the second gather saw existing buffers (first = 0) and the kernel read uninitialised memory
    (utils/optimizer.py:231-243: the first step SETS buf = d_p).  The caching allocator is primed with NaN-filled blocks of the
    buffers' sizes so that such a read cannot pass by luck."""
    from alignq_amd import config
    from alignq_amd.admm import ADMM
    from alignq_amd.optimizer import ADMM_OPT, SGD, sgd_admm_step
    old = config.args.bitW
    config.args.bitW = 4
    try:
        sizes = [432, 36864, 16, 640, 2304, 9216][: 2 * n_groups]

        def world():
            g = torch.Generator(device="cpu").manual_seed(11)
            ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in sizes]
            for p in ps:
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            m = ADMM(16).to(dev)
            with torch.no_grad():
                m.alterD.copy_(torch.rand(16, 16, generator=g))
                m.gamma.copy_(torch.rand(16, 16, generator=g))
            m.alterD.grad, m.gamma.grad = torch.zeros_like(m.alterD), torch.zeros_like(m.gamma)
            m.D = (torch.randn(16, 16, generator=g) * 0.05).to(dev)
            groups = [dict(params=ps[2 * i:2 * i + 2], lr=0.1 / (i + 1)) for i in range(n_groups)]
            sgd = SGD(groups, lr=0.1, momentum=0.9, weight_decay=5e-4)
            opt = ADMM_OPT([m.alterD, m.gamma])
            sargs = ([], [], [], 1.0, 4.0)
            aargs = ([0], [1], [m.D], [m.alterD], [m.gamma], [m.mu], [m.rho])
            return ps, m, sgd, opt, sargs, aargs

        def poison():
            junk = [torch.full((s,), float("nan"), device=dev) for s in sizes for _ in range(2)]
            del junk

        ps, m, sgd, opt, sargs, aargs = world()
        sgd.step(*sargs)
        opt.step(*aargs)
        want = [npy(p) for p in ps] + [npy(sgd.state[p]["momentum_buffer"]) for p in ps] + [npy(m.alterD), npy(m.gamma)]
        ps, m, sgd, opt, sargs, aargs = world()
        poison()
        sgd_admm_step(sgd, sargs, opt, aargs)
        got = [npy(p) for p in ps] + [npy(sgd.state[p]["momentum_buffer"]) for p in ps] + [npy(m.alterD), npy(m.gamma)]
        for w, g_ in zip(want, got):
            assert np.isfinite(g_).all()
            assert np.array_equal(w, g_)
    finally:
        config.args.bitW = old


# ------------------------------------------------------------------------------------------------ ADVICE r3 (low)
def test_folded_batchnorm_eligibility_is_per_batch_slice(dev):
    """fused._bn_nhwc_ok with groups: every slice needs >= 2 values per channel and the batch must divide (alignq_bnq_* take
    P = (B / groups) * H * W and return ALIGNQ_EINVAL below 2); bn_only / bn_act_relu then fall back to the per-slice modules."""
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    bn = torch.nn.BatchNorm2d(8).to(dev).train()
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
    # (a slice with fewer than two values per channel needs H * W == 1, and a 1 x 1 tensor is never channels-last to torch: the
    #  divisibility is the reachable half of the finding)
    z = cl(torch.randn(2, 8, 1, 2, device=dev))
    assert fused._bn_nhwc_ok(bn, z) and fused._bn_nhwc_ok(bn, z, 2) and not fused._bn_nhwc_ok(bn, z, 4)
    z3 = cl(torch.randn(3, 8, 4, 4, device=dev))
    assert fused._bn_nhwc_ok(bn, z3) and not fused._bn_nhwc_ok(bn, z3, 2)
    with pytest.raises(ValueError):
        fused.bn_only(bn, z3, groups=2)
    z4 = cl(torch.randn(4, 8, 4, 4, device=dev))
    old = config.args.abitW
    config.args.abitW = 8
    try:
        act = NO.activation_quantize_fn(8, "aligned").to(dev)
        bn2 = torch.nn.BatchNorm2d(8).to(dev).train()
        y = fused.bn_act_relu(bn, act, z4, 0, relu=True, groups=2)
        want = torch.cat([torch.relu(act(bn2(z4[:2]))), torch.relu(act(bn2(z4[2:])))], 0)
        assert float((y - want).abs().max()) <= 2.0 / 255 + 1e-6          # tie-zone flips between the folded and MIOpen's BN
        np.testing.assert_allclose(npy(bn.running_mean), npy(bn2.running_mean), atol=1e-6)
    finally:
        config.args.abitW = old


def test_office_dual_traversal_without_admm_terms(dev):
    """ResNet.forward(groups=2) with sites that return the number 0 as their loss (abitW == 32: W-only quantisation,
    dann_office/model/quantization.py:121-123) - torch.stack on floats raised a TypeError in round 3."""
    from alignq_amd import config
    from alignq_amd.resnet_office import DANN, Bottleneck, ResNet
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW, config.args.abitW, config.args.train_batch_size = 8, 32, 4
    try:
        torch.manual_seed(3)
        m = DANN(lambda w, a, s: ResNet(w, a, s, Bottleneck, [1, 1, 1, 1]), 8, 32, "aligned").to(dev).train()
        x = torch.randn(8, 3, 64, 64, device=dev)
        feat, tl = m.feature(x, groups=2)
        assert feat.shape[0] == 8 and float(tl) == 0.0
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ VERDICT r3 item 2
@pytest.mark.parametrize("tree,formula", [("admm", 0), ("cdf", 1)])
def test_hip_bins_vs_reference_at_scale(dev, tree, formula, record_property):
    """G3L on the HIP path: alignq_act_quant_fwd's int32 bins, the packed (int8 / int16 / uint8) indices of
    alignq_act_quant_fwd_packed and the ADMM site's x_q against the REFERENCE's integer bins on 2^20 captured elements per bit
    width: equal outside the reference's own tie zone, at most one off inside, flips <= 16 per 2^20 and reported; and the HIP
    bins equal the C oracle's on every element (shared NERF32 specification)."""
    from alignq_amd import _lib as L, ops
    from tests.test_oracle_c import g3l_check, load_g3l
    assert (O.FORMULA_ADMM, O.FORMULA_CDF) == (0, 1)
    x, g = load_g3l(tree)
    r = float(g["act_range"])
    xd = cu(x, dev)
    for k in (2, 4, 8):
        n = 2 ** k - 1
        xq = torch.empty_like(xd)
        bins = torch.empty(xd.shape, dtype=torch.int32, device=dev)
        L.check(L.load().alignq_act_quant_fwd(L.ptr(xd), L.ptr(xq), L.ptr(bins), xd.numel(), k, r, formula, L.stream_ptr()),
                "alignq_act_quant_fwd")
        n_tie, flips = g3l_check(npy(bins), g, k)
        record_property(f"g3l_hip_{tree}_k{k}", {"tie_zone": n_tie, "flips": flips})
        print(f"G3L HIP {tree} k={k}: {n_tie} in the tie zone, {flips} bins differ from the reference")
        assert flips <= 16
        oq, _, obins = O.act_quant_fwd(x, k, r, formula)
        assert np.array_equal(npy(bins), obins) and np.array_equal(npy(xq).view(np.uint32), oq.view(np.uint32))
        packed = ops.act_quant_pack(xd, k, r, formula)
        assert np.array_equal(npy(packed).astype(np.int32), obins)
        if tree == "admm":          # the fused ADMM site quantises with the same arithmetic: [128, 8192] as one site
            xs = xd.view(128, -1)
            A = torch.rand(128, 128, device=dev)
            xq_s, _, _ = ops.SiteFn.apply(xs, A, A.clone(), k, r, 0.0, 0.2, 0.3)
            assert np.array_equal(np.rint(npy(xq_s).astype(np.float64).ravel() * n).astype(np.int32), obins)


# ------------------------------------------------------------------------------------------------ round 4: ReLU mask as bits
@pytest.mark.parametrize("Bt,C,H,groups", [(6, 4, 3, 1), (6, 4, 3, 2), (56, 64, 28, 2), (10, 2048, 2, 2), (7, 16, 5, 1)])
def test_bnq_backward_with_the_relu_mask_as_bits_equals_the_fp32_y_form(dev, Bt, C, H, groups):
    """alignq_bnq_fwd's one-bit-per-element ReLU mask (round 4) against the mask taken from the fp32 y: the same dz, dgamma,
    dbeta bit for bit (the bits ARE [y > 0]), at vec counts that are not multiples of a 64-quad chunk, with batch slices, at
    512-thread channel counts; and the bits themselves against y."""
    from alignq_amd import _lib as L
    lib = L.load()
    st, p = L.stream_ptr(), L.ptr
    torch.manual_seed(Bt * C + H)
    P = (Bt // groups) * H * H
    z = (torch.randn(Bt, H, H, C, device=dev) * 1.3 + 0.1)
    g = torch.randn(Bt, H, H, C, device=dev) * 0.01
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    ab, save = torch.empty(groups, 2, C, device=dev), torch.empty(groups, 2, C, device=dev)
    y = torch.empty_like(z)
    ws = torch.empty(lib.alignq_bnq_ws_bytes(C, groups), dtype=torch.uint8, device=dev)
    nbytes = lib.alignq_bnq_mask_bytes(P, C, groups)
    nvec = P * C // 4
    assert nbytes == groups * ((nvec + 63) // 64) * 32
    mask = torch.full((nbytes,), 0xAA, dtype=torch.uint8, device=dev)
    L.check(lib.alignq_bnq_fwd(p(z), P, C, groups, p(gam), p(bet), p(rm), p(rv), p(nbt), 0.1, 1e-5, 8, 2.0, 0, 1, None, p(ab), p(save), p(y),
                               p(mask), p(ws), st), "alignq_bnq_fwd")
    # the bits: per group and chunk of 64 quads four little-endian 64-bit words, word = quad component, bit = quad within the chunk
    yq = npy(y).reshape(groups, nvec, 4) > 0
    mb = np.unpackbits(npy(mask).reshape(groups, -1, 4, 8), axis=-1, bitorder="little").reshape(groups, -1, 4, 64)
    got = mb.transpose(0, 1, 3, 2).reshape(groups, -1, 4)[:, :nvec]
    assert np.array_equal(got, yq)
    outs = []
    for use_bits in (False, True):
        dz, dg, db = torch.empty_like(z), torch.empty(C, device=dev), torch.empty(C, device=dev)
        L.check(lib.alignq_bnq_bwd(p(g), p(z), None if use_bits else p(y), p(mask) if use_bits else None, p(ab), p(save), P, C, groups,
                                   2.0, 1, p(dz), None, p(dg), p(db), p(ws), st), "alignq_bnq_bwd")
        outs.append((npy(dz), npy(dg), npy(db)))
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][0]).max() > 0


# ------------------------------------------------------------------------------------------------ round 4: GradFork
@pytest.mark.parametrize("B,C,H,groups", [(28, 256, 14, 2), (6, 64, 8, 1), (10, 2048, 2, 2), (6, 16, 4, 1)])
def test_forked_block_input_gives_the_bits_of_autograds_own_sum(dev, B, C, H, groups):
    """fused.GradFork: where a folded site's output feeds the next block's convolution branch AND its shortcut, the two gradients
    reach the site kernel as two pointers (g + g2 on load) instead of being added by an elementwise pass of autograd's: every
    gradient of the producing site (dz, d residual, dgamma, dbeta, dalterD, dgamma_admm) bit for bit, with batch slices, with
    C % 32 != 0 (the form without the second pointer adds on the host side)."""
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    Bt = B * groups
    old = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = 8, B
    try:
        torch.manual_seed(B + C)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
        z0 = cl(torch.randn(Bt, C, H, H, device=dev) * 1.2 + 0.2)
        r0 = cl(torch.relu(torch.randn(Bt, C, H, H, device=dev)))
        w1, w2 = cl(torch.randn(Bt, C, H, H, device=dev) * 0.01), cl(torch.randn(Bt, C, H, H, device=dev) * 0.01)
        outs = []
        for fork in (False, True):
            torch.manual_seed(1)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
            admm = NO.ADMM(B).to(dev)
            act = NO.activation_quantize_fn2(8, "aligned", admm).to(dev)
            z, res = z0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            got = fused.bn_site_res_relu(bn, act, z, res, 1e-5, groups)
            assert got is not None
            y, loss = got
            assert getattr(y, "_alignq_site_tok", None) is not None
            a, b = fused.fork_block_input(y) if fork else (y, y)
            if fork:
                assert a is not y and b is not y
            ((a * w1).sum() + (b * w2).sum() + loss).backward()
            assert "extra" not in y._alignq_site_tok
            outs.append([npy(t) for t in (z.grad, res.grad, bn.weight.grad, bn.bias.grad, admm.alterD.grad, admm.gamma.grad)])
        for u, v in zip(*outs):
            assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
    finally:
        config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ VERDICT r3 item 3
def test_cdf_only_resnet20_full_size_step_on_the_fast_path(dev, monkeypatch):
    """Configuration 1 (cdf_alignment/resnet-20-cifar-10: model/resnet.py:63-79,134, main.py:269-315) at FULL size - ResNet-20,
    batch 128, 8W/8A - on the HIP fast path: channels-last, Conv2d_Q's body convolutions on alignq_conv3x3_nhwc, every
    `act_q(bn(.))` folded (fused.bn_act_relu -> alignq_bnq_fwd / _bwd, formula 1).  Three sites (stem; a block's bn0 with the
    ReLU; the last block's bn1 with the shortcut and the ReLU in the same pass) are checked teacher-forced against the C oracle on the tensors the step itself produced:
    x_q bit-exact outside a near-tie band (the device's (a, b) differ by ~1e-6 from the oracle's), at most one level inside;
    the loss is finite, every site really took the folded path, and the captured HIP graph reproduces eager iterations."""
    import alignq_amd.resnet as RN
    from alignq_amd import config, fused
    from alignq_amd.train_step import TrainStep
    old = (config.args.bitW, config.args.abitW, config.args.train_batch_size)
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    try:
        k, r, n = 8, float(config.args.act_range), 255
        calls = []
        real = fused.bn_act_relu

        def spy(bn, act, z, formula, relu=True, groups=1, residual=None):
            assert formula == 1 and fused.bnq_fusable(bn, act, z, groups)
            out = real(bn, act, z, formula, relu=relu, groups=groups, residual=residual)
            calls.append((bn, z.detach(), out.detach(), relu, None if residual is None else residual.detach()))
            return out
        monkeypatch.setattr(RN, "bn_act_relu", spy)

        def make():
            torch.manual_seed(3)
            return RN.resnet20_quant(8, 8, tree="cdf").to(dev).train()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(128, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (128,), generator=g).to(dev)
        net = make()
        step = TrainStep(net, channels_last=True)
        assert not step.admms
        # the batch-norm parameters / statistics the recorded calls saw (the optimizer step changes them afterwards)
        before = {id(m): (npy(m.weight), npy(m.bias)) for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)}
        logits, ce, tl = step(x, y)
        torch.cuda.synchronize()
        assert torch.isfinite(logits).all() and torch.isfinite(ce) and (tl is None or float(tl) == 0.0)
        assert len(calls) == 21                                  # 1 stem + 9 blocks x 2 + 2 shortcut sites
        for idx in (0, 1, len(calls) - 1):
            bn, z, out, relu, resid = calls[idx]
            B, C, H, W = z.shape
            gam, bet = before[id(bn)]
            zm = np.ascontiguousarray(npy(z).transpose(0, 2, 3, 1)).reshape(B, -1)
            ab_o, _, _ = O.bn_fold_ab(zm, C, 1, gam, bet, bn.eps)
            x_o = O.bn_apply(zm, C, 1, ab_o)
            q_o, c_o, _ = O.act_quant_fwd(x_o, k, r, O.FORMULA_CDF)
            if resid is not None:         # `out += shortcut` rides in the same pass (resnet.py:76-78)
                q_o = q_o + np.ascontiguousarray(npy(resid).transpose(0, 2, 3, 1)).reshape(B, -1)
            y_o = np.maximum(q_o, 0.0) if relu else q_o
            frac = c_o.astype(np.float64) * n
            near = np.abs(frac - np.floor(frac) - 0.5) < 2e-3
            got = np.ascontiguousarray(npy(out).transpose(0, 2, 3, 1)).reshape(B, -1)
            diff = np.abs(got - y_o) * n / (2.0 * r)             # in levels: x_q = r * (2 * bin / n - 1)
            assert np.all(diff[~near] == 0), (idx, int(np.count_nonzero(diff[~near])))
            assert np.all(diff[near] <= 1.0 + 1e-3)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
        # ---- graph == eager over three iterations from the same initial state: BIT FOR BIT (round 6; every kernel of the step reduces
        # in a fixed order, so the replay is the same computation as the eager iterations the oracle comparisons above ran on)
        from tests.test_gpu_round6 import differing, full_state
        monkeypatch.setattr(RN, "bn_act_relu", real)
        n1, n2 = make(), make()
        s1, s2 = TrainStep(n1, channels_last=True), TrainStep(n2, channels_last=True)
        for _ in range(3):
            l1, c1, _ = s1(x, y)
        s2.capture(x, y, warmup=2)          # two real iterations, then the captured third
        l2, c2, _ = s2(x, y)
        torch.cuda.synchronize()
        assert np.array_equal(npy(c1), npy(c2)) and np.array_equal(npy(l1), npy(l2))
        bad = differing(full_state(n1, s1, []), full_state(n2, s2, []))
        assert not bad, bad[:6]
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size = old


# ------------------------------------------------------------------------------------------------ round 4: residual in the fold
@pytest.mark.parametrize("B,C,H,relu", [(128, 16, 32, True), (100, 64, 8, True), (28, 128, 28, True), (9, 8, 5, False)])
def test_bn_folded_plain_quantiser_with_residual_equals_the_composition(dev, B, C, H, relu):
    """`out = act_q1(bn1(z)); out += shortcut; out = F.relu(out)` of the CDF-only block (cdf_alignment/resnet-20-cifar-10/model/
    resnet.py:73-78) as ONE folded chain (fused.bn_act_relu(residual=): the shortcut joins in the apply pass, its gradient is the
    apply pass's second output) against the fold WITHOUT the residual followed by torch's add and ReLU: the same bits forward and
    backward (same kernels' arithmetic on the same data, fp32 add and max are exact operations), small (in-kernel finalisation) and
    large sites alike."""
    import alignq_amd.cdf_alignment as NC
    from alignq_amd import config, fused
    old = config.args.abitW
    config.args.abitW = 8
    try:
        torch.manual_seed(B + C + H)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
        z0 = cl(torch.randn(B, C, H, H, device=dev) * 1.3 + 0.2)
        r0 = cl(torch.randn(B, C, H, H, device=dev))
        g0 = cl(torch.randn(B, C, H, H, device=dev) * 0.01)
        outs = []
        for folded in (False, True):
            torch.manual_seed(5)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
            act = NC.activation_quantize_fn(8, "second").to(dev)
            z, res = z0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            if folded:
                y = fused.bn_act_relu(bn, act, z, 1, relu=relu, residual=res)
            else:
                y = fused.bn_act_relu(bn, act, z, 1, relu=False) + res
                if relu:
                    y = torch.relu(y)
            y.backward(g0)
            outs.append([npy(t) for t in (y, z.grad, res.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)])
        for u, v in zip(*outs):
            assert np.array_equal(u, v)            # by value: a ReLU may return either zero for a negative zero
    finally:
        config.args.abitW = old


def test_small_site_finalisation_inside_the_apply_kernels_equals_the_launches(dev):
    """Small single-group sites finalise their batch-norm statistics inside the apply kernels (bnq_kernels.hip: fin_small); sites
    with several batch slices run the finalisation launches.  One seeded problem through both forms in one process (round 5: no
    environment switch any more): the slice alone (in-kernel finalisation) against the same slice twice as two groups (launches):
    y, (a, b), the saved statistics, dz of group 0, dgamma / dbeta (twice the single slice's) - the partial counts differ, so the
    sums agree to double rounding and the fp32 results to 1 ulp of the statistics: 1e-6 relative, y at bin-flip scale."""
    from alignq_amd import _lib as L
    lib = L.load()
    st, p = L.stream_ptr(), L.ptr
    torch.manual_seed(0)
    B, C, H = 128, 16, 32
    P = B * H * H
    z1 = torch.randn(B, H, H, C, device=dev) * 1.3 + 0.1
    g1 = torch.randn(B, H, H, C, device=dev) * 0.01
    r1 = torch.randn(B, H, H, C, device=dev)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    res = {}
    for G in (1, 2):
        z, g, rs = (torch.cat([t] * G, 0).contiguous() for t in (z1, g1, r1))
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
        y, dz, dres = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
        mask = torch.zeros(lib.alignq_bnq_mask_bytes(P, C, G), dtype=torch.uint8, device=dev)
        L.check(lib.alignq_bnq_fwd(p(z), P, C, G, p(gam), p(bet), p(rm), p(rv), p(nbt), 0.1, 1e-5, 8, 2.0, 1, 1, p(rs), p(ab), p(save),
                                   p(y), p(mask), p(ws), st), "fwd")
        L.check(lib.alignq_bnq_bwd(p(g), p(z), None, p(mask), p(ab), p(save), P, C, G, 2.0, 1, p(dz), p(dres), p(dg), p(db), p(ws), st),
                "bwd")
        torch.cuda.synchronize()
        res[G] = dict(y=npy(y[:B]), ab=npy(ab[0]), save=npy(save[0]), nbt=int(nbt), dz=npy(dz[:B]), dres=npy(dres[:B]),
                      dg=npy(dg) / G, db=npy(db) / G)
    a, b = res[2], res[1]
    assert a["nbt"] == 2 and b["nbt"] == 1
    for key in ("ab", "save", "dg", "db"):
        np.testing.assert_allclose(b[key], a[key], rtol=2e-6, atol=1e-7, err_msg=key)
    np.testing.assert_allclose(b["dz"], a["dz"], rtol=1e-4, atol=1e-7)
    lev = 4.0 / 255                                               # one level of x_q = r * (2 bin / n - 1)
    dy = np.abs(b["y"] - a["y"])
    assert dy.max() <= lev * 1.001 and np.count_nonzero(dy) < 1e-4 * dy.size      # a 1-ulp (a, b) moves a few near-tie elements
    same = dy == 0
    assert np.array_equal(b["dres"][same], a["dres"][same])


# ------------------------------------------------------------------------------------------------ round 4: per-column BN sums
@pytest.mark.parametrize("B,C,H,groups", [(6, 64, 8, 1), (28, 256, 14, 2), (5, 16, 3, 1), (28, 2048, 7, 2)])
def test_small_batch_site_backward_with_per_column_bn_sums_equals_the_sums_pass(dev, B, C, H, groups):
    """alignq_site1_groups_bwd_bn (the site kernel leaves sum_b dx and sum_b dx * zhat per feature column, zhat formed from
    x = gamma * zhat + beta; a small reduction over the columns replaces alignq_bnq_bwd_dx's pass over dx and z) against
    alignq_site1_groups_bwd + alignq_bnq_bwd_dx on the same inputs: dx is the same kernel's (bit-identical), so dz / dgamma / dbeta
    may differ by the summation order and by the one rounding of x = a*z + b only.  Two channels have gamma == 0: x carries no
    trace of z there and the finalisation must sum them from dx and z directly (their dgamma is NOT zero).  The oracle comparison
    of the default path is test_bn_folded_small_batch_admm_site_vs_oracle."""
    import alignq_amd.office as NO
    from alignq_amd import config, fused
    Bt = B * groups
    old = (config.args.abitW, config.args.train_batch_size, fused._S1_BN_COLS)
    config.args.abitW, config.args.train_batch_size = 8, B
    try:
        torch.manual_seed(B + C)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
        z0 = cl(torch.randn(Bt, C, H, H, device=dev) * 1.2 + 0.2)
        r0 = cl(torch.relu(torch.randn(Bt, C, H, H, device=dev)))
        g0 = cl(torch.randn(Bt, C, H, H, device=dev) * 0.01)
        outs = []
        for cols in (False, True):
            fused._S1_BN_COLS = cols
            torch.manual_seed(1)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
                bn.weight[1] = 0.0
                bn.weight[C - 3] = 0.0
                # round 5: tiny NON-zero gamma beside a beta of ordinary size - x = gamma * zhat + beta keeps zhat only to
                # eps * |beta| / |gamma| (1e-8: nothing of it, 1e-5: 3e-3): such channels must take the direct sum as well
                # (alignq_bn_col_ill: |gamma| < 1e-2 |beta|)
                for ch, gv in ((2, 1e-8), (3, 1e-5), (4, 1e-3), (5, -1e-6)):
                    bn.weight[ch] = gv
                    bn.bias[ch] = 0.5
            admm = NO.ADMM(B).to(dev)
            act = NO.activation_quantize_fn2(8, "aligned", admm).to(dev)
            z, res = z0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            y, loss = fused.bn_site_res_relu(bn, act, z, res, 1e-5, groups)
            torch.autograd.backward([y, loss], [g0, torch.ones((), device=dev)])
            outs.append([npy(t) for t in (z.grad, res.grad, bn.weight.grad, bn.bias.grad)])
        (dz_a, dr_a, dg_a, db_a), (dz_b, dr_b, dg_b, db_b) = outs
        assert np.array_equal(dr_a, dr_b)
        assert abs(dg_a[1]) > 0 and abs(dg_a[C - 3]) > 0 and all(abs(dg_a[ch]) > 0 for ch in (2, 3, 4, 5))
        scale = float(np.abs(dg_a).max())
        np.testing.assert_allclose(dg_b, dg_a, rtol=2e-5, atol=2e-6 * scale)
        np.testing.assert_allclose(db_b, db_a, rtol=2e-5, atol=2e-6 * float(np.abs(db_a).max()))
        np.testing.assert_allclose(dz_b, dz_a, rtol=1e-4, atol=1e-6 * float(np.abs(dz_a).max()) + 1e-9)
    finally:
        config.args.abitW, config.args.train_batch_size, fused._S1_BN_COLS = old


# ------------------------------------------------------------------------------------------------ ADMM update above dim = 128
@pytest.mark.parametrize("dim,b,S,scale", [(256, 256, 3, 0.05), (520, 300, 2, 0.05), (1024, 1024, 2, 0.02), (200, 200, 2, 1e-6)])
def test_admm_update_above_128_rows_on_many_workgroups_vs_oracle(dev, dim, b, S, scale):
    """ADMM_OPT.step for ADMM(dim > 128) - the exact-global correlation builds ADMM(dim = B_g) - runs 64 workgroups per site in two
    launches (alignq_admm_update_ws) instead of one workgroup per site (1 ms at dim = 1024): alterD / gamma against the C oracle's
    statement of utils/optimizer.py:97-124, a short batch (b < dim: D zero-padded), S sites in one call, and the branch
    |V|_F <= mu / rho (scale 1e-6 with gamma = 0: alterD becomes 0); in place."""
    from alignq_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(dim + b)
    mu, rho = 0.2, 0.3
    Ds = [(rng.standard_normal((b, b)) * scale).astype(np.float32) for _ in range(S)]
    As = [rng.random((dim, dim)).astype(np.float32) for _ in range(S)]
    Gs = [(rng.random((dim, dim)) * (0.0 if scale < 1e-5 else 1.0)).astype(np.float32) for _ in range(S)]
    want = [O.admm_update(D.copy(), A.copy(), G.copy(), mu, rho) for D, A, G in zip(Ds, As, Gs)]
    dD, dA, dG = [cu(v, dev) for v in Ds], [cu(v, dev) for v in As], [cu(v, dev) for v in Gs]
    ws = torch.empty(lib.alignq_admm_update_ws_bytes(S, dim), dtype=torch.uint8, device=dev)
    ptrs = [t.data_ptr() for t in dA + dG]
    L.check(lib.alignq_admm_update_ws(L.ptr_array(dD), L.ptr_array(dA), L.ptr_array(dG), S, b, dim, mu, rho, L.ptr(ws), L.stream_ptr()),
            "alignq_admm_update_ws")
    assert ptrs == [t.data_ptr() for t in dA + dG]
    for s in range(S):
        np.testing.assert_allclose(npy(dA[s]), want[s][0], atol=TOL, rtol=1e-5)
        np.testing.assert_allclose(npy(dG[s]), want[s][1], atol=TOL, rtol=1e-5)
    if scale < 1e-5:
        assert all(not npy(a).any() for a in dA)
    # the one-workgroup entry on the same inputs: same values up to the summation order of the norm
    eA, eG = [cu(v, dev) for v in As], [cu(v, dev) for v in Gs]
    L.check(lib.alignq_admm_update(L.ptr_array(dD), L.ptr_array(eA), L.ptr_array(eG), S, b, dim, mu, rho, L.stream_ptr()), "alignq_admm_update")
    for s in range(S):
        np.testing.assert_allclose(npy(dA[s]), npy(eA[s]), atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(npy(dG[s]), npy(eG[s]), atol=1e-6, rtol=1e-5)
