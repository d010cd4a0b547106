"""Round 5: Conv2d_Q's convolution at the ResNet-50 / Office-31 shapes on the exact-product GEMM kernels
(csrc/qgemm_kernels.hip, alignq_qconv_fwd / _dgrad / _wgrad) against an fp64 convolution of the same fp32 inputs.
Reference op: cdf_alignment_admm/dann_office/model/quantization.py:164-181 (F.conv2d(input, weight_q, ...)), shapes of
model/resnet.py:31-41,104-110,131-156 (Bottleneck) and :122-126 (downsample)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CL = torch.channels_last


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _wq(cout, cin, ks, k, dev, seed):
    g = torch.Generator().manual_seed(seed)
    n = 2 ** k - 1
    w = torch.round(torch.tanh(torch.randn(cout, cin, ks, ks, generator=g)) * n) / n
    return w.to(dev).contiguous(memory_format=CL)


def _levels(shape, n_a, r, dev, seed):
    """relu(act_q(.)) of the ADMM / Office formula: idx / n_a with integer idx in [0, r * n_a]"""
    g = torch.Generator().manual_seed(seed)
    idx = torch.clamp(torch.round(torch.randn(shape, generator=g) * 0.6 * n_a), 0, r * n_a)
    return (idx / n_a).to(dev).contiguous(memory_format=CL)


# every 1x1 convolution of ResNet-50's bottlenecks (C_in, C_out, H_in, stride): conv1 / conv3 of each layer, conv1 of the first
# block of a layer (reads the previous layer's resolution), the downsample convolutions
R50_1X1 = [(64, 64, 56, 1), (64, 256, 56, 1), (256, 64, 56, 1), (256, 128, 56, 1), (128, 512, 28, 1), (512, 128, 28, 1),
           (256, 512, 56, 2), (512, 256, 28, 1), (256, 1024, 14, 1), (1024, 256, 14, 1), (512, 1024, 28, 2), (1024, 512, 14, 1),
           (512, 2048, 7, 1), (2048, 512, 7, 1), (1024, 2048, 14, 2)]
R50_3X3 = [(64, 56, 1), (128, 56, 2), (128, 28, 1), (256, 28, 2), (256, 14, 1), (512, 14, 2), (512, 7, 1)]


def _check(dev, B, cin, cout, H, ks, stride, k, level, seed):
    from alignq_amd import ops
    pad = (ks - 1) // 2
    wq = _wq(cout, cin, ks, k, dev, seed).requires_grad_(True)
    if level:
        x = _levels((B, cin, H, H), 255.0, 2, dev, seed + 1)
    else:
        x = (torch.randn(B, cin, H, H, generator=torch.Generator().manual_seed(seed + 1)) * 1.3).to(dev).contiguous(memory_format=CL)
    x.requires_grad_(True)
    assert ops.qconv_gemm_supported(x, wq, (stride, stride), (pad, pad), (1, 1), 1, None, k)
    y = ops.QConvGemmFn.apply(x, wq, k, stride, 255.0 if level else 0.0)
    assert y.is_contiguous(memory_format=CL)
    gy = (torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 2)) * 1e-3).to(dev).contiguous(memory_format=CL)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(True), wq.detach().double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, stride=stride, padding=pad)
    yd.backward(gy.double())
    y32 = torch.nn.functional.conv2d(x.detach(), wq.detach(), stride=stride, padding=pad)
    floor = 2e-6 * float(yd.abs().max())
    err = float((y.detach() - yd).abs().max())
    assert err <= max(float((y32 - yd).abs().max()), floor), ("fwd", err, floor)
    if level:       # integer operands: the sum is an exact integer below 2^24, one correctly rounded division
        n_w = 2 ** k - 1
        acc = torch.nn.functional.conv2d(torch.round(xd.detach() * 255.0), torch.round(wd.detach() * n_w), stride=stride, padding=pad)
        if float(acc.abs().max()) < 2 ** 24:
            want = (acc / (255.0 * n_w)).float()
            assert torch.equal(y.detach(), want), "level operands: bit-exact quotient of the integer sum"
    dx32, dw32 = torch.ops.aten.convolution_backward(gy, x.detach(), wq.detach(), None, (stride, stride), (pad, pad), (1, 1), False,
                                                      (0, 0), 1, (True, True, False))[:2]
    floor = 2e-6 * float(xd.grad.abs().max())
    err = float((x.grad - xd.grad).abs().max())
    assert err <= max(float((dx32 - xd.grad).abs().max()), floor), ("dgrad", err, floor)
    floor = 2e-6 * float(wd.grad.abs().max())
    err = float((wq.grad - wd.grad).abs().max())
    assert err <= max(float((dw32 - wd.grad).abs().max()), floor), ("wgrad", err, floor, float((dw32 - wd.grad).abs().max()))


@pytest.mark.parametrize("cin,cout,H,stride", R50_1X1)
@pytest.mark.parametrize("level", [False, True])
def test_qconv1x1_matches_fp64_at_every_resnet50_shape(dev, cin, cout, H, stride, level):
    """B = 3 (rows not a multiple of the 128-row tile at 14x14 / 7x7; odd batch): forward, data gradient, filter gradient."""
    _check(dev, 3, cin, cout, H, 1, stride, 8, level, seed=cin + cout + H)


@pytest.mark.parametrize("c,H,stride", R50_3X3)
@pytest.mark.parametrize("level", [False, True])
def test_qconv3x3_matches_fp64_at_every_resnet50_shape(dev, c, H, stride, level):
    _check(dev, 3, c, c, H, 3, stride, 8, level, seed=c + H)


def test_qconv3x3_stride2_data_gradient_runs_on_the_parity_class_kernel(dev, monkeypatch):
    """round 5: the three 3x3 stride-2 layers' data gradients (even grids) are alignq_qconv_dgrad's (KM 3: one group of rows per
    parity class of the input pixel, 1 / 2 / 2 / 4 taps each) - torch's conv2d_input is not called; an odd grid (no ResNet-50 layer
    has one at 224 x 224) keeps torch's.  Values: _check against fp64, every class's border (last row / column: the tap that
    reaches dy row H/2 contributes nothing), a batch whose class size is not a multiple of the row tile."""
    def boom(*a, **k):
        raise AssertionError("torch.nn.grad.conv2d_input called for an even grid")
    monkeypatch.setattr(torch.nn.grad, "conv2d_input", boom)
    for B, c, H in ((3, 128, 56), (5, 64, 6), (2, 256, 10), (56, 256, 14)):
        _check(dev, B, c, c, H, 3, 2, 8, False, seed=B + c + H)
    _check(dev, 3, 64, 128, 12, 3, 2, 4, True, seed=11)
    monkeypatch.undo()
    _check(dev, 3, 64, 64, 9, 3, 2, 8, False, seed=5)          # odd grid: torch's data gradient, same bars


@pytest.mark.parametrize("cin,cout,H,ks,stride", [(512, 2048, 7, 1, 1), (512, 512, 7, 3, 1), (512, 1024, 14, 1, 2), (256, 256, 14, 3, 1),
                                                  (512, 512, 14, 3, 2), (64, 64, 56, 3, 1)])
def test_qconv_data_gradient_split_k_matches_one_workgroup_per_tile(dev, cin, cout, H, ks, stride):
    """alignq_qconv_dgrad with its scratch (layers with few row tiles and a long contraction: 2-4 workgroups per tile over disjoint
    k ranges + a closing pass that adds the raw sums in split order and divides) against the same call without it, B = 56: every
    form (1x1, scattering 1x1 stride 2, halo 3x3, parity-class 3x3 stride 2).  Same products, another grouping of the fp32 sums:
    equal to 4e-6 of the largest element (each is within 2e-6 of fp64, _check); the last shape takes no split and is bit-equal.
    Deterministic: a second call reproduces the first bit for bit."""
    from alignq_amd import _lib as L, ops
    lib = L.load()
    B, Ho = 56, (H - 1) // stride + 1
    w = _wq(cout, cin, ks, 8, dev, 3)
    wb = ops.pack_filter_bins([w], 8)[0][0]
    gy = (torch.randn(B, cout, Ho, Ho, generator=torch.Generator().manual_seed(4)) * 1e-3).to(dev).contiguous(memory_format=CL)
    nws = lib.alignq_qconv_dgrad_ws_bytes(B, H, H, cin, cout, ks, stride)
    # round 6: the query answers the bytes of the split the launch WILL take (2-4 images of dx), 0 when it takes none
    # (layer1's 45 MB data gradient has 3136 row tiles; layers whose tiles already fill the chip)
    out_bytes = B * cin * H * H * 4
    assert nws in (0, 2 * out_bytes, 3 * out_bytes, 4 * out_bytes) and ((cin, H, ks) != (64, 56, 3) or nws == 0)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws else None
    outs = []
    for scratch in (ws, None, ws):
        dx = torch.full((B, cin, H, H), float("nan"), device=dev).contiguous(memory_format=CL)
        L.check(lib.alignq_qconv_dgrad(L.ptr(gy), L.ptr(wb), L.ptr(dx), B, H, H, cin, cout, ks, stride, 8, L.ptr(scratch), None), "dgrad")
        outs.append(dx)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[2])
    scale = float(outs[1].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-6 * scale
    if nws == 0:
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cin,cout,H,ks,stride,k", [(256, 64, 56, 1, 1, 8), (512, 1024, 28, 1, 2, 4), (256, 256, 14, 3, 1, 2),
                                                    (2048, 512, 7, 1, 1, 8)])
def test_qconv_full_batch_of_config5(dev, cin, cout, H, ks, stride, k):
    """the merged source + target batch of the Office step (2 x 28 images)"""
    _check(dev, 56, cin, cout, H, ks, stride, k, level=(ks == 3), seed=7)


def test_qconv_bn_partials_are_the_column_sums(dev):
    """forward epilogue: per-tile per-channel {sum y, sum y^2} (double), two groups that must not share a tile"""
    from alignq_amd import ops
    B, cin, cout, H = 6, 128, 256, 14
    wq = _wq(cout, cin, 1, 8, dev, 3)
    x = (torch.randn(B, cin, H, H, generator=torch.Generator().manual_seed(4)) * 1.3).to(dev).contiguous(memory_format=CL)
    y = ops.QConvGemmFn.apply_with_stats(x, wq, 8, 1, 0.0, 2)
    part, n_parts, groups = y._alignq_bnq_part
    assert groups == 2 and tuple(part.shape) == (2, n_parts, cout, 2) and n_parts in ((3 * H * H + 127) // 128, (3 * H * H + 63) // 64)
    yg = y.detach().permute(0, 2, 3, 1).reshape(2, 3 * H * H, cout).double()
    tot = part.sum(1)
    np.testing.assert_allclose(tot[..., 0].cpu().numpy(), yg.sum(1).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(tot[..., 1].cpu().numpy(), (yg * yg).sum(1).cpu().numpy(), rtol=1e-6, atol=1e-4)
    y2 = ops.QConvGemmFn.apply(x, wq, 8, 1, 0.0)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("groups", [1, 2])
def test_folded_bn_chains_take_the_convolution_partials(dev, groups):
    """bn_act_relu / bn_only / bn_site_res_relu fed with the convolution epilogue's partial sums (alignq_bnq_fwd_parts /
    alignq_bnq_stats_parts) against the same chains making their own statistics pass: same a, b up to the summation order"""
    from alignq_amd import config, fused, office as Q, ops
    from alignq_amd.admm import ADMM
    saved = (config.args.abitW, config.args.train_batch_size)
    config.args.abitW, config.args.train_batch_size = 8, 4
    try:
        B, cin, cout, H = 4 * groups, 128, 256, 14
        wq = _wq(cout, cin, 1, 8, dev, 5)
        x = (torch.randn(B, cin, H, H, generator=torch.Generator().manual_seed(6)) * 1.3).to(dev).contiguous(memory_format=CL)
        res = torch.relu(torch.randn(B, cout, H, H, generator=torch.Generator().manual_seed(7))).to(dev).contiguous(memory_format=CL)
        outs = []
        for with_parts in (False, True):
            torch.manual_seed(0)
            bn = torch.nn.BatchNorm2d(cout).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.1)
            act = Q.activation_quantize_fn(a_bit=8, stage="aligned")
            admm = ADMM(4).to(dev)
            act3 = Q.activation_quantize_fn2(a_bit=8, stage="aligned", admm=admm)
            z = (ops.QConvGemmFn.apply_with_stats(x, wq, 8, 1, 0.0, groups) if with_parts else ops.QConvGemmFn.apply(x, wq, 8, 1, 0.0))
            assert (fused.conv_partials(z, groups) is not None) == with_parts
            y1 = fused.bn_act_relu(bn, act, z, 0, True, groups)
            y2 = fused.bn_only(bn, z, groups)
            y3, loss = fused.bn_site_res_relu(bn, act3, z, res, 1e-5, groups)
            outs.append((y1.detach(), y2.detach(), y3.detach(), loss.detach(), bn.running_mean.clone(), bn.running_var.clone()))
        a, b = outs
        # quantiser outputs may differ by one level where a*z+b sits on a rounding boundary: compare through the tie band
        assert float((a[0] - b[0]).abs().max()) <= 1.0 / 255 + 1e-6 and float(((a[0] - b[0]).abs() > 1e-6).float().mean()) < 1e-3
        np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=0, atol=2e-5)
        assert float(((a[2] - b[2]).abs() > 1e-5).float().mean()) < 1e-3
        np.testing.assert_allclose(float(a[3].sum()), float(b[3].sum()), rtol=1e-4)
        np.testing.assert_allclose(a[4].cpu().numpy(), b[4].cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(a[5].cpu().numpy(), b[5].cpu().numpy(), rtol=1e-5, atol=1e-6)
    finally:
        config.args.abitW, config.args.train_batch_size = saved


def test_qconv_rejects_what_it_does_not_take(dev):
    from alignq_amd import _lib as L
    lib = L.load()
    assert lib.alignq_qconv_supported(2, 8, 8, 48, 64, 1, 1) == 0
    assert lib.alignq_qconv_supported(2, 8, 8, 64, 64, 5, 1) == 0
    assert lib.alignq_qconv_supported(2, 8, 8, 64, 64, 3, 3) == 0
    x = torch.zeros(2, 8, 8, 64, device=dev)
    assert lib.alignq_qconv_dgrad(L.ptr(x), L.ptr(x), L.ptr(x), 2, 7, 8, 64, 64, 3, 2, 8, None, None) == -2       # ALIGNQ_EUNSUPPORTED: odd grid
    assert lib.alignq_qconv_fwd(L.ptr(x), L.ptr(x), L.ptr(x), 2, 8, 8, 64, 64, 1, 1, 9, 0.0, 0, 1, None, None) == -1    # w_bit
    assert lib.alignq_qconv_fwd(L.ptr(x), L.ptr(x), L.ptr(x), 2, 8, 8, 64, 64, 1, 1, 8, 0.0, 2, 1, None, None) == -1    # indices need x_levels


@pytest.mark.parametrize("cin,cout,H,ks,stride", [(64, 256, 56, 1, 1), (128, 128, 28, 3, 1), (128, 128, 56, 3, 2), (512, 2048, 7, 1, 1),
                                                  (256, 256, 14, 3, 1)])
def test_qconv_reads_int16_level_indices_bit_for_bit(dev, cin, cout, H, ks, stride):
    """N2 on the Office path: the level operand as int16 indices (x_bin_bytes = 2) gives the SAME bits as the fp32 level tensor
    (the f16 operand terms are the same integers) - forward, filter gradient - and the data gradient does not depend on it."""
    from alignq_amd import fused, ops
    B = 5
    wq = _wq(cout, cin, ks, 8, dev, 21)
    x = _levels((B, cin, H, H), 255.0, 2, dev, 22)
    xb = torch.round(x * 255.0).to(torch.int16).contiguous(memory_format=CL)
    gy = None
    res = []
    for packed in (False, True):
        w = wq.clone(memory_format=CL).requires_grad_(True)
        if packed:
            xin = fused.packed_handle(x.shape, dev).requires_grad_(True)
            y = ops.QConvGemmFn.apply(xin, w, 8, stride, 255.0, 1, False, None, xb)
        else:
            xin = x.clone(memory_format=CL).requires_grad_(True)
            y = ops.QConvGemmFn.apply(xin, w, 8, stride, 255.0)
        if gy is None:
            gy = (torch.randn(y.shape, generator=torch.Generator().manual_seed(23)) * 1e-3).to(dev).contiguous(memory_format=CL)
        y.backward(gy)
        res.append((y.detach(), w.grad.clone(), xin.grad.clone()))
    for u, v in zip(*res):
        assert torch.equal(u, v)


def test_folded_quantiser_emits_int16_indices_and_a_handle(dev):
    """fused.bn_act_relu(pack=True): the int16 indices are round(y * n) of the fp32 form bit for bit, the handle carries them and the
    level tag, fused.materialize gives the fp32 tensor back, and the backward (through the one-bit mask) is unchanged."""
    from alignq_amd import config, fused, office as Q
    saved = config.args.abitW
    config.args.abitW = 8
    try:
        B, C, H = 6, 128, 14
        z0 = (torch.randn(B, C, H, H, generator=torch.Generator().manual_seed(31)) * 1.3 + 0.2).to(dev).contiguous(memory_format=CL)
        g0 = torch.randn(B, C, H, H, generator=torch.Generator().manual_seed(32)).to(dev).contiguous(memory_format=CL)
        outs = []
        for pack in (False, True):
            torch.manual_seed(0)
            bn = torch.nn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
            act = Q.activation_quantize_fn(a_bit=8, stage="aligned")
            z = z0.clone(memory_format=CL).requires_grad_(True)
            y = fused.bn_act_relu(bn, act, z, 0, True, 2, None, pack)
            assert ops_level(y) == 255.0
            if pack:
                bins, a_bit = y._alignq_bins
                assert bins.dtype == torch.int16 and a_bit == 8 and y.stride() == (0, 0, 0, 0)
                val = fused.materialize(y)
            else:
                bins, val = None, y
            val.backward(g0)
            outs.append((val.detach(), bins, z.grad.clone(), bn.weight.grad.clone()))
        (y0, _, dz0, dg0), (y1, b1, dz1, dg1) = outs
        assert torch.equal(y0, y1) and torch.equal(dz0, dz1) and torch.equal(dg0, dg1)
        assert torch.equal(b1.float(), torch.round(y0 * 255.0)) and int(b1.min()) >= 0
    finally:
        config.args.abitW = saved


def ops_level(t):
    from alignq_amd import ops
    return ops.level_count(t)


def test_filter_bins_pack_is_exact(dev):
    """alignq_qconv_pack_weights: bf16 and f16 bit patterns of rint(W_q * n), multi-tensor, any layout"""
    from alignq_amd import ops
    ws = [_wq(64, 64, 3, 8, dev, 1), _wq(128, 64, 1, 4, dev, 2).contiguous(), _wq(64, 256, 1, 8, dev, 3)]
    for k, w in ((8, ws[0]), (4, ws[1]), (8, ws[2])):
        (bf, hf), = ops.pack_filter_bins([w], k)
        want = torch.round(w * (2 ** k - 1))
        assert bf.stride() == w.stride() and hf.stride() == w.stride()
        assert torch.equal(bf.view(torch.bfloat16).float(), want) and torch.equal(hf.view(torch.float16).float(), want)
    many = ops.pack_filter_bins([ws[0]] * 70, 8)          # more filters than one launch takes
    assert all(torch.equal(b.view(torch.bfloat16).float(), torch.round(ws[0] * 255)) for b, _ in many)


@pytest.mark.parametrize("B,H,W,k", [(3, 224, 224, 8), (2, 64, 48, 8), (4, 37, 41, 4), (2, 8, 8, 2), (56, 32, 32, 8)])
def test_qconv_stem7_matches_fp64_and_leaves_the_batch_norm_sums(dev, B, H, W, k):
    """The Office stem Conv2d_Q(3, 64, 7, stride 2, padding 3) (dann_office/model/resnet.py:193-195) on alignq_qconv_stem7_fwd:
    against fp64 (bar: MIOpen's own fp32 error or 2e-6 of the largest output), borders (odd sizes, an 8 x 8 image whose every
    pixel is a border pixel), the epilogue's per-workgroup {sum y, sum y^2} for 1 and 2 batch slices against the column sums of y,
    and the filter gradient through the Function against fp64."""
    from alignq_amd import _lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(B + H + W)
    x = (torch.randn(B, 3, H, W, generator=g) * 1.7).to(dev).contiguous(memory_format=CL)
    wq = _wq(64, 3, 7, k, dev, 9).requires_grad_(True)
    assert ops.qconv_stem7_supported(x, wq, (2, 2), (3, 3), (1, 1), 1, None, k)
    yd = torch.nn.functional.conv2d(x.double(), wq.detach().double(), stride=2, padding=3)
    y32 = torch.nn.functional.conv2d(x, wq.detach(), stride=2, padding=3)
    for groups in ((1, 2) if B % 2 == 0 else (1,)):
        y = ops.QConvStem7Fn.apply_with_stats(x, wq, k, groups)
        assert y.shape == yd.shape and y.is_contiguous(memory_format=CL)
        err, floor = float((y.detach() - yd).abs().max()), 2e-6 * float(yd.abs().max())
        assert err <= max(float((y32 - yd).abs().max()), floor), ("fwd", err, floor)
        part, n_parts, gr = y._alignq_bnq_part
        assert gr == groups and tuple(part.shape) == (groups, n_parts, 64, 2)
        ys = y.detach().double().reshape(groups, B // groups, 64, -1)
        want = torch.stack([ys.sum(dim=(1, 3)), (ys * ys).sum(dim=(1, 3))], dim=-1)          # [groups][64][2]
        got = part.sum(dim=1)
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
    gy = (torch.randn(y.shape, generator=g) * 1e-3).to(dev).contiguous(memory_format=CL)
    y.backward(gy)
    wd = wq.detach().double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), wd, stride=2, padding=3).backward(gy.double())
    dw32 = torch.ops.aten.convolution_backward(gy, x, wq.detach(), None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    err, floor = float((wq.grad - wd.grad).abs().max()), 2e-6 * float(wd.grad.abs().max())
    assert err <= max(float((dw32 - wd.grad).abs().max()), floor), ("wgrad", err, floor)
