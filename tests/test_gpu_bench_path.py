"""GPU parity of the kernels bench.py's timed step actually launches, DIRECTLY against the plain-C oracle
(oracle/alignq_oracle.c: oq_bn_fold_ab / oq_bn_site_fwd / oq_bn_site_bwd), through the C ABI:

  * alignq_site_partials_bn  (site_fwd4_kernel<TF,true> with the batch-norm / ReLU / shortcut fold), NCHW and channels-last,
    with (a,b) given, finalised in-kernel from alignq_bn_partial_stats[_nhwc], and finalised from the convolution
    epilogue's float partials (conv_parts > 0);
  * alignq_site_reduce_loss -> alignq_site_prep_fused -> alignq_site_bwd_apply_bn (site_bwd4_kernel<TF,true,true>) ->
    alignq_bn_bwd_apply;
  * the lazy batch-norm form of alignq_conv3x3_nhwc_bwd (dz formed on load, per-tile sums reduced in-kernel);
  * full-size ResNet-20 (S=21) and ResNet-56 (S=57: two chunks of alignq_site_reduce_loss_multi / _prep_fused_multi)
    steps: deferred multi-site launches == per-site launches bit for bit.

Bars: x_q bit-exact given (a,b); D / loss / dz / dgamma / dbeta within 1e-5 (+1e-4 relative)."""
import ctypes

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


def npy(t):
    return t.detach().cpu().numpy()


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _mem(t, nhwc):
    """the [B,F] matrix of a logical [B,C,H,W] tensor in the memory order under test"""
    B = t.shape[0]
    return npy(t.permute(0, 2, 3, 1) if nhwc else t).reshape(B, -1)


def _dev_like(a, shape, nhwc, dev):
    """numpy [B,F] in memory order -> device tensor of logical shape [B,C,H,W] stored in that order"""
    B, C, H, W = shape
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.view(B, H, W, C).permute(0, 3, 1, 2) if nhwc else t.view(B, C, H, W)


class _SiteRun:
    """One BN-folded site through the C ABI; tensors are plain torch allocations."""

    def __init__(self, dev, z, gamma, beta, k, relu, res, nhwc, r=2.0, eps=0.0, bn_eps=1e-5, momentum=0.1):
        from alignq_amd import _lib as L
        self.L, self.lib, self.dev = L, L.load(), dev
        self.z, self.gamma, self.beta, self.k, self.relu, self.res, self.nhwc = z, gamma, beta, k, relu, res, nhwc
        self.r, self.eps, self.bn_eps, self.momentum = r, eps, bn_eps, momentum
        self.B, self.C, H, W = z.shape
        self.HW, self.F = H * W, self.C * H * W

    mask_from_bins = False

    def forward(self, mode, ab_in=None, save_in=None, conv_part=None, want_bins=False):
        L, lib, dev = self.L, self.lib, self.dev
        B, C, HW, F = self.B, self.C, self.HW, self.F
        st = L.stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        self.ab = torch.empty(2, C, **f32) if ab_in is None else torch.from_numpy(ab_in).to(dev)
        self.save = torch.empty(2, C, **f32) if save_in is None else torch.from_numpy(save_in).to(dev)
        self.rm, self.rv = torch.zeros(C, **f32), torch.ones(C, **f32)
        self.nbt = torch.zeros((), dtype=torch.int64, device=dev)
        part, conv_parts = None, 0
        if mode == "stats":
            if self.nhwc:
                part = torch.empty(lib.alignq_bn_nhwc_ws_bytes(C), dtype=torch.uint8, device=dev)
                L.check(lib.alignq_bn_partial_stats_nhwc(L.ptr(self.z), B, C, HW, L.ptr(part), st), "bn_partial_stats_nhwc")
            else:
                part = torch.empty(lib.alignq_bn_ws_bytes(C), dtype=torch.uint8, device=dev)
                L.check(lib.alignq_bn_partial_stats(L.ptr(self.z), B, C, HW, L.ptr(part), st), "bn_partial_stats")
        elif mode == "conv":
            part, conv_parts = conv_part
        self.y = torch.empty_like(self.z)
        self.bins = None
        if want_bins:
            nb = lib.alignq_bin_bytes(self.k, self.r, 0)
            self.bins = torch.empty_strided(self.z.shape, self.z.stride(), dtype={1: torch.int8, 2: torch.int16}[nb], device=dev)
        self.stats = torch.empty(4, F, **f32)
        self.ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
        L.check(lib.alignq_site_partials_bn(L.ptr(self.z), L.ptr(part), L.ptr(self.gamma), L.ptr(self.beta), L.ptr(self.rm),
                                            L.ptr(self.rv), L.ptr(self.nbt), self.momentum, self.bn_eps, L.ptr(self.ab),
                                            L.ptr(self.save), C, HW, B, F, self.k, self.r, self.eps, int(self.relu),
                                            L.ptr(self.res), int(self.nhwc), int(conv_parts), L.ptr(self.y), L.ptr(self.bins),
                                            L.ptr(self.stats), L.ptr(self.ws), st), "alignq_site_partials_bn")
        return self

    def reduce_loss(self, A, Gm, mu=0.2, rho=0.3):
        L, lib = self.L, self.lib
        self.A, self.Gm, self.mu = A, Gm, mu
        self.D = torch.empty(self.B, self.B, dtype=torch.float32, device=self.dev)
        self.scal = torch.empty(4, dtype=torch.float32, device=self.dev)
        L.check(lib.alignq_site_reduce_loss(L.ptr(self.ws), self.B, self.F, L.ptr(self.D), L.ptr(A), L.ptr(Gm), A.shape[0],
                                            mu, rho, L.ptr(self.scal), L.stream_ptr()), "alignq_site_reduce_loss")
        return self

    def backward(self, g, g_loss):
        L, lib, dev = self.L, self.lib, self.dev
        B, C, HW, F = self.B, self.C, self.HW, self.F
        st = L.stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        gl = torch.tensor(g_loss, **f32)
        self.S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, **f32)      # fp32 S + its bf16 fragment image
        self.dA, self.dG = torch.empty_like(self.A), torch.empty_like(self.Gm)
        L.check(lib.alignq_site_prep_fused(L.ptr(self.D), L.ptr(self.A), L.ptr(self.Gm), self.A.shape[0], L.ptr(self.scal),
                                           self.mu, L.ptr(gl), B, F, L.ptr(self.S), L.ptr(self.dA), L.ptr(self.dG), st),
                "alignq_site_prep_fused")
        self.dx = torch.empty_like(self.z)
        self.part = torch.empty(lib.alignq_site_bn_part_bytes(F, int(self.nhwc)), dtype=torch.uint8, device=dev)
        self.dres = torch.empty_like(self.z) if (self.res is not None and self.relu) else None
        L.check(lib.alignq_site_bwd_apply_bn(L.ptr(g), L.ptr(self.S), L.ptr(self.z), L.ptr(self.ab), L.ptr(self.save), C, HW,
                                             int(self.nhwc), L.ptr(self.y) if (self.relu and not self.mask_from_bins) else None,
                                             L.ptr(self.bins) if self.mask_from_bins else None,
                                             self.bins.element_size() if self.mask_from_bins else 0, L.ptr(self.dres),
                                             L.ptr(self.stats), B, F, self.r, self.eps, L.ptr(self.dx), L.ptr(self.part), st),
                "alignq_site_bwd_apply_bn")
        return self

    def bn_backward(self):
        L, lib = self.L, self.lib
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.dz = torch.empty_like(self.z)
        self.dgam, self.dbet = torch.empty(self.C, **f32), torch.empty(self.C, **f32)
        L.check(lib.alignq_bn_bwd_apply(L.ptr(self.dx), L.ptr(self.z), L.ptr(self.ab), L.ptr(self.save), L.ptr(self.part),
                                        self.B, self.C, self.HW, int(self.nhwc), L.ptr(self.dz), L.ptr(self.dgam),
                                        L.ptr(self.dbet), L.stream_ptr()), "alignq_bn_bwd_apply")
        return self


def _expected_y(x, k, r, res, relu):
    xq, _, bins = O.act_quant_fwd(x, k, r, O.FORMULA_ADMM)
    y = xq
    if res is not None:
        y = (y + res).astype(np.float32)
    if relu:
        y = np.maximum(y, np.float32(0.0))
    return y, bins


@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("B,C,H", [(128, 16, 32), (128, 32, 16), (128, 64, 8), (128, 16, 48), (100, 32, 32)])
@pytest.mark.parametrize("nhwc", [0, 1])
def test_bn_folded_site_kernels_vs_oracle(dev, nhwc, B, C, H, k):
    """site_fwd4_kernel<TF,true> / site_bwd4_kernel<TF,true,true> / bn_bwd_apply at the ResNet-20/56 site shapes
    128 x {16384, 8192, 4096}, both layouts, against the C oracle; F = 36864 and 32768 (a short batch) run the forward's
    multi-tile instantiation (more tiles than workgroups: accumulators carried over the tile loop)."""
    if F_big := (C * H * H > 16384):
        if k == 4 and not nhwc:
            pytest.skip("k = 4 at the large shapes runs in the channels-last layout only (oracle time)")
    rng = np.random.default_rng(1000 * nhwc + 10 * C + k)
    relu, with_res = (k != 4), (k == 8 or C == 32)
    shape = (B, C, H, H)
    F = C * H * H
    zm = (rng.standard_normal((B, F)) * 1.7 + 0.3).astype(np.float32)
    gm = (rng.standard_normal((B, F)) * 0.01).astype(np.float32)
    rm_ = (rng.standard_normal((B, F)) * 0.7).astype(np.float32) if with_res else None
    gamma = (rng.random(C) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.2).astype(np.float32)
    A0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    G0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    z = _dev_like(zm, shape, nhwc, dev)
    g = _dev_like(gm, shape, nhwc, dev)
    res = _dev_like(rm_, shape, nhwc, dev) if with_res else None
    tg, tb = torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev)
    A, Gm = torch.from_numpy(A0).to(dev), torch.from_numpy(G0).to(dev)

    # ---- the oracle, once per case ------------------------------------------------------------------------------
    ab_o, save_o, vu_o = O.bn_fold_ab(zm, C, nhwc, gamma, beta, 1e-5)
    y_o, D_o, x_o = O.bn_site_fwd(zm, C, nhwc, ab_o, k, 2.0, 0.0, rm_, relu)
    loss_o, dD_o, dA_o, dG_o = O.admm_loss(D_o, A0, G0, 0.2, 0.3)
    g_loss = 0.7
    dz_o, dgam_o, dbet_o, dres_o, dx_o = O.bn_site_bwd(gm, dD_o * np.float32(g_loss), zm, C, nhwc, ab_o, save_o,
                                                      y_o if relu else None, 2.0, 0.0)

    # ---- (a,b) given: the whole chain against the oracle -----------------------------------------------------------
    run = _SiteRun(dev, z, tg, tb, k, relu, res, nhwc).forward("ab_in", ab_o.copy(), save_o.copy()).reduce_loss(A, Gm)
    assert bits_equal(_mem(run.y, nhwc), y_o), "x_q differs from the oracle at the oracle's (a,b)"
    np.testing.assert_allclose(npy(run.D), D_o, atol=TOL)
    np.testing.assert_allclose(float(run.scal[0]), loss_o, atol=TOL)
    run.backward(g, g_loss).bn_backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(npy(run.dA), dA_o * g_loss, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(npy(run.dG), dG_o * g_loss, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(_mem(run.dx, nhwc), dx_o, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(_mem(run.dz, nhwc), dz_o, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(npy(run.dgam), dgam_o, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(npy(run.dbet), dbet_o, atol=TOL, rtol=1e-4)
    if run.dres is not None:
        assert bits_equal(_mem(run.dres, nhwc), dres_o)

    # ---- (a,b) finalised in-kernel from the statistics kernel's partials --------------------------------------------
    run2 = _SiteRun(dev, z, tg, tb, k, relu, res, nhwc).forward("stats").reduce_loss(A, Gm)
    torch.cuda.synchronize()
    ab_k = npy(run2.ab)
    np.testing.assert_allclose(ab_k, ab_o, rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(npy(run2.save), save_o, rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(npy(run2.rm), 0.1 * save_o[0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(npy(run2.rv), 0.9 + 0.1 * vu_o, rtol=1e-5)
    assert int(run2.nbt) == 1
    y_k, _ = _expected_y(O.bn_apply(zm, C, nhwc, ab_k), k, 2.0, rm_, relu)
    assert bits_equal(_mem(run2.y, nhwc), y_k), "x_q differs from the oracle at the kernel's (a,b)"
    np.testing.assert_allclose(npy(run2.D), D_o, atol=TOL)
    np.testing.assert_allclose(float(run2.scal[0]), loss_o, atol=TOL)
    n = 2 ** k - 1            # and against the oracle's own (a,b): tie-zone bin flips only
    flips = np.abs(_mem(run2.y, nhwc) - y_o) * n
    assert flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 1e-4


@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 8), (128, 64, 8, 4), (128, 16, 32, 2), (100, 32, 16, 8)])
def test_conv_parts_site_and_lazy_bn_conv_backward_vs_oracle(dev, B, C, H, k):
    """The exact launch chain of the timed step for one body layer, channels-last: alignq_conv3x3_nhwc (+ BN partials in the
    epilogue) -> alignq_site_partials_bn(conv_parts) -> reduce+loss -> prep -> alignq_site_bwd_apply_bn ->
    alignq_conv3x3_nhwc_bwd (lazy BN: reduces the per-tile sums, forms dz on load, writes dgamma / dbeta) -> slab reduce.
    Oracle: oq_bn_* on the convolution's output as the kernel produced it; the convolution's own gradients from the
    oracle's dz through an fp64 convolution."""
    from alignq_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr()
    rng = np.random.default_rng(B + C + k)
    torch.manual_seed(B + C + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    f32 = dict(dtype=torch.float32, device=dev)
    x = (torch.randn(B, C, H, H, device=dev) * 1.1).contiguous(memory_format=cl)
    wq = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl)
    z = torch.empty_like(x)
    n_parts = lib.alignq_conv3x3_bn_parts(B, H, H, C)
    assert n_parts > 0
    part = torch.empty(C, n_parts, 2, **f32)
    L.check(lib.alignq_conv3x3_nhwc(L.ptr(x), L.ptr(wq), L.ptr(z), B, H, H, C, k, 0, None, L.ptr(part), None, 0, 0, st), "conv fwd")
    torch.cuda.synchronize()
    zm = _mem(z, 1)
    F = C * H * H
    gamma = (rng.random(C) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.1).astype(np.float32)
    gm = (rng.standard_normal((B, F)) * 0.01).astype(np.float32)
    rsm = (rng.standard_normal((B, F)) * 0.7).astype(np.float32)
    A0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    G0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    shape = (B, C, H, H)
    g, res = _dev_like(gm, shape, 1, dev), _dev_like(rsm, shape, 1, dev)
    tg, tb = torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev)
    A, Gm = torch.from_numpy(A0).to(dev), torch.from_numpy(G0).to(dev)

    run = _SiteRun(dev, z, tg, tb, k, True, res, 1).forward("conv", conv_part=(part, n_parts)).reduce_loss(A, Gm)
    torch.cuda.synchronize()
    ab_o, save_o, vu_o = O.bn_fold_ab(zm, C, 1, gamma, beta, 1e-5)
    ab_k, save_k = npy(run.ab), npy(run.save)
    # float partial sums of the convolution epilogue: (a,b) agree with the double-precision statistics to ~1e-6
    np.testing.assert_allclose(ab_k, ab_o, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(save_k, save_o, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(npy(run.rv), 0.9 + 0.1 * vu_o, rtol=1e-5)
    y_o, D_o, _ = O.bn_site_fwd(zm, C, 1, ab_k, k, 2.0, 0.0, rsm, True)           # at the kernel's (a,b)
    assert bits_equal(_mem(run.y, 1), y_o), "x_q differs from the oracle at the kernel's (a,b)"
    np.testing.assert_allclose(npy(run.D), D_o, atol=TOL)
    loss_o, dD_o, _, _ = O.admm_loss(D_o, A0, G0, 0.2, 0.3)
    np.testing.assert_allclose(float(run.scal[0]), loss_o, atol=TOL)

    run.backward(g, 1.0)
    dz_o, dgam_o, dbet_o, dres_o, dx_o = O.bn_site_bwd(gm, dD_o, zm, C, 1, ab_k, save_k, y_o, 2.0, 0.0)
    np.testing.assert_allclose(_mem(run.dx, 1), dx_o, atol=TOL, rtol=1e-4)
    assert bits_equal(_mem(run.dres, 1), dres_o)
    # lazy form: dy = g w.r.t. the BN output (run.dx), per-tile sums in run.part; ktot = NULL
    dxc, dw = torch.empty_like(x), torch.empty_like(wq)
    ws = torch.empty(lib.alignq_conv3x3_wgrad_ws_bytes(C), dtype=torch.uint8, device=dev)
    dgam, dbet = torch.empty(C, **f32), torch.empty(C, **f32)
    ns = ctypes.c_int(0)
    L.check(lib.alignq_conv3x3_nhwc_bwd(L.ptr(x), L.ptr(run.dx), L.ptr(wq), L.ptr(dxc), L.ptr(ws), B, H, H, C, k,
                                        ctypes.byref(ns), None, L.ptr(z), L.ptr(run.ab), L.ptr(run.save), None,
                                        L.ptr(run.part), L.ptr(dgam), L.ptr(dbet), None, 0, 0, st), "alignq_conv3x3_nhwc_bwd")
    L.check(lib.alignq_conv3x3_wgrad_reduce_multi(1, L.ptr_array([ws]), L.ptr_array([dw]), (ctypes.c_int * 1)(ns.value),
                                                  (ctypes.c_int * 1)(9 * C * C), st), "wgrad_reduce_multi")
    torch.cuda.synchronize()
    np.testing.assert_allclose(npy(dgam), dgam_o, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(npy(dbet), dbet_o, atol=TOL, rtol=1e-4)
    dzt = _dev_like(dz_o, shape, 1, dev).double()
    dx_ref = torch.nn.grad.conv2d_input(x.shape, wq.double(), dzt, padding=1)
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), wq.shape, dzt, padding=1)
    sx, sw = float(dx_ref.abs().max()), float(dw_ref.abs().max())
    np.testing.assert_allclose(npy(dxc), npy(dx_ref.float()), atol=2e-5 * sx + 1e-7, rtol=1e-4)
    np.testing.assert_allclose(npy(dw), npy(dw_ref.float()), atol=2e-5 * sw + 1e-7, rtol=2e-4)


def _run_model_step(dev, depth_units, k, batch, deferred, seed=0):
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = k
    config.args.train_batch_size = batch
    torch.manual_seed(seed)
    net = PreActResNet(PreActBlock_conv_Q, depth_units, k, k, "second", 10).to(dev).train()
    step = TrainStep(net, channels_last=True, qconv=True, defer_losses=deferred)
    torch.manual_seed(seed + 1)
    x = torch.randn(batch, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (batch,), device=dev)
    logits, ce, tl = step._forward_backward(x, y, set_to_none=True)
    torch.cuda.synchronize()
    out = dict(logits=npy(logits), ce=float(ce.detach()), tl=float(tl.detach()), D=[npy(m.D) for m in step.admms],
               grads={n_: npy(p.grad) for n_, p in net.named_parameters() if p.grad is not None})
    return out, step


@pytest.mark.parametrize("name,units,k,S", [("resnet20", [3, 3, 3], 8, 21), ("resnet56", [9, 9, 9], 4, 57)])
def test_full_size_model_deferred_multi_equals_per_site(dev, name, units, k, S):
    """BASELINE configs 2 and 4 at model level, B=128: resnet20_quant(8,8) (S=21) and resnet56_quant(4,4) (S=57 > the 32-site
    chunk of alignq_site_reduce_loss_multi / alignq_site_prep_fused_multi, i.e. two launches each).  The deferred
    multi-site launches must reproduce the per-site launches bit for bit: D of every site, every parameter gradient, logits;
    the loss sum up to the order of one fp32 sum."""
    from alignq_amd import config
    try:
        a, step_a = _run_model_step(dev, units, k, 128, deferred=False)
        b, step_b = _run_model_step(dev, units, k, 128, deferred=True)
        assert len(step_a.admms) == S and len(a["D"]) == S
        assert step_b._deferred is not None and len(step_b._deferred.records) == S
        assert np.array_equal(a["logits"], b["logits"]) and a["ce"] == b["ce"]
        np.testing.assert_allclose(b["tl"], a["tl"], rtol=2e-6)
        for i, (da, db) in enumerate(zip(a["D"], b["D"])):
            assert np.array_equal(da, db), f"site {i} D"
            assert np.isfinite(da).all()
        assert a["grads"].keys() == b["grads"].keys()
        for n_ in a["grads"]:
            assert np.array_equal(a["grads"][n_], b["grads"][n_]), n_
            assert np.isfinite(a["grads"][n_]).all(), n_
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


# ------------------------------------------------------------------------------------------------ N2 in the bench path
@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 4), (128, 64, 8, 8), (100, 32, 16, 2)])
def test_site_emits_level_indices_and_consumers_read_them(dev, B, C, H, k):
    """SURVEY 8f-N2 on the kernels of the captured step: alignq_site_partials_bn(bins_out) stores the level index of
    relu(x_q) (int16 for 8-bit: 1021 levels, int8 for <= 4-bit) — bit-exact against the oracle's bins clamped at 0, and
    idx / n is exactly the fp32 output of the same launch; alignq_site_bwd_apply_bn masks by the index exactly as by y;
    alignq_conv3x3_nhwc / _wgrad / _bwd reading the indices agree with the fp32-input kernels and with an fp64 convolution."""
    from alignq_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr()
    rng = np.random.default_rng(B + C + k)
    F = C * H * H
    shape = (B, C, H, H)
    zm = (rng.standard_normal((B, F)) * 1.5 + 0.2).astype(np.float32)
    gm = (rng.standard_normal((B, F)) * 0.01).astype(np.float32)
    gamma = (rng.random(C) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.2).astype(np.float32)
    A0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    G0 = (rng.standard_normal((128, 128)) * 0.05).astype(np.float32)
    z, g = _dev_like(zm, shape, 1, dev), _dev_like(gm, shape, 1, dev)
    tg, tb = torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev)
    A, Gm = torch.from_numpy(A0).to(dev), torch.from_numpy(G0).to(dev)
    n = 2 ** k - 1
    run = _SiteRun(dev, z, tg, tb, k, True, None, 1).forward("stats", want_bins=True).reduce_loss(A, Gm)
    torch.cuda.synchronize()
    assert run.bins.dtype == (torch.int16 if 2 * n > 127 else torch.int8)
    ab_k = npy(run.ab)
    x = O.bn_apply(zm, C, 1, ab_k)
    xq_o, _, bins_o = O.act_quant_fwd(x, k, 2.0, O.FORMULA_ADMM)
    bins = _mem(run.bins, 1).astype(np.int32)
    assert np.array_equal(bins, np.maximum(bins_o, 0)), "stored level index != oracle bins (clamped by the ReLU)"
    y = _mem(run.y, 1)
    assert bits_equal(y, np.maximum(xq_o, np.float32(0)))
    assert np.array_equal(y, (bins.astype(np.float32) / np.float32(n)))                 # dequantised value == idx / n exactly
    # backward: mask from the index == mask from y, bit for bit
    tf = 64 if F >= 16384 else 32
    n_part = (F // tf) * min(C, tf) * 2                  # floats the backward writes: [tiles][channels of a tile][2]
    run.backward(g, 1.0)
    dx_y, part_y = npy(run.dx).copy(), npy(run.part).view(np.float32)[:n_part].copy()
    run.mask_from_bins = True
    run.backward(g, 1.0)
    torch.cuda.synchronize()
    assert np.array_equal(npy(run.dx), dx_y) and np.array_equal(npy(run.part).view(np.float32)[:n_part], part_y)
    # convolution consumers: forward and filter gradient from the indices vs from the fp32 tensor vs fp64
    torch.manual_seed(k)
    wq = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * 255) / 255).to(dev).contiguous(memory_format=torch.channels_last)
    f32 = dict(dtype=torch.float32, device=dev)
    ya, yb = torch.empty_like(run.y), torch.empty_like(run.y)
    nb = run.bins.element_size()
    L.check(lib.alignq_conv3x3_nhwc(L.ptr(run.y), L.ptr(wq), L.ptr(ya), B, H, H, C, 8, 0, None, None, None, 0, 0, st), "conv fp32")
    L.check(lib.alignq_conv3x3_nhwc(None, L.ptr(wq), L.ptr(yb), B, H, H, C, 8, 0, None, None, L.ptr(run.bins), nb, k, st), "conv bins")
    yd = torch.nn.functional.conv2d(run.y.double(), wq.double(), padding=1)
    torch.cuda.synchronize()
    ea, eb = float((ya - yd).abs().max()), float((yb - yd).abs().max())
    assert eb <= max(ea, 2e-6 * float(yd.abs().max())) * 1.05, (ea, eb)                 # at least as accurate as the fp32 form
    dy = torch.randn_like(run.y)
    ws = torch.empty(lib.alignq_conv3x3_wgrad_ws_bytes(C), dtype=torch.uint8, device=dev)
    dwa, dwb = torch.empty_like(wq), torch.empty_like(wq)
    L.check(lib.alignq_conv3x3_nhwc_wgrad(L.ptr(run.y), L.ptr(dy), L.ptr(dwa), L.ptr(ws), B, H, H, C, None, None, 0, 0, st), "wgrad fp32")
    L.check(lib.alignq_conv3x3_nhwc_wgrad(None, L.ptr(dy), L.ptr(dwb), L.ptr(ws), B, H, H, C, None, L.ptr(run.bins), nb, k, st), "wgrad bins")
    dwd = torch.nn.grad.conv2d_weight(run.y.double(), wq.shape, dy.double(), padding=1)
    torch.cuda.synchronize()
    sw = float(dwd.abs().max())
    np.testing.assert_allclose(npy(dwb), npy(dwd.float()), atol=2e-5 * sw, rtol=2e-4)
    np.testing.assert_allclose(npy(dwb), npy(dwa), atol=2e-5 * sw, rtol=2e-4)


def test_trainstep_with_packed_activations_tracks_the_fp32_form(dev):
    """TrainStep(pack_bins=True) (the default: site0 of every block hands conv1 its level indices, no fp32 copy exists)
    against pack_bins=False on full-size resnet20_quant(8,8) at B=128: the same forward up to convolution rounding (the
    index form sums exact integers), so D of every site, the losses and every gradient agree closely."""
    from alignq_amd import config
    from alignq_amd.resnet import resnet20_quant
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    try:
        outs = []
        for pack in (False, True):
            torch.manual_seed(0)
            net = resnet20_quant(8, 8).to(dev).train()
            step = TrainStep(net, channels_last=True, qconv=True, pack_bins=pack)
            assert any(getattr(m, "pack_bins", False) for m in net.modules()) == pack
            torch.manual_seed(1)
            x = torch.randn(128, 3, 32, 32, device=dev)
            y = torch.randint(0, 10, (128,), device=dev)
            logits, ce, tl = step._forward_backward(x, y, set_to_none=True)
            torch.cuda.synchronize()
            outs.append(dict(logits=npy(logits), ce=float(ce.detach()), tl=float(tl.detach()), D=[npy(m.D) for m in step.admms],
                             grads={n_: npy(p.grad) for n_, p in net.named_parameters() if p.grad is not None}))
        a, b = outs
        np.testing.assert_allclose(b["logits"], a["logits"], atol=2e-2)        # 8-bit bins flip on 1e-6 convolution differences
        np.testing.assert_allclose(b["tl"], a["tl"], rtol=1e-3)
        for i, (da, db) in enumerate(zip(a["D"], b["D"])):
            np.testing.assert_allclose(db, da, atol=2e-3, err_msg=f"site {i}")
        for n_ in a["grads"]:
            ga, gb = a["grads"][n_].ravel(), b["grads"][n_].ravel()
            if ga.size >= 64 and "alterD" not in n_ and "gamma" not in n_:
                cos = float(np.dot(ga, gb) / (np.linalg.norm(ga) * np.linalg.norm(gb) + 1e-30))
                assert cos > 0.97, (n_, cos)          # (the stem, 20 layers of 8-bit bin flips upstream, sits lowest: 0.989)
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128
