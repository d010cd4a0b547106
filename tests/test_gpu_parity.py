"""GPU parity tests (run on the MI355X box with -m gpu): the HIP kernels, called through the C ABI
(alignq_amd._lib / ops), against
  (1) the plain-C oracle on the same seeded inputs        -> bit-exact for bins and dequantised values
      (same ALIGNQ-ERF32 spec), 1e-5 for reductions / Gram / gradients;
  (2) the golden vectors captured from the reference       -> bins exact outside the erf tie zone
      (|frac(t*n) - 1/2| < 1e-4), dequantised values and residuals within 1e-5;
  (3) size-independent properties at BASELINE.json's full sizes.
"""
import numpy as np
import pytest
import torch

from tests import oracle_c as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

TIE = 1e-4
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
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def check_bins_vs_ref(q_ours, q_ref, y_ref, n, scale=1.0):
    frac = y_ref - np.floor(y_ref)
    tie = np.abs(frac - 0.5) < TIE
    diff = np.abs(q_ours - q_ref) * n / scale
    assert np.all(diff[~tie] == 0)
    assert np.all(diff[tie] <= 1.0 + 1e-3)


def test_act_quant_with_fused_relu_equals_the_composition(dev):
    """ops.ActQuantReluFn (alignq_act_quant_relu_fwd / _bwd) == relu(ActQuantFn(x)) bit for bit, forward and backward
    (the Office bottleneck's `self.relu(self.act_q1(...))`)."""
    from alignq_amd import ops
    torch.manual_seed(0)
    for shape, k in (((28, 64, 56, 56), 8), ((5, 7, 4), 4), ((3, 1024 + 4), 2)):
        x0 = torch.randn(*shape, device=dev) * 1.3
        g = torch.randn(*shape, device=dev)
        xa = x0.clone().requires_grad_(True)
        ya = torch.relu(ops.ActQuantFn.apply(xa, k, 2.0, 0))
        ya.backward(g)
        xb = x0.clone().requires_grad_(True)
        yb = ops.ActQuantReluFn.apply(xb, k, 2.0, 0)
        yb.backward(g)
        assert bits_equal(npy(ya), npy(yb)) and bits_equal(npy(xa.grad), npy(xb.grad)), (shape, k)
        assert float(yb.min()) >= 0.0 and float((yb == 0).float().mean()) > 0.2


@pytest.mark.parametrize("B,shape", [(28, (64, 14, 14)), (16, (8, 6, 6)), (28, (256, 7, 9))])
def test_small_batch_site_with_folded_residual_and_relu(dev, B, shape):
    """ops.SiteFn(residual=, relu=True) (alignq_site_partials_res: the Office bottleneck's `out += identity; out =
    self.relu(out)` inside the site forward kernel) against relu(SiteFn(x)[0] + residual): forward bit for bit; gradients of
    x, of the residual and of the ADMM state equal (the backward runs the same kernels on the ReLU-masked gradient)."""
    from alignq_amd import ops
    torch.manual_seed(B)
    x0 = torch.randn(B, *shape, device=dev) * 1.2
    r0 = torch.randn(B, *shape, device=dev)
    g = torch.randn(B, *shape, device=dev) * 0.01
    A0, G0 = torch.rand(32, 32, device=dev), torch.rand(32, 32, device=dev)
    outs = []
    for fold in (False, True):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        A, Gm = A0.clone().requires_grad_(True), G0.clone().requires_grad_(True)
        if fold:
            assert ops.site_res_supported(x, r)
            y, loss, D = ops.SiteFn.apply(x, A, Gm, 8, 2.0, 1e-5, 0.2, 0.3, None, None, None, r, True)
        else:
            xq, loss, D = ops.SiteFn.apply(x, A, Gm, 8, 2.0, 1e-5, 0.2, 0.3)
            y = torch.relu(xq + r)
        (loss + (y * g).sum()).backward()
        outs.append(dict(y=npy(y), D=npy(D), loss=float(loss.detach()), dx=npy(x.grad), dr=npy(r.grad), dA=npy(A.grad),
                         dG=npy(Gm.grad)))
    a, b = outs
    assert bits_equal(a["y"], b["y"]) and bits_equal(a["D"], b["D"]) and a["loss"] == b["loss"]
    assert bits_equal(a["dr"], b["dr"])
    np.testing.assert_allclose(b["dx"], a["dx"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(b["dA"], a["dA"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(b["dG"], a["dG"], rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------------------------------- R1/R4 plain
@pytest.mark.parametrize("formula", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 4, 8, 32])
def test_act_quant_bit_exact_vs_oracle(dev, formula, k):
    from alignq_amd import ops
    rng = np.random.default_rng(10 + k)
    x = np.concatenate([rng.standard_normal(1 << 18) * 1.7, rng.uniform(-6, 6, 4099),
                        np.array([0.0, -0.0, 0.875 * 1.4142135, 4.0 * 1.4142135, 30.0, -30.0, 1e-30]),
                        # region boundaries of ERF32 (|z| = 0.875, 4) approached from both sides, +-inf, huge
                        np.nextafter(np.float32(0.875 * 2 ** 0.5), np.float32([0, 9] * 4)) * np.float32([1, 1, -1, -1] * 2),
                        np.linspace(5.6568, 5.6570, 201), -np.linspace(5.6568, 5.6570, 201),
                        np.linspace(1.23743, 1.23745, 201), -np.linspace(1.23743, 1.23745, 201),
                        np.array([np.inf, -np.inf, 1e30, -1e30, 3e38, -3e38])]).astype(np.float32)
    xq, bins = ops.act_quant_bins(cu(x, dev), k, 2.0, formula)
    oq, ot, ob = O.act_quant_fwd(x, k, 2.0, formula)
    assert bits_equal(npy(xq), oq)
    nanq, _ = ops.act_quant_bins(cu(np.array([np.nan, 1.0], np.float32), dev), k, 2.0, formula)
    # NaN propagates (k == 1 is sign(): torch.sign(nan) == 0, formula 1 then maps 0 -> -r)
    assert (np.isnan(npy(nanq)[0]) if k != 1 else not np.isnan(npy(nanq)[0])) and not np.isnan(npy(nanq)[1])
    if k not in (32,):
        assert np.array_equal(npy(bins), ob)


def test_act_quant_ragged_and_tiny(dev):
    from alignq_amd import ops
    for n in (1, 2, 3, 5, 1023, 1025):
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        xq, bins = ops.act_quant_bins(cu(x, dev), 4, 2.0, 0)
        oq, _, ob = O.act_quant_fwd(x, 4, 2.0, 0)
        assert bits_equal(npy(xq), oq) and np.array_equal(npy(bins), ob)


@pytest.mark.parametrize("tree,fname,formula", [("admm", "g3_act_quant_admm", 0), ("cdf", "g3_act_quant_cdfonly", 1)])
def test_act_quant_vs_reference_golden(dev, tree, fname, formula):
    from alignq_amd import ops
    g = load_golden(fname)
    r = float(g["act_range"])
    pre = g["t"] if tree == "admm" else g["c"]
    for k in (2, 4, 8):
        n = 2 ** k - 1
        x = cu(g["x"], dev).requires_grad_(True)
        xq = ops.ActQuantFn.apply(x, k, r, formula)
        xq.backward(cu(g["g"], dev))
        check_bins_vs_ref(npy(xq), g[f"xq_k{k}"], pre * n, n, 1.0 if tree == "admm" else 2.0 * r)
        np.testing.assert_allclose(npy(xq), g[f"xq_k{k}"], atol=(1.0 if tree == "admm" else 2.0 * r) / n + TOL)
        np.testing.assert_allclose(npy(x.grad), g[f"dx_k{k}"], atol=TOL, rtol=1e-4)


@pytest.mark.parametrize("k", [1, 2, 4, 8, 32])
def test_uniform_quantize_golden(dev, k):
    from alignq_amd.quantization import uniform_quantize
    g = load_golden("g1_uniform_quantize")
    x = cu(g["x"], dev).requires_grad_(True)
    y = uniform_quantize(k)(x)
    y.backward(cu(g[f"gy_k{k}"], dev))
    assert bits_equal(npy(y), g[f"y_k{k}"])
    assert bits_equal(npy(x.grad), g[f"gx_k{k}"])


# ------------------------------------------------------------------------------------------- R3 weights
@pytest.mark.parametrize("tree,fname,formula", [("admm", "g2_weight_quant_admm", 0), ("cdf", "g2_weight_quant_cdfonly", 1)])
def test_weight_quant(dev, tree, fname, formula):
    from alignq_amd import ops
    g = load_golden(fname)
    si = 0
    while f"W_s{si}" in g:
        W = g[f"W_s{si}"]
        ms = npy(ops.weight_stats(cu(W, dev)))
        np.testing.assert_allclose(ms[0], g[f"m_s{si}"], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(ms[1], g[f"s_s{si}"], rtol=2e-6)
        ms_ref = np.array([g[f"m_s{si}"], g[f"s_s{si}"]], np.float32)
        for k in (2, 4, 8):
            n = 2 ** k - 1
            # (a) given the reference's (m, s): bit-exact vs the C oracle, tie-zone rule vs the reference
            q, c, pdf, bins = ops.weight_quant_given_stats(cu(W, dev), cu(ms_ref, dev), k, formula, True, True)
            oq, oc, opdf, ob = O.weight_quant_fwd(W, ms_ref, k, formula)
            assert bits_equal(npy(q), oq) and bits_equal(npy(c), oc) and np.array_equal(npy(bins), ob)
            np.testing.assert_allclose(npy(pdf), opdf, rtol=1e-6, atol=1e-7)
            check_bins_vs_ref(npy(q), g[f"Wq_s{si}_k{k}"], g[f"cdf_s{si}"] * n, n, 1.0 if tree == "admm" else 2.0)
            np.testing.assert_allclose(npy(c), g[f"cdf_s{si}"], atol=3e-7)
            np.testing.assert_allclose(npy(pdf), g[f"pdf_s{si}"], rtol=3e-6, atol=1e-6)
            # (b) end to end through the autograd Function (own stats), incl. backward through mean/std
            Wt = cu(W, dev).requires_grad_(True)
            q2, c2, p2 = ops.WeightQuantFn.apply(Wt, k, formula)
            q2.backward(cu(g[f"g_s{si}"], dev))
            np.testing.assert_allclose(npy(c2), g[f"cdf_s{si}"], atol=1e-6)
            np.testing.assert_allclose(npy(q2), g[f"Wq_s{si}_k{k}"], atol=(1.0 if tree == "admm" else 2.0) / n + TOL)
            np.testing.assert_allclose(npy(Wt.grad), g[f"dW_s{si}_k{k}"], atol=2e-5, rtol=1e-4)
        si += 1


# ------------------------------------------------------------------------------------------- R5 corr
@pytest.mark.parametrize("fname,eps", [("g4_corr_noeps", 0.0), ("g4_corr_eps", 1e-5)])
def test_corr_vs_reference(dev, fname, eps):
    from alignq_amd import ops
    g = load_golden(fname)
    ci = 0
    while f"x_c{ci}" in g:
        x = cu(g[f"x_c{ci}"], dev).requires_grad_(True)
        G = ops.CorrFn.apply(x, eps)
        G.backward(cu(g[f"dG_c{ci}"], dev))
        np.testing.assert_allclose(npy(G), g[f"G_c{ci}"], atol=TOL, rtol=0)
        ref = g[f"dx_c{ci}"]
        np.testing.assert_allclose(npy(x.grad), ref, atol=TOL * max(1.0, np.abs(ref).max()), rtol=1e-4)
        ci += 1


@pytest.mark.parametrize("B,F", [(128, 4096), (128, 16384), (28, 1568), (64, 640), (10, 100), (33, 70), (3, 64), (2, 64)])
def test_corr_vs_oracle_shapes(dev, B, F):
    """ragged shapes: F not a multiple of the 64-feature tile or of 4, B not a multiple of 32; B = 2 is the smallest batch the
    ABI advertises (include/alignq.h: 2 <= B).

    The bar on dx scales with the size of the terms it is a difference of: dx_f = rho_f * (dXh - mean(dXh) - xh * proj), each
    term of magnitude |S| |xh| rho_f / F with rho_f = 1 / std_f.  At B = 2 every standardised column is exactly +-1/sqrt(2)
    and the analytic gradient is ZERO (the three terms cancel), while rho_f reaches 10^2 for columns whose two samples
    nearly coincide; the kernels evaluate S * Xh on split-bf16 operands (16 mantissa bits, 2^-16 relative to the TERMS, the
    accuracy that keeps D within 6e-7), so the residue is 2^-16 * max-term, not 1e-5 absolute.  (Round 1 saw 2.2e-5 here
    and dropped the case; it is the conditioning of the expression, the same code passes 1e-5 wherever rho is O(1).)"""
    from alignq_amd import ops
    rng = np.random.default_rng(B * 1000 + F)
    x = (rng.standard_normal((B, F)) * 0.7 + 0.2).astype(np.float32)
    dG = rng.standard_normal((B, B)).astype(np.float32)
    xt = cu(x, dev).requires_grad_(True)
    G = ops.CorrFn.apply(xt, 0.0)
    G.backward(cu(dG, dev))
    np.testing.assert_allclose(npy(G), O.corr_fwd(x, 0.0), atol=TOL, rtol=0)
    ref = O.corr_bwd(dG, x, 0.0)
    rho = 1.0 / x.astype(np.float64).std(0, ddof=1)
    term = np.abs(dG).max() * 2 * np.sqrt(B) / F * rho                   # per-column magnitude of the cancelling terms
    atol = np.maximum(TOL * max(1.0, np.abs(ref).max()), 2.0 ** -15 * term)[None, :]
    assert (np.abs(npy(xt.grad) - ref) <= atol + 1e-4 * np.abs(ref)).all(), float(np.abs(npy(xt.grad) - ref).max())


# ------------------------------------------------------------------------------------------- R4+R5+R6 site
def _site(dev, x, g, A0, G0, k, r, eps, mu=0.2, rho=0.3):
    from alignq_amd import ops
    xt = cu(x, dev).requires_grad_(True)
    A = cu(A0, dev).requires_grad_(True)
    Gm = cu(G0, dev).requires_grad_(True)
    xq, loss, D = ops.SiteFn.apply(xt, A, Gm, k, r, eps, mu, rho)
    (loss + (xq * cu(g, dev)).sum()).backward()
    return npy(xq), float(loss), npy(D), npy(xt.grad), npy(A.grad), npy(Gm.grad)


@pytest.mark.parametrize("name", ["a", "b", "short"])
def test_site_vs_reference(dev, name):
    g = load_golden("g5_g6_admm_site")
    k, x = int(g[f"k_{name}"]), g[f"x_{name}"]
    n = 2 ** k - 1
    xq, loss, D, dx, dA, dG = _site(dev, x, g[f"g_{name}"], g[f"alterD0_{name}"], g[f"gamma0_{name}"], k, 2.0, 0.0)
    oq, ot, _ = O.act_quant_fwd(x, k, 2.0, 0)
    assert bits_equal(xq, oq)                                   # same spec => same bits as the C oracle
    check_bins_vs_ref(xq, g[f"xq_{name}"], ot.astype(np.float64) * n, n)
    np.testing.assert_allclose(D, g[f"D_{name}"], atol=TOL, rtol=0)
    np.testing.assert_allclose(loss, g[f"loss_{name}"], atol=TOL)
    np.testing.assert_allclose(dx, g[f"dx_{name}"], atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(dA, g[f"dalterD_{name}"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(dG, g[f"dgamma_{name}"], atol=1e-7, rtol=1e-4)


def test_site_office_vs_reference(dev):
    g = load_golden("g5_office_site")
    k, x, r = int(g["k"]), g["x"], float(g["act_range"])
    xq, loss, D, dx, dA, dG = _site(dev, x, g["g"], g["alterD0"], g["gamma0"], k, r, 1e-5)
    np.testing.assert_allclose(D, g["D"], atol=TOL, rtol=0)
    np.testing.assert_allclose(loss, g["loss"], atol=TOL)
    np.testing.assert_allclose(dx, g["dx"], atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(dA, g["dalterD"], atol=1e-7, rtol=1e-4)


@pytest.mark.parametrize("B,C,H,W,k", [(128, 16, 32, 32, 8), (128, 64, 8, 8, 2), (28, 8, 14, 14, 8), (64, 3, 5, 7, 4),
                                      (100, 16, 16, 16, 4)])
def test_site_vs_oracle_shapes(dev, B, C, H, W, k):
    """CIFAR site shapes (incl. the largest, 128x16384), an Office-like B=28 site, ragged F, eval batch 100."""
    rng = np.random.default_rng(B + C + H)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    gq = (rng.standard_normal(x.shape) * 0.01).astype(np.float32)
    A0, G0 = rng.random((128, 128)).astype(np.float32), rng.random((128, 128)).astype(np.float32)
    xq, loss, D, dx, dA, dG = _site(dev, x, gq, A0, G0, k, 2.0, 0.0)
    oq, oD = O.site_fwd(x, k, 2.0, 0.0)
    assert bits_equal(xq, oq)
    np.testing.assert_allclose(D, oD, atol=TOL, rtol=0)
    ol, odD, odA, odG = O.admm_loss(oD, A0, G0, 0.2, 0.3)
    np.testing.assert_allclose(loss, ol, atol=TOL)
    odx = O.site_bwd(gq, odD, x, 2.0, 0.0)
    np.testing.assert_allclose(dx, odx, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(dA, odA, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(dG, odG, atol=1e-7, rtol=1e-4)


def test_site_properties_full_size(dev):
    """Size-independent properties at config-5's largest site (28 x 802816): D symmetric, zero diagonal-sum
    identity trace(corr)= (B-1) for both matrices => trace(D) = 0, x_q on the k-bit lattice, D invariant under
    a per-feature affine map of x in the corr(x,x) term (checked through corr alone)."""
    from alignq_amd import ops
    torch.manual_seed(0)
    B, F, k = 28, 802816, 8
    x = torch.randn(B, F, device=dev)
    A = torch.rand(B, B, device=dev)
    Gm = torch.rand(B, B, device=dev)
    xq, loss, D = ops.SiteFn.apply(x, A, Gm, k, 2.0, 1e-5, 0.2, 0.3)
    D = npy(D)
    assert np.allclose(D, D.T, atol=1e-6)
    assert abs(np.trace(D)) < 1e-3
    lat = npy(xq[:, :4096]) * 255.0
    assert np.all(np.abs(lat - np.rint(lat)) < 1e-3) and np.abs(lat).max() <= 510
    G1 = npy(ops.CorrFn.apply(x[:, :100352].contiguous(), 0.0))
    G2 = npy(ops.CorrFn.apply((x[:, :100352] * 3.0 + 1.5).contiguous(), 0.0))
    np.testing.assert_allclose(G1, G2, atol=2e-5)
    assert abs(np.trace(G1) - (B - 1)) < 1e-3


# ------------------------------------------------------------------------------------------- R6/R7
def test_admm_loss_and_update(dev):
    from alignq_amd.admm import ADMM
    from alignq_amd.optimizer import ADMM_OPT
    from alignq_amd import config
    g = load_golden("g5_g6_admm_site")
    config.args.bitW = 8
    for name in ("a", "b", "short", "small"):
        dim = g[f"alterD0_{name}"].shape[0]
        admm = ADMM(dim).to(dev)
        with torch.no_grad():
            admm.alterD.copy_(cu(g[f"alterD0_{name}"], dev))
            admm.gamma.copy_(cu(g[f"gamma0_{name}"], dev))
        D = cu(g[f"D_{name}"], dev).requires_grad_(True)
        loss = admm(D)
        loss.backward()
        np.testing.assert_allclose(float(loss), g[f"loss_{name}"], atol=TOL)
        np.testing.assert_allclose(npy(admm.alterD.grad), g[f"dalterD_{name}"], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(npy(admm.gamma.grad), g[f"dgamma_{name}"], atol=1e-6, rtol=1e-4)
        if name == "small":
            np.testing.assert_allclose(npy(D.grad), g["dD_small"], atol=1e-6)
        opt = ADMM_OPT([admm.alterD, admm.gamma])
        a_ptr, g_ptr = admm.alterD.data_ptr(), admm.gamma.data_ptr()
        opt.step([0], [1], [admm.D], [admm.alterD], [admm.gamma], [admm.mu], [admm.rho])
        np.testing.assert_allclose(npy(admm.alterD), g[f"alterD1_{name}"], atol=TOL)
        np.testing.assert_allclose(npy(admm.gamma), g[f"gamma1_{name}"], atol=TOL)
        assert (admm.alterD.data_ptr(), admm.gamma.data_ptr()) == (a_ptr, g_ptr)    # in place: graph-safe
    assert np.all(g["alterD1_small"] == 0)


def test_admm_opt_skips_params_without_grad(dev):
    from alignq_amd.admm import ADMM
    from alignq_amd.optimizer import ADMM_OPT
    admm = ADMM(8).to(dev)
    before = npy(admm.alterD).copy()
    opt = ADMM_OPT([admm.alterD, admm.gamma])
    opt.step([0], [1], [torch.zeros(8, 8, device=dev)], [admm.alterD], [admm.gamma], [0.2], [0.3])
    assert np.array_equal(npy(admm.alterD), before)


# ------------------------------------------------------------------------------------------- R8
def test_sgd_step_vs_reference(dev):
    from alignq_amd.optimizer import SGD
    from alignq_amd import config
    g = load_golden("g7_sgd_step")
    config.args.bitW = int(g["bitW"])
    ps = [torch.nn.Parameter(cu(g[f"p{i}_0"], dev)) for i in range(3)]
    opt = SGD(ps, lr=0.04, momentum=0.9, weight_decay=1e-4)
    for step in (1, 2):
        for i, p in enumerate(ps):
            p.grad = cu(g[f"grad{i}_{step}"], dev)
        opt.step([1], [cu(g["w_cdf"], dev)], [cu(g["w_pdf"], dev)], float(g["lam"]), float(g["lam2"]))
        for i, p in enumerate(ps):
            np.testing.assert_allclose(npy(p), g[f"p{i}_{step}"], atol=1e-6)
            np.testing.assert_allclose(npy(opt.state[p]["momentum_buffer"]), g[f"buf{i}_{step}"], atol=1e-6, rtol=1e-6)
            np.testing.assert_allclose(npy(p.grad), g[f"gradout{i}_{step}"], atol=1e-5, rtol=1e-5)
    config.args.bitW = 8
    assert "momentum_buffer" in opt.state_dict()["state"][0]


# ------------------------------------------------------------------------------------------- whole model
def _load_ref_state(net, g, prefix):
    sd = net.state_dict()
    for key, v in g.items():
        if key.startswith(prefix):
            sd[key[len(prefix):]] = torch.from_numpy(v)
    net.load_state_dict(sd, strict=True)


def _ref_to_oracle_name(n):
    if ".opt." in n or n.startswith("act_q"):
        return None
    n = n.replace("admm_skip.", "site_skip.admm.")
    return n.replace("admm0.", "site0.admm.").replace("admm1.", "site1.admm.")


def test_tiny_resnet_two_steps_vs_reference(dev):
    """G8: PreActResNet([1,1,1]) B=8 k=4, two full iterations in the reference's order; the harness model keeps the
    reference's parameter names, so the captured state_dict loads as is.

    Two comparisons:
      * against the golden captured from the reference on torch-CPU.  The convolutions (MIOpen here, oneDNN there)
        differ by ~1e-6, which flips a handful of the 4-bit activation bins per forward, each flip moving a logit
        by ~1e-3: whole-network values are therefore compared at bin-flip scale (1e-2), the smooth quantities
        (trans_loss, D) tighter;
      * against the eager-torch oracle run ON THE SAME GPU (same convolutions): torch-GPU itself divides by the
        level count with a reciprocal multiply, so weights/activations already differ in the last ulp and the same
        bin-flip amplification applies (measured: one flip at the third site moves the logits by 2e-3); it is a
        sanity bound, not a tight check.  Tight parity is established per site, teacher-forced, in the tests above.
    """
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    from oracle import torch_ref as R
    g = load_golden("g8_tiny_resnet_admm")
    config.args.bitW = config.args.abitW = 4
    config.args.train_batch_size = 8
    try:
        net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10)
        _load_ref_state(net, g, "init/")
        net = net.to(dev).train()
        step = TrainStep(net)
        cfg = R.Config(tree="admm", bitW=4, abitW=4, train_batch_size=8)
        onet = R.PreActResNet(cfg, [1, 1, 1], 4, 4)
        onet.load_state_dict({_ref_to_oracle_name(k[5:]): torch.from_numpy(v) for k, v in g.items()
                              if k.startswith("init/") and _ref_to_oracle_name(k[5:]) is not None}, strict=True)
        onet = onet.to(dev).train()
        ostep = R.TrainStep(onet, cfg)
        for it in range(2):
            x, y = cu(g["xs"][it], dev), cu(g["ys"][it], dev)
            logits, ce, tl = step(x, y)
            ologits, oce, otl = ostep(x, y)
            # vs the reference golden (CPU convolutions)
            # iteration 1 starts from weights that already differ at bin-flip scale: sanity bound only
            np.testing.assert_allclose(npy(logits), g[f"logits_{it}"], atol=2e-2 if it == 0 else 0.15)
            np.testing.assert_allclose(float(ce), g[f"ce_{it}"], atol=5e-3 if it == 0 else 3e-2)
            np.testing.assert_allclose(float(tl), g[f"trans_{it}"], atol=2e-3)
            for si, m in enumerate(step.admms):
                np.testing.assert_allclose(npy(m.D), g[f"D_{it}_{si}"], atol=2e-3)
            # vs the oracle on the same GPU
            d = np.abs(npy(logits) - npy(ologits))
            assert np.median(d) < (5e-3 if it == 0 else 3e-2) and d.max() < (2e-2 if it == 0 else 0.15), (np.median(d), d.max())
            np.testing.assert_allclose(float(tl), float(otl), atol=2e-4)
            np.testing.assert_allclose(float(ce), float(oce), atol=2e-3 if it == 0 else 3e-2)
            got, ogot = net.state_dict(), onet.state_dict()
            for key, v in g.items():
                if key.startswith(f"after{it}/") and "num_batches" not in key:
                    name = key[len(f"after{it}/"):]
                    np.testing.assert_allclose(npy(got[name]), v, atol=2e-2, rtol=2e-2, err_msg=key)
                    on = _ref_to_oracle_name(name)
                    if on is not None:
                        ref_v = npy(ogot[on])
                        dd = np.abs(npy(got[name]) - ref_v) / (np.abs(ref_v) + 0.1)
                        assert np.median(dd) < 2e-2, (name, np.median(dd), dd.max())
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


def test_cdf_only_tree_two_steps_vs_oracle(dev):
    """BASELINE config 1's tree (cdf_alignment/resnet-20-cifar-10: CDF quantisers, no ADMM sites, plain momentum SGD because
    main.py:308 crashes as shipped — SURVEY F6a): a small PreActResNet, two full iterations of TrainStep against the
    eager-torch oracle on the same GPU from the same initial state.  The two differ by last-ulp quantiser arithmetic, so
    whole-network values are compared at bin-flip scale (see test_tiny_resnet_two_steps_vs_reference)."""
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    from oracle import torch_ref as R
    config.args.bitW = config.args.abitW = 4
    config.args.train_batch_size = 8
    try:
        torch.manual_seed(21)
        net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10, tree="cdf").to(dev).train()
        cfg = R.Config(tree="cdf", bitW=4, abitW=4, train_batch_size=8)
        onet = R.PreActResNet(cfg, [1, 1, 1], 4, 4)
        onet.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, strict=True)
        onet = onet.to(dev).train()
        step, ostep = TrainStep(net), R.TrainStep(onet, cfg)
        assert not step.admms
        g = torch.Generator().manual_seed(5)
        for it in range(2):
            x = torch.randn(8, 3, 32, 32, generator=g).to(dev)
            y = torch.randint(0, 10, (8,), generator=g).to(dev)
            logits, ce, tl = step(x, y)
            ologits, oce, _ = ostep(x, y)
            assert tl is None or float(tl) == 0.0
            d = np.abs(npy(logits) - npy(ologits))
            assert np.median(d) < (5e-3 if it == 0 else 3e-2) and d.max() < (3e-2 if it == 0 else 0.15), (np.median(d), d.max())
            np.testing.assert_allclose(float(ce.detach()), float(oce.detach()), atol=3e-3 if it == 0 else 3e-2)
        got, ogot = net.state_dict(), onet.state_dict()
        for name, v in ogot.items():
            if "num_batches" in name:
                continue
            ref_v = npy(v)
            dd = np.abs(npy(got[name]) - ref_v) / (np.abs(ref_v) + 0.1)
            assert np.median(dd) < 2e-2, (name, float(np.median(dd)), float(dd.max()))
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


def test_graph_capture_matches_eager(dev):
    """The captured HIP graph of the full iteration reproduces eager iterations."""
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 4
    config.args.train_batch_size = 16
    try:
        torch.manual_seed(1)
        nets = []
        for _ in range(2):
            torch.manual_seed(1)
            nets.append(PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10).to(dev).train())
        x = torch.randn(16, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (16,), device=dev)
        eager, graphed = TrainStep(nets[0]), TrainStep(nets[1])
        for _ in range(3):
            eager(x, y)
        graphed.capture(x, y, warmup=3)          # 3 real warm-up iterations
        for _ in range(2):
            eager(x, y)
            graphed(x, y)
        torch.cuda.synchronize()
        # eager did 5 iterations; graphed did 3 warm-up + 1 capture pass (not executed) + 2 replays = 5.
        # NAMED, BOUNDED EXCEPTION to round 6's bit-for-bit rule (tests/test_gpu_round6.py asserts equality on every benchmarked
        # configuration, whose convolutions are this repository's kernels): this small NCHW step runs F.conv2d on MIOpen, which
        # picks its filter-gradient algorithm per call context (measured on MI355X: conv weights differ by 1-3e-8 = one ulp after five
        # steps, nothing else does) - library code outside this repository.  Bound: 1e-6 absolute / 1e-5 relative.
        from tests.test_gpu_round6 import differing, full_state
        sa, sb = full_state(nets[0], eager, eager.admms), full_state(nets[1], graphed, graphed.admms)
        bad = differing(sa, sb)
        print("MIOpen-path graph vs eager: tensors differing bitwise:", [(b_[0], b_[3]) for b_ in bad])
        for key in sa:
            np.testing.assert_allclose(sa[key], sb[key], atol=1e-6, rtol=1e-5, err_msg=key)
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


def test_product_path_refuses_cpu_tensors():
    from alignq_amd import ops
    with pytest.raises(RuntimeError):
        ops.ActQuantFn.apply(torch.randn(8), 4, 2.0, 0)


def test_prequantize_all_weights_matches_per_tensor(dev):
    """fused.prequantize_weights (multi-tensor kernels) == per-tensor WeightQuantFn, forward bits and gradients."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import ops
    from alignq_amd.fused import prequantize_weights
    torch.manual_seed(3)
    Conv = A.conv2d_Q_fn(4, "second")
    convs = [Conv(3, 16, 3, 1, 1, bias=False), Conv(16, 32, 3, 2, 1, bias=False), Conv(32, 64, 1, 2, 0, bias=False),
             Conv(64, 64, 3, 1, 1, bias=False)]
    convs = [c.to(dev) for c in convs]
    gs = [torch.randn_like(c.weight) for c in convs]
    prequantize_weights(convs)
    qs = [c.quantize_fn(c.weight) for c in convs]
    assert all(c.quantize_fn._pre is None for c in convs)
    sum((q * g).sum() for q, g in zip(qs, gs)).backward()
    for c, q, g in zip(convs, qs, gs):
        w2 = c.weight.detach().clone().requires_grad_(True)
        q2, c2, p2 = ops.WeightQuantFn.apply(w2, 4, 0)
        q2.backward(g)
        assert bits_equal(npy(q), npy(q2)) and bits_equal(npy(c.quantize_fn.weight_cdf), npy(c2))
        np.testing.assert_allclose(npy(c.quantize_fn.weight_pdf), npy(p2), rtol=1e-6)
        np.testing.assert_allclose(npy(c.weight.grad), npy(w2.grad), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("channels_last", [False, True])
def test_office_dann_harness_runs_and_matches_eager_under_graph(dev, channels_last):
    """Config-5 harness (ResNet-Bottleneck + DANN head, Office tree ops: eps-corr, activation_quantize_fn2, two passes per
    step, three SGD groups incl. alterD/gamma): a small instance runs, every ADMM site takes the TARGET pass's D, and
    the HIP-graph replay reproduces eager iterations."""
    from alignq_amd import config
    from alignq_amd.resnet_office import DANN, Bottleneck, ResNet
    from alignq_amd.train_step import OfficeTrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = config.args.eval_batch_size = 6
    try:
        def make():
            torch.manual_seed(7)
            m = DANN(lambda w, a, s: ResNet(w, a, s, Bottleneck, [1, 1, 1, 1]), 8, 8, "aligned").to(dev).train()
            return m
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(6, 3, 64, 64, generator=g).to(dev)
        xt = torch.randn(6, 3, 64, 64, generator=g).to(dev)
        ys = torch.randint(0, 31, (6,), generator=g).to(dev)
        cl = dict(channels_last=channels_last)
        OfficeTrainStep(make(), lr=0.004, **cl)(xs, ys, xt)      # throw-away: lets MIOpen settle its solver choice per shape
        m0 = make()
        OfficeTrainStep(m0, lr=0.004, **cl)                        # (applies the memory format, marks the blocks)
        outs = []
        for fuse in (False, True):                                 # quantiser + ReLU in one launch: same forward values
            for mod in m0.modules():
                if hasattr(mod, "fuse_relu"):
                    mod.fuse_relu = fuse
            outs.append([npy(t) for t in m0(xs.contiguous(memory_format=torch.channels_last) if channels_last else xs, 0.5)[:2]])
        # (bit-exact per operator, test_act_quant_with_fused_relu_equals_the_composition; across the network MIOpen's strided
        # convolutions already differ by 1e-5 between two identical calls, which flips 8-bit bins: bin-flip scale here)
        np.testing.assert_allclose(outs[1][0], outs[0][0], atol=5e-2)
        np.testing.assert_allclose(outs[1][1], outs[0][1], atol=5e-2)
        m1, m2 = make(), make()
        s1, s2 = OfficeTrainStep(m1, lr=0.004, **cl), OfficeTrainStep(m2, lr=0.004, **cl)
        a0 = npy(m1.feature.layer1[0].admm0.alterD).copy()
        for _ in range(2):
            cls, loss, tl = s1(xs, ys, xt)
        assert torch.isfinite(loss) and torch.isfinite(tl) and torch.isfinite(cls).all()
        assert not np.array_equal(npy(m1.feature.layer1[0].admm0.alterD), a0)
        assert m1.feature.layer1[0].admm0.D.shape == (6, 6)
        s2.capture(xs, ys, xt, warmup=2)       # 2 real warm-up iterations
        s1(xs, ys, xt)
        s2(xs, ys, xt)
        torch.cuda.synchronize()
        # MIOpen's weight-gradient kernels accumulate with atomics and 8-bit bins flip on 1e-6 perturbations, so the two
        # trajectories agree only at bin-flip scale (see test_tiny_resnet...): median tight, worst element loose
        for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            d = np.abs(npy(p1) - npy(p2))
            assert np.median(d) < 1e-3 and d.max() < 2e-2, (n1, float(np.median(d)), float(d.max()))
    finally:
        config.args.train_batch_size, config.args.eval_batch_size = 128, 100


@pytest.mark.parametrize("relu,residual", [(False, False), (True, False), (True, True), (False, True)])
@pytest.mark.parametrize("B,C,H,W,k", [(128, 16, 32, 32, 8), (128, 64, 8, 8, 4), (100, 32, 16, 16, 8)])
def test_bn_folded_site_matches_unfused(dev, B, C, H, W, k, relu, residual):
    """fused.bn_site (batch-norm folded into the site kernels, training mode) against act(bn(z)) with torch's BatchNorm2d
    + the unfused site: x_q equal up to tie-zone bin flips (x differs by one fma rounding), D / loss / dz / dgamma / dbeta /
    running statistics within fp32 tolerance."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import config
    from alignq_amd.fused import bn_site, bn_site_fusable
    config.args.bitW = config.args.abitW = k
    torch.manual_seed(B + C)
    z = (torch.randn(B, C, H, W, device=dev) * 1.7 + 0.3)
    gq = torch.randn(B, C, H, W, device=dev) * 0.01
    res0 = torch.randn(B, C, H, W, device=dev) * 0.7
    outs = []
    for fused in (False, True):
        torch.manual_seed(1)
        res = res0.clone().requires_grad_(True) if residual else None
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, device=dev) + 0.5)
            bn.bias.copy_(torch.randn(C, device=dev) * 0.2)
        admm = A.ADMM(128).to(dev)
        act = A.activation_quantize_fn(k, "second", admm)
        zz = z.clone().requires_grad_(True)
        if fused:
            assert bn_site_fusable(bn, act, zz)
            xq, loss = bn_site(bn, act, zz, relu=relu, residual=res)
        else:
            xq, loss = act(bn(zz))
            if residual:
                xq = xq + res
            if relu:
                xq = torch.nn.functional.relu(xq)
        (loss + (xq * gq).sum()).backward()
        outs.append(dict(xq=npy(xq), loss=float(loss.detach()), D=npy(admm.D), dz=npy(zz.grad), dw=npy(bn.weight.grad),
                         db=npy(bn.bias.grad), rm=npy(bn.running_mean), rv=npy(bn.running_var),
                         nbt=int(bn.num_batches_tracked), dA=npy(admm.alterD.grad),
                         dres=npy(res.grad) if residual else None))
    u, f = outs
    n = 2 ** k - 1
    flips = np.abs(u["xq"] - f["xq"]) * n
    assert flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 1e-3          # tie-zone flips only
    np.testing.assert_allclose(f["D"], u["D"], atol=TOL)
    np.testing.assert_allclose(f["loss"], u["loss"], atol=TOL)
    np.testing.assert_allclose(f["rm"], u["rm"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(f["rv"], u["rv"], atol=1e-6, rtol=1e-5)
    assert f["nbt"] == u["nbt"] == 1
    np.testing.assert_allclose(f["dA"], u["dA"], atol=1e-7, rtol=1e-4)
    # gradients see the flipped bins only through g (STE), so they agree to tolerance
    np.testing.assert_allclose(f["dz"], u["dz"], atol=2e-5, rtol=1e-3)
    np.testing.assert_allclose(f["dw"], u["dw"], atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(f["db"], u["db"], atol=2e-4, rtol=1e-3)
    if residual:        # gq masked by relu(x_q + res) > 0: differs only where a flipped bin moves the sum across 0
        assert (f["dres"] != u["dres"]).mean() < 1e-3
    config.args.bitW = config.args.abitW = 8


@pytest.mark.parametrize("B", [128, 100])
def test_batched_deferred_sites_match_per_site_launches(dev, B):
    """fused.DeferredLosses(batch=True): one alignq_site_reduce_loss_multi / alignq_site_prep_fused_multi launch for all
    sites of a step must give bit-identical D, x_q and gradients to the per-site launches (same kernel bodies), and the
    same loss sum up to the order of one fp32 sum.  Mixed plain (SiteFn) and BN-folded (BNSiteFn) sites, different F,
    upstream loss scale != 1."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import config
    from alignq_amd.fused import DeferredLosses, bn_site
    config.args.bitW = config.args.abitW = 8
    shapes = [(16, 32, 32), (32, 16, 16), (64, 8, 8), (3, 8, 8)]
    torch.manual_seed(5)
    zs = [torch.randn(B, *s, device=dev) * 1.3 + 0.1 for s in shapes]
    gqs = [torch.randn(B, *s, device=dev) * 0.01 for s in shapes]
    res = []
    for batch in (False, True):
        torch.manual_seed(2)
        admms = [A.ADMM(128).to(dev) for _ in shapes]
        for a in admms:
            with torch.no_grad():
                a.alterD.copy_(torch.randn(128, 128, device=dev) * 0.05)
                a.gamma.copy_(torch.randn(128, 128, device=dev) * 0.05)
        acts = [A.activation_quantize_fn(8, "second", a) for a in admms]
        bns = [torch.nn.BatchNorm2d(s[0]).to(dev).train() for s in shapes]
        xs = [z.clone().requires_grad_(True) for z in zs]
        d = DeferredLosses(batch=batch)
        with d:
            outs = []
            for i, (x, act, bn) in enumerate(zip(xs, acts, bns)):
                if i % 2 == 0:
                    xq, l = act(x)
                else:
                    xq, l = bn_site(bn, act, x, relu=(i == 1))
                assert l == 0.0
                outs.append(xq)
            total = d.total()
        assert len(d.records) == (len(shapes) if batch else 0)
        obj = 0.7 * total + sum((o * g).sum() for o, g in zip(outs, gqs))
        obj.backward()
        res.append(dict(total=float(total.detach()), xq=[npy(o) for o in outs], D=[npy(a.D) for a in admms],
                        dx=[npy(x.grad) for x in xs], dA=[npy(a.alterD.grad) for a in admms],
                        dG=[npy(a.gamma.grad) for a in admms],
                        dw=[npy(bn.weight.grad) for i, bn in enumerate(bns) if i % 2 == 1]))
    u, b = res
    np.testing.assert_allclose(b["total"], u["total"], rtol=1e-6)
    for key in ("xq", "D", "dx", "dA", "dG", "dw"):
        for x, y in zip(u[key], b[key]):
            assert np.array_equal(x, y), key


@pytest.mark.parametrize("relu,residual", [(False, False), (True, True)])
@pytest.mark.parametrize("B,C,H,W,k", [(128, 16, 32, 32, 8), (128, 64, 8, 8, 4), (100, 32, 16, 16, 8), (128, 8, 4, 4, 8)])
def test_bn_folded_site_channels_last_matches_nchw(dev, B, C, H, W, k, relu, residual):
    """The channels-last (torch.channels_last) form of the BN fold against the unfused composition on contiguous tensors:
    the same logical tensors in the other memory layout must give the same results (tie-zone bin flips only; D is
    invariant under the feature permutation up to summation order)."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import config
    from alignq_amd.fused import bn_site, bn_site_fusable
    config.args.bitW = config.args.abitW = k
    torch.manual_seed(B + C + 1)
    z = (torch.randn(B, C, H, W, device=dev) * 1.7 + 0.3)
    gq = torch.randn(B, C, H, W, device=dev) * 0.01
    res0 = torch.randn(B, C, H, W, device=dev) * 0.7
    outs = []
    for nhwc in (False, True):
        fmt = torch.channels_last if nhwc else torch.contiguous_format
        torch.manual_seed(1)
        res = res0.clone(memory_format=fmt).requires_grad_(True) if residual else None
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, device=dev) + 0.5)
            bn.bias.copy_(torch.randn(C, device=dev) * 0.2)
        admm = A.ADMM(128).to(dev)
        act = A.activation_quantize_fn(k, "second", admm)
        zz = z.clone(memory_format=fmt).requires_grad_(True)
        if nhwc:
            assert bn_site_fusable(bn, act, zz) and not zz.is_contiguous()
            for _ in range(2):          # twice: no state may leak between calls
                bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
                xq, loss = bn_site(bn, act, zz, relu=relu, residual=res)
            assert xq.is_contiguous(memory_format=torch.channels_last)
        else:
            xq, loss = act(bn(zz))
            if residual:
                xq = xq + res
            if relu:
                xq = torch.nn.functional.relu(xq)
        (loss + (xq * gq).sum()).backward()
        outs.append(dict(xq=npy(xq), loss=float(loss.detach()), D=npy(admm.D), dz=npy(zz.grad), dw=npy(bn.weight.grad),
                         db=npy(bn.bias.grad), rm=npy(bn.running_mean), rv=npy(bn.running_var),
                         nbt=int(bn.num_batches_tracked), dA=npy(admm.alterD.grad),
                         dres=npy(res.grad) if residual else None))
    u, f = outs
    n = 2 ** k - 1
    flips = np.abs(u["xq"] - f["xq"]) * n
    assert flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 1e-3
    np.testing.assert_allclose(f["D"], u["D"], atol=TOL)
    np.testing.assert_allclose(f["loss"], u["loss"], atol=TOL)
    np.testing.assert_allclose(f["rm"], u["rm"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(f["rv"], u["rv"], atol=1e-6, rtol=1e-5)
    assert f["nbt"] == u["nbt"] == 1
    np.testing.assert_allclose(f["dA"], u["dA"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(f["dz"], u["dz"], atol=2e-5, rtol=1e-3)
    np.testing.assert_allclose(f["dw"], u["dw"], atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(f["db"], u["db"], atol=2e-4, rtol=1e-3)
    if residual:
        assert (f["dres"] != u["dres"]).mean() < 1e-3
    config.args.bitW = config.args.abitW = 8


@pytest.mark.parametrize("k", [2, 4, 8])
def test_uniform_admm_ablation(dev, k):
    """alignq_amd.uniform_admm (model/quantization_uniform_admm.py, the use_cdf=False ablation): plain uniform quantisers,
    D = corr(x,x) - corr(x,x) == 0, trans_loss = admm(0); STE gradient; alterD / gamma receive the loss gradient."""
    import alignq_amd.uniform_admm as U
    from alignq_amd import config
    import oracle.torch_ref as R
    torch.manual_seed(k)
    B = 64
    x = torch.randn(B, 8, 6, 6, device=dev).requires_grad_(True)
    admm = U.ADMM(B).to(dev)
    act = U.activation_quantize_fn(k, "second", admm)
    xq, loss = act(x)
    g = torch.randn_like(xq)
    (loss + (xq * g).sum()).backward()
    n = 2 ** k - 1
    ref_q = torch.round(x.detach().cpu() * n) / n              # IEEE division on the CPU (torch-GPU multiplies by a reciprocal)
    assert bits_equal(npy(xq), npy(ref_q))
    assert np.array_equal(npy(admm.D), np.zeros((B, B), np.float32))
    A0, G0 = admm.alterD.detach().cpu().requires_grad_(True), admm.gamma.detach().cpu().requires_grad_(True)
    ref_loss = R.admm_loss(torch.zeros(B, B), A0, G0, admm.mu, admm.rho)
    ref_loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=1e-6)
    np.testing.assert_allclose(npy(admm.alterD.grad), A0.grad.numpy(), atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(npy(admm.gamma.grad), G0.grad.numpy(), atol=1e-9, rtol=1e-5)
    assert np.array_equal(npy(x.grad), npy(g))                     # pure straight-through
    w = torch.randn(16, 8, 3, 3, device=dev) * 0.3
    conv = U.conv2d_Q_fn(k, "second")(8, 16, 3, 1, 1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    conv(x.detach())
    assert bits_equal(npy(conv.quantize_fn.weight_q), npy(torch.round(w.cpu() * n) / n))


def test_set_lr_recaptures_graph(dev):
    """TrainStep.set_lr on a captured step (the reference's per-epoch StepLR): the re-captured graph must use the new rate.
    With lr = 0 no trainable tensor may move; alterD/gamma still follow their closed-form update."""
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 4
    config.args.train_batch_size = 16
    try:
        torch.manual_seed(2)
        net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10).to(dev).train()
        x = torch.randn(16, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (16,), device=dev)
        step = TrainStep(net, lr=0.04, momentum=0.0, weight_decay=0.0).capture(x, y, warmup=2)
        w0 = net.layers[0].conv0.weight.detach().clone()
        step(x, y)
        torch.cuda.synchronize()
        assert not torch.equal(w0, net.layers[0].conv0.weight)
        step.set_lr(0.0)
        w1 = net.layers[0].conv0.weight.detach().clone()
        a1 = net.layers[0].admm0.alterD.detach().clone()
        step(x, y)
        step(x, y)
        torch.cuda.synchronize()
        assert torch.equal(w1, net.layers[0].conv0.weight)
        assert not torch.equal(a1, net.layers[0].admm0.alterD)
    finally:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = 128


def test_flat_bucket_native_pack_unpack(dev):
    """dp.FlatBucket on the GPU: one multi-tensor launch each way; dense tensors of any layout round-trip bit-exactly and
    the packed image is the storage order of each tensor (what the elementwise all-reduce needs)."""
    from alignq_amd.dp import FlatBucket
    torch.manual_seed(0)
    ts = [torch.randn(16, 8, 3, 3, device=dev).contiguous(memory_format=torch.channels_last), torch.randn(7, device=dev),
          torch.randn(128, 128, device=dev), torch.randn(64, 16, 1, 1, device=dev)] + [torch.randn(5, 3, device=dev)
                                                                                    for _ in range(60)]
    b = FlatBucket([t.shape for t in ts], dev)
    b.pack(ts)
    off = 0
    for t in ts[:3]:
        n = t.numel()
        raw = torch.as_strided(t, (n,), (1,))          # storage order
        assert torch.equal(b.flat[off:off + n], raw)
        off += n
    b.flat.mul_(2.0)
    want = [t.clone() * 2.0 for t in ts]
    b.unpack(ts)
    for t, w in zip(ts, want):
        assert torch.equal(t, w)


@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 4), (128, 64, 8, 8), (3, 16, 32, 2), (5, 64, 8, 1),
                                     (2, 32, 16, 8)])
def test_qconv3x3_matches_fp64_convolution(dev, B, C, H, k):
    """alignq_conv3x3_nhwc (Conv2d_Q's F.conv2d for the ResNet body) forward and data gradient against an fp64 convolution of
    the same fp32 inputs: exact products, so the error is fp32 accumulation error only — required here to stay below the
    error of MIOpen's own fp32 convolution on the same data (+ a small floor)."""
    from alignq_amd import ops
    torch.manual_seed(C + B + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    x = (torch.randn(B, C, H, H, device=dev) * 1.3).contiguous(memory_format=cl).requires_grad_(True)
    wq = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    assert ops.qconv3x3_supported(x, wq, (1, 1), (1, 1), (1, 1), 1, None, k)
    y = ops.QConv3x3Fn.apply(x, wq, k)
    assert y.is_contiguous(memory_format=cl)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(True), wq.detach().double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, padding=1)
    yd.backward(gy.double())
    y32 = torch.nn.functional.conv2d(x.detach(), wq.detach(), padding=1)
    floor = 2e-6 * float(yd.abs().max())
    assert float((y.detach() - yd).abs().max()) <= max(float((y32 - yd).abs().max()), floor)
    dx32 = torch.nn.grad.conv2d_input(x.shape, wq.detach(), gy, padding=1)
    floor = 2e-6 * float(xd.grad.abs().max())
    assert float((x.grad - xd.grad).abs().max()) <= max(float((dx32 - xd.grad).abs().max()), floor)
    np.testing.assert_allclose(npy(wq.grad), wd.grad.float().cpu().numpy(), rtol=2e-4, atol=1e-3 * float(wd.grad.abs().max()))


@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 4), (128, 64, 8, 8), (4, 64, 8, 2)])
def test_qconv3x3_fused_backward_matches_separate_launches(dev, B, C, H, k):
    """alignq_conv3x3_nhwc_bwd (data gradient + filter-gradient slabs in one launch, reduction deferred to
    fused.DeferredWgrads.flush) must reproduce the separate launches bit for bit (same device code per role)."""
    from alignq_amd import ops
    from alignq_amd.fused import DeferredWgrads
    torch.manual_seed(C + B + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    x0 = (torch.randn(B, C, H, H, device=dev) * 1.3).contiguous(memory_format=cl)
    w0 = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl)
    gy = torch.randn_like(x0)
    res = []
    for fused_mode in (False, True):
        x, w = x0.clone(memory_format=cl).requires_grad_(True), w0.clone(memory_format=cl).requires_grad_(True)
        if fused_mode:
            # the consumer of a deferred filter gradient must flush before reading it (in a model that is the weight
            # quantiser's backward); here a pass-through node plays that role
            class Consumer(torch.autograd.Function):
                @staticmethod
                def forward(ctx, t):
                    return t.view_as(t)

                @staticmethod
                def backward(ctx, g):
                    from alignq_amd.fused import active_wgrads
                    assert len(active_wgrads().items) == 1
                    active_wgrads().flush()
                    return g
            with DeferredWgrads():
                ops.QConv3x3Fn.apply(x, Consumer.apply(w), k).backward(gy)
        else:
            ops.QConv3x3Fn.apply(x, w, k).backward(gy)
        res.append((npy(x.grad), npy(w.grad)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_trainstep_gradients_with_own_convolutions_match_miopen(dev):
    """Whole-step guard for the deferred filter-gradient reduction and the fused convolution backward: after one
    forward+backward of the same model on the same batch, every parameter gradient of TrainStep(channels_last, qconv) must
    point the same way as with MIOpen convolutions (cosine > 0.999; rounding-level differences and rare bin flips only)."""
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    grads = []
    torch.manual_seed(11)       # (the batch used to come from whatever state the previous tests left the generator in)
    x = torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)
    for qconv in (False, True):
        torch.manual_seed(3)
        net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
        step = TrainStep(net, channels_last=True, qconv=qconv)
        step._forward_backward(x, y, set_to_none=True)
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().float().flatten().clone() for n, p in net.named_parameters() if p.grad is not None})
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        # a handful of bins flip between the two convolution implementations (fp32 rounding of z); a 16..64-element
        # batch-norm vector averages over fewer of them than a filter does
        assert cos > (0.999 if a.numel() >= 256 else 0.997), (n, cos)
        assert abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0) < 0.02, n


def test_fast_path_trajectory_tracks_plain_path(dev):
    """End-to-end guard for everything TrainStep fuses (BN/ReLU/shortcut fold, channels-last, own convolutions forward and
    backward, deferred reductions, HIP graph): ten training iterations on a fixed batch against the plain per-module path
    (NCHW, MIOpen convolutions and batch-norm, eager launches — the path the per-op and two-step reference tests pin).  The
    two runs differ by rounding and rare tie-zone bin flips only, so the loss curves and the trained weights must stay close."""
    import copy
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    torch.manual_seed(11)
    base = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
    x = torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)
    runs = []
    for fast in (False, True):
        net = copy.deepcopy(base)
        if fast:      # two eager warm-up iterations (real steps: MIOpen plans its five convolutions), then graph replays
            step = TrainStep(net, lr=0.02, channels_last=True, qconv=True, fuse_bn=True).capture(x, y, warmup=2)
        else:
            step = TrainStep(net, lr=0.02, channels_last=False, fuse_bn=False, defer_losses=False)
            for _ in range(2):
                step(x, y)
        ces, tls = [], []
        for _ in range(8):
            _, ce, tl = step(x, y)
            ces.append(float(ce.detach())); tls.append(float(tl.detach()))
        torch.cuda.synchronize()
        runs.append((ces, tls, {n: p.detach().float().flatten().clone() for n, p in net.named_parameters()}))
    (ce_a, tl_a, w_a), (ce_b, tl_b, w_b) = runs
    assert ce_a[-1] < ce_a[0] and ce_b[-1] < ce_b[0]                       # both actually train
    np.testing.assert_allclose(ce_b, ce_a, rtol=0.05, atol=0.02)
    np.testing.assert_allclose(tl_b, tl_a, rtol=0.02)
    for n in w_a:
        if "alterD" in n or "gamma" in n or w_a[n].numel() < 64:
            continue
        cos = float(torch.dot(w_a[n], w_b[n]) / (w_a[n].norm() * w_b[n].norm() + 1e-30))
        assert cos > 0.995, (n, cos)


@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 8), (128, 64, 8, 4), (72, 16, 32, 8)])
def test_conv_epilogue_bn_statistics_feed_the_fold(dev, B, C, H, k):
    """The convolution's epilogue leaves per-workgroup per-channel {sum y, sum y^2}; (1) they add up to the statistics of y,
    (2) fused.bn_site consuming them (no statistics pass over y) gives the same site results as with its own statistics
    kernel: identical up to the float-vs-double rounding of the partial sums (tie-zone bin flips only)."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import config, ops
    from alignq_amd.fused import bn_site
    config.args.bitW = config.args.abitW = k
    torch.manual_seed(B + C)
    n = 2 ** k - 1
    cl = torch.channels_last
    x = (torch.randn(B, C, H, H, device=dev) * 1.1).contiguous(memory_format=cl)
    wq = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl)
    y = ops.QConv3x3Fn.apply_with_stats(x, wq, k)
    part, n_parts = y._alignq_bn_part[:2]
    assert part.shape == (C, n_parts, 2)
    yd = y.double()
    np.testing.assert_allclose(npy(part[:, :, 0].double().sum(1)), npy(yd.sum((0, 2, 3))), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(npy(part[:, :, 1].double().sum(1)), npy((yd * yd).sum((0, 2, 3))), rtol=1e-5)
    gq = torch.randn_like(y) * 0.01
    outs = []
    for with_part in (False, True):
        torch.manual_seed(1)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        admm = A.ADMM(128).to(dev)
        act = A.activation_quantize_fn(k, "second", admm)
        z = y.detach().clone(memory_format=cl).requires_grad_(True)
        if with_part:
            z._alignq_bn_part = (part, n_parts)
        xq, loss = bn_site(bn, act, z, relu=True)
        (loss + (xq * gq).sum()).backward()
        outs.append(dict(xq=npy(xq), D=npy(admm.D), loss=float(loss.detach()), dz=npy(z.grad), rm=npy(bn.running_mean),
                         rv=npy(bn.running_var)))
    a, b = outs
    flips = np.abs(a["xq"] - b["xq"]) * n
    assert flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 1e-3
    np.testing.assert_allclose(b["D"], a["D"], atol=TOL)
    np.testing.assert_allclose(b["loss"], a["loss"], atol=TOL)
    np.testing.assert_allclose(b["rm"], a["rm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b["rv"], a["rv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b["dz"], a["dz"], atol=2e-5, rtol=1e-3)
    config.args.bitW = config.args.abitW = 8


@pytest.mark.parametrize("B,C,H,k", [(128, 16, 32, 8), (128, 32, 16, 8), (128, 64, 8, 4), (80, 32, 16, 8)])
def test_lazy_bn_backward_inside_the_fused_convolution_backward(dev, B, C, H, k):
    """conv (alignq_conv3x3_nhwc) -> folded BN + site.  Inside a fused.DeferredWgrads context the site backward hands its
    per-tile sums to alignq_conv3x3_nhwc_bwd, which reduces them itself, forms dz on load and writes the batch-norm parameter
    gradients; outside it the batch-norm input gradient is materialised (alignq_bn_bwd_apply) and the convolution gradients
    run as separate launches.  Same numbers up to summation order."""
    import alignq_amd.cdf_alignment_admm as A
    from alignq_amd import config, ops
    from alignq_amd.fused import DeferredWgrads, bn_site
    config.args.bitW = config.args.abitW = k
    torch.manual_seed(B + C + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    x0 = (torch.randn(B, C, H, H, device=dev) * 1.1).contiguous(memory_format=cl)
    w0 = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl)
    gq = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=cl) * 0.01
    outs = []
    for fused_bwd in (False, True):
        torch.manual_seed(1)
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
        admm = A.ADMM(128).to(dev)
        act = A.activation_quantize_fn(k, "second", admm)
        x = x0.clone(memory_format=cl).requires_grad_(True)
        w = w0.clone(memory_format=cl).requires_grad_(True)

        class Consumer(torch.autograd.Function):      # stands in for the weight quantiser's backward: flushes before it reads dW
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, g):
                from alignq_amd.fused import active_wgrads
                if active_wgrads() is not None:
                    active_wgrads().flush()
                return g
        z = ops.QConv3x3Fn.apply_with_stats(x, Consumer.apply(w), k)
        assert z._alignq_bn_part[2] == 2
        xq, loss = bn_site(bn, act, z, relu=True)
        total = loss + (xq * gq).sum()
        if fused_bwd:
            with DeferredWgrads(fresh_grads=True) as wg:       # all .grad are None here
                total.backward()
                wg.flush()
        else:
            total.backward()
        torch.cuda.synchronize()
        outs.append(dict(dx=npy(x.grad), dw=npy(w.grad), dg=npy(bn.weight.grad), db=npy(bn.bias.grad)))
    a, b = outs
    for key in ("dg", "db"):
        np.testing.assert_allclose(b[key], a[key], rtol=2e-5, atol=1e-6 * float(np.abs(a[key]).max() + 1e-12), err_msg=key)
    np.testing.assert_allclose(b["dx"], a["dx"], rtol=1e-4, atol=1e-5 * float(np.abs(a["dx"]).max()))
    np.testing.assert_allclose(b["dw"], a["dw"], rtol=1e-4, atol=1e-5 * float(np.abs(a["dw"]).max()))
    config.args.bitW = config.args.abitW = 8


@pytest.mark.parametrize("B,CIN,COUT,H,ks,k", [(128, 16, 32, 32, 3, 8), (128, 16, 32, 32, 1, 8), (128, 32, 64, 16, 3, 4),
                                               (128, 32, 64, 16, 1, 8), (8, 16, 32, 32, 3, 2), (8, 32, 64, 16, 1, 8)])
def test_qconv_transition_forward_matches_fp64(dev, B, CIN, COUT, H, ks, k):
    """alignq_conv_gen_nhwc_fwd (stride-2 3x3 and 1x1 shortcut convolutions) against fp64, its batch-norm partials against the
    statistics of its output, and its data / filter gradients (alignq_conv_gen_nhwc_dgrad / _wgrad) against MIOpen's."""
    from alignq_amd import ops
    torch.manual_seed(CIN + ks + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    pad = 1 if ks == 3 else 0
    x = (torch.randn(B, CIN, H, H, device=dev) * 1.2).contiguous(memory_format=cl).requires_grad_(True)
    wq = (torch.round(torch.tanh(torch.randn(COUT, CIN, ks, ks)) * n) / n).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    assert ops.qconv_gen_supported(x, wq, (2, 2), (pad, pad), (1, 1), 1, None, k)
    y = ops.QConvGenFn.apply_with_stats(x, wq, k, pad)
    assert y.shape == (B, COUT, H // 2, H // 2) and y.is_contiguous(memory_format=cl)
    yd = torch.nn.functional.conv2d(x.detach().double(), wq.detach().double(), stride=2, padding=pad)
    y32 = torch.nn.functional.conv2d(x.detach(), wq.detach(), stride=2, padding=pad)
    floor = 2e-6 * float(yd.abs().max())
    assert float((y.detach() - yd).abs().max()) <= max(float((y32 - yd).abs().max()), floor)
    part, n_parts, lazy_ok = y._alignq_bn_part[:3]
    assert lazy_ok and part.shape == (COUT, n_parts, 2)
    np.testing.assert_allclose(npy(part[:, :, 0].double().sum(1)), npy(yd.sum((0, 2, 3))), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(npy(part[:, :, 1].double().sum(1)), npy((yd * yd).sum((0, 2, 3))), rtol=1e-5)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(True), wq.detach().clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, stride=2, padding=pad).backward(gy)
    np.testing.assert_allclose(npy(x.grad), npy(xr.grad), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(npy(wq.grad), npy(wr.grad), rtol=1e-3, atol=1e-3 * float(wr.grad.abs().max()))
    # tap: the input alias as a second output; its gradient is added in the data-gradient kernel's epilogue
    xt = x.detach().clone().requires_grad_(True)
    yt, xa = ops.QConvGenFn.apply_with_stats(xt, wq.detach(), k, pad, True)
    assert xa.data_ptr() == xt.data_ptr() and bits_equal(npy(yt), npy(y))
    gt = torch.randn_like(xt)
    torch.autograd.backward([yt, xa], [gy, gt])
    np.testing.assert_allclose(npy(xt.grad), npy(x.grad + gt), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,k", [(128, 32, 8), (6, 32, 4), (3, 8, 8)])
def test_qconv_stem_matches_fp64(dev, B, H, k):
    """alignq_conv_stem_nhwc_fwd / _wgrad (3 -> 16 channels) against fp64 / MIOpen, incl. the batch-norm partials."""
    from alignq_amd import ops
    torch.manual_seed(B + k)
    n = 2 ** k - 1
    cl = torch.channels_last
    x = torch.randn(B, 3, H, 32, device=dev).contiguous(memory_format=cl)
    wq = (torch.round(torch.tanh(torch.randn(16, 3, 3, 3)) * n) / n).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    assert ops.qconv_stem_supported(x, wq, (1, 1), (1, 1), (1, 1), 1, None, k)
    y = ops.QConvStemFn.apply_with_stats(x, wq, k)
    yd = torch.nn.functional.conv2d(x.double(), wq.detach().double(), padding=1)
    y32 = torch.nn.functional.conv2d(x, wq.detach(), padding=1)
    floor = 2e-6 * float(yd.abs().max())
    assert float((y.detach() - yd).abs().max()) <= max(float((y32 - yd).abs().max()), floor)
    part, n_parts, _ = y._alignq_bn_part[:3]
    np.testing.assert_allclose(npy(part[:, :, 0].double().sum(1)), npy(yd.sum((0, 2, 3))), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(npy(part[:, :, 1].double().sum(1)), npy((yd * yd).sum((0, 2, 3))), rtol=1e-5)
    gy = torch.randn_like(y)
    y.backward(gy)
    wd = wq.detach().double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), wd, padding=1).backward(gy.double())
    np.testing.assert_allclose(npy(wq.grad), wd.grad.float().cpu().numpy(), rtol=2e-4, atol=1e-4 * float(wd.grad.abs().max()))


def test_fused_head_matches_torch(dev):
    """fused.HeadCEFn (avgpool + linear + mean cross-entropy, one launch each way) against the PyTorch composition."""
    from alignq_amd.fused import HeadCEFn, head_ce_supported
    torch.manual_seed(0)
    B, C, H, K = 128, 64, 8, 10
    feat0 = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    lin = torch.nn.Linear(C, K).to(dev)
    y = torch.randint(0, K, (B,), device=dev)
    f1 = feat0.clone(memory_format=torch.channels_last).requires_grad_(True)
    assert head_ce_supported(f1, lin.weight, y)
    logits, ce = HeadCEFn.apply(f1, lin.weight, lin.bias, y)
    (ce * 1.7).backward()
    got = (npy(logits), float(ce.detach()), npy(f1.grad), npy(lin.weight.grad), npy(lin.bias.grad))
    lin.zero_grad()
    f2 = feat0.clone(memory_format=torch.channels_last).requires_grad_(True)
    lg = lin(torch.nn.functional.adaptive_avg_pool2d(f2, 1).view(B, -1))
    ce2 = torch.nn.functional.cross_entropy(lg, y)
    (ce2 * 1.7).backward()
    np.testing.assert_allclose(got[0], npy(lg), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(got[1], float(ce2.detach()), rtol=1e-6)
    np.testing.assert_allclose(got[2], npy(f2.grad), atol=1e-9, rtol=1e-4)
    np.testing.assert_allclose(got[3], npy(lin.weight.grad), atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(got[4], npy(lin.bias.grad), atol=1e-7, rtol=1e-4)


def test_eval_forward_fast_layout_matches_plain(dev):
    """The reference's test() pass (main.py: model.eval(), forward under no_grad, eval batch != train batch): batch-norm
    uses its running statistics, so nothing is folded, and the ADMM sites slice alterD/gamma to the evaluation batch.  The
    channels-last network on this repository's convolutions must give the logits of the plain NCHW / MIOpen network."""
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    torch.manual_seed(5)
    net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
    x = torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)
    step = TrainStep(net, channels_last=True, qconv=True)       # marks the convolutions for the own kernels
    for _ in range(3):
        step(x, y)                                              # running statistics away from their initial values
    net.eval()
    xe = torch.randn(100, 3, 32, 32, device=dev)                # evaluation batch of the reference's CIFAR runs
    with torch.no_grad():
        fast, tl_fast = net(xe.contiguous(memory_format=torch.channels_last))
        convs = [m for m in net.modules() if hasattr(m, "use_qconv")]
        assert convs and all(m.use_qconv for m in convs)
        for m in convs:
            m.use_qconv = False
        fuse = net.fuse_bn
        try:
            plain, tl_plain = net(xe)
        finally:
            for m in convs:
                m.use_qconv = True
            net.fuse_bn = fuse
    assert fast.shape == plain.shape == (100, 10)
    a, b = npy(fast).ravel(), npy(plain).ravel()
    assert np.isfinite(a).all() and np.isfinite(b).all()
    cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.9995, cos                                    # rounding + rare tie-zone bin flips only
    assert np.median(np.abs(a - b)) < 2e-3 * max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(float(tl_fast), float(tl_plain), rtol=2e-3)
    net.train()


def test_short_last_batch_runs_through_the_fast_path(dev):
    """CIFAR's 50000 images leave a last batch of 80 at batch 128: ADMM(dim=128) slices its state to [80,80]
    (utils/admm.py:21-27) and ADMM_OPT zero-pads D back to dim (utils/optimizer.py:95-103).  One eager iteration of the fast
    path (fold, channels-last, own convolutions) at B=80 against the plain path on the same weights."""
    import copy
    from alignq_amd import config
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    torch.manual_seed(9)
    base = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
    x = torch.randn(80, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (80,), device=dev)
    outs = []
    for fast in (False, True):
        net = copy.deepcopy(base)
        step = (TrainStep(net, channels_last=True, qconv=True) if fast else
                TrainStep(net, channels_last=False, fuse_bn=False, defer_losses=False))
        logits, ce, tl = step(x, y)
        torch.cuda.synchronize()
        assert logits.shape == (80, 10)
        outs.append((float(ce.detach()), float(tl.detach()), {n: p.detach().clone() for n, p in net.named_parameters()}))
    (ce_a, tl_a, w_a), (ce_b, tl_b, w_b) = outs
    np.testing.assert_allclose(ce_b, ce_a, rtol=2e-3)
    np.testing.assert_allclose(tl_b, tl_a, rtol=2e-3)
    for n in w_a:
        a, b = w_a[n].float().flatten(), w_b[n].float().flatten()
        assert torch.isfinite(b).all(), n
        if "alterD" in n or "gamma" in n:
            # rows/columns beyond the batch see a zero-padded D: identical update on both paths
            np.testing.assert_allclose(npy(b), npy(a), atol=2e-4, err_msg=n)
        elif a.numel() >= 64:
            # batch-norm biases start at 0, so after one step they ARE -lr * gradient: same bound as the gradient guard above
            cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            assert cos > 0.999, (n, cos)
