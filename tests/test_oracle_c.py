"""The plain-C oracle (oracle/alignq_oracle.c) against tensors captured from the reference (tests/golden).

Tolerances (BASELINE.json north_star): integer bins bit-exact, dequantised values / residuals within 1e-5.
Bins: exact outside the documented erf tie zone |frac(y) - 1/2| < TIE (y = pre-round value from the
reference), at most one bin off inside it."""
import numpy as np
import pytest

from tests import oracle_c as O
from tests.conftest import load_golden

TIE = 1e-4
TOL = 1e-5


def check_bins(q_ours, q_ref, y_ref, n, scale=1.0):
    """q = bin/n*scale (+offset) on both sides; y_ref = reference pre-round value in bin units."""
    frac = y_ref - np.floor(y_ref)
    tie = np.abs(frac - 0.5) < TIE
    diff = np.abs(q_ours - q_ref) * n / scale
    assert np.all(diff[~tie] == 0), f"bin mismatch outside tie zone: {np.count_nonzero(diff[~tie])}"
    assert np.all(diff[tie] <= 1.0 + 1e-3)
    return int(tie.sum()), int(np.count_nonzero(diff))


def test_nerf32_accuracy():
    """nerf32(y) ~ erf(y/sqrt 2): what decides a bin is the ABSOLUTE error in units of ulp(1) = 2^-24, because every consumer
    forms 1 + nerf32 first (tests/native/verify_nerf.c walks every fp32: 0.564); relative to the result's own ulp it
    stays below 2 (worst near |y| = 1/16, where a node boundary meets small results)."""
    import scipy.special as sp
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(1 << 20) * 1.5, rng.uniform(-6.2, 6.2, 1 << 20),
                        rng.uniform(-0.2, 0.2, 1 << 18)]).astype(np.float32)
    y = O.nerf32(x)
    ref = sp.erf(x.astype(np.float64) / np.sqrt(2.0))
    assert (np.abs(y - ref) / 2.0 ** -24).max() < 0.6
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert (np.abs(y - ref) / ulp).max() < 2.0
    assert np.array_equal(O.nerf32(np.array([0.0, np.inf, -np.inf, 5.7, -7.0, 5.625, 1e30], np.float32)),
                          np.array([0.0, 1.0, -1.0, 1.0, -1.0, 1.0, 1.0], np.float32))
    assert np.signbit(O.nerf32(np.array([-0.0], np.float32)))[0]
    assert np.isnan(O.nerf32(np.array([np.nan], np.float32)))[0]
    # odd, and the node boundaries (ties of the index rounding go to the even node) are seamless to 1 ulp(1)
    assert np.array_equal(O.nerf32(-x), -y)
    edges = (np.arange(1, 90, dtype=np.float64) / 16.0).astype(np.float32)
    for e in (edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(9))):
        assert (np.abs(O.nerf32(e) - sp.erf(e.astype(np.float64) / np.sqrt(2.0))) / 2.0 ** -24).max() < 0.6
    xe = rng.uniform(-30, 5, 1 << 20).astype(np.float32)
    re = np.exp(xe.astype(np.float64))
    assert (np.abs(O.exp32(xe) - re) / np.spacing(re.astype(np.float32))).max() < 1.2


@pytest.mark.parametrize("tree,fname,formula", [("admm", "g3_act_quant_admm", O.FORMULA_ADMM),
                                                ("cdf", "g3_act_quant_cdfonly", O.FORMULA_CDF)])
def test_act_quant_vs_reference(tree, fname, formula):
    g = load_golden(fname)
    r = float(g["act_range"])
    pre = g["t"] if tree == "admm" else g["c"]
    for k in (2, 4, 8):
        n = 2 ** k - 1
        xq, t, bins = O.act_quant_fwd(g["x"], k, r, formula)
        np.testing.assert_allclose(t, pre, atol=3e-7, rtol=0)
        scale = 1.0 if tree == "admm" else 2.0 * r
        check_bins(xq, g[f"xq_k{k}"], pre * n, n, scale)
        # bins returned really are the integers behind x_q
        if tree == "admm":
            assert np.array_equal(xq, (bins / np.float32(n)).astype(np.float32))
            assert bins.min() >= -r * n and bins.max() <= r * n
        else:
            assert bins.min() >= 0 and bins.max() <= n
        dx = O.act_quant_bwd(g["g"], g["x"], r)
        np.testing.assert_allclose(dx, g[f"dx_k{k}"], atol=1e-6, rtol=1e-5)


def test_known_answer_bins():
    x = np.array([-2, -1, -0.3, 0, 0.3, 1, 2.0], np.float32)
    for formula, k, want in [(O.FORMULA_CDF, 2, [0, 0, 1, 2, 2, 3, 3]), (O.FORMULA_ADMM, 2, [-6, -4, -1, 0, 1, 4, 6]),
                             (O.FORMULA_CDF, 8, [6, 40, 97, 128, 158, 215, 249]),
                             (O.FORMULA_ADMM, 8, [-487, -348, -120, 0, 120, 348, 487])]:
        assert O.act_quant_fwd(x, k, 2.0, formula)[2].tolist() == want


@pytest.mark.parametrize("k", [1, 2, 4, 8, 32])
def test_uniform_quantize_edge_widths(k):
    """k=1 (sign) and k=32 (identity) paths of uniform_quantize, via the weight path with formula ADMM."""
    g = load_golden("g1_uniform_quantize")
    # feed values through the raw rounding: use act path inverse is not available, so check the rounding rule
    # on exact bin inputs: y = round(x*n)/n for x in golden (pure rounding, no erf involved)
    x = g["x"]
    if k == 32:
        want = x
    elif k == 1:
        want = np.sign(x)
    else:
        n = np.float32(2 ** k - 1)
        want = np.rint(x * n) / n
    assert np.array_equal(want.astype(np.float32), g[f"y_k{k}"])


@pytest.mark.parametrize("tree,fname,formula", [("admm", "g2_weight_quant_admm", O.FORMULA_ADMM),
                                                ("cdf", "g2_weight_quant_cdfonly", O.FORMULA_CDF)])
def test_weight_quant_vs_reference(tree, fname, formula):
    g = load_golden(fname)
    si = 0
    while f"W_s{si}" in g:
        W = g[f"W_s{si}"]
        ms = O.weight_stats(W)
        np.testing.assert_allclose(ms[0], g[f"m_s{si}"], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(ms[1], g[f"s_s{si}"], rtol=2e-6)
        ms_ref = np.array([g[f"m_s{si}"], g[f"s_s{si}"]], np.float32)
        for k in (2, 4, 8):
            n = 2 ** k - 1
            q, c, pdf, bins = O.weight_quant_fwd(W, ms_ref, k, formula)
            np.testing.assert_allclose(c, g[f"cdf_s{si}"], atol=3e-7, rtol=0)
            np.testing.assert_allclose(pdf, g[f"pdf_s{si}"], rtol=3e-6, atol=1e-6)
            scale = 1.0 if tree == "admm" else 2.0
            check_bins(q, g[f"Wq_s{si}_k{k}"], g[f"cdf_s{si}"] * n, n, scale)
            dW = O.weight_quant_bwd(g[f"g_s{si}"], W, ms_ref)
            np.testing.assert_allclose(dW, g[f"dW_s{si}_k{k}"], atol=2e-5, rtol=1e-4)
        si += 1


@pytest.mark.parametrize("fname,eps", [("g4_corr_noeps", 0.0), ("g4_corr_eps", 1e-5)])
def test_corr_vs_reference(fname, eps):
    g = load_golden(fname)
    ci = 0
    while f"x_c{ci}" in g:
        G = O.corr_fwd(g[f"x_c{ci}"], eps)
        np.testing.assert_allclose(G, g[f"G_c{ci}"], atol=TOL, rtol=0)
        dx = O.corr_bwd(g[f"dG_c{ci}"], g[f"x_c{ci}"], eps)
        ref = g[f"dx_c{ci}"]
        np.testing.assert_allclose(dx, ref, atol=TOL * max(1.0, np.abs(ref).max()), rtol=1e-4)
        ci += 1


@pytest.mark.parametrize("name", ["a", "b", "short"])
def test_site_vs_reference(name):
    g = load_golden("g5_g6_admm_site")
    k, x = int(g[f"k_{name}"]), g[f"x_{name}"]
    mu, rho = float(g["mu"]), float(g["rho"])
    xq, D = O.site_fwd(x, k, 2.0)
    n = 2 ** k - 1
    _, t, _ = O.act_quant_fwd(x, k, 2.0, O.FORMULA_ADMM)
    check_bins(xq, g[f"xq_{name}"], t.astype(np.float64) * n, n)
    np.testing.assert_allclose(D, g[f"D_{name}"], atol=TOL, rtol=0)
    loss, dD, dA, dg = O.admm_loss(g[f"D_{name}"], g[f"alterD0_{name}"], g[f"gamma0_{name}"], mu, rho)
    np.testing.assert_allclose(loss, g[f"loss_{name}"], atol=TOL)
    np.testing.assert_allclose(dA, g[f"dalterD_{name}"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(dg, g[f"dgamma_{name}"], atol=1e-7, rtol=1e-4)
    dx = O.site_bwd(g[f"g_{name}"], dD, x, 2.0)
    np.testing.assert_allclose(dx, g[f"dx_{name}"], atol=TOL, rtol=1e-4)
    A1, G1 = O.admm_update(g[f"D_{name}"], g[f"alterD0_{name}"], g[f"gamma0_{name}"], mu, rho)
    np.testing.assert_allclose(A1, g[f"alterD1_{name}"], atol=TOL)
    np.testing.assert_allclose(G1, g[f"gamma1_{name}"], atol=TOL)


def test_admm_small_norm_branch():
    g = load_golden("g5_g6_admm_site")
    mu, rho = float(g["mu"]), float(g["rho"])
    loss, dD, dA, dg = O.admm_loss(g["D_small"], g["alterD0_small"], g["gamma0_small"], mu, rho)
    np.testing.assert_allclose(loss, g["loss_small"], atol=1e-6)
    np.testing.assert_allclose(dD, g["dD_small"], atol=1e-6)
    np.testing.assert_allclose(dA, g["dalterD_small"], atol=1e-6)
    np.testing.assert_allclose(dg, g["dgamma_small"], atol=1e-6)
    A1, G1 = O.admm_update(g["D_small"], g["alterD0_small"], g["gamma0_small"], mu, rho)
    assert np.all(A1 == 0)
    np.testing.assert_allclose(G1, g["gamma1_small"], atol=1e-6)


def test_office_site_vs_reference():
    g = load_golden("g5_office_site")
    k, x, r = int(g["k"]), g["x"], float(g["act_range"])
    n = 2 ** k - 1
    xq, D = O.site_fwd(x, k, r, 1e-5)
    _, t, _ = O.act_quant_fwd(x, k, r, O.FORMULA_ADMM)
    check_bins(xq, g["xq"], t.astype(np.float64) * n, n)
    check_bins(xq, g["xq_plain"], t.astype(np.float64) * n, n)
    np.testing.assert_allclose(D, g["D"], atol=TOL, rtol=0)
    loss, dD, dA, dg = O.admm_loss(g["D"], g["alterD0"], g["gamma0"], 0.2, 0.3)
    np.testing.assert_allclose(loss, g["loss"], atol=TOL)
    dx = O.site_bwd(g["g"], dD, x, r, 1e-5)
    np.testing.assert_allclose(dx, g["dx"], atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(O.act_quant_bwd(g["g"], x, r), g["dx_plain"], atol=1e-6, rtol=1e-5)


def test_sgd_step_vs_reference():
    g = load_golden("g7_sgd_step")
    bitW, lam, lam2 = int(g["bitW"]), float(g["lam"]), float(g["lam2"])
    for i in range(3):
        p, buf = g[f"p{i}_0"], None
        for step in (1, 2):
            p, d, buf = O.sgd_step(p, g[f"grad{i}_{step}"], buf, 0.04, 0.9, 0.0, 1e-4, 0, step == 1)
            np.testing.assert_allclose(p, g[f"p{i}_{step}"], atol=1e-6)
            np.testing.assert_allclose(buf, g[f"buf{i}_{step}"], atol=1e-6, rtol=1e-6)
            if i == 1:
                d = O.sgd_grad_approx(d, g["w_cdf"], g["w_pdf"], bitW, lam, lam2)
            np.testing.assert_allclose(d, g[f"gradout{i}_{step}"], atol=1e-5, rtol=1e-5)


def test_fma_division_is_ieee_division_exhaustive(tmp_path):
    """Proof obligation of alignq_math.h::div_levels (the HIP kernels divide a level index by the level count with an fma
    sequence and OR the index's sign bit back): equality with IEEE division over every index, tests/native/verify_div.c."""
    import os
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "native", "verify_div.c")
    exe = str(tmp_path / "verify_div")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", src, "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "levels: mismatches 0" in out.stdout


def test_nerf32_exhaustive(tmp_path):
    """ALIGNQ-NERF32 against erf(y/sqrt 2) in double over EVERY non-negative fp32 (tests/native/verify_nerf.c, ~20 s on 8
    cores): |error| < 0.6 * 2^-24, the special values, and the program's own bound."""
    import os
    import re
    import subprocess
    here = os.path.dirname(__file__)
    exe = str(tmp_path / "verify_nerf")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fopenmp", os.path.join(here, "native", "verify_nerf.c"),
                    os.path.join(here, "..", "oracle", "alignq_oracle.c"), "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    m = re.search(r"max \|err\| = ([0-9.]+) \* 2\^-24", out.stdout)
    assert m and float(m.group(1)) < 0.6, out.stdout
    assert "nerf32(0)=0 nerf32(-0)=-0 nerf32(inf)=1 nerf32(-7)=-1 nerf32(nan)=nan" in out.stdout


@pytest.mark.parametrize("nhwc", [0, 1])
@pytest.mark.parametrize("relu,residual", [(False, False), (True, True)])
def test_bn_folded_site_oracle_vs_torch_batchnorm(nhwc, relu, residual):
    """oq_bn_fold_ab / oq_bn_site_fwd / oq_bn_site_bwd (the BN-folded ADMM site the GPU bench path runs) against the
    composition the reference's block executes (cdf_alignment_admm/resnet-56-cifar-10/model/resnet.py:87-96): torch's own
    training-mode nn.BatchNorm2d followed by the golden-pinned eager restatement of activation_quantize_fn (+ shortcut add
    + relu), forward and autograd backward."""
    import torch
    from oracle import torch_ref as R
    torch.manual_seed(7 + nhwc)
    B, C, H, W, k, r = 12, 8, 6, 4, 4, 2.0
    cfg = R.Config(tree="admm", abitW=k, train_batch_size=B)
    z = (torch.randn(B, C, H, W) * 1.6 + 0.2)
    res = torch.randn(B, C, H, W) * 0.7 if residual else None
    gq = torch.randn(B, C, H, W) * 0.01
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    admm = R.ADMM(B)
    zz = z.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    xq, loss = R.act_quant(bn(zz), k, "second", cfg, admm)
    y = xq + rr if residual else xq
    y = torch.relu(y) if relu else y
    (loss + (y * gq).sum()).backward()

    def mem(t):     # the [B,F] matrix in the memory order under test
        t = t.detach()
        return (t.permute(0, 2, 3, 1) if nhwc else t).reshape(B, -1).numpy()

    def unmem(a):
        return (a.reshape(B, H, W, C).transpose(0, 3, 1, 2) if nhwc else a.reshape(B, C, H, W))

    zm = mem(z)
    ab, save, vu = O.bn_fold_ab(zm, C, nhwc, bn.weight.detach().numpy(), bn.bias.detach().numpy(), bn.eps)
    np.testing.assert_allclose(save[0], z.mean((0, 2, 3)).numpy(), atol=1e-6)
    np.testing.assert_allclose(vu, z.transpose(0, 1).reshape(C, -1).var(1, unbiased=True).numpy(), rtol=1e-5)
    np.testing.assert_allclose(0.1 * vu + 0.9, bn.running_var.numpy(), rtol=1e-5)
    yo, Do, xo = O.bn_site_fwd(zm, C, nhwc, ab, k, r, 0.0, None if res is None else mem(res), relu)
    np.testing.assert_allclose(unmem(xo), bn(z).detach().numpy(), atol=2e-6)          # fma vs torch's mul+add
    flips = np.abs(unmem(yo) - y.detach().numpy()) * (2 ** k - 1)
    assert flips.max() <= 1.0 + 1e-3 and (flips > 0.5).mean() < 2e-3                   # tie-zone flips only
    np.testing.assert_allclose(Do, admm.D.detach().numpy(), atol=TOL)
    _, dD, _, _ = O.admm_loss(Do, admm.alterD.detach().numpy(), admm.gamma.detach().numpy(), 0.2, 0.3)
    dz, dg, db, dres, _ = O.bn_site_bwd(mem(gq), dD, zm, C, nhwc, ab, save, yo if relu else None, r, 0.0)
    np.testing.assert_allclose(unmem(dz), zz.grad.numpy(), atol=TOL, rtol=1e-3)
    np.testing.assert_allclose(dg, bn.weight.grad.numpy(), atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(db, bn.bias.grad.numpy(), atol=1e-4, rtol=1e-3)
    if residual:
        assert (unmem(dres) != rr.grad.numpy()).mean() < 2e-3


@pytest.mark.parametrize("fname,eps,cases", [("g4b_corr_xy_noeps", 0.0, 3), ("g4b_corr_xy_eps", 1e-5, 2)])
def test_general_corr_xy_vs_reference(fname, eps, cases):
    """oq_corr_xy_fwd / _bwd against the reference's corr(x, y) called with two DIFFERENT matrices (fixture captured from
    model/quantization.py:134-137 and the Office tree's :158-161)."""
    g = load_golden(fname)
    for ci in range(cases):
        x, y, dG = g[f"x_c{ci}"], g[f"y_c{ci}"], g[f"dG_c{ci}"]
        G = O.corr_xy_fwd(x, y, eps)
        np.testing.assert_allclose(G, g[f"G_c{ci}"], atol=TOL, rtol=0)
        assert np.abs(G - G.T).max() > 1e-3                    # really the non-symmetric product
        dx, dy = O.corr_xy_bwd(dG, x, y, eps)
        np.testing.assert_allclose(dx, g[f"dx_c{ci}"], atol=TOL, rtol=1e-4)
        np.testing.assert_allclose(dy, g[f"dy_c{ci}"], atol=TOL, rtol=1e-4)
        # corr(x, x) through the general path equals the SYRK oracle
        np.testing.assert_allclose(O.corr_xy_fwd(x, x, eps), O.corr_fwd(x, eps), atol=1e-6)


def test_teacher_forced_sites_of_the_tiny_resnet():
    """G8b: three activation sites of the reference's tiny PreActResNet captured IN MODEL CONTEXT by forward hooks (input =
    the BN output, x_q, trans loss, D, ADMM state): the C oracle on exactly those inputs (teacher forcing) — bins exact outside
    the tie zone, D and loss within 1e-5."""
    g = load_golden("g8b_tiny_resnet_sites")
    k, r = int(g["k"]), float(g["act_range"])
    n = 2 ** k - 1
    for name in ("stem", "b0q1", "b2q0"):
        x = g[f"{name}/x"]
        xq, D = O.site_fwd(x, k, r, 0.0)
        _, t, _ = O.act_quant_fwd(x, k, r, O.FORMULA_ADMM)
        check_bins(xq, g[f"{name}/xq"], t.astype(np.float64) * n, n)
        np.testing.assert_allclose(D, g[f"{name}/D"], atol=TOL, rtol=0)
        loss, _, _, _ = O.admm_loss(D, g[f"{name}/alterD"], g[f"{name}/gamma"], 0.2, 0.3)
        np.testing.assert_allclose(loss, g[f"{name}/loss"], atol=TOL)


def test_office_bottleneck_sites_fixture_vs_oracle():
    """G13 (captured from the reference's own Bottleneck inside the tiny DANN, dann_office/model/resnet.py:131-156): the C
    oracle on the recorded site inputs — plain quantisers act_q1 / act_q2 (bins exact outside the tie zone), the ADMM site
    act_q3 with the Office tree's eps corr (x_q, D, trans loss), and the batch-norm fold's (a, b) / affine against the
    reference's own BatchNorm2d output."""
    g = load_golden("g13_office_bottleneck_sites")
    k, r = int(g["k"]), float(g["act_range"])
    n = 2 ** k - 1
    for name in ("q1", "q2", "q3"):
        x = g[f"{name}/x"]
        xq, t, _ = O.act_quant_fwd(x.reshape(-1), k, r, O.FORMULA_ADMM)
        frac = t.astype(np.float64) * n
        tie = np.abs(frac - np.floor(frac) - 0.5) < 1e-4
        diff = np.abs(xq - g[f"{name}/xq"].reshape(-1)) * n
        assert np.all(diff[~tie] == 0) and np.all(diff[tie] <= 1.0 + 1e-3), name
    B = g["q3/x"].shape[0]
    _, D = O.site_fwd(g["q3/x"].reshape(B, -1), k, r, 1e-5)
    np.testing.assert_allclose(D, g["q3/D"], atol=1e-5)
    loss, _, _, _ = O.admm_loss(D, g["q3/alterD"], g["q3/gamma"], 0.2, 0.3)
    np.testing.assert_allclose(loss, float(g["q3/loss"]), atol=1e-5)
    for bn in ("bn1", "bn3", "bnd"):
        z = g[f"{bn}/z"]
        C = z.shape[1]
        zm = np.ascontiguousarray(z.transpose(0, 2, 3, 1)).reshape(z.shape[0], -1)          # channels-last memory order
        ab, save, vu = O.bn_fold_ab(zm, C, 1, g[f"{bn}/weight"], g[f"{bn}/bias"], float(g[f"{bn}/eps"]))
        x = O.bn_apply(zm, C, 1, ab).reshape(z.shape[0], z.shape[2], z.shape[3], C).transpose(0, 3, 1, 2)
        np.testing.assert_allclose(x, g[f"{bn}/out"], atol=3e-5, rtol=1e-5)
        m = float(g[f"{bn}/momentum"])
        np.testing.assert_allclose(m * save[0], g[f"{bn}/running_mean"], atol=1e-6)           # running_mean started at 0
        np.testing.assert_allclose((1 - m) * 1.0 + m * vu, g[f"{bn}/running_var"], atol=1e-5, rtol=1e-5)


def g3l_check(bins_ours, g, k):
    """G3L rule (tests/golden/gen_goldens.py:_g3l): the reference's bins, exactly, outside its own tie zone; at most one bin
    off inside it.  Returns (elements in the tie zone, flips)."""
    diff = bins_ours.astype(np.int64).ravel() - g[f"bins_k{k}"].astype(np.int64)
    tie = np.zeros(diff.size, bool)
    tie[g[f"tie_idx_k{k}"]] = True
    bad = np.flatnonzero((diff != 0) & ~tie)
    assert bad.size == 0, f"k={k}: {bad.size} bins differ from the reference outside the tie zone, first at {bad[:5]}"
    assert np.abs(diff[tie]).max(initial=0) <= 1
    return int(tie.sum()), int(np.count_nonzero(diff))


def load_g3l(tree):
    import hashlib
    ga = load_golden("g3l_act_bins_admm")
    g = ga if tree == "admm" else load_golden("g3l_act_bins_cdfonly")
    x = ga["x"]
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["x_sha256"]) and x.size == 1 << 20
    return x, g


@pytest.mark.parametrize("tree,formula", [("admm", O.FORMULA_ADMM), ("cdf", O.FORMULA_CDF)])
def test_act_quant_bins_vs_reference_at_scale(tree, formula, record_property):
    """"bit-exact bins against the reference" pinned on 2^20 reference-captured elements per tree and bit
    width instead of G3's 8,192: INTEGER bins equal outside the reference's own erf tie zone, at most one off inside, and the
    number of flips is bounded (<= 16 per 2^20 at k = 8) and reported.  model/quantization.py:49-59,102-110 (ADMM tree),
    cdf_alignment/.../quantization.py:37-50,91-103 (CDF-only tree)."""
    x, g = load_g3l(tree)
    r = float(g["act_range"])
    for k in (2, 4, 8):
        _, _, bins = O.act_quant_fwd(x, k, r, formula)
        n_tie, flips = g3l_check(bins, g, k)
        record_property(f"g3l_{tree}_k{k}", {"tie_zone": n_tie, "flips": flips})
        print(f"G3L {tree} k={k}: {n_tie} of 2^20 elements in the tie zone, {flips} bins differ from the reference")
        assert flips <= 16
        assert 100 < n_tie < 400            # 2 * TIE of the unit interval per bin: ~210 expected


def _unpack_nan(g, key, shape):
    return np.unpackbits(g[key])[: int(np.prod(shape))].astype(bool).reshape(shape)


@pytest.mark.parametrize("name", ["a", "b"])
def test_constant_column_gives_the_reference_nan_pattern(name):
    """SURVEY H5 / F9 (fixture G14): the CIFAR trees' corr has no epsilon (cdf_alignment_admm/resnet-20-cifar-10/model/
    quantization.py:134-137).  With two columns constant over the batch the reference gives: corr all NaN; corr's dx for a finite
    dG NaN in exactly those columns and the usual values elsewhere; the ADMM site: D, loss, dx, dalterD, dgamma all NaN, x_q
    untouched.  The C oracle reproduces each of them (IEEE arithmetic, no clamping); with eps = 1e-5 everything is finite."""
    g = load_golden("g14_constant_column")
    x = g[f"x_{name}"].astype(np.float32)
    B, F = x.shape
    k, head = int(g["k"]), int(g["head"])
    n = 2 ** k - 1
    cols = [int(c) for c in g["const_cols"]]
    assert all(np.all(x[:, c] == v) for c, v in zip(cols, g["const_vals"]))
    # corr alone
    G = O.corr_fwd(x, 0.0)
    assert np.array_equal(np.isnan(G), _unpack_nan(g, f"G_isnan_{name}", (B, B))) and np.isnan(G).all()
    dx = O.corr_bwd(g[f"dG_{name}"], x, 0.0)
    want_nan = _unpack_nan(g, f"corr_dx_isnan_{name}", (B, F))
    assert np.array_equal(np.isnan(dx), want_nan)
    assert want_nan[:, cols].all() and want_nan.sum() == B * len(cols)
    ref = g[f"corr_dx_head_{name}"]
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(dx[:, :head][ok], ref[ok], atol=TOL * max(1.0, float(np.abs(ref[ok]).max())), rtol=1e-4)
    # the whole site
    xq, D = O.site_fwd(x, k, 2.0, 0.0)
    assert np.isnan(D).all() and _unpack_nan(g, f"D_isnan_{name}", (B, B)).all()
    _, t, _ = O.act_quant_fwd(x, k, 2.0, O.FORMULA_ADMM)
    bins_ref = g[f"bins_head_{name}"].astype(np.float32) / n
    check_bins(xq[:, :head], bins_ref, t[:, :head].astype(np.float64) * n, n)
    assert np.isfinite(xq).all()
    loss, dD, dA, dg = O.admm_loss(D, g[f"alterD0_{name}"], g[f"gamma0_{name}"], 0.2, 0.3)
    assert np.isnan(loss) and np.isnan(g[f"loss_{name}"])
    for got, key, shape in ((dD, None, None), (dA, "dalterD", (B, B)), (dg, "dgamma", (B, B))):
        assert np.isnan(got).all()
        if key:
            assert _unpack_nan(g, f"{key}_isnan_{name}", shape).all()
    gq = (np.random.default_rng(3).standard_normal((B, F)) * 0.01).astype(np.float32)
    sdx = O.site_bwd(gq, dD, x, 2.0, 0.0)
    assert np.isnan(sdx).all() and _unpack_nan(g, f"dx_isnan_{name}", (B, F)).all()
    # Office form (dann_office/model/quantization.py:158-161): the guarded std keeps everything finite
    xq_e, D_e = O.site_fwd(x, k, 2.0, 1e-5)
    assert np.isfinite(D_e).all() and np.array_equal(xq_e, xq)
    assert np.isfinite(O.corr_bwd(g[f"dG_{name}"], x, 1e-5)).all()
