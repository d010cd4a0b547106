"""The eager-torch restatement (oracle/torch_ref.py) against tensors captured from the reference's own
Python (tests/golden).  Elementwise results must be bit-identical (same ATen op sequence); reductions
that go through BLAS/threaded sums are compared to 1e-6."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests.conftest import load_golden


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bits(a, b, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape, what
    assert np.array_equal(bits(a), bits(b)), f"{what}: {np.abs(a - b).max()}"


@pytest.mark.parametrize("k", [1, 2, 4, 8, 32])
def test_g1_uniform_quantize(k):
    g = load_golden("g1_uniform_quantize")
    x = T(g["x"]).requires_grad_(True)
    y = R.quantize_ste(x, k)
    y.backward(T(g[f"gy_k{k}"]))
    assert_bits(y, g[f"y_k{k}"])
    assert_bits(x.grad, g[f"gx_k{k}"])


@pytest.mark.parametrize("tree,fname", [("admm", "g2_weight_quant_admm"), ("cdf", "g2_weight_quant_cdfonly")])
def test_g2_weight_quant(tree, fname):
    g = load_golden(fname)
    cfg = R.Config(tree=tree)
    si = 0
    while f"W_s{si}" in g:
        for k in (2, 4, 8):
            W = T(g[f"W_s{si}"]).requires_grad_(True)
            Wq, c, pdf = R.weight_quant(W, k, cfg)
            Wq.backward(T(g[f"g_s{si}"]))
            assert_bits(c, g[f"cdf_s{si}"], "cdf")
            assert_bits(pdf, g[f"pdf_s{si}"], "pdf")
            assert_bits(Wq, g[f"Wq_s{si}_k{k}"], "Wq")
            np.testing.assert_allclose(W.grad.numpy(), g[f"dW_s{si}_k{k}"], rtol=0, atol=1e-6)
        si += 1
    assert si >= 2


@pytest.mark.parametrize("tree,fname", [("admm", "g3_act_quant_admm"), ("cdf", "g3_act_quant_cdfonly")])
def test_g3_act_quant(tree, fname):
    g = load_golden(fname)
    cfg = R.Config(tree=tree, act_range=float(g["act_range"]), method="plain")
    for k in (2, 4, 8):
        x = T(g["x"]).requires_grad_(True)
        xq, tl = R.act_quant(x, k, "second", cfg, None)
        assert tl == 0
        xq.backward(T(g["g"]))
        assert_bits(xq, g[f"xq_k{k}"])
        assert_bits(x.grad, g[f"dx_k{k}"])


def test_known_answer_bins():
    # SURVEY.md §8c known-answer table
    x = torch.tensor([-2, -1, -0.3, 0, 0.3, 1, 2.0])
    for tree, k, want in [("cdf", 2, [0, 0, 1, 2, 2, 3, 3]), ("admm", 2, [-6, -4, -1, 0, 1, 4, 6]),
                          ("cdf", 8, [6, 40, 97, 128, 158, 215, 249]),
                          ("admm", 8, [-487, -348, -120, 0, 120, 348, 487])]:
        cfg = R.Config(tree=tree)
        t, _ = R.cdf_transform(x, torch.zeros(1), torch.ones(1), "a", cfg)
        n = 2 ** k - 1
        assert torch.round(t * n).int().tolist() == want


@pytest.mark.parametrize("fname,eps", [("g4_corr_noeps", 0.0), ("g4_corr_eps", 1e-5)])
def test_g4_corr(fname, eps):
    g = load_golden(fname)
    ci = 0
    while f"x_c{ci}" in g:
        x = T(g[f"x_c{ci}"]).requires_grad_(True)
        G = R.corr(x, x, eps)
        G.backward(T(g[f"dG_c{ci}"]))
        np.testing.assert_allclose(G.detach().numpy(), g[f"G_c{ci}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"dx_c{ci}"], rtol=1e-5, atol=1e-6)
        ci += 1
    assert ci >= 2


@pytest.mark.parametrize("name", ["a", "b", "short"])
def test_g5_g6_site(name):
    g = load_golden("g5_g6_admm_site")
    k = int(g[f"k_{name}"])
    dim = g[f"alterD0_{name}"].shape[0]
    cfg = R.Config(tree="admm")
    admm = R.ADMM(dim)
    with torch.no_grad():
        admm.alterD.copy_(T(g[f"alterD0_{name}"]))
        admm.gamma.copy_(T(g[f"gamma0_{name}"]))
    x = T(g[f"x_{name}"]).requires_grad_(True)
    xq, tl = R.act_quant(x, k, "second", cfg, admm)
    (tl + (xq * T(g[f"g_{name}"])).sum()).backward()
    assert_bits(xq, g[f"xq_{name}"])
    np.testing.assert_allclose(admm.D.detach().numpy(), g[f"D_{name}"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(tl.item(), g[f"loss_{name}"], atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g[f"dx_{name}"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(admm.alterD.grad.numpy(), g[f"dalterD_{name}"], atol=1e-7, rtol=1e-5)
    np.testing.assert_allclose(admm.gamma.grad.numpy(), g[f"dgamma_{name}"], atol=1e-7, rtol=1e-5)
    opt = R.ADMM_OPT([admm.alterD, admm.gamma])
    opt.step([0], [1], [admm.D], [admm.alterD], [admm.gamma], [admm.mu], [admm.rho])
    np.testing.assert_allclose(admm.alterD.detach().numpy(), g[f"alterD1_{name}"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(admm.gamma.detach().numpy(), g[f"gamma1_{name}"], atol=1e-6, rtol=0)


def test_g6_small_norm_branch():
    g = load_golden("g5_g6_admm_site")
    admm = R.ADMM(4)
    with torch.no_grad():
        admm.alterD.copy_(T(g["alterD0_small"]))
        admm.gamma.copy_(T(g["gamma0_small"]))
    D = T(g["D_small"]).requires_grad_(True)
    loss = admm(D)
    loss.backward()
    assert_bits(loss.detach().reshape(()), g["loss_small"])
    assert_bits(D.grad, g["dD_small"])
    assert_bits(admm.alterD.grad, g["dalterD_small"])
    assert_bits(admm.gamma.grad, g["dgamma_small"])
    opt = R.ADMM_OPT([admm.alterD, admm.gamma])
    opt.step([0], [1], [admm.D], [admm.alterD], [admm.gamma], [admm.mu], [admm.rho])
    assert np.all(g["alterD1_small"] == 0)          # the fixture really exercises the branch
    assert_bits(admm.alterD.detach(), g["alterD1_small"])
    assert_bits(admm.gamma.detach(), g["gamma1_small"])


def test_g5_office_site():
    g = load_golden("g5_office_site")
    k = int(g["k"])
    cfg = R.Config(tree="office", act_range=float(g["act_range"]))
    x = T(g["x"]).requires_grad_(True)
    xq, tl = R.act_quant(x, k, "aligned", cfg, None)
    xq.backward(T(g["g"]))
    assert_bits(xq, g["xq_plain"])
    assert_bits(x.grad, g["dx_plain"])
    admm = R.ADMM(28)
    with torch.no_grad():
        admm.alterD.copy_(T(g["alterD0"]))
        admm.gamma.copy_(T(g["gamma0"]))
    x = T(g["x"]).requires_grad_(True)
    xq, tl = R.act_quant(x, k, "aligned", cfg, admm)
    (tl + (xq * T(g["g"])).sum()).backward()
    assert_bits(xq, g["xq"])
    np.testing.assert_allclose(admm.D.detach().numpy(), g["D"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(tl.item(), g["loss"], atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=1e-6, rtol=1e-5)


def test_g7_sgd_step():
    g = load_golden("g7_sgd_step")
    ps = [torch.nn.Parameter(T(g[f"p{i}_0"]).clone()) for i in range(3)]
    opt = R.SGD(ps, lr=0.04, momentum=0.9, weight_decay=1e-4, bitW=int(g["bitW"]))
    for step in (1, 2):
        for i, p in enumerate(ps):
            p.grad = T(g[f"grad{i}_{step}"]).clone()
        opt.step([1], [T(g["w_cdf"])], [T(g["w_pdf"])], float(g["lam"]), float(g["lam2"]))
        for i, p in enumerate(ps):
            assert_bits(p.detach(), g[f"p{i}_{step}"], f"p{i}")
            assert_bits(opt.state[p]["momentum_buffer"], g[f"buf{i}_{step}"], f"buf{i}")
            assert_bits(p.grad, g[f"gradout{i}_{step}"], f"grad{i}")


def ref_to_oracle_name(n):
    """reference module names (model/resnet.py:48-64,113) -> oracle/torch_ref.py names"""
    if ".opt." in n or n.startswith("act_q"):
        return None                                   # aliases of admmN.* in the reference state_dict
    n = n.replace("admm_skip.", "site_skip.admm.")
    n = n.replace("admm0.", "site0.admm.").replace("admm1.", "site1.admm.")
    return n


def test_g8_tiny_resnet_two_steps():
    g = load_golden("g8_tiny_resnet_admm")
    cfg = R.Config(tree="admm", bitW=4, abitW=4, train_batch_size=8)
    net = R.PreActResNet(cfg, [1, 1, 1], 4, 4)
    sd = {}
    for key, v in g.items():
        if key.startswith("init/"):
            n = ref_to_oracle_name(key[5:])
            if n is not None:
                sd[n] = T(v)
    net.load_state_dict(sd, strict=True)
    net.train()
    step = R.TrainStep(net, cfg)
    for it in range(2):
        logits, ce, tl = step(T(g["xs"][it]), T(g["ys"][it]))
        np.testing.assert_allclose(logits.detach().numpy(), g[f"logits_{it}"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(ce.item(), g[f"ce_{it}"], atol=1e-5)
        np.testing.assert_allclose(tl.item(), g[f"trans_{it}"], atol=1e-5)
        got = net.state_dict()
        for key, v in g.items():
            if key.startswith(f"after{it}/"):
                n = ref_to_oracle_name(key[len(f"after{it}/"):])
                if n is not None:
                    np.testing.assert_allclose(got[n].numpy(), v, atol=2e-5, rtol=1e-4, err_msg=n)


def test_office_tiny_dann_two_iterations_match_the_reference():
    """G10: the eager restatement of the Office harness (OfficeDANN + OfficeTrainStep: two passes per iteration, summed
    loss, three SGD groups incl. alterD/gamma, ADMM_OPT on the TARGET pass's D, per-epoch SGD re-creation) against the
    reference's own dann_office model driven through main.py:343-456's sequence — same ATen ops, so values match to
    rounding (1e-6) over both iterations."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from det_init import det_init_, sample
    from oracle import torch_ref as R
    g = load_golden("g10_office_tiny_dann")
    torch.set_num_threads(4)
    cfg = R.Config(tree="office", bitW=4, abitW=4, train_batch_size=6)
    torch.manual_seed(0)
    net = R.OfficeDANN(cfg, 4, 4, str(g["stage"]), (1, 1, 1, 1), width_per_group=8).train()
    assert [n for n, _ in net.named_parameters()] == list(g["names"])
    det_init_(net)
    step = R.OfficeTrainStep(net, cfg, lr=float(g["lr"]), alpha=float(g["alpha"]))
    named = list(net.named_parameters())
    for it, epoch in enumerate((1, 2)):
        rate = step.new_epoch(epoch, int(g["num_epochs"]), float(g["lr"]))
        assert abs(rate - float(g[f"rate_{it}"])) < 1e-12
        out = step(torch.from_numpy(g["xs"][it]), torch.from_numpy(g["ys"][it]), torch.from_numpy(g["xt"][it]))
        for key in ("cls_s", "dom_s", "dom_t", "tl_s", "tl_t", "loss"):
            np.testing.assert_allclose(out[key].detach().numpy(), g[f"{key}_{it}"], atol=2e-5, rtol=1e-5, err_msg=key)
        for bi, b in enumerate(net.feature.blocks()):
            np.testing.assert_allclose(b.admm0.D.detach().numpy(), g[f"D_{it}_{bi}"], atol=1e-6)      # the TARGET pass's D
            np.testing.assert_allclose(step.D_src[bi].numpy(), g[f"Dsrc_{it}_{bi}"], atol=1e-6)
            assert np.abs(g[f"D_{it}_{bi}"] - g[f"Dsrc_{it}_{bi}"]).max() > 1e-4                       # and they do differ
        for j, (n, p) in enumerate(named):
            np.testing.assert_allclose(sample(p).numpy(), g[f"after_{it}/{j}"], atol=2e-6, rtol=1e-5, err_msg=n)
            if f"buf_{it}/{j}" in g:
                np.testing.assert_allclose(sample(step.opt_t.state[p]["momentum_buffer"]).numpy(), g[f"buf_{it}/{j}"],
                                           atol=2e-6, rtol=1e-5, err_msg=n)


@pytest.mark.parametrize("tree", ["admm", "cdf"])
def test_g3l_bins_at_scale(tree):
    """The torch restatement (the cpu_baseline leg of bench.py) reproduces the reference's integer bins on all 2^20 G3L
    elements with NO flips - it runs the reference's own op sequence on the same torch build."""
    from tests.test_oracle_c import load_g3l
    x, g = load_g3l(tree)
    r = float(g["act_range"])
    cfg = R.Config(tree=tree, act_range=r, method="plain")
    for k in (2, 4, 8):
        n = 2 ** k - 1
        xq, _ = R.act_quant(T(x.reshape(128, -1)), k, "second", cfg, None)
        xq = xq.numpy().astype(np.float64).ravel()
        bins = np.rint(xq * n) if tree == "admm" else np.rint((xq / r + 1.0) * 0.5 * n)
        assert np.array_equal(bins.astype(np.int16), g[f"bins_k{k}"])


@pytest.mark.parametrize("tree,fname", [("admm", "g11b_cdf_live_stats_admm"), ("cdf", "g11b_cdf_live_stats_cdfonly")])
def test_g11b_cdf_with_live_statistics(tree, fname):
    """Round 5: the oracle's cdf_transform with m and s IN the autograd graph against the reference's
    cdf(m, s, src) module - both outputs' gradients w.r.t. tensor, m, s, and dW through mean / std (model/quantization.py:49-59,78)."""
    g = load_golden(fname)
    cfg = R.Config(tree=tree, act_range=float(g["act_range"]))
    v0, gc, gp = torch.tensor(g["v"]), torch.tensor(g["gc"]), torch.tensor(g["gp"])
    for src in ("w", "a"):
        v = v0.clone().requires_grad_(True)
        m = torch.tensor(float(g[f"m_{src}"]), requires_grad=True)
        s = torch.tensor(float(g[f"s_{src}"]), requires_grad=True)
        c, pdf = R.cdf_transform(v, m, s, src, cfg)
        torch.autograd.backward([c, pdf], [gc, gp])
        np.testing.assert_allclose(c.detach().numpy(), g[f"cdf_{src}"], atol=1e-6)
        np.testing.assert_allclose(pdf.detach().numpy(), g[f"pdf_{src}"], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(v.grad.numpy(), g[f"dv_{src}"], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(float(m.grad), float(g[f"dm_{src}"]), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(float(s.grad), float(g[f"ds_{src}"]), rtol=1e-5, atol=1e-5)
    w = v0.clone().requires_grad_(True)
    c, pdf = R.cdf_transform(w, torch.mean(w), torch.std(w), "w", cfg)
    c.backward(gc)
    np.testing.assert_allclose(c.detach().numpy(), g["cdf_ms"], atol=1e-6)
    np.testing.assert_allclose(w.grad.numpy(), g["dW_ms"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", ["a", "b"])
def test_g14_constant_column_nan_pattern(name):
    """corr without epsilon (cdf_alignment_admm/resnet-20-cifar-10/model/quantization.py:134-137) on a batch with
    constant columns - the restatement gives the reference's NaN pattern: corr all NaN, its dx NaN in those columns only, the ADMM
    site's D / loss / every gradient NaN, x_q bit-identical to the recorded level indices."""
    g = load_golden("g14_constant_column")
    x0 = T(g[f"x_{name}"].astype(np.float32))
    B, F = x0.shape
    k, head = int(g["k"]), int(g["head"])
    n = 2 ** k - 1
    unpack = lambda key, shape: np.unpackbits(g[key])[: int(np.prod(shape))].astype(bool).reshape(shape)     # noqa: E731
    x = x0.clone().requires_grad_(True)
    G = R.corr(x, x, 0.0)
    G.backward(T(g[f"dG_{name}"]))
    assert np.array_equal(np.isnan(G.detach().numpy()), unpack(f"G_isnan_{name}", (B, B)))
    dx = x.grad.numpy()
    assert np.array_equal(np.isnan(dx), unpack(f"corr_dx_isnan_{name}", (B, F)))
    ref = g[f"corr_dx_head_{name}"]
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(dx[:, :head][ok], ref[ok], rtol=1e-5, atol=1e-6)
    cfg = R.Config(tree="admm")
    admm = R.ADMM(B)
    with torch.no_grad():
        admm.alterD.copy_(T(g[f"alterD0_{name}"]))
        admm.gamma.copy_(T(g[f"gamma0_{name}"]))
    shape = tuple(int(v) for v in g[f"shape_{name}"])
    x = x0.clone().view(shape).requires_grad_(True)
    xq, tl = R.act_quant(x, k, "second", cfg, admm)
    (tl + (xq * 0.01).sum()).backward()
    # (by value: the fixture keeps integer level indices, which cannot tell -0 from +0)
    assert np.array_equal(xq.detach().view(B, F)[:, :head].numpy(), (g[f"bins_head_{name}"].astype(np.float64) / n).astype(np.float32))
    assert torch.isnan(tl) and np.isnan(g[f"loss_{name}"])
    for t, key, shp in ((admm.D, "D", (B, B)), (x.grad, "dx", (B, F)), (admm.alterD.grad, "dalterD", (B, B)), (admm.gamma.grad, "dgamma", (B, B))):
        assert np.array_equal(np.isnan(t.detach().numpy().reshape(shp)), unpack(f"{key}_isnan_{name}", shp)), key
