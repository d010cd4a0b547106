#!/usr/bin/env python3
"""Golden-vector generator: imports the *reference's own Python* on CPU and records
input/output tensors of the hot-path ops as small .npz fixtures.

Runs only in the build container (needs /root/reference); the fixtures it writes
are committed next to it.  Nothing but tensors leaves the reference: no source text.

Import recipe: argparse runs at import in the reference
(utils/options.py:95), so sys.argv is set first; the module-global `device`
(model/quantization.py:16, model/resnet.py:25) is overwritten with cpu.  One process
per variant directory because the module names collide.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py            # all variants
    python tests/golden/gen_goldens.py --variant admm_cifar                 # one variant (child)

torch build used for capture is recorded in every file (`torch_version`).
"""
import argparse
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ALIGNQ_REFERENCE", "/root/reference")

VARIANTS = {
    # name -> (directory under the reference, options module)
    "admm_cifar": ("cdf_alignment_admm/resnet-56-cifar-10", "utils.options"),
    "cdf_only": ("cdf_alignment/resnet-20-cifar-10", "utils.options"),
    "office": ("cdf_alignment_admm/dann_office", "utils.options_office"),
}


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    import torch
    arrays["torch_version"] = np.array(torch.__version__)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def _enter(variant, extra_argv):
    """Perform the import recipe; returns (quantization module, args)."""
    import torch
    sys.dont_write_bytecode = True
    vdir, optmod = VARIANTS[variant]
    vdir = os.path.join(REF, vdir)
    sys.argv = ["main.py"] + extra_argv
    os.chdir(vdir)
    sys.path.insert(0, vdir)
    import importlib
    q = importlib.import_module("model.quantization")
    q.device = torch.device("cpu")
    args = importlib.import_module(optmod).args
    return q, args


# ----------------------------------------------------------------------------------------------
G3L_SEED, G3L_SHAPE, G3L_TIE = 20261004, (128, 8192), 1e-4


def _g3l(q, args, tree):
    """G3L: the reference's activation quantiser on 2^20 N(0,1) elements (post-batch-norm statistics), k in {2,4,8}: the
    INTEGER BINS behind its output as int16, plus - for the documented erf tie zone (|frac(pre-round value in bin units) - 1/2|
    < 1e-4) - the indices of the elements inside it and the reference's own fp32 pre-round value there.
    The same seeded x in both trees (stored once, in the ADMM-tree file; the CDF-only file carries its SHA-256).
    A generator of its own and called LAST in each variant, so the older fixtures' random streams are unchanged."""
    import hashlib
    import torch
    g = torch.Generator().manual_seed(G3L_SEED)
    x = torch.randn(G3L_SHAPE, generator=g)
    r = float(args.act_range)
    out = {"act_range": np.array(r, dtype=np.float32), "tie": np.array(G3L_TIE),
           "x_sha256": np.array(hashlib.sha256(_np(x).tobytes()).hexdigest())}
    pre, _ = q.cdf(torch.zeros(1), torch.ones(1), "a")(x)      # ADMM tree: t = r (2 Phi - 1); CDF-only tree: c = Phi
    pre = _np(pre).astype(np.float32).ravel()
    for k in (2, 4, 8):
        n = 2 ** k - 1
        if tree == "admm":
            xq, tl = q.activation_quantize_fn(k, "second", None)(x)
            assert tl == 0
            b = np.rint(_np(xq).astype(np.float64) * n)
        else:
            xq = q.activation_quantize_fn(k, "second")(x)
            b = np.rint((_np(xq).astype(np.float64) / r + 1.0) * 0.5 * n)
        assert np.abs(b).max() < 32768
        out[f"bins_k{k}"] = b.astype(np.int16).ravel()
        y = pre * np.float32(n)                                  # what torch.round saw (fp32, as the reference forms it)
        tie = np.flatnonzero(np.abs((y - np.floor(y)) - 0.5) < G3L_TIE)
        out[f"tie_idx_k{k}"] = tie.astype(np.int32)
        out[f"tie_pre_k{k}"] = pre[tie]
    if tree == "admm":
        out["x"] = _np(x).ravel()
    _save("g3l_act_bins_" + ("admm" if tree == "admm" else "cdfonly"), **out)


# ----------------------------------------------------------------------------------------------
def gen_admm_cifar():
    import torch
    q, args = _enter("admm_cifar", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "8"])
    from utils.admm import ADMM
    from utils.optimizer import SGD, ADMM_OPT
    import model.resnet as r
    r.device = torch.device("cpu")
    g = torch.Generator().manual_seed(1234)

    # G1 uniform_quantize (model/quantization.py:19-38)
    x = torch.randn(4096, generator=g)
    ties = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 0.0, 1.0 / 6.0, 0.5 / 3.0, 0.5 / 15, 1.5 / 15, 2.5 / 255, -3.5 / 255])
    x = torch.cat([x, ties])
    out = {"x": _np(x)}
    for k in (1, 2, 4, 8, 32):
        xi = x.clone().requires_grad_(True)
        y = q.uniform_quantize(k)(xi)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out[f"y_k{k}"] = _np(y)
        out[f"gy_k{k}"] = _np(gy)
        out[f"gx_k{k}"] = _np(xi.grad)
    _save("g1_uniform_quantize", **out)

    # G2 weight quant, ADMM tree (model/quantization.py:61-85)
    out = {}
    for si, shape in enumerate([(16, 3, 3, 3), (64, 64, 3, 3), (32, 16, 1, 1)]):
        W = torch.randn(shape, generator=g) * 0.05 + 0.01
        gq = torch.randn(shape, generator=g)
        out[f"W_s{si}"] = _np(W)
        out[f"g_s{si}"] = _np(gq)
        for k in (2, 4, 8):
            Wi = W.clone().requires_grad_(True)
            fn = q.weight_quantize_fn(k, "second")
            Wq = fn(Wi)
            Wq.backward(gq)
            out[f"m_s{si}"] = _np(torch.mean(W))
            out[f"s_s{si}"] = _np(torch.std(W))
            out[f"cdf_s{si}"] = _np(fn.weight_cdf)
            out[f"pdf_s{si}"] = _np(fn.weight_pdf)
            out[f"Wq_s{si}_k{k}"] = _np(Wq)
            out[f"dW_s{si}_k{k}"] = _np(Wi.grad)
    _save("g2_weight_quant_admm", **out)

    # G3 activation quant, ADMM-tree formula without the corr/ADMM branch
    # (model/quantization.py:102-132 with args.method != 'ours').
    out = {}
    args.method = "plain"
    x = torch.randn(8, 16, 8, 8, generator=g) * 1.3
    x.view(-1)[:7] = torch.tensor([-2.0, -1.0, -0.3, 0.0, 0.3, 1.0, 2.0])
    gq = torch.randn(x.shape, generator=g)
    out["x"] = _np(x)
    out["g"] = _np(gq)
    out["act_range"] = np.array(float(args.act_range), dtype=np.float32)
    for k in (2, 4, 8):
        xi = x.clone().requires_grad_(True)
        fn = q.activation_quantize_fn(k, "second", None)
        # pre-round transform (what uniform_q sees), for tie-zone masks
        t, _ = q.cdf(torch.zeros(1), torch.ones(1), "a")(x)
        xq, tl = fn(xi)
        assert tl == 0
        xq.backward(gq)
        out["t"] = _np(t)
        out[f"xq_k{k}"] = _np(xq)
        out[f"dx_k{k}"] = _np(xi.grad)
    args.method = "ours"
    _save("g3_act_quant_admm", **out)

    # G4 corr, no eps (model/quantization.py:134-137)
    out = {}
    for ci, (B, Fdim) in enumerate([(16, 256), (128, 1024), (8, 96)]):
        x = torch.randn(B, Fdim, generator=g) * 0.8 + 0.1
        dG = torch.randn(B, B, generator=g)
        xi = x.clone().requires_grad_(True)
        G = q.corr(xi, xi)
        G.backward(dG)
        out[f"x_c{ci}"] = _np(x)
        out[f"dG_c{ci}"] = _np(dG)
        out[f"G_c{ci}"] = _np(G)
        out[f"dx_c{ci}"] = _np(xi.grad)
    _save("g4_corr_noeps", **out)

    # G5 ADMM site end to end + G6 ADMM_OPT.step (utils/admm.py:24-33, utils/optimizer.py:60-135)
    out = {}
    cases = [  # (name, dim, batch, C, H, W, k)
        ("a", 16, 16, 4, 8, 8, 4),
        ("b", 128, 128, 16, 8, 8, 8),
        ("short", 16, 10, 4, 8, 8, 2),   # last, short batch: D is [10,10] inside dim 16
    ]
    for name, dim, B, C, H, W, k in cases:
        torch.manual_seed(77)
        admm = ADMM(dim)
        fn = q.activation_quantize_fn(k, "second", admm)
        x = torch.randn(B, C, H, W, generator=g)
        gq = torch.randn(x.shape, generator=g) * 0.01
        xi = x.clone().requires_grad_(True)
        alterD0, gamma0 = _np(admm.alterD), _np(admm.gamma)
        xq, tl = fn(xi)
        (tl + (xq * gq).sum()).backward()
        out[f"x_{name}"] = _np(x)
        out[f"g_{name}"] = _np(gq)
        out[f"k_{name}"] = np.array(k)
        out[f"alterD0_{name}"] = alterD0
        out[f"gamma0_{name}"] = gamma0
        out[f"xq_{name}"] = _np(xq)
        out[f"loss_{name}"] = _np(tl)
        out[f"D_{name}"] = _np(admm.D)
        out[f"dx_{name}"] = _np(xi.grad)
        out[f"dalterD_{name}"] = _np(admm.alterD.grad)
        out[f"dgamma_{name}"] = _np(admm.gamma.grad)
        opt = ADMM_OPT([admm.alterD, admm.gamma])
        opt.step([0], [1], [admm.D], [admm.alterD], [admm.gamma], [admm.mu], [admm.rho])
        out[f"alterD1_{name}"] = _np(admm.alterD)
        out[f"gamma1_{name}"] = _np(admm.gamma)
    # the ||V||_F <= mu/rho branch (optimizer.py:109-112)
    admm = ADMM(4)
    with torch.no_grad():
        admm.alterD.mul_(0.1)
        admm.gamma.mul_(0.01)
    D = (torch.randn(4, 4, generator=g) * 0.05)
    Di = D.clone().requires_grad_(True)
    out["alterD0_small"], out["gamma0_small"], out["D_small"] = _np(admm.alterD), _np(admm.gamma), _np(D)
    loss = admm(Di)
    loss.backward()
    out["loss_small"] = _np(loss)
    out["dD_small"] = _np(Di.grad)
    out["dalterD_small"] = _np(admm.alterD.grad)
    out["dgamma_small"] = _np(admm.gamma.grad)
    opt = ADMM_OPT([admm.alterD, admm.gamma])
    opt.step([0], [1], [admm.D], [admm.alterD], [admm.gamma], [admm.mu], [admm.rho])
    out["alterD1_small"], out["gamma1_small"] = _np(admm.alterD), _np(admm.gamma)
    out["mu"], out["rho"] = np.array(admm.mu, dtype=np.float32), np.array(admm.rho, dtype=np.float32)
    _save("g5_g6_admm_site", **out)

    # G7 SGD.step (utils/optimizer.py:196-262): 2 steps, 3 tensors, tensor 1 in idx
    out = {}
    args.bitW = 4
    ps = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.1) for s in [(8, 4, 3, 3), (16, 8, 3, 3), (10,)]]
    opt = SGD(ps, lr=0.04, momentum=0.9, weight_decay=1e-4)
    w_cdf = [torch.rand(16, 8, 3, 3, generator=g) * 2 - 1]
    w_pdf = [torch.rand(16, 8, 3, 3, generator=g) * 8]
    out["w_cdf"], out["w_pdf"] = _np(w_cdf[0]), _np(w_pdf[0])
    out["bitW"] = np.array(4)
    for i, p in enumerate(ps):
        out[f"p{i}_0"] = _np(p)
    for step in (1, 2):
        for i, p in enumerate(ps):
            p.grad = torch.randn(p.shape, generator=g)
            out[f"grad{i}_{step}"] = _np(p.grad)
        opt.step([1], w_cdf, w_pdf, float(args.lam), float(args.lam2))
        for i, p in enumerate(ps):
            out[f"p{i}_{step}"] = _np(p)
            out[f"buf{i}_{step}"] = _np(opt.state[p]["momentum_buffer"])
            out[f"gradout{i}_{step}"] = _np(p.grad)
    out["lam"], out["lam2"] = np.array(float(args.lam)), np.array(float(args.lam2))
    args.bitW = 8
    _save("g7_sgd_step", **out)

    # G8 tiny PreActResNet([1,1,1]) — two full training iterations in the reference's order
    # (cdf_alignment_admm/resnet-20-cifar-10/main.py:288-374: zero_grad x2 -> fwd -> CE+trans -> bwd ->
    #  SGD.step -> ADMM_OPT.step), k=4, B=8.
    out = {}
    args.bitW = 4
    args.abitW = 4
    args.train_batch_size = 8
    torch.manual_seed(5)
    net = r.PreActResNet(r.PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10)
    net.train()
    for n_, p_ in net.state_dict().items():
        out["init/" + n_] = _np(p_)
    named = list(net.named_parameters())
    param_t = [(n, p) for n, p in named if "alterD" not in n and "gamma" not in n]
    param_admm = [(n, p) for n, p in named if "alterD" in n or "gamma" in n]
    opt_t = SGD([p for _, p in param_t], lr=0.04, momentum=0.9, weight_decay=1e-4)
    opt_a = ADMM_OPT([p for _, p in param_admm])
    ce = torch.nn.CrossEntropyLoss()
    xs = torch.randn(2, 8, 3, 32, 32, generator=g)
    ys = torch.randint(0, 10, (2, 8), generator=g)
    out["xs"], out["ys"] = _np(xs), _np(ys)
    for it in range(2):
        opt_t.zero_grad()
        opt_a.zero_grad()
        logits, tl = net(xs[it])
        loss_ce = ce(logits, ys[it])
        (loss_ce + tl).backward()
        idx = [j for j, (n, _) in enumerate(param_t) if "conv" in n and "weight" in n][1:]
        w_cdf, w_pdf = [], []
        for layer in net.layers:
            for conv in (layer.conv0, layer.conv1, layer.skip_conv):
                if conv is not None:
                    w_cdf.append(conv.quantize_fn.weight_cdf)
                    w_pdf.append(conv.quantize_fn.weight_pdf)
        a_idx = [j for j, (n, _) in enumerate(param_admm) if "alterD" in n]
        g_idx = [j for j, (n, _) in enumerate(param_admm) if "gamma" in n]
        mods = [net.admm0]
        for layer in net.layers:
            mods += [layer.admm0, layer.admm1]
            if layer.skip_conv is not None:
                mods.append(layer.admm_skip)
        out[f"logits_{it}"] = _np(logits)
        out[f"ce_{it}"] = _np(loss_ce)
        out[f"trans_{it}"] = _np(tl)
        for si, m in enumerate(mods):
            out[f"D_{it}_{si}"] = _np(m.D)
        opt_t.step(idx, w_cdf, w_pdf, float(args.lam), float(args.lam2))
        opt_a.step(a_idx, g_idx, [m.D for m in mods], [m.alterD for m in mods], [m.gamma for m in mods],
                   [m.mu for m in mods], [m.rho for m in mods])
        for n_, p_ in net.state_dict().items():
            out[f"after{it}/" + n_] = _np(p_)
    _save("g8_tiny_resnet_admm", **out)


# ----------------------------------------------------------------------------------------------
def gen_cdf_only():
    import torch
    q, args = _enter("cdf_only", ["--bitW", "8", "--abitW", "8"])
    g = torch.Generator().manual_seed(4321)
    out = {}
    for si, shape in enumerate([(16, 3, 3, 3), (64, 64, 3, 3)]):
        W = torch.randn(shape, generator=g) * 0.05 - 0.02
        gq = torch.randn(shape, generator=g)
        out[f"W_s{si}"], out[f"g_s{si}"] = _np(W), _np(gq)
        for k in (2, 4, 8):
            Wi = W.clone().requires_grad_(True)
            Wq = q.weight_quantize_fn(k, "second")(Wi)
            Wq.backward(gq)
            c, pdf = q.cdf(torch.mean(W), torch.std(W), "w")(W)
            out[f"cdf_s{si}"], out[f"pdf_s{si}"] = _np(c), _np(pdf)
            out[f"m_s{si}"], out[f"s_s{si}"] = _np(torch.mean(W)), _np(torch.std(W))
            out[f"Wq_s{si}_k{k}"] = _np(Wq)
            out[f"dW_s{si}_k{k}"] = _np(Wi.grad)
    _save("g2_weight_quant_cdfonly", **out)

    out = {}
    x = torch.randn(8, 16, 8, 8, generator=g) * 1.3
    x.view(-1)[:7] = torch.tensor([-2.0, -1.0, -0.3, 0.0, 0.3, 1.0, 2.0])
    gq = torch.randn(x.shape, generator=g)
    out["x"], out["g"] = _np(x), _np(gq)
    out["act_range"] = np.array(float(args.act_range), dtype=np.float32)
    c, _ = q.cdf(torch.zeros(1), torch.ones(1), "a")(x)
    out["c"] = _np(c)
    for k in (2, 4, 8):
        xi = x.clone().requires_grad_(True)
        xq = q.activation_quantize_fn(k, "second")(xi)
        xq.backward(gq)
        out[f"xq_k{k}"] = _np(xq)
        out[f"dx_k{k}"] = _np(xi.grad)
    _save("g3_act_quant_cdfonly", **out)


# ----------------------------------------------------------------------------------------------
def gen_office():
    import torch
    q, args = _enter("office", ["--bitW", "8", "--abitW", "8"])
    from utils.admm import ADMM
    g = torch.Generator().manual_seed(999)
    out = {}
    for ci, (B, Fdim) in enumerate([(28, 1024), (16, 256)]):
        x = torch.randn(B, Fdim, generator=g) * 0.8 + 0.1
        if ci == 1:
            x[:, 5] = 0.25          # a constant feature: finite only thanks to the +1e-5
        dG = torch.randn(B, B, generator=g)
        xi = x.clone().requires_grad_(True)
        G = q.corr(xi, xi)
        G.backward(dG)
        out[f"x_c{ci}"], out[f"dG_c{ci}"], out[f"G_c{ci}"], out[f"dx_c{ci}"] = _np(x), _np(dG), _np(G), _np(xi.grad)
    _save("g4_corr_eps", **out)

    out = {}
    torch.manual_seed(3)
    admm = ADMM(28)
    k = 8
    x = torch.randn(28, 8, 4, 4, generator=g)
    gq = torch.randn(x.shape, generator=g) * 0.01
    # plain (no ADMM) Office activation quant: returns a tensor (dann_office/model/quantization.py:87-110)
    xi = x.clone().requires_grad_(True)
    xq = q.activation_quantize_fn(k, "aligned")(xi)
    xq.backward(gq)
    out["x"], out["g"], out["k"] = _np(x), _np(gq), np.array(k)
    out["act_range"] = np.array(float(args.act_range), dtype=np.float32)
    out["xq_plain"], out["dx_plain"] = _np(xq), _np(xi.grad)
    # ADMM site (activation_quantize_fn2, :112-156) with the eps corr
    xi = x.clone().requires_grad_(True)
    out["alterD0"], out["gamma0"] = _np(admm.alterD), _np(admm.gamma)
    xq, tl = q.activation_quantize_fn2(k, "aligned", admm)(xi)
    (tl + (xq * gq).sum()).backward()
    out["xq"], out["loss"], out["D"], out["dx"] = _np(xq), _np(tl), _np(admm.D), _np(xi.grad)
    out["dalterD"], out["dgamma"] = _np(admm.alterD.grad), _np(admm.gamma.grad)
    _save("g5_office_site", **out)


def gen_office_keys():
    """state_dict key names / shapes and the SGD state_dict layout of the reference's ResNet-50-DANN (config 5): what a
    checkpoint written by dann_office/main.py:165-183 contains (checkpoint-name compatibility)."""
    import importlib
    import torch
    q, args = _enter("office", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "28"])
    r = importlib.import_module("model.resnet")
    r.device = torch.device("cpu")
    r.load_state_dict_from_url = lambda *a, **k: {}          # DANN defaults to pretrained=True (no network here)
    torch.manual_seed(0)
    net = r.resnet50_dann(wbit=8, abit=8, stage=args.stage)
    sd = net.state_dict()
    keys = list(sd.keys())
    from utils.optimizer import SGD
    opt = SGD([{"params": net.feature.parameters()},
               {"params": net.class_classifier.parameters(), "lr": 0.01},
               {"params": net.domain_classifier.parameters(), "lr": 0.01}], lr=0.001, momentum=0.9, weight_decay=5e-4)
    osd = opt.state_dict()
    _save("g9_office_state_keys", keys=np.array(keys), shapes=np.array([",".join(map(str, sd[k].shape)) for k in keys]),
          named_parameters=np.array([n for n, _ in net.named_parameters()]),
          sgd_group_sizes=np.array([len(g["params"]) for g in osd["param_groups"]]),
          sgd_group_keys=np.array(sorted(osd["param_groups"][0].keys())), stage=np.array(str(args.stage)))


# ----------------------------------------------------------------------------------------------
# Round-2 additions.  Each lives in its OWN variant (own process, own generators) so that the fixtures above stay
# byte-identical when the script is re-run.
def gen_corr_xy_admm():
    """G4b: the GENERAL corr(x, y), y is not x (model/quantization.py:134-137), and G11: the `cdf` nn.Module used
    stand-alone (model/quantization.py:41-59) for both quant_src values, with its autograd gradient."""
    import torch
    q, args = _enter("admm_cifar", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "8"])
    g = torch.Generator().manual_seed(2024)
    out = {}
    for ci, (B, Fdim) in enumerate([(16, 256), (128, 1024), (5, 70)]):
        x = torch.randn(B, Fdim, generator=g) * 0.8 + 0.1
        y = torch.randn(B, Fdim, generator=g) * 1.3 - 0.2 + 0.3 * x
        dG = torch.randn(B, B, generator=g)
        xi, yi = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        G = q.corr(xi, yi)
        G.backward(dG)
        out[f"x_c{ci}"], out[f"y_c{ci}"], out[f"dG_c{ci}"] = _np(x), _np(y), _np(dG)
        out[f"G_c{ci}"], out[f"dx_c{ci}"], out[f"dy_c{ci}"] = _np(G), _np(xi.grad), _np(yi.grad)
    _save("g4b_corr_xy_noeps", **out)

    out = {"act_range": np.array(float(args.act_range), dtype=np.float32)}
    v = torch.randn(4, 6, 5, 5, generator=g) * 1.1
    gc = torch.randn(v.shape, generator=g)
    out["v"], out["gc"] = _np(v), _np(gc)
    for src, m, s in (("a", 0.0, 1.0), ("w", 0.07, 0.6)):
        vi = v.clone().requires_grad_(True)
        c, pdf = q.cdf(torch.tensor(m), torch.tensor(s), src)(vi)
        c.backward(gc)
        out[f"m_{src}"], out[f"s_{src}"] = np.array(m, np.float32), np.array(s, np.float32)
        out[f"cdf_{src}"], out[f"pdf_{src}"], out[f"dv_{src}"] = _np(c), _np(pdf), _np(vi.grad)
    _save("g11_cdf_module_admm", **out)


def gen_corr_xy_office():
    """G4b, Office tree: corr(x, y) with the +1e-5 (dann_office/model/quantization.py:158-161)."""
    import torch
    q, args = _enter("office", ["--bitW", "8", "--abitW", "8"])
    g = torch.Generator().manual_seed(2025)
    out = {}
    for ci, (B, Fdim) in enumerate([(28, 512), (7, 96)]):
        x = torch.randn(B, Fdim, generator=g) * 0.8 + 0.1
        y = torch.randn(B, Fdim, generator=g) * 1.3 - 0.2 + 0.3 * x
        if ci == 1:
            y[:, 3] = -0.5          # a constant feature of y: finite only thanks to the +1e-5
        dG = torch.randn(B, B, generator=g)
        xi, yi = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        G = q.corr(xi, yi)
        G.backward(dG)
        out[f"x_c{ci}"], out[f"y_c{ci}"], out[f"dG_c{ci}"] = _np(x), _np(y), _np(dG)
        out[f"G_c{ci}"], out[f"dx_c{ci}"], out[f"dy_c{ci}"] = _np(G), _np(xi.grad), _np(yi.grad)
    _save("g4b_corr_xy_eps", **out)


def gen_office_tiny_dann():
    """G10 (SURVEY.md §8f-N3 at value level): a tiny DANN — ResNet(Bottleneck, [1,1,1,1], width_per_group=8) + both heads,
    4W/4A, batch 6, 64x64 inputs — through TWO iterations of dann_office/main.py:343-456 (source pass, target pass, summed
    loss, SGD over the three groups incl. alterD/gamma, ADMM_OPT.step on the TARGET pass's D), with the per-epoch SGD
    re-creation of main.py:321-328 between them (epoch 1 -> epoch 2: new learning rates, momentum buffers start over).
    Initial parameters come from tests/golden/det_init.py (same function on the test side), so no state_dict is stored."""
    import importlib
    import math
    import torch
    sys.path.insert(0, HERE)
    from det_init import det_init_, sample
    q, args = _enter("office", ["--bitW", "4", "--abitW", "4", "--train_batch_size", "6"])
    r = importlib.import_module("model.resnet")
    r.device = torch.device("cpu")
    from utils.optimizer import SGD, ADMM_OPT
    torch.manual_seed(0)
    net = r.DANN(lambda w, a, s: r.ResNet(w, a, s, r.Bottleneck, [1, 1, 1, 1], width_per_group=8), 4, 4, args.stage)
    net.train()
    det_init_(net)
    g = torch.Generator().manual_seed(31)
    xs = torch.randn(2, 6, 3, 64, 64, generator=g)
    xt = torch.randn(2, 6, 3, 64, 64, generator=g) * 1.2 + 0.1
    ys = torch.randint(0, 31, (2, 6), generator=g)
    out = dict(xs=_np(xs), xt=_np(xt), ys=_np(ys), stage=np.array(str(args.stage)), lr=np.array(0.004),
               num_epochs=np.array(10), alpha=np.array(0.5), names=np.array([n for n, _ in net.named_parameters()]))
    named = list(net.named_parameters())
    param_admm = [(n, p) for n, p in named if "alterD" in n or "gamma" in n]
    opt_a = ADMM_OPT([p for _, p in param_admm])
    ce = torch.nn.CrossEntropyLoss()
    blocks = [b for layer in (net.feature.layer1, net.feature.layer2, net.feature.layer3, net.feature.layer4) for b in layer]
    for it, epoch in enumerate((1, 2)):
        rate = 0.004 / math.pow(1 + 10 * (epoch - 1) / 10, 0.75)                    # main.py:321
        opt_t = SGD([{"params": net.feature.parameters()},                          # main.py:324-328 (new every epoch)
                     {"params": net.class_classifier.parameters(), "lr": rate},
                     {"params": net.domain_classifier.parameters(), "lr": rate}],
                    lr=rate / 10, momentum=0.9, weight_decay=5e-4)
        opt_t.zero_grad()
        opt_a.zero_grad()
        cls_s, dom_s, tl_s = net(xs[it], alpha=0.5)
        D_src = [_np(b.admm0.D) for b in blocks]
        l_cls = ce(cls_s, ys[it])
        l_ds = ce(dom_s, torch.zeros(6, dtype=torch.long))
        _, dom_t, tl_t = net(xt[it], alpha=0.5)
        l_dt = ce(dom_t, torch.ones(6, dtype=torch.long))
        loss = l_cls + l_ds + l_dt + tl_s + tl_t
        loss.backward()
        idx = [j for j, (n, _) in enumerate(named) if ("conv" in n or "downsample.0" in n) and "weight" in n][1:]
        w_cdf, w_pdf = [], []
        for b in blocks:
            for k, conv in enumerate([b.conv1, b.conv2, b.conv3, b.downsample]):
                if conv is not None:
                    conv = conv[0] if k == 3 else conv
                    w_cdf.append(conv.quantize_fn.weight_cdf)
                    w_pdf.append(conv.quantize_fn.weight_pdf)
        a_idx = [j for j, (n, _) in enumerate(param_admm) if "alterD" in n]
        g_idx = [j for j, (n, _) in enumerate(param_admm) if "gamma" in n]
        out[f"cls_s_{it}"], out[f"dom_s_{it}"], out[f"dom_t_{it}"] = _np(cls_s), _np(dom_s), _np(dom_t)
        out[f"tl_s_{it}"], out[f"tl_t_{it}"], out[f"loss_{it}"] = _np(tl_s), _np(tl_t), _np(loss)
        for bi, b in enumerate(blocks):
            out[f"Dsrc_{it}_{bi}"] = D_src[bi]
            out[f"D_{it}_{bi}"] = _np(b.admm0.D)                                    # the TARGET pass's D
        for j, (n, p) in enumerate(named):
            if p.grad is not None:                      # feature.fc is never used by DANN.forward
                out[f"grad_{it}/{j}"] = _np(sample(p.grad))
        opt_t.step(idx, w_cdf, w_pdf, float(args.lam), float(args.lam2))
        opt_a.step(a_idx, g_idx, [b.admm0.D for b in blocks], [b.admm0.alterD for b in blocks],
                   [b.admm0.gamma for b in blocks], [b.admm0.mu for b in blocks], [b.admm0.rho for b in blocks])
        for j, (n, p) in enumerate(named):
            out[f"after_{it}/{j}"] = _np(sample(p))
            st = opt_t.state.get(p, {})
            if "momentum_buffer" in st:
                out[f"buf_{it}/{j}"] = _np(sample(st["momentum_buffer"]))
        out[f"rate_{it}"] = np.array(rate)
    _save("g10_office_tiny_dann", **out)


def gen_tiny_resnet_sites():
    """G8b ("capture per-site inputs/outputs via forward hooks on 3 sites"): the tiny PreActResNet([1,1,1]) of G8
    (same seed, same first batch) with forward hooks on three activation sites — the stem's, the first block's act_q1 (whose
    output joins the shortcut) and the last block's act_q0 — recording, in model context, the site's input (the BN output),
    its x_q, its trans loss, its D and its ADMM state.  The GPU test teacher-forces exactly these inputs through the HIP site,
    which makes the whole-model comparison of G8 tight where it matters (per site) instead of bin-flip scale end to end."""
    import torch
    q, args = _enter("admm_cifar", ["--bitW", "4", "--abitW", "4", "--train_batch_size", "8"])
    import model.resnet as r
    r.device = torch.device("cpu")
    g = torch.Generator().manual_seed(1234)
    # replay the generator consumption of gen_admm_cifar up to G8's inputs is not needed: G8's own file holds xs / init
    g8 = np.load(os.path.join(HERE, "g8_tiny_resnet_admm.npz"))
    torch.manual_seed(5)
    net = r.PreActResNet(r.PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10)
    net.train()
    sd = {k[len("init/"):]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith("init/")}
    net.load_state_dict(sd)
    sites = {"stem": net.act_q0, "b0q1": net.layers[0].act_q1, "b2q0": net.layers[2].act_q0}
    rec = {}

    def hook(name):
        def fn(mod, inp, out):
            rec[name] = dict(x=_np(inp[0]), xq=_np(out[0]), loss=_np(out[1]), D=_np(mod.opt.D), alterD=_np(mod.opt.alterD),
                             gamma=_np(mod.opt.gamma))
        return fn
    hs = [m.register_forward_hook(hook(n)) for n, m in sites.items()]
    logits, tl = net(torch.from_numpy(g8["xs"][0]))
    for h in hs:
        h.remove()
    out = {"k": np.array(4), "act_range": np.array(float(args.act_range), dtype=np.float32), "logits": _np(logits), "trans": _np(tl)}
    for n, d in rec.items():
        for k_, v in d.items():
            out[f"{n}/{k_}"] = v
    assert np.allclose(out["logits"], g8["logits_0"], atol=1e-6), "G8b must replay G8's first forward"
    _save("g8b_tiny_resnet_sites", **out)


def gen_office_bottleneck_sites():
    """G13: the Office tree's counterpart of G8b.  The tiny DANN of G10 (same det_init, same first source
    batch) with forward hooks on ONE Bottleneck (feature.layer4[0]: downsample branch, 2048 output channels) recording, in
    model context (dann_office/model/resnet.py:131-156): the inputs of bn1 / bn2 / bn3 (the convolutions' outputs), the
    batch-norm parameters, the inputs and outputs of act_q1 / act_q2 (plain quantisers) and of act_q3 (the ADMM site: x_q,
    trans loss, D, alterD, gamma), the downsample branch's output and the block's output.  The GPU tests teacher-force these
    inputs through the HIP paths (plain, ReLU-fused and batch-norm-folded)."""
    import importlib
    import torch
    sys.path.insert(0, HERE)
    from det_init import det_init_
    q, args = _enter("office", ["--bitW", "4", "--abitW", "4", "--train_batch_size", "6"])
    r = importlib.import_module("model.resnet")
    r.device = torch.device("cpu")
    torch.manual_seed(0)
    net = r.DANN(lambda w, a, s: r.ResNet(w, a, s, r.Bottleneck, [1, 1, 1, 1], width_per_group=8), 4, 4, args.stage)
    net.train()
    det_init_(net)
    g = torch.Generator().manual_seed(31)
    xs = torch.randn(2, 6, 3, 64, 64, generator=g)          # G10's generator stream: xs[0] is its first source batch
    blk = net.feature.layer4[0]
    rec = {}

    def bn_hook(name):
        def fn(mod, inp, out):
            rec[name + "/z"] = _np(inp[0])
            rec[name + "/out"] = _np(out)
            rec[name + "/weight"], rec[name + "/bias"] = _np(mod.weight), _np(mod.bias)
            rec[name + "/running_mean"], rec[name + "/running_var"] = _np(mod.running_mean), _np(mod.running_var)
            rec[name + "/eps"], rec[name + "/momentum"] = np.array(mod.eps), np.array(mod.momentum)
        return fn

    def act_hook(name):
        def fn(mod, inp, out):
            rec[name + "/x"] = _np(inp[0])
            if isinstance(out, tuple):
                rec[name + "/xq"], rec[name + "/loss"] = _np(out[0]), _np(out[1])
                rec[name + "/D"], rec[name + "/alterD"], rec[name + "/gamma"] = _np(mod.opt.D), _np(mod.opt.alterD), _np(mod.opt.gamma)
            else:
                rec[name + "/xq"] = _np(out)          # (before the in-place ReLU that follows)
        return fn
    hs = [blk.bn1.register_forward_hook(bn_hook("bn1")), blk.bn2.register_forward_hook(bn_hook("bn2")),
          blk.bn3.register_forward_hook(bn_hook("bn3")), blk.downsample[1].register_forward_hook(bn_hook("bnd")),
          blk.act_q1.register_forward_hook(act_hook("q1")), blk.act_q2.register_forward_hook(act_hook("q2")),
          blk.act_q3.register_forward_hook(act_hook("q3")),
          blk.register_forward_hook(lambda m, i, o: rec.__setitem__("block/out", _np(o[0])))]
    net(xs[0], alpha=0.5)
    for h in hs:
        h.remove()
    out = {"k": np.array(4), "act_range": np.array(float(args.act_range), dtype=np.float32), "stage": np.array(str(args.stage))}
    out.update(rec)
    _save("g13_office_bottleneck_sites", **out)


def gen_g3l_admm():
    q, args = _enter("admm_cifar", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "8"])
    args.method = "plain"          # model/quantization.py:112: the quantiser alone, no corr / ADMM branch
    _g3l(q, args, "admm")


def gen_g3l_cdfonly():
    q, args = _enter("cdf_only", ["--bitW", "8", "--abitW", "8"])
    _g3l(q, args, "cdf")


def _g11b(q, args, tree):
    """G11b (round 5): the `cdf` module with LIVE statistics - m and s are tensors of the autograd graph, as in the reference's own
    use cdf(torch.mean(x), torch.std(x), 'w')(x) (ADMM tree model/quantization.py:78, CDF tree :70): the gradients of BOTH
    outputs w.r.t. the tensor, m and s (free m, s), and dW of the composed form through mean / std."""
    import torch
    g = torch.Generator().manual_seed(20261005)
    out = {"act_range": np.array(float(getattr(args, "act_range", 2)), dtype=np.float32)}
    v = torch.randn(6, 4, 3, 3, generator=g) * 0.7 + 0.05
    gc, gp = torch.randn(v.shape, generator=g), torch.randn(v.shape, generator=g) * 0.3
    out["v"], out["gc"], out["gp"] = _np(v), _np(gc), _np(gp)
    for src, m0, s0 in (("w", 0.07, 0.6), ("a", -0.2, 1.3)):
        vi = v.clone().requires_grad_(True)
        m, s = torch.tensor(m0, requires_grad=True), torch.tensor(s0, requires_grad=True)
        c, pdf = q.cdf(m, s, src)(vi)
        torch.autograd.backward([c, pdf], [gc, gp])
        out[f"m_{src}"], out[f"s_{src}"] = np.array(m0, np.float32), np.array(s0, np.float32)
        out[f"cdf_{src}"], out[f"pdf_{src}"] = _np(c), _np(pdf)
        out[f"dv_{src}"], out[f"dm_{src}"], out[f"ds_{src}"] = _np(vi.grad), _np(m.grad), _np(s.grad)
    # the composed form the reference's weight quantiser builds: gradient of the first output only, through mean and std
    wi = v.clone().requires_grad_(True)
    c, pdf = q.cdf(torch.mean(wi), torch.std(wi), "w")(wi)
    c.backward(gc)
    out["cdf_ms"], out["pdf_ms"], out["dW_ms"] = _np(c), _np(pdf), _np(wi.grad)
    _save("g11b_cdf_live_stats_" + tree, **out)


def gen_g11b_admm():
    q, args = _enter("admm_cifar", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "8"])
    _g11b(q, args, "admm")


def gen_g11b_cdfonly():
    q, args = _enter("cdf_only", ["--bitW", "8", "--abitW", "8"])
    _g11b(q, args, "cdfonly")

def gen_g14_constant_column():
    """G14: the CIFAR trees' `corr` divides by an unguarded std (model/quantization.py:134-137), so a feature
    that is constant over the batch gives 0/0 = NaN in the standardised matrix, and through x_hat x_hat^T / F every entry of the
    correlation becomes NaN.  `corr` alone with a finite upstream dG: dx is NaN in the constant columns only (column j of the
    gradient only sees column j of x_hat).  The whole ADMM site: D, the loss and EVERY gradient (dx, dalterD, dgamma) are NaN, while
    x_q (elementwise) is untouched.  Recorded: the reference's own outputs at [128, 4096] and [28, 1568] with two constant columns
    (value 0 and value 0.75).  Kept small: x is generated fp16-representable and stored as fp16; x_q as its int16 level index
    (x_q = idx / n exactly) and corr's dx by value for the first 64 columns (both constant ones among them); everything else as
    NaN masks (the site's upstream gradient is any finite tensor: not stored)."""
    import torch
    q, args = _enter("admm_cifar", ["--bitW", "8", "--abitW", "8", "--train_batch_size", "8"])
    from utils.admm import ADMM
    g = torch.Generator().manual_seed(1414)
    k, n = 8, 255
    out = {"const_cols": np.array([5, 17]), "const_vals": np.array([0.0, 0.75], dtype=np.float32), "k": np.array(k), "head": np.array(64)}
    for name, B, C, H, W in (("a", 128, 16, 16, 16), ("b", 28, 8, 14, 14)):
        x = (torch.randn(B, C * H * W, generator=g) * 0.8 + 0.1).half().float()
        x[:, 5] = 0.0
        x[:, 17] = 0.75
        out[f"x_{name}"] = _np(x).astype(np.float16)
        assert np.array_equal(out[f"x_{name}"].astype(np.float32), _np(x))
        out[f"shape_{name}"] = np.array([B, C, H, W])
        # corr alone
        dG = torch.randn(B, B, generator=g)
        xi = x.clone().requires_grad_(True)
        G = q.corr(xi, xi)
        G.backward(dG)
        cdx = _np(xi.grad)
        out[f"dG_{name}"] = _np(dG)
        out[f"G_isnan_{name}"] = np.packbits(np.isnan(_np(G)))
        out[f"corr_dx_isnan_{name}"] = np.packbits(np.isnan(cdx))
        out[f"corr_dx_head_{name}"] = cdx[:, :64].copy()
        # the whole ADMM site (model/quantization.py:102-132, utils/admm.py:24-33)
        torch.manual_seed(141)
        admm = ADMM(B)
        fn = q.activation_quantize_fn(k, "second", admm)
        gq = (torch.randn(B, C, H, W, generator=g) * 0.01).half().float()
        xi = x.clone().view(B, C, H, W).requires_grad_(True)
        out[f"alterD0_{name}"], out[f"gamma0_{name}"] = _np(admm.alterD), _np(admm.gamma)
        xq, tl = fn(xi)
        (tl + (xq * gq).sum()).backward()
        bins = np.rint(_np(xq).astype(np.float64) * n).reshape(B, -1)
        assert np.array_equal((bins / n).astype(np.float32), _np(xq).reshape(B, -1))
        out[f"bins_head_{name}"] = bins[:, :64].astype(np.int16)       # (the upstream gradient is not stored: any finite one gives NaN)
        out[f"loss_{name}"] = _np(tl)
        for key, t in (("D", admm.D), ("dx", xi.grad), ("dalterD", admm.alterD.grad), ("dgamma", admm.gamma.grad)):
            out[f"{key}_isnan_{name}"] = np.packbits(np.isnan(_np(t)))
    _save("g14_constant_column", **out)


GEN = {"g11b_admm": gen_g11b_admm, "g11b_cdfonly": gen_g11b_cdfonly, "g3l_admm": gen_g3l_admm, "g3l_cdfonly": gen_g3l_cdfonly, "admm_cifar": gen_admm_cifar, "cdf_only": gen_cdf_only, "office": gen_office, "office_keys": gen_office_keys,
       "corr_xy_admm": gen_corr_xy_admm, "corr_xy_office": gen_corr_xy_office, "office_tiny_dann": gen_office_tiny_dann, "tiny_resnet_sites": gen_tiny_resnet_sites,
       "office_bottleneck_sites": gen_office_bottleneck_sites,
       "g14_constant_column": gen_g14_constant_column}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", choices=sorted(GEN), default=None)
    a = ap.parse_args()
    if a.variant is None:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for v in GEN:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", v], check=True, env=env)
        return
    GEN[a.variant]()


if __name__ == "__main__":
    main()
