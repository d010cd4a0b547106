"""The fused transition launches (alignq_transition_nhwc_fwd / _bwd, ops.QTransitionFn) against the separate launches they
replace (alignq_conv_gen_nhwc_fwd / _dgrad / _wgrad), which tests/test_gpu_parity.py pins to the oracle: outputs, batch-norm
partials and filter gradients bit for bit, the data gradient to fp32 accumulation order."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, B, CIN, COUT, H, W, w_bit, seed):
    from alignq_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    cl = torch.channels_last
    x = torch.randn(B, CIN, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    n = (1 << w_bit) - 1
    w3 = (torch.randint(-n, n + 1, (COUT, CIN, 3, 3), generator=g).float() / n).to(dev).contiguous(memory_format=cl)
    w1 = (torch.randint(-n, n + 1, (COUT, CIN, 1, 1), generator=g).float() / n).to(dev).contiguous(memory_format=cl)
    gy3 = torch.randn(B, COUT, H // 2, W // 2, generator=g).to(dev).contiguous(memory_format=cl)
    gy1 = torch.randn(B, COUT, H // 2, W // 2, generator=g).to(dev).contiguous(memory_format=cl)
    return x, w3, w1, gy3, gy1


@pytest.mark.parametrize("B,CIN,COUT,H,W", [(128, 16, 32, 32, 32), (128, 32, 64, 16, 16), (8, 16, 32, 32, 32), (6, 32, 64, 16, 16)])
@pytest.mark.parametrize("w_bit", [8, 4])
def test_transition_launches_equal_the_separate_launches(B, CIN, COUT, H, W, w_bit):
    from alignq_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    x, w3, w1, gy3, gy1 = _mk(dev, B, CIN, COUT, H, W, w_bit, 7 + CIN + w_bit)
    st, p = L.stream_ptr(), L.ptr
    cl = torch.channels_last
    f32 = dict(dtype=torch.float32, device=dev)

    def empty_y():
        return torch.empty((B, COUT, H // 2, W // 2), **f32).contiguous(memory_format=cl)

    n3 = lib.alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, 3, 2)
    n1 = lib.alignq_conv_gen_bn_parts(B, H, W, CIN, COUT, 1, 2)
    assert n3 > 0 and n1 > 0
    # ---- forward
    y3r, y1r, y3, y1 = empty_y(), empty_y(), empty_y(), empty_y()
    p3r, p1r = torch.empty(COUT, n3, 2, **f32), torch.empty(COUT, n1, 2, **f32)
    p3, p1 = torch.empty_like(p3r), torch.empty_like(p1r)
    L.check(lib.alignq_conv_gen_nhwc_fwd(p(x), p(w3), p(y3r), B, H, W, CIN, COUT, 3, 2, w_bit, p(p3r), st), "fwd3")
    L.check(lib.alignq_conv_gen_nhwc_fwd(p(x), p(w1), p(y1r), B, H, W, CIN, COUT, 1, 2, w_bit, p(p1r), st), "fwd1")
    L.check(lib.alignq_transition_nhwc_fwd(p(x), p(w3), p(w1), p(y3), p(y1), B, H, W, CIN, COUT, w_bit, p(p3), p(p1), st), "fwd")
    torch.cuda.synchronize()
    assert torch.equal(y3, y3r) and torch.equal(y1, y1r)
    assert torch.equal(p3, p3r) and torch.equal(p1, p1r)
    # ---- backward, plain gradients and with an `add` operand
    add = torch.randn_like(x)
    ws = lambda ks: torch.empty(lib.alignq_conv_gen_wgrad_ws_bytes(CIN, COUT, ks) // 4, **f32)
    for a in (None, add):
        dx1r, dxr = torch.empty_like(x), torch.empty_like(x)
        none7 = [None] * 7
        L.check(lib.alignq_conv_gen_nhwc_dgrad(p(gy1), p(w1), p(dx1r), B, H, W, CIN, COUT, 1, 2, w_bit, p(a), *none7, st), "dg1")
        L.check(lib.alignq_conv_gen_nhwc_dgrad(p(gy3), p(w3), p(dxr), B, H, W, CIN, COUT, 3, 2, w_bit, p(dx1r), *none7, st), "dg3")
        dw3r, dw1r = torch.empty_like(w3), torch.empty_like(w1)
        none5 = [None] * 5
        L.check(lib.alignq_conv_gen_nhwc_wgrad(p(x), p(gy3), p(dw3r), p(ws(3)), B, H, W, CIN, COUT, 3, 2, None, *none5, st), "wg3")
        L.check(lib.alignq_conv_gen_nhwc_wgrad(p(x), p(gy1), p(dw1r), p(ws(1)), B, H, W, CIN, COUT, 1, 2, None, *none5, st), "wg1")
        dx, dw3, dw1 = torch.empty_like(x), torch.empty_like(w3), torch.empty_like(w1)
        ws3, ws1 = ws(3), ws(1)
        ns3, ns1 = ctypes.c_int(0), ctypes.c_int(0)
        L.check(lib.alignq_transition_nhwc_bwd(p(x), p(gy3), p(gy1), p(w3), p(w1), p(dx), p(ws3), p(ws1), B, H, W, CIN, COUT,
                                               w_bit, ctypes.byref(ns3), ctypes.byref(ns1), p(a), *([None] * 14), st), "bwd")
        L.check(lib.alignq_conv3x3_wgrad_reduce_multi(2, L.ptr_array([ws3, ws1]), L.ptr_array([dw3, dw1]),
                                                      (ctypes.c_int * 2)(ns3.value, ns1.value),
                                                      (ctypes.c_int * 2)(9 * CIN * COUT, CIN * COUT), st), "reduce")
        torch.cuda.synchronize()
        assert torch.equal(dw3, dw3r) and torch.equal(dw1, dw1r)
        # dx: one accumulation chain over ten taps instead of nine + a separately rounded 1x1 term
        err = (dx - dxr).abs().max().item()
        assert err <= 2e-5 * max(1.0, dxr.abs().max().item()), err


def test_transition_function_matches_the_two_convolution_functions():
    """ops.QTransitionFn (with the lazy batch-norm links unused: plain gradients) == QConvGenFn x 2 through autograd."""
    from alignq_amd import ops
    dev = torch.device("cuda:0")
    B, CIN, COUT, H, W, w_bit = 16, 16, 32, 32, 32, 8
    x, w3, w1, gy3, gy1 = _mk(dev, B, CIN, COUT, H, W, w_bit, 3)
    xa, w3a, w1a = (t.clone().requires_grad_(True) for t in (x, w3, w1))
    y3, y1 = ops.QTransitionFn.apply(xa, w3a, w1a, w_bit)
    torch.autograd.backward([y3, y1], [gy3, gy1])
    xb, w3b, w1b = (t.clone().requires_grad_(True) for t in (x, w3, w1))
    z3, xt = ops.QConvGenFn.apply(xb, w3b, w_bit, 1, True)
    z1 = ops.QConvGenFn.apply(xt, w1b, w_bit, 0)
    torch.autograd.backward([z3, z1], [gy3, gy1])
    torch.cuda.synchronize()
    assert torch.equal(y3, z3) and torch.equal(y1, z1)
    assert torch.equal(w3a.grad, w3b.grad) and torch.equal(w1a.grad, w1b.grad)
    assert (xa.grad - xb.grad).abs().max().item() <= 2e-5 * max(1.0, xb.grad.abs().max().item())


def test_head_forward_means_and_merged_head_prep_backward():
    """alignq_head_ce_fwd's in-kernel batch mean / site-loss total against torch, twice (the ticket re-arms), and
    alignq_head_ce_bwd_site_prep against the two launches it replaces, bit for bit."""
    import torch.nn.functional as F
    from alignq_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    f32 = dict(dtype=torch.float32, device=dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    B, C, H, W, K, S, dim = 128, 64, 8, 8, 10, 21, 128
    feat = torch.randn(B, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    Wt, bias = torch.randn(K, C, generator=g).to(dev) * 0.2, torch.randn(K, generator=g).to(dev) * 0.1
    target = torch.randint(0, K, (B,), generator=g).to(dev)
    scal_all = torch.rand(64, 4, generator=g).to(dev)
    st, p = L.stream_ptr(), L.ptr
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    ref_logits = F.adaptive_avg_pool2d(feat, 1).flatten(1) @ Wt.t() + bias
    ref_ce = F.cross_entropy(ref_logits, target)
    for _ in range(2):
        pooled, logits, probs = torch.empty(B, C, **f32), torch.empty(B, K, **f32), torch.empty(B, K, **f32)
        loss, ce, tr = torch.empty(B, **f32), torch.empty((), **f32), torch.empty((), **f32)
        L.check(lib.alignq_head_ce_fwd(p(feat), p(Wt), p(bias), p(target), B, H * W, C, K, p(pooled), p(logits), p(probs), p(loss),
                                       p(ce), p(counter), p(scal_all), S, p(tr), st), "head fwd")
        torch.cuda.synchronize()
        assert abs(ce.item() - ref_ce.item()) <= 2e-6 * max(1.0, abs(ref_ce.item()))
        assert abs(ce.item() - loss.double().mean().item()) <= 1e-6
        assert abs(tr.item() - scal_all[:S, 0].double().sum().item()) <= 1e-5
        assert counter.item() == 0
    # ---- backward: merged launch == head backward + site preparation
    gce, gtr = torch.full((), 1.0, **f32), torch.full((), 0.7, **f32)
    D = [torch.randn(B, B, generator=g).to(dev) * 0.05 for _ in range(S)]
    A = [torch.randn(dim, dim, generator=g).to(dev) * 0.05 for _ in range(S)]
    Gm = [torch.rand(dim, dim, generator=g).to(dev) * 0.01 for _ in range(S)]
    scal = [scal_all[i] for i in range(S)]
    Fs = [16384] * 7 + [8192] * 7 + [4096] * 7
    nS = lib.alignq_site_bwd_ws_bytes(B) // 4

    def outs():
        return ([torch.zeros(nS, **f32) for _ in range(S)], [torch.empty(dim, dim, **f32) for _ in range(S)],
                [torch.empty(dim, dim, **f32) for _ in range(S)])

    S1, dA1, dG1 = outs()
    S2, dA2, dG2 = outs()
    df1, dW1, db1 = torch.empty_like(feat), torch.empty_like(Wt), torch.empty(K, **f32)
    df2, dW2, db2 = torch.empty_like(feat), torch.empty_like(Wt), torch.empty(K, **f32)
    site = lambda So, dAo, dGo: (S, L.ptr_array(D), L.ptr_array(A), L.ptr_array(Gm), L.ptr_array(scal), p(gtr), L.i64_array(Fs),
                                 B, dim, 0.01, L.ptr_array(So), L.ptr_array(dAo), L.ptr_array(dGo), st)
    L.check(lib.alignq_head_ce_bwd(p(gce), p(probs), p(target), p(pooled), p(Wt), B, H * W, C, K, p(df1), p(dW1), p(db1), st), "hb")
    L.check(lib.alignq_site_prep_fused_multi(*site(S1, dA1, dG1)), "prep")
    L.check(lib.alignq_head_ce_bwd_site_prep(p(gce), p(probs), p(target), p(pooled), p(Wt), B, H * W, C, K, p(df2), p(dW2), p(db2),
                                             *site(S2, dA2, dG2)), "merged")
    torch.cuda.synchronize()
    assert torch.equal(df1, df2) and torch.equal(dW1, dW2) and torch.equal(db1, db2)
    for i in range(S):
        assert torch.equal(S1[i], S2[i]) and torch.equal(dA1[i], dA2[i]) and torch.equal(dG1[i], dG2[i])
