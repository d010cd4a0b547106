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
