"""alignq_site1_groups_fwd against _fwd_m (with the one-bit ReLU mask) and _bwd_bn against _bwd_bn_m on ONE box, config 5's two
extreme tail shapes (2 x [28, 802816], 2 x [28, 100352]); back-to-back launches over 4 rotating operand sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alignq_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
p = L.ptr
G, B, k, R = 2, 28, 8, 4


def t(fn, n=200):
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % R)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, H in ((256, 56), (2048, 7)):
    F, P = C * H * H, B * H * H
    zs = [torch.randn(G * B, H, H, C, device=dev) * 1.1 + 0.15 for _ in range(R)]
    rs = [torch.relu(torch.randn(G * B, H, H, C, device=dev)) for _ in range(R)]
    gs = [torch.randn(G * B, H, H, C, device=dev) * 0.01 for _ in range(R)]
    ys, dxs, drs = ([torch.empty(G * B, H, H, C, device=dev) for _ in range(R)] for _ in range(3))
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
    ws_bn = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
    L.check(lib.alignq_bnq_stats(p(zs[0]), P, C, G, p(gam), p(bet), None, None, None, 0.1, 1e-5, p(ab), p(save), p(ws_bn), None), "stats")
    stats = torch.empty(G, 4, F, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F) * G, dtype=torch.uint8, device=dev)
    mask = torch.empty(lib.alignq_site1_mask_bytes(B, F, G), dtype=torch.uint8, device=dev)
    cols = torch.empty(lib.alignq_site1_cols_bytes(F, G), dtype=torch.uint8, device=dev)
    S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) * G, dtype=torch.uint8, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    f0 = lambda i: L.check(lib.alignq_site1_groups_fwd(p(zs[i]), p(ab), C, B, F, G, k, 2.0, 1e-5, p(rs[i]), 1, p(ys[i]), p(stats), p(ws), None), "f")
    f1 = lambda i: L.check(lib.alignq_site1_groups_fwd_m(p(zs[i]), p(ab), C, B, F, G, k, 2.0, 1e-5, p(rs[i]), 1, p(ys[i]), p(stats), p(ws), p(mask), None), "fm")
    b0 = lambda i: L.check(lib.alignq_site1_groups_bwd_bn(p(gs[i]), None, p(ys[i]), p(S), p(zs[i]), p(ab), p(save), C, p(stats), B, F, G, 2.0, 1e-5,
                                                          p(dxs[i]), p(drs[i]), p(dg), p(db), p(cols), p(ws_bn), None), "b")
    b1 = lambda i: L.check(lib.alignq_site1_groups_bwd_bn_m(p(gs[i]), None, p(mask), p(S), p(zs[i]), p(ab), p(save), C, p(stats), B, F, G, 2.0, 1e-5,
                                                            p(dxs[i]), p(drs[i]), p(dg), p(db), p(cols), p(ws_bn), None), "bm")
    r = [t(f0), t(f1), t(f0), t(f1), t(b0), t(b1), t(b0), t(b1)]
    print(f"2x[28,{F}]  fwd {r[0]:.1f} / {r[2]:.1f}   fwd_m {r[1]:.1f} / {r[3]:.1f}   bwd_bn {r[4]:.1f} / {r[6]:.1f}   bwd_bn_m {r[5]:.1f} / {r[7]:.1f} us")
