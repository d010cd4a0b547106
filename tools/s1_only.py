#!/usr/bin/env python3
"""The bottleneck tail's two kernels alone at one of configuration 5's shapes (for rocprofv3 --pmc passes and quick timings):
    python3 tools/s1_only.py [C H] [reps]     default 256 56 (2 x [28, 802816]); 2048 7 is the other extreme"""
import sys, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); p = L.ptr
C, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 56)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, G, k, r, eps = 28, 2, 8, 2.0, 1e-5
F, P = C * H * H, B * H * H
g = torch.Generator().manual_seed(0)
z = (torch.randn(G * B, H, H, C, generator=g) * 1.2 + 0.1).to(dev)
res = (torch.relu(torch.randn(G * B, H, H, C, generator=g)) - 0.3).to(dev)
gy = (torch.randn(G * B, H, H, C, generator=g) * 1e-2).to(dev)
gam, bet = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
ws_bn = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
L.check(lib.alignq_bnq_stats(p(z), P, C, G, p(gam), p(bet), None, None, None, 0.1, 1e-5, p(ab), p(save), p(ws_bn), None), "stats")
stats = torch.empty(G, 4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F) * G, dtype=torch.uint8, device=dev)
y = torch.empty_like(z)
mask = torch.zeros(lib.alignq_site1_mask_bytes(B, F, G), dtype=torch.uint8, device=dev)
sb = lib.alignq_site_bwd_ws_bytes(B)
Sbuf = torch.zeros(sb * G, dtype=torch.uint8, device=dev)
dz, dres = torch.empty_like(z), torch.empty_like(z)
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
cols = torch.empty(lib.alignq_site1_cols_bytes(F, G), dtype=torch.uint8, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(reps + 2):
    ev[0].record()
    L.check(lib.alignq_site1_groups_fwd_m(p(z), p(ab), C, B, F, G, k, r, eps, p(res), 1, p(y), p(stats), p(ws), p(mask), None), "fwd_m")
    ev[1].record()
    L.check(lib.alignq_site1_groups_bwd_bn_m(p(gy), None, p(mask), p(Sbuf), p(z), p(ab), p(save), C, p(stats), B, F, G, r, eps, p(dz), p(dres),
                                             p(dg), p(db), p(cols), p(ws_bn), None), "bwd")
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
n = G * B * F
print(f"C={C} H={H}: site1 fwd {tf / reps * 1e3:.1f} us = {12.14 * n / (tf / reps * 1e-3) / 8e12:.3f} of 8 TB/s (12.14 B/el) | bwd chain {tb / reps * 1e3:.1f} us")
