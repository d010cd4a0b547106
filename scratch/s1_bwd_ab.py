"""site1 backward at [28, 802816], plain and the Office step's folded dual form: python3 scratch/s1_bwd_ab.py  (ALIGNQ_SO=... for A/B)"""
import sys, torch
sys.path.insert(0, '.')
import bench
from alignq_amd import _lib as L
lib = L.load(); st = L.stream_ptr(); p = L.ptr
dev = torch.device('cuda:0')
B, F = 28, 802816
R = 4
xs = [torch.randn(B, F, device=dev) for _ in range(R)]; gs = [torch.randn(B, F, device=dev) * 0.01 for _ in range(R)]
dx = torch.empty(B, F, device=dev); xq = torch.empty(B, F, device=dev)
stats = torch.empty(4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
D, A, Gm = torch.empty(B, B, device=dev), torch.rand(B, B, device=dev), torch.rand(B, B, device=dev)
scal, one = torch.empty(4, device=dev), torch.ones((), device=dev)
S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
dA, dG = torch.empty_like(A), torch.empty_like(Gm)
lib.alignq_site_partials(p(xs[0]), B, F, 8, 2.0, 1e-5, p(xq), p(stats), p(ws), st)
lib.alignq_site_reduce_loss(p(ws), B, F, p(D), p(A), p(Gm), B, 0.2, 0.3, p(scal), st)
lib.alignq_site_prep_fused(p(D), p(A), p(Gm), B, p(scal), 0.2, p(one), B, F, p(S), p(dA), p(dG), st)
t = bench.time_call_rot(lambda i: lib.alignq_site_bwd_apply(p(gs[i]), p(S), p(xs[i]), p(stats), B, F, 2.0, 1e-5, p(dx), st), 24, R)
print("plain [28,802816] bwd %.1f us (%.3f of 8 TB/s)" % (t * 1e6, 12.0 * B * F / t / 8e12))
