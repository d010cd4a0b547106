"""time the fused site forward / backward at one shape (HIP events, median of N launches)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = L.stream_ptr(); p = L.ptr
B, F = int(sys.argv[1]), int(sys.argv[2])
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
R = 3
xs = [torch.randn(B, F, device=dev) for _ in range(R)]; gs = [torch.randn(B, F, device=dev) * 0.01 for _ in range(R)]
xq, dx = torch.empty_like(xs[0]), torch.empty_like(xs[0])
stats = torch.empty(4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
def t(fn, n=20):
    for i in range(3): fn(i)
    ev = [torch.cuda.Event(True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(i); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]) * 1e3)
f = t(lambda i: lib.alignq_site_partials(p(xs[i % R]), B, F, 8, 2.0, eps, p(xq), p(stats), p(ws), st))
b = t(lambda i: lib.alignq_site_bwd_apply(p(gs[i % R]), p(S), p(xs[i % R]), p(stats), B, F, 2.0, eps, p(dx), st))
print(f"[{B},{F}] fwd {f:.1f} us ({8.0 * B * F / f / 1e6:.2f} TB/s, {8.0 * B * F / f / 8e6:.3f})  bwd {b:.1f} us ({12.0 * B * F / b / 1e6:.2f} TB/s, {12.0 * B * F / b / 8e6:.3f})")
