"""time the folded batch-norm + quantiser chain at Office shapes: rocprof-free, HIP events around each entry point"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = L.stream_ptr(); p = L.ptr
def t(fn, n=12):
    for i in range(3): fn()
    ev = [torch.cuda.Event(True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]) * 1e3)
for (G, Pimg, C) in ((2, 28 * 56 * 56, 64), (2, 28 * 56 * 56, 256), (2, 28 * 28 * 28, 128), (2, 28 * 28 * 28, 512), (2, 28 * 14 * 14, 1024), (2, 28 * 7 * 7, 2048), (2, 28 * 112 * 112, 64)):
    n = G * Pimg * C
    z = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01
    y, dz = torch.empty_like(z), torch.empty_like(z)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
    dga, dbe = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
    f = t(lambda: L.check(lib.alignq_bnq_fwd(p(z), Pimg, C, G, p(gam), p(bet), p(rm), p(rv), None, 0.1, 1e-5, 8, 2.0, 0, 1, p(ab), p(save), p(y), p(ws), st), "f"))
    s = t(lambda: L.check(lib.alignq_bnq_stats(p(z), Pimg, C, G, p(gam), p(bet), p(rm), p(rv), None, 0.1, 1e-5, p(ab), p(save), p(ws), st), "s"))
    b = t(lambda: L.check(lib.alignq_bnq_bwd(p(g), p(z), p(y), p(ab), p(save), Pimg, C, G, 2.0, 1, p(dz), p(dga), p(dbe), p(ws), st), "b"))
    mb = n * 4 / 1e6
    print(f"[{G}x{Pimg},{C}] {mb:7.1f} MB  stats {s:6.1f} us ({mb / s:5.2f} TB/s)  fwd(3 launches, 12 B/el) {f:6.1f} us ({3 * mb / f:5.2f} TB/s)  bwd(3 launches, 28 B/el) {b:6.1f} us ({7 * mb / b:5.2f} TB/s)")
