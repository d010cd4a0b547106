import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from alignq_amd import _lib as L
lib = L.load(); st = L.stream_ptr(); p = L.ptr
dev = torch.device('cuda:0')
for B, F in ((256, 16384), (512, 16384), (1024, 16384)):
    x = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01
    xq, dx = torch.empty_like(x), torch.empty_like(x)
    D = torch.empty(B, B, device=dev); stats = torch.empty(4, F, device=dev)
    dD = torch.randn(B, B, device=dev) * 1e-3
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    wsb = torch.empty(lib.alignq_site_bwd_ws_bytes(B), dtype=torch.uint8, device=dev)
    f = lambda: L.check(lib.alignq_site_fwd(p(x), B, F, 8, 2.0, 0.0, p(xq), p(D), p(stats), p(ws), st), 'f')
    b = lambda: L.check(lib.alignq_site_bwd(p(g), p(dD), None, p(x), p(stats), B, F, 2.0, 0.0, p(dx), p(wsb), st), 'b')
    f(); b()
    print(B, F, 'site fwd %.1f us  bwd %.1f us' % (bench.time_call(f, 20) * 1e6, bench.time_call(b, 20) * 1e6))
