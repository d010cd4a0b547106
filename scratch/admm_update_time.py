import sys, torch
sys.path.insert(0, '.')
import bench
from alignq_amd import _lib as L
lib = L.load(); st = L.stream_ptr()
dev = torch.device('cuda:0')
S = 21
for dim in (128, 256, 512, 1024):
    b = dim
    Ds = [torch.randn(b, b, device=dev) * 0.05 for _ in range(S)]
    As = [torch.rand(dim, dim, device=dev) for _ in range(S)]
    Gs = [torch.rand(dim, dim, device=dev) for _ in range(S)]
    pD, pA, pG = L.ptr_array(Ds), L.ptr_array(As), L.ptr_array(Gs)
    ws = torch.empty(lib.alignq_admm_update_ws_bytes(S, dim), dtype=torch.uint8, device=dev)
    f0 = lambda: L.check(lib.alignq_admm_update(pD, pA, pG, S, b, dim, 0.2, 0.3, st), "u")
    f = lambda: L.check(lib.alignq_admm_update_ws(pD, pA, pG, S, b, dim, 0.2, 0.3, L.ptr(ws), st), "u")
    f0(); print(dim, "one workgroup per site: %.1f us" % (bench.time_call(f0, 10) * 1e6))
    f()
    print(dim, 'admm_update x21 sites: %.1f us' % (bench.time_call(f, 10) * 1e6))
