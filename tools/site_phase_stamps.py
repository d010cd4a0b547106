import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
L.SO_PATH = 'tools/lib/libalignq_stamps.so'
lib = L.load()
lib.alignq_debug_read_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
B, k = 128, 8
for F in (16384, 8192, 4096):
    x = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01
    xq, dx = torch.empty_like(x), torch.empty_like(x)
    D = torch.empty(B, B, device=dev); stats = torch.empty(4, F, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)   # fp32 S + bf16 image, zero (timing only)
    st = L.stream_ptr(); p = L.ptr
    res = []
    for it in range(5):
        lib.alignq_site_partials(p(x), B, F, k, 2.0, 0.0, p(xq), p(stats), p(ws), st)
        lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, 0.0, p(dx), st)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        lib.alignq_debug_read_stamps(buf)
        a = np.array(buf[:16], dtype=np.int64)
        res.append(a)
    a = res[-1]
    f = (a[1:6] - a[0:5]) * 0.01   # 100 MHz -> us
    b = (a[11:16] - a[10:15]) * 0.01
    print(f"F={F} fwd phases us: load+erf {f[0]:.2f} | stats {f[1]:.2f} | standardise->LDS {f[2]:.2f} | MFMA {f[3]:.2f} | combine+slab {f[4]:.2f} | total {sum(f):.2f}")
    print(f"        bwd phases us: cols+load+erf+LDS {b[0]:.2f} | MFMA {b[1]:.2f} | proj {b[2]:.2f} | assemble {b[3]:.2f} | copy-out {b[4]:.2f} | total {sum(b):.2f}")
