"""In-kernel phase breakdown of the B=128 site kernels (workgroup 0's wall-clock stamps; diagnostic build:
`make -C alignq_amd/csrc stamps`).  Times the entry points the training step uses: channels-last BN fold finalised from the
convolution epilogue's float partials (conv_parts), ReLU, with and without the residual operand."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
L.SO_PATH = 'tools/lib/libalignq_stamps.so'
lib = L.load()
lib.alignq_debug_read_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
B, k = 128, 8
for F, C, n_parts in ((16384, 16, 512), (8192, 32, 256), (4096, 64, 256)):
    HW = F // C
    z = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01; res = torch.randn(B, F, device=dev)
    y, dx, dres = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
    stats = torch.empty(4, F, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
    n = B * HW
    part = torch.stack([torch.randn(C, n_parts, device=dev) * 3 + 0.1 * n / n_parts * 0, (torch.rand(C, n_parts, device=dev) + 0.5) * n / n_parts], 2).contiguous()
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    ab, save = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    bpart = torch.empty(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
    st = L.stream_ptr(); p = L.ptr
    for with_res in (False, True):
        res_f, res_b = [], []
        for it in range(6):
            L.check(lib.alignq_site_partials_bn(p(z), p(part), p(gam), p(bet), None, None, None, 0.1, 1e-5, p(ab), p(save), C, HW, B, F, k,
                                                2.0, 0.0, 1, p(res) if with_res else None, 1, n_parts, p(y), None, p(stats), p(ws), st), "fwd")
            L.check(lib.alignq_site_bwd_apply_bn(p(g), p(S), p(z), p(ab), p(save), C, HW, 1, p(y), None, 0, p(dres) if with_res else None,
                                                 p(stats), B, F, 2.0, 0.0, p(dx), p(bpart), st), "bwd")
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 64)()
            lib.alignq_debug_read_stamps(buf)
            a = np.array(buf[:16], dtype=np.int64)
            res_f.append((a[1:6] - a[0:5]) * 0.01); res_b.append((a[11:16] - a[10:15]) * 0.01)   # 100 MHz -> us
        f, b = np.median(res_f[1:], 0), np.median(res_b[1:], 0)
        print(f"F={F} C={C} res={int(with_res)} fwd us: finalise+load+erf {f[0]:.2f} | stats {f[1]:.2f} | stage {f[2]:.2f} | MFMA {f[3]:.2f} | combine+slab {f[4]:.2f} | total {sum(f):.2f}")
        print(f"                     bwd us: load+erf+stage+S {b[0]:.2f} | MFMA {b[1]:.2f} | proj {b[2]:.2f} | assemble {b[3]:.2f} | copy-out {b[4]:.2f} | total {sum(b):.2f}")
