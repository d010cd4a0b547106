"""Per-workgroup entry/exit times of the B=128 site kernels (diagnostic build, see tools/README.md).
Shows the dispatch ramp, per-block residency and the kernel's span as seen from inside the kernel."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
L.SO_PATH = 'tools/lib/libalignq_stamps.so'
lib = L.load()
lib.alignq_debug_read_stamps.argtypes = [ctypes.c_void_p]
lib.alignq_debug_read_block_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
B, k = 128, 8
NHWC = int(os.environ.get('NHWC', '0'))
p = L.ptr


def blocks(kern, nblk):
    buf = (ctypes.c_ulonglong * (2 * 2 * 2048))()
    lib.alignq_debug_read_block_stamps(buf)
    a = np.array(buf, dtype=np.int64).reshape(2, 2, 2048)[kern, :, :nblk] * 0.01   # 100 MHz -> us
    t0 = a[0].min()
    ent, ext = a[0] - t0, a[1] - t0
    return ent, ext


for (C, HW) in ((16, 1024), (32, 256), (64, 64)):
    F = C * HW
    z = torch.randn(B, C, HW, device=dev); g = torch.randn(B, F, device=dev) * 0.01
    xq, dx = torch.empty(B, F, device=dev), torch.empty(B, F, device=dev)
    stats = torch.empty(4, F, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    ws_bn = torch.empty(lib.alignq_bn_nhwc_ws_bytes(C) if NHWC else lib.alignq_bn_ws_bytes(C), dtype=torch.uint8, device=dev)
    ab, save = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    part = torch.empty(lib.alignq_site_bn_part_bytes(F, NHWC), dtype=torch.uint8, device=dev)
    S = torch.rand(B, B, device=dev) * 1e-6
    st = L.stream_ptr()
    for it in range(4):
        (lib.alignq_bn_partial_stats_nhwc if NHWC else lib.alignq_bn_partial_stats)(p(z), B, C, HW, p(ws_bn), st)
        lib.alignq_site_partials_bn(p(z), p(ws_bn), p(gam), p(bet), None, None, None, 0.1, 1e-5, p(ab), p(save), C, HW, B, F,
                                    k, 2.0, 0.0, 1, None, NHWC, 0, p(xq), None, p(stats), p(ws), st)
        torch.cuda.synchronize()
        lib.alignq_site_bwd_apply_bn(p(g), p(S), p(z), p(ab), p(save), C, HW, NHWC, p(xq), None, 0, None, p(stats), B, F, 2.0, 0.0, p(dx),
                                     p(part), st)
        torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.alignq_debug_read_stamps(buf)
    a = np.array(buf[:16], dtype=np.int64)
    bb = (a[11:16] - a[10:15]) * 0.01
    ff = (a[1:6] - a[0:5]) * 0.01
    blkbuf = (ctypes.c_ulonglong * (2 * 2 * 2048))()
    lib.alignq_debug_read_block_stamps(blkbuf)
    e0 = np.array(blkbuf, dtype=np.int64).reshape(2, 2, 2048)
    print(f"F={F} block 0 fwd: entry->first stamp {(a[0] - e0[0, 0, 0]) * 0.01:.2f} | load+erf {ff[0]:.2f} | stats {ff[1]:.2f} | "
          f"standardise->LDS {ff[2]:.2f} | MFMA {ff[3]:.2f} | combine+slab {ff[4]:.2f}")
    print(f"F={F} block 0 bwd: entry->tile loop {(a[10] - e0[1, 0, 0]) * 0.01:.2f} | load+erf+LDS(+S frags) {bb[0]:.2f} | MFMA {bb[1]:.2f} | "
          f"proj {bb[2]:.2f} | assemble {bb[3]:.2f} | copy-out {bb[4]:.2f} | last stamp->exit {(e0[1, 1, 0] - a[15]) * 0.01:.2f}")
    nf = min((F + (64 if F >= 16384 else 32 if F >= 8192 else 16) - 1) // (64 if F >= 16384 else 32 if F >= 8192 else 16), 256)
    tfb = 64 if F >= 16384 else 32
    nb = (F + tfb - 1) // tfb
    for name, kern, n in (("fwd", 0, nf), ("bwd", 1, nb)):
        ent, ext = blocks(kern, n)
        d = ext - ent
        print(f"F={F} {name}: blocks {n} | entry ramp p50 {np.median(ent):.2f} max {ent.max():.2f} us | residency min {d.min():.2f} "
              f"p50 {np.median(d):.2f} max {d.max():.2f} us | span (first entry -> last exit) {ext.max():.2f} us")
