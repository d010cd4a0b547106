"""In-kernel phases of the MULTI-TILE site forward and of the one-tile-per-workgroup backward at the roofline-sized shape
[128, 524288] (diagnostic build: `make -C alignq_amd/csrc stamps`).  Workgroup 0 overwrites its stamps every tile, so the
forward's phases are those of its LAST tile and stamp1 - stamp0 is everything before it (31 tiles + the last transform)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
L.SO_PATH = 'tools/lib/libalignq_stamps.so'
lib = L.load()
lib.alignq_debug_read_stamps.argtypes = [ctypes.c_void_p]
lib.alignq_debug_read_block_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
B, k, F = 128, 8, 524288
x = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01
xq, dx = torch.empty_like(x), torch.empty_like(x)
stats = torch.empty(4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
st = L.stream_ptr(); p = L.ptr
rf, rb, tf_, tb_ = [], [], [], []
for it in range(6):
    e0, e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    L.check(lib.alignq_site_partials(p(x), B, F, k, 2.0, 0.0, p(xq), p(stats), p(ws), st), "fwd")
    e1.record()
    L.check(lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, 0.0, p(dx), st), "bwd")
    e2.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.alignq_debug_read_stamps(buf)
    a = np.array(buf[:16], dtype=np.int64)
    rf.append((a[1:6] - a[0:5]) * 0.01); rb.append((a[11:16] - a[10:15]) * 0.01)
    tf_.append(e0.elapsed_time(e1) * 1e3); tb_.append(e1.elapsed_time(e2) * 1e3)
    blk = (ctypes.c_ulonglong * (2 * 2 * 2048))()
    lib.alignq_debug_read_block_stamps(blk)
    bb = np.array(blk[:], dtype=np.int64).reshape(2, 2, 2048)
f, b = np.median(rf[1:], 0), np.median(rb[1:], 0)
print(f"fwd launch {np.median(tf_[1:]):.1f} us; WG0: before last tile's stats {f[0]:.2f} | stats {f[1]:.2f} | stage {f[2]:.2f} | MFMA {f[3]:.2f} | combine+slab {f[4]:.2f}")
print(f"bwd launch {np.median(tb_[1:]):.1f} us; WG0 tile: load+erf+stage+S {b[0]:.2f} | MFMA {b[1]:.2f} | proj {b[2]:.2f} | assemble {b[3]:.2f} | copy-out {b[4]:.2f} | total {sum(b):.2f}")
for kern, name in ((0, "fwd"), (1, "bwd")):
    ent, ext = bb[kern, 0], bb[kern, 1]
    n = 256 if kern == 0 else 2048
    d = (ext[:n] - ent[:n]) * 0.01
    t0 = ent[:n].min()
    print(f"{name}: first {n} workgroups: residency median {np.median(d):.2f} us (min {d.min():.2f}, max {d.max():.2f}); entries spread {(ent[:n].max() - t0) * 0.01:.2f} us; last exit at {(ext[:n].max() - t0) * 0.01:.2f} us")
