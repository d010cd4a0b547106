"""Launch time of the 3x3 convolution kernels against the batch size (workgroups per CU): shows whether the workgroups of a
launch are co-resident (flat) or run in rounds (linear)."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load()
dev = torch.device('cuda:0')
cl = torch.channels_last
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for C, Wd in ((16, 32), (32, 16), (64, 8)):
    for B in (32, 64, 128, 256, 512):
        x = [torch.randn(B, C, Wd, Wd, device=dev).contiguous(memory_format=cl) for _ in range(4)]
        w = (torch.randint(-255, 256, (C, C, 3, 3), device=dev).float() / 255).contiguous(memory_format=cl)
        y = [torch.empty_like(x[0]) for _ in range(4)]
        n_parts = lib.alignq_conv3x3_bn_parts(B, Wd, Wd, C)
        part = torch.empty(C, n_parts, 2, device=dev)
        st = L.stream_ptr(); p = L.ptr
        i = [0]
        def fwd():
            k = i[0] % 4; i[0] += 1
            L.check(lib.alignq_conv3x3_nhwc(p(x[k]), p(w), p(y[k]), B, Wd, Wd, C, 8, 0, None, p(part), None, 0, 0, st), "fwd")
        ws = torch.empty(lib.alignq_conv3x3_wgrad_ws_bytes(C) // 4, device=dev)
        dx = [torch.empty_like(x[0]) for _ in range(4)]
        ns = ctypes.c_int(0)
        def bwd():
            k = i[0] % 4; i[0] += 1
            L.check(lib.alignq_conv3x3_nhwc_bwd(p(x[k]), p(y[k]), p(w), p(dx[k]), p(ws), B, Wd, Wd, C, 8, ctypes.byref(ns), None,
                                                None, None, None, None, None, None, None, None, 0, 0, st), "bwd")
        tf, tb = timeit(fwd), timeit(bwd)
        print(f"C={C:2d} B={B:3d} workgroups fwd {n_parts:5d}  fwd {tf:6.2f} us   bwd {tb:6.2f} us")
