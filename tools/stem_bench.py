"""The Office stem (Conv2d_Q(3, 64, 7, 2, 3)) at B = 56, 224 x 224: alignq_qconv_stem7_fwd / _wgrad against MIOpen (HIP events, back to back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alignq_amd import _lib as L, ops

dev = torch.device("cuda:0")
CL = torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 56
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=CL)
w = (torch.round(torch.tanh(torch.randn(64, 3, 7, 7, device=dev)) * 255) / 255).contiguous(memory_format=CL).requires_grad_(True)
gy = (torch.randn(B, 64, 112, 112, device=dev) * 1e-3).contiguous(memory_format=CL)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


bins = ops.pack_filter_bins([w.detach()], 8)[0]
print("stem7 fwd (no stats)  %.1f us" % timeit(lambda: ops.QConvStem7Fn.apply(x, w, 8, 1, False, bins)))
print("stem7 fwd (+ stats)   %.1f us" % timeit(lambda: ops.QConvStem7Fn.apply_with_stats(x, w, 8, 2, bins)))
print("MIOpen fwd            %.1f us" % timeit(lambda: torch.nn.functional.conv2d(x, w.detach(), stride=2, padding=3)))
y = ops.QConvStem7Fn.apply(x, w, 8, 1, False, bins)
print("stem7 wgrad (+reduce) %.1f us" % timeit(lambda: torch.autograd.grad(y, w, gy, retain_graph=True)))
print("MIOpen wrw            %.1f us" % timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w.detach(), None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))))
