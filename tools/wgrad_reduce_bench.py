"""alignq_conv3x3_wgrad_reduce_multi on ResNet-20-sized slab sets: warm (same buffers every launch) vs rotating buffer sets."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
from bench import time_call
lib = L.load(); dev = torch.device('cuda:0')
# (slabs, elements) of ResNet-20's 16 body convolutions: C=16 -> 256 x 2304, C=32 -> 256 x 9216, C=64 -> 64 x 36864 (= 110 MB)
shapes = [(256, 2304)] * 6 + [(256, 9216)] * 5 + [(64, 36864)] * 5
def make():
    ws = [torch.randn(n * e, device=dev) for n, e in shapes]
    dw = [torch.empty(e, device=dev) for _, e in shapes]
    return ws, dw
sets = [make() for _ in range(6)]
T = len(shapes)
ns = (ctypes.c_int * T)(*[n for n, _ in shapes]); ne = (ctypes.c_int * T)(*[e for _, e in shapes])
tabs = [(L.ptr_array(ws), L.ptr_array(dw)) for ws, dw in sets]
st = L.stream_ptr()
i = [0]
def warm():
    lib.alignq_conv3x3_wgrad_reduce_multi(T, tabs[0][0], tabs[0][1], ns, ne, st)
def rot():
    k = i[0] % len(tabs); i[0] += 1
    lib.alignq_conv3x3_wgrad_reduce_multi(T, tabs[k][0], tabs[k][1], ns, ne, st)
mb = sum(n * e for n, e in shapes) * 4 / 1e6
tw, tr = time_call(warm, 50), time_call(rot, 60)
print(f"{mb:.1f} MB of slabs: warm {tw*1e6:.1f} us ({mb/tw/1e6:.2f} TB/s), rotating over {len(tabs)} sets {tr*1e6:.1f} us ({mb/tr/1e6:.2f} TB/s)")
