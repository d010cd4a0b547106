"""N2: HBM bytes of one ResNet-20 stage-1 site / convolution pair with the activation stored as fp32 vs as its
int16 level index.  Launch order (each x3): site fwd fp32-out | site fwd index-out | conv fwd fp32-in | conv fwd index-in |
filter gradient fp32-x | filter gradient index-x | site bwd mask-from-y | site bwd mask-from-index.  Run under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/n2_bytes.py`; tools/make_profiles_n2.sh tabulates."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from alignq_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
st, p = L.stream_ptr(), L.ptr
B, C, H, k = 128, 16, 32, 8
HW, F = H * H, C * H * H
cl = torch.channels_last
torch.manual_seed(0)
z = (torch.randn(B, C, H, H, device=dev) * 1.5).contiguous(memory_format=cl)
g = torch.randn_like(z) * 0.01
gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
ab, save = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
wsb = torch.empty(lib.alignq_bn_nhwc_ws_bytes(C), dtype=torch.uint8, device=dev)
lib.alignq_bn_partial_stats_nhwc(p(z), B, C, HW, p(wsb), st)
y = torch.empty_like(z)
bins = torch.empty_strided(z.shape, z.stride(), dtype=torch.int16, device=dev)
stats = torch.empty(4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
dx = torch.empty_like(z)
part = torch.empty(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
wq = (torch.round(torch.tanh(torch.randn(C, C, 3, 3)) * 255) / 255).to(dev).contiguous(memory_format=cl)
out, dw = torch.empty_like(z), torch.empty_like(wq)
wsw = torch.empty(lib.alignq_conv3x3_wgrad_ws_bytes(C), dtype=torch.uint8, device=dev)


def site_fwd(xq, b):
    L.check(lib.alignq_site_partials_bn(p(z), p(wsb), p(gam), p(bet), None, None, None, 0.1, 1e-5, p(ab), p(save), C, HW, B, F, k, 2.0,
                                        0.0, 1, None, 1, 0, p(xq), p(b), p(stats), p(ws), st), "site fwd")


def site_bwd(yy, b):
    L.check(lib.alignq_site_bwd_apply_bn(p(g), p(S), p(z), p(ab), p(save), C, HW, 1, p(yy), p(b), 2 if b is not None else 0, None,
                                         p(stats), B, F, 2.0, 0.0, p(dx), p(part), st), "site bwd")


steps = [lambda: site_fwd(y, None), lambda: site_fwd(None, bins),
         lambda: L.check(lib.alignq_conv3x3_nhwc(p(y), p(wq), p(out), B, H, H, C, 8, 0, None, None, None, 0, 0, st), "conv"),
         lambda: L.check(lib.alignq_conv3x3_nhwc(None, p(wq), p(out), B, H, H, C, 8, 0, None, None, p(bins), 2, k, st), "conv"),
         lambda: L.check(lib.alignq_conv3x3_nhwc_wgrad(p(y), p(g), p(dw), p(wsw), B, H, H, C, None, None, 0, 0, st), "wgrad"),
         lambda: L.check(lib.alignq_conv3x3_nhwc_wgrad(None, p(g), p(dw), p(wsw), B, H, H, C, None, p(bins), 2, k, st), "wgrad"),
         lambda: site_bwd(y, None), lambda: site_bwd(None, bins)]
site_fwd(y, bins)                      # both outputs exist before anything is measured
torch.cuda.synchronize()
for f in steps:
    for _ in range(3):
        f()
    torch.cuda.synchronize()
print("n2_bytes done: elements", B * F)
