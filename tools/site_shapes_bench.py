import sys, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
from bench import time_call
lib = L.load(); dev = torch.device('cuda:0'); st = L.stream_ptr(); p = L.ptr
for B, F in [(28, 802816), (28, 401408), (28, 200704), (28, 100352), (64, 65536), (128, 524288)]:
    x = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01
    xq, dx = torch.empty_like(x), torch.empty_like(x)
    D = torch.empty(B, B, device=dev); stats = torch.empty(4, F, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)   # fp32 S + bf16 image, zero (timing only)
    A = torch.rand(B, B, device=dev); Gm = torch.rand(B, B, device=dev); scal = torch.empty(4, device=dev)
    tp = time_call(lambda: lib.alignq_site_partials(p(x), B, F, 8, 2.0, 1e-5, p(xq), p(stats), p(ws), st), 20)
    def pair():
        lib.alignq_site_partials(p(x), B, F, 8, 2.0, 1e-5, p(xq), p(stats), p(ws), st)
        lib.alignq_site_reduce_loss(p(ws), B, F, p(D), p(A), p(Gm), B, 0.2, 0.3, p(scal), st)
    tr = time_call(pair, 20) - tp
    tb = time_call(lambda: lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, 1e-5, p(dx), st), 20)
    tq = time_call(lambda: lib.alignq_act_quant_fwd(p(x), p(xq), None, B * F, 8, 2.0, 0, st), 20)
    n = B * F
    print(f"B={B:4d} F={F:7d} ({n*4/1e6:6.1f} MB)  partials {tp*1e6:7.1f} us {8*n/tp/1e9:6.0f} GB/s | reduce+loss {tr*1e6:6.1f} us | bwd {tb*1e6:7.1f} us {12*n/tb/1e9:6.0f} GB/s | plain quant fwd {tq*1e6:7.1f} us {8*n/tq/1e9:6.0f} GB/s")
