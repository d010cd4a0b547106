import sys, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = L.stream_ptr(); p = L.ptr
B, F = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(B, F, device=dev); g = torch.randn(B, F, device=dev) * 0.01
xq, dx = torch.empty_like(x), torch.empty_like(x)
stats = torch.empty(4, F, device=dev)
ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)   # fp32 S + bf16 image, zero (timing only)
for _ in range(5):
    lib.alignq_site_partials(p(x), B, F, 8, 2.0, 1e-5, p(xq), p(stats), p(ws), st)
    lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, 1e-5, p(dx), st)
    lib.alignq_act_quant_fwd(p(x), p(xq), None, B * F, 8, 2.0, 0, st)
torch.cuda.synchronize()
