import sys, torch
sys.path.insert(0, '.')
import bench
from alignq_amd import _lib as L
lib = L.load(); p = L.ptr; st = L.stream_ptr(); dev = torch.device('cuda')
for B, F in ((28, 802816), (28, 100352)):
    x = torch.randn(B, F, device=dev); xq = torch.empty_like(x); g = torch.randn(B, F, device=dev) * 0.01; dx = torch.empty_like(x)
    stats = torch.empty(4, F, device=dev); G = torch.empty(B, B, device=dev)
    ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
    S = torch.zeros(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
    n = B * F
    t_q = bench.time_call(lambda: lib.alignq_act_quant_fwd(p(x), p(xq), None, n, 8, 2.0, 0, st), 20)
    t_c = bench.time_call(lambda: lib.alignq_corr_fwd(p(x), B, F, 1e-5, p(G), p(stats), p(ws), st), 20)
    t_s = bench.time_call(lambda: lib.alignq_site_partials(p(x), B, F, 8, 2.0, 1e-5, p(xq), p(stats), p(ws), st), 20)
    t_sn = bench.time_call(lambda: lib.alignq_site_partials(p(x), B, F, 8, 2.0, 1e-5, None, p(stats), p(ws), st), 20)
    t_b = bench.time_call(lambda: lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, 1e-5, p(dx), st), 20)
    t_bc = bench.time_call(lambda: lib.alignq_corr_bwd(p(G), p(x), p(stats), B, F, 1e-5, p(dx), p(S), st), 20)
    t_qb = bench.time_call(lambda: lib.alignq_act_quant_bwd(p(g), p(x), p(dx), n, 2.0, st), 20)
    print(f"[{B},{F}] plain quant fwd {t_q*1e6:.1f} us | corr fwd (x only, + reduce) {t_c*1e6:.1f} | site partials {t_s*1e6:.1f} | site partials without x_q store {t_sn*1e6:.1f} | site bwd {t_b*1e6:.1f} | corr bwd (prep + x only) {t_bc*1e6:.1f} | plain quant bwd {t_qb*1e6:.1f}")
