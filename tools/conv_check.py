"""Developer check: alignq_conv3x3_nhwc vs torch (MIOpen) conv2d / its data gradient, and timings."""
import sys, torch
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); p = L.ptr
torch.backends.cudnn.benchmark = True
def t_call(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, C, H) in ((128, 16, 32), (128, 32, 16), (128, 64, 8), (3, 16, 32), (5, 64, 8)):
    torch.manual_seed(C + B)
    x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.round(torch.tanh(torch.randn(C, C, 3, 3, device=dev)).cpu() * 255) / 255).to(dev).contiguous(memory_format=torch.channels_last)   # 8-bit quantised values b/255
    y = torch.empty_like(x)
    st = L.stream_ptr()
    rc = lib.alignq_conv3x3_nhwc(p(x), p(w), p(y), B, H, H, C, 8, 0, None, None, None, 0, 0, st); assert rc == 0, rc
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1).float()
    ref32 = torch.nn.functional.conv2d(x, w, padding=1)
    err = (y - ref).abs().max().item(); err32 = (ref32 - ref).abs().max().item()
    dy = torch.randn_like(x)
    dx = torch.empty_like(x)
    rc = lib.alignq_conv3x3_nhwc(p(dy), p(w), p(dx), B, H, H, C, 8, 1, None, None, None, 0, 0, st); assert rc == 0, rc
    dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1).float()
    derr = (dx - dref).abs().max().item()
    dw = torch.empty_like(w)
    ws = torch.empty(lib.alignq_conv3x3_wgrad_ws_bytes(C), dtype=torch.uint8, device=dev)
    rc = lib.alignq_conv3x3_nhwc_wgrad(p(x), p(dy), p(dw), p(ws), B, H, H, C, None, None, 0, 0, st); assert rc == 0, rc
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    w32 = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
    werr, werr32 = (dw - wref.float()).abs().max().item(), (w32 - wref.float()).abs().max().item()
    line = f"B={B} C={C} H={H}: wgrad err {werr:.2e} (MIOpen {werr32:.2e}, |dW|max {wref.abs().max():.0f})"
    if B == 128:
        t5 = t_call(lambda: lib.alignq_conv3x3_nhwc_wgrad(p(x), p(dy), p(dw), p(ws), B, H, H, C, None, None, 0, 0, st))
        t6 = t_call(lambda: torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1))
        line += f" wgrad {t5:.1f} us vs MIOpen {t6:.1f}"
    print(line)
    line = f"B={B} C={C} H={H}: fwd max err {err:.2e} (MIOpen fp32 {err32:.2e}, |y|max {ref.abs().max():.1f}) dgrad err {derr:.2e}"
    if B == 128:
        t1 = t_call(lambda: lib.alignq_conv3x3_nhwc(p(x), p(w), p(y), B, H, H, C, 8, 0, None, None, None, 0, 0, st))
        t2 = t_call(lambda: torch.nn.functional.conv2d(x, w, padding=1))
        t3 = t_call(lambda: lib.alignq_conv3x3_nhwc(p(dy), p(w), p(dx), B, H, H, C, 8, 1, None, None, None, 0, 0, st))
        t4 = t_call(lambda: torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1))
        line += f" | fwd {t1:.1f} us vs MIOpen {t2:.1f} | dgrad {t3:.1f} vs {t4:.1f}"
    print(line)
