"""plain CDF quantiser on cold and on re-used operands (HIP events), 2^26 elements"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = L.stream_ptr(); p = L.ptr
n = 1 << 26
R = 4
xs = [torch.randn(n, device=dev) for _ in range(R)]; gs = [torch.randn(n, device=dev) for _ in range(R)]; ys = [torch.empty(n, device=dev) for _ in range(R)]
def t(fn, reps=24):
    for i in range(4): fn(i)
    e = [torch.cuda.Event(True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn(i); e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)]) * 1e3)
fc = t(lambda i: lib.alignq_act_quant_fwd(p(xs[i % R]), p(ys[i % R]), None, n, 8, 2.0, 0, st))
bc = t(lambda i: lib.alignq_act_quant_bwd(p(gs[i % R]), p(xs[i % R]), p(ys[i % R]), n, 2.0, st))
fw = t(lambda i: lib.alignq_act_quant_fwd(p(xs[0]), p(ys[0]), None, n, 8, 2.0, 0, st))
bw = t(lambda i: lib.alignq_act_quant_bwd(p(gs[0]), p(xs[0]), p(ys[0]), n, 2.0, st))
cc = t(lambda i: ys[i % R].copy_(xs[i % R]))
print(f"cold fwd {fc:.1f} us ({8*n/fc/8e6:.3f})  bwd {bc:.1f} us ({12*n/bc/8e6:.3f})  | re-used fwd {fw:.1f} ({8*n/fw/8e6:.3f}) bwd {bw:.1f} ({12*n/bw/8e6:.3f}) | torch copy cold {cc:.1f}")
