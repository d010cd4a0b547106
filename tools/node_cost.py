"""Cost of one dependent kernel node in a captured HIP graph vs eager stream launches (tiny kernels: pure hand-over)."""
import time, torch
dev = torch.device('cuda:0')
x = torch.zeros(64, device=dev)
big = torch.zeros(1 << 22, device=dev)      # 16 MB: a kernel that dirties L2
def chain(n, t):
    for _ in range(n):
        t.add_(1.0)
for name, t in (("tiny (64 floats)", x), ("16 MB tensor", big)):
    for n in (50, 200):
        chain(n, t); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(n, t)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / reps / n * 1e6
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record(); chain(n, t); e.record(); torch.cuda.synchronize()
        print(f"{name:18s} n={n:4d}: graph {tg:6.2f} us per node, eager {s.elapsed_time(e) / n * 1e3:6.2f} us per launch")
