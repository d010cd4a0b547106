#!/usr/bin/env python3
"""Probe (round 6): does an EXTERNAL event recorded inside a captured HIP graph order an eagerly enqueued consumer on another stream
behind the graph's producer, on every replay?  (What an all-reduce overlapped with a captured backward needs.)
Run under `timeout`: a wrong answer is a wrong number, but a driver that mishandles the flag could also hang."""
import sys
import torch

dev = torch.device("cuda:0")
n = 1 << 24
a = torch.zeros(n, device=dev)
b = torch.zeros(n, device=dev)
out = torch.zeros(n, device=dev)
step = torch.zeros((), device=dev)
side = torch.cuda.Stream()
try:
    ev = torch.cuda.Event(external=True)
except Exception as e:      # noqa: BLE001
    print("external events unsupported by this torch:", e); sys.exit(0)

cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    step += 1
    for _ in range(20):
        a.copy_(a * 0 + step)          # producer: a long chain that ends with a == step everywhere
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
        step += 1
        for _ in range(20):
            a.copy_(a * 0 + step)
        ev.record()                     # external: becomes an event-record node
        for _ in range(20):
            b.copy_(b * 0 + step)      # later graph work the consumer should overlap with
torch.cuda.synchronize()
ok = True
for it in range(5):
    g.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        out.copy_(a)                    # consumer: must see THIS replay's a
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = float(step)
    got = (float(out.min()), float(out.max()))
    print("replay", it, "step", want, "consumer saw", got)
    ok &= got == (want, want)
print("EXTERNAL_EVENT_OK" if ok else "EXTERNAL_EVENT_WRONG")
