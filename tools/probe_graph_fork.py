#!/usr/bin/env python3
"""Probe (round 6): what does a fork inside a captured HIP graph cost the MAIN chain?  A chain of 2 N dependent kernels (~10 us each,
launch-latency-sized like the CIFAR step's nodes) against the same chain with N side kernels forked off it (one per pair, each
waiting for its chain kernel only, all joined at the end) - the shape a split "quantise on the chain, Gram beside it" site forward
would have.  Also: the side kernels appended to the chain (no fork) for reference."""
import time
import torch

dev = torch.device("cuda:0")
N = 21
n = 1 << 21                     # 8 MB per tensor: an elementwise kernel of ~8-10 us
a = [torch.randn(n, device=dev) for _ in range(2)]
side_buf = [torch.randn(n, device=dev) for _ in range(N)]
side_out = [torch.empty(n, device=dev) for _ in range(N)]


def chain_kernel(i):
    torch.add(a[i % 2], 1.0, out=a[(i + 1) % 2])


def side_kernel(j):
    torch.mul(side_buf[j], a[0], out=side_out[j])


def build(mode):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            chain_kernel(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for j in range(N):
                chain_kernel(2 * j)
                if mode == "fork":
                    side.wait_stream(s)
                    with torch.cuda.stream(side):
                        side_kernel(j)
                elif mode == "inline":
                    side_kernel(j)
                chain_kernel(2 * j + 1)
            if mode == "fork":
                s.wait_stream(side)
    torch.cuda.synchronize()
    return g


for mode in ("chain", "inline", "fork", "chain", "fork"):
    g = build(mode)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{mode:7s}: {us:8.1f} us per replay ({2 * N} chain kernels{', ' + str(N) + ' side kernels' if mode != 'chain' else ''})")
