#!/usr/bin/env python3
"""CPU baseline (kind "port": the eager-torch restatement of the reference, oracle/torch_ref.py) for BASELINE.json configs 4 and
5, timed on THIS box's host cores next to the GPU numbers of bench.py's other_configs (every config's GPU
number gets a same-box CPU number beside it).  3 timed steps after 1 warm-up, a short sweep of thread counts, the fastest reported.
    python tools/cpu_baseline_configs.py > profiles/r05_cpu_baseline_configs.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as R  # noqa: E402


def timed(make_step, run, threads, steps=3):
    best = None
    tried = {}
    for th in threads:
        torch.set_num_threads(th)
        torch.manual_seed(0)
        step = make_step()
        run(step)                                    # warm-up
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            run(step)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        tried[str(th)] = med
        if best is None or med < best[1]:
            best = (th, med)
        print(f"  threads {th}: {med:.2f} s/step", file=sys.stderr, flush=True)
    return best, tried


def main():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = sorted({min(avail, t) for t in (16, 32, 64)})
    out = {"host_cores": os.cpu_count(), "usable_cores": avail, "torch": torch.__version__, "kind": "port"}
    gen = torch.Generator().manual_seed(0)
    # config 4 (index 3): ResNet-56 4W/4A CDF+ADMM, 128 images per GPU
    cfg = R.Config(tree="admm", bitW=4, abitW=4, train_batch_size=128)
    x = torch.randn(128, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (128,), generator=gen)
    (th, med), tried = timed(lambda: R.TrainStep(R.resnet56(cfg).train(), cfg), lambda s: s(x, y), threads)
    out["resnet56_4w4a_b128"] = {"images_per_sec": 128 / med, "s_per_step": med, "cores": th, "s_per_step_by_threads": tried,
                                 "sample": "3 full training steps (median) after 1 warm-up per thread count"}
    # config 5 (index 4): ResNet-50-DANN 8W/8A, 28 source + 28 target images per GPU
    cfg5 = R.Config(tree="office", bitW=8, abitW=8, train_batch_size=28)
    xs, xt = torch.randn(28, 3, 224, 224, generator=gen), torch.randn(28, 3, 224, 224, generator=gen)
    ys = torch.randint(0, 31, (28,), generator=gen)
    (th, med), tried = timed(lambda: R.OfficeTrainStep(R.OfficeDANN(cfg5, 8, 8).train(), cfg5, lr=0.004, alpha=0.5),
                             lambda s: s(xs, ys, xt), threads)
    out["resnet50_dann_8w8a_b28"] = {"images_per_sec": 56 / med, "s_per_step": med, "cores": th, "s_per_step_by_threads": tried,
                                     "sample": "3 full DANN iterations (source + target pass, median) after 1 warm-up per thread count"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
