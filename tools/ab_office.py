"""A/B of Python-level variants of config 5's captured step ON ONE BOX (box-to-box spread is ~2 %, more than most single changes):
   python3 tools/ab_office.py base no_site1_batch ...
Each arm builds resnet50_dann(8, 8) + OfficeTrainStep afresh, captures, replays REPS x STEPS steps and prints the median ms/step.
Arms are monkeypatches applied here, not switches of the product path."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ARMS = {
    "base": lambda: None,
    "no_site1_batch": lambda: setattr(__import__("alignq_amd.fused", fromlist=["x"]), "active_site1", lambda: None),
    "no_rmask": lambda: setattr(__import__("alignq_amd.fused", fromlist=["x"]), "_S1_RMASK", False),
}


def run(arm, steps=20, reps=5):
    import importlib
    import alignq_amd.quantization  # noqa: F401
    from alignq_amd import config, fused
    importlib.reload(fused) if False else None
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = config.args.eval_batch_size = 28
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = resnet50_dann(8, 8).to(dev).train()
    step = OfficeTrainStep(net, lr=0.004, channels_last=True)
    xs = torch.randn(28, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    xt = torch.randn(28, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    ys = torch.randint(0, 31, (28,), device=dev)
    step.capture(xs, ys, xt, warmup=2)
    for _ in range(3):
        step(xs, ys, xt)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step(xs, ys, xt)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / steps)
    del step, net
    torch.cuda.empty_cache()
    return statistics.median(ms), min(ms)


if __name__ == "__main__":
    from alignq_amd import fused
    saved = {k: getattr(fused, k) for k in ("active_site1", "_S1_RMASK")}
    for arm in sys.argv[1:] or list(ARMS):
        for k, v in saved.items():
            setattr(fused, k, v)
        ARMS[arm]()
        med, mn = run(arm)
        print(f"{arm:20s} median {med:.3f} ms/step   min {mn:.3f}", flush=True)
