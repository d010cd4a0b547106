"""Child process of tests/test_gpu_round2.py::test_capture_after_an_eager_exact_global_phase: the sequence that ended in a
segmentation fault inside hipStreamEndCapture in round 4 (attach(global_corr=True) -> eager steps -> detach -> capture of the same
step object), run apart from the test process so that a crash or a hang costs one child (hard timeout in the parent).
Prints one line `OK <ms per replay>` on success."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ["MASTER_ADDR"] = "127.0.0.1"                      # (not the parent's rendezvous: it has a group of its own)
    os.environ["MASTER_PORT"] = sys.argv[1] if len(sys.argv) > 1 else "29561"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from alignq_amd import config, dp
    from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
    from alignq_amd.train_step import TrainStep
    config.args.bitW = config.args.abitW = 8
    config.args.train_batch_size = 128
    torch.manual_seed(5)
    x = torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)
    net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
    step = TrainStep(net, lr=0.01, channels_last=True)
    dp.attach(step, force=True, global_corr=True)
    before = [p.detach().clone() for p in net.parameters()]
    try:
        step.capture(x, y, warmup=1)
        raise SystemExit("capture() took the exact-global mode")
    except RuntimeError as e:
        assert "eager-only" in str(e), e
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters())), "a refused capture must not touch the model"
    out = step(x, y)                                   # eager exact-global iteration; `out` stays referenced on purpose
    assert torch.isfinite(out[1]).item() and torch.isfinite(out[2]).item()
    dp.detach(step)
    out2 = step(x, y)                                  # per-rank sites again, eagerly; also kept
    assert torch.isfinite(out2[1]).item()
    # a graph of an earlier iteration that IS still referenced: refused with its reason, before the warm-up, instead of a crash
    held = step._forward_backward(x, y)                # (internal: returns the iteration's tensors with their history)
    before = [p.detach().clone() for p in net.parameters()]
    try:
        step.capture(x, y, warmup=1)
        raise SystemExit("capture() did not see the retained graph")
    except RuntimeError as e:
        assert "still referenced by the autograd graph" in str(e), e
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    del held
    step.capture(x, y, warmup=1)                       # round 4: segmentation fault inside hipStreamEndCapture
    sx, sy = step.static_inputs()
    for _ in range(3):
        res = step(sx, sy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        res = step(sx, sy)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    assert torch.isfinite(res[1]).item() and res[1].grad_fn is None
    dist.destroy_process_group()
    print(f"OK {ms:.3f}")


if __name__ == "__main__":
    main()
