"""Diagnostic: OfficeTrainStep on the G10 tiny DANN vs the fixture; prints the observed differences (calibrates the bars of
tests/test_gpu_round2.py::test_office_tiny_dann_two_iterations_vs_reference)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from det_init import det_init_, sample  # noqa: E402
from alignq_amd import config  # noqa: E402
from alignq_amd.resnet_office import DANN, Bottleneck, ResNet  # noqa: E402
from alignq_amd.train_step import OfficeTrainStep  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g10_office_tiny_dann.npz")))
dev = torch.device("cuda:0")
config.args.bitW = config.args.abitW = 4
config.args.train_batch_size = config.args.eval_batch_size = 6
for cl, fr in ((False, False), (True, False), (False, True), (True, True)):
    torch.manual_seed(0)
    net = DANN(lambda w, a, s: ResNet(w, a, s, Bottleneck, [1, 1, 1, 1], width_per_group=8), 4, 4, "aligned")
    det_init_(net)
    net = net.to(dev).train()
    step = OfficeTrainStep(net, lr=0.004, alpha=0.5, channels_last=cl, fuse_relu=fr)
    named = list(net.named_parameters())
    print(f"== channels_last={cl} fuse_relu={fr}")
    for it, epoch in enumerate((1, 2)):
        step.new_epoch(epoch, 10, 0.004)
        f = lambda a: torch.from_numpy(a).to(dev)
        cls_s, loss, tl = step(f(g["xs"][it]), f(g["ys"][it]), f(g["xt"][it]))
        torch.cuda.synchronize()
        print(f" it{it}: cls max|d| {np.abs(cls_s.detach().cpu().numpy() - g[f'cls_s_{it}']).max():.4f}  "
              f"tl {float(tl):.5f} vs {float(g[f'tl_s_{it}']) + float(g[f'tl_t_{it}']):.5f}  loss {float(loss):.4f} vs {float(g[f'loss_{it}']):.4f}")
        for bi, b in enumerate(step.blocks):
            D = b.admm0.D.detach().cpu().numpy()
            print(f"   block{bi}: |D-Dtgt| {np.abs(D - g[f'D_{it}_{bi}']).max():.2e}  |D-Dsrc| {np.abs(D - g[f'Dsrc_{it}_{bi}']).max():.2e}")
        worst = []
        for j, (n, p) in enumerate(named):
            d = np.abs(sample(p).cpu().numpy() - g[f"after_{it}/{j}"]).max()
            upd = np.abs(g[f"after_{it}/{j}"] - (g[f"after_{it-1}/{j}"] if it else g[f"after_{it}/{j}"])).max()
            worst.append((d, n))
        worst.sort(reverse=True)
        print("   worst params:", [(f"{d:.2e}", n) for d, n in worst[:5]])
