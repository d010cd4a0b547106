"""Diagnostic: every Conv2d_Q of the full-size config-5 forward, the GEMM kernel's output against F.conv2d (fp64) on the SAME
input (teacher-forced), with the level tag it was given."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from det_init import det_init_
from alignq_amd import config, ops, fused
from alignq_amd.resnet_office import resnet50_dann
from alignq_amd.train_step import OfficeTrainStep
dev = torch.device("cuda:0")
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = config.args.eval_batch_size = 28
g = torch.Generator().manual_seed(11)
xs = torch.randn(28, 3, 224, 224, generator=g).to(dev); xt = torch.randn(28, 3, 224, 224, generator=g).to(dev)
ys = torch.randint(0, 31, (28,), generator=g).to(dev)
net = det_init_(resnet50_dann(8, 8)).to(dev).train()
step = OfficeTrainStep(net, lr=0.004, channels_last=True)
real_apply = ops.QConvGemmFn.apply
names = {id(m): n for n, m in net.named_modules()}
rows = []
def spy_stats(x, w, w_bit, stride, x_levels=0.0, groups=1, bins=None):
    y = real_with(x, w, w_bit, stride, x_levels, groups, bins)
    with torch.no_grad():
        ks = w.shape[2]
        ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=(ks - 1) // 2)
        err = float((y.double() - ref).abs().max()); sc = float(ref.abs().max())
        onlev = float((x * x_levels - torch.round(x * x_levels)).abs().max()) if x_levels else -1.0
        part = getattr(y, "_alignq_bnq_part", None)
        perr = -1.0
        if part is not None:
            p, npart, grp = part
            yg = y.permute(0, 2, 3, 1).reshape(grp, -1, y.shape[1]).double()
            perr = float(((p.sum(1)[..., 0] - yg.sum(1)).abs() / (yg.abs().sum(1) + 1e-9)).max())
        rows.append((tuple(x.shape), tuple(w.shape), stride, x_levels, err, sc, onlev, perr))
    return y
real_with = ops.QConvGemmFn.apply_with_stats
ops.QConvGemmFn.apply_with_stats = staticmethod(spy_stats)
step._forward_backward(xs, ys, xt)
torch.cuda.synchronize()
for r in rows:
    print("x", r[0], "w", r[1], "s", r[2], "lev", r[3], "maxerr %.3e of %.3e" % (r[4], r[5]), "offgrid %.2e" % r[6], "bnpart relerr %.2e" % r[7])
