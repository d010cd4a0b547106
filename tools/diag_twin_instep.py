"""Diagnostic (round 6): the eager CIFAR step with the backward twin launches against the same step with two single launches per pair, eight
times: number of parameter tensors that differ after one step (0 expected; 19-25 in 2 of 8 runs while the F = 8192 pair ran two workgroups
per CU)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import config, fused, _lib as L
from alignq_amd.resnet import resnet20_quant
from alignq_amd.train_step import TrainStep
dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = 8; config.args.train_batch_size = 128
g = torch.Generator().manual_seed(13)
x = torch.randn(128, 3, 32, 32, generator=g).to(dev); y = torch.randint(0, 10, (128,), generator=g).to(dev)
real_add = fused._BwdTwin.add
def add_single(self, args, keep, st):
    self._single(args, st)
def mk(sync_before, sync_after):
    def f(self, args, keep, st):
        second = self.pending is not None
        if second and sync_before: torch.cuda.synchronize()
        r = real_add(self, args, keep, st)
        if second and sync_after: torch.cuda.synchronize()
        return r
    return f
def run(fn):
    fused._BwdTwin.add = fn
    torch.manual_seed(7)
    m = resnet20_quant(8, 8).to(dev).train()
    s = TrainStep(m, channels_last=True)
    for it in range(1):
        s(x, y)
    torch.cuda.synchronize()
    return {n: p.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
def add_delayed(self, args, keep, st):
    if self.pending is None:
        self.pending = (args, keep, st); fused._bwd_twins.open.append(self); return
    pa, _pk, pst = self.pending; self.pending = None
    if self in fused._bwd_twins.open: fused._bwd_twins.open.remove(self)
    self._single(pa, pst); self._single(args, st)
ref = run(add_single)
for name, fn in [("twin", real_add)] * 8:
    v = run(fn)
    bad = [n for n in v if not np.array_equal(v[n], ref[n])]
    print(name, "differs in", len(bad), "tensors", [b for b in bad if "admm" not in b][:40])
