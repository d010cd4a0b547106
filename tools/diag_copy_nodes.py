"""Which device-to-device copies does one eager Office iteration issue, and from where?  (torch.profiler with python stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alignq_amd.quantization  # noqa: F401
from alignq_amd import config
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
for _ in range(2):
    step(xs, ys, xt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(xs, ys, xt)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name == "aten::copy_":
        st = [f for f in (e.stack or []) if "torch/" not in f and "<built-in" not in f][:4]
        shp = getattr(e, "input_shapes", None)
        cnt[tuple(st)] += 1
for st, c in cnt.most_common(12):
    print(c, " <- ".join(st))
