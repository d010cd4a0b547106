"""Is configuration 5's iteration reproducible run to run?  Two fresh models from the same initial state, 2 eager iterations each, on
the same inputs: which parameters differ bit for bit, and by how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alignq_amd.quantization  # noqa: F401
from alignq_amd import config
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = config.args.eval_batch_size = 28
from alignq_amd.resnet_office import resnet50_dann
from alignq_amd.train_step import OfficeTrainStep
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
xs = torch.randn(28, 3, 224, 224, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
xt = torch.randn(28, 3, 224, 224, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
ys = torch.randint(0, 31, (28,), generator=g).to(dev)
res = []
for run in range(2):
    torch.manual_seed(0)
    net = resnet50_dann(8, 8).to(dev).train()
    step = OfficeTrainStep(net, lr=4e-5, channels_last=True)
    for _ in range(2):
        out = step(xs, ys, xt)
    torch.cuda.synchronize()
    res.append({n: p.detach().clone() for n, p in net.named_parameters()})
    del step, net
bad = [(n, float((res[0][n] - res[1][n]).abs().max())) for n in res[0] if not torch.equal(res[0][n], res[1][n])]
print("parameters:", len(res[0]), "differing:", len(bad))
for n, d in bad[:12]:
    print("  ", n, d)
