import sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import config
from alignq_amd.resnet_office import DANN, Bottleneck, ResNet
from alignq_amd.train_step import OfficeTrainStep
dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = config.args.eval_batch_size = 6
def make():
    torch.manual_seed(7)
    return DANN(lambda w, a, s: ResNet(w, a, s, Bottleneck, [1, 1, 1, 1]), 8, 8, "aligned").to(dev).train()
g = torch.Generator().manual_seed(0)
xs = torch.randn(6, 3, 64, 64, generator=g).to(dev); xt = torch.randn(6, 3, 64, 64, generator=g).to(dev)
ys = torch.randint(0, 31, (6,), generator=g).to(dev)
def diff(m1, m2, tag):
    worst = max(((float((p1 - p2).abs().max()), n) for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters())))
    print(tag, 'max param diff', worst)
m1, m2 = make(), make(); s1, s2 = OfficeTrainStep(m1), OfficeTrainStep(m2)
diff(m1, m2, 'init')
for it in range(3):
    o1 = s1(xs, ys, xt); o2 = s2._iteration(xs, ys, xt, set_to_none=False)
    print(it, float(o1[1]), float(o2[1]))
    diff(m1, m2, f'eager vs eager(set_to_none=False) it{it}')
m3, m4 = make(), make(); s3, s4 = OfficeTrainStep(m3), OfficeTrainStep(m4)
for it in range(2): s3(xs, ys, xt)
s4.capture(xs, ys, xt, warmup=2)
diff(m3, m4, 'after 2 its (eager vs capture-warmup)')
s3(xs, ys, xt); s4(xs, ys, xt); torch.cuda.synchronize()
diff(m3, m4, 'after replay')
