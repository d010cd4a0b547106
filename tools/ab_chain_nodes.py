"""A/B on one box (round 6): the captured CIFAR step with / without the twin launches, the head role of the closing reduction launch and
the one-launch weight quantiser (ALIGNQ_SO of an older build for the latter).  python3 tools/ab_chain_nodes.py [20|56] [bits]"""
import contextlib, sys, time, torch
sys.path.insert(0, '.')
from alignq_amd import config, fused, resnet as R
from alignq_amd.train_step import TrainStep
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = bits; config.args.train_batch_size = 128
g = torch.Generator().manual_seed(13)
x = torch.randn(128, 3, 32, 32, generator=g).to(dev); y = torch.randint(0, 10, (128,), generator=g).to(dev)
real_twin = R.twin_sites
def run(twin, head, steps=300):
    R.twin_sites = real_twin if twin else contextlib.nullcontext
    fused._HEAD_ROLE = head
    torch.manual_seed(7)
    m = (R.resnet20_quant if depth == 20 else R.resnet56_quant)(bits, bits).to(dev).train()
    s = TrainStep(m, channels_last=True, qconv=True, fuse_bn=True)
    s.capture(x, y, warmup=3)
    for _ in range(20): s(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(2):
    for twin, head in ((True, True), (False, True), (True, False), (False, False)):
        print(f"resnet{depth} {bits} bit: twin {int(twin)} head-role {int(head)}: {run(twin, head):.4f} ms/step", flush=True)
