"""where the data-parallel Office step spends its extra time at world size 1: graph 1 (forward + backward + pack), the eager
all-reduces, graph 2 (unpack + optimizer steps), each timed alone (HIP events, 10 repetitions)"""
import os, sys, time, torch
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev); dist.barrier()
from alignq_amd import config, dp
from alignq_amd.resnet_office import resnet50_dann
from alignq_amd.train_step import OfficeTrainStep
torch.backends.cudnn.benchmark = True
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = config.args.eval_batch_size = 28
torch.manual_seed(0)
model = resnet50_dann(8, 8).to(dev).train()
step = OfficeTrainStep(model, lr=0.004, channels_last=True)
hook = dp.attach_office(step, force=True)
xs, xt = torch.randn(28, 3, 224, 224, device=dev), torch.randn(28, 3, 224, 224, device=dev)
ys = torch.randint(0, 31, (28,), device=dev)
step.capture(xs, ys, xt, warmup=2)
for _ in range(3): step(xs, ys, xt)
def t(fn, n=10):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("whole step %.2f ms" % t(lambda: step(xs, ys, xt)))
print("graph 1    %.2f ms" % t(lambda: step._graph.replay()))
print("reduce     %.2f ms" % t(lambda: hook.reduce()))
print("graph 2    %.2f ms" % t(lambda: step._graph2.replay()))
print("buckets:", [b.flat.numel() * 4 // 2**20 for b, _ in hook._phase], "MiB")
dist.barrier(); dist.destroy_process_group()
