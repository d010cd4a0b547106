"""Which torch (non-alignq) operators launch kernels inside one TrainStep iteration: every one is a HIP-graph node of ~3-5 us
on the in-order chain.  Prints the aten ops of one eager iteration with their call stacks' innermost alignq_amd frame."""
import sys
import torch
sys.path.insert(0, '.')
from alignq_amd import config
from alignq_amd.resnet import resnet20_quant
from alignq_amd.train_step import TrainStep
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = 128
torch.manual_seed(0)
net = resnet20_quant(8, 8).to(dev).train()
step = TrainStep(net, channels_last=True, qconv=True)
x = torch.randn(128, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 10, (128,), device=dev)
for _ in range(3):
    step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, y)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_device_time_total", row_limit=60, max_name_column_width=40,
                                                   max_src_column_width=90))
