"""ResNet-20 CDF+ADMM step with the exact-global correlation at world size 1 (B_g = per-GPU batch): eager, timed, to find cliffs."""
import os, sys, time, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from alignq_amd import config, dp
from alignq_amd.resnet import resnet20_quant
from alignq_amd.train_step import TrainStep
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = config.args.eval_batch_size = B
torch.manual_seed(0)
model = resnet20_quant(8, 8).to(dev).train()
st = TrainStep(model, lr=0.04, channels_last=True)
dp.attach(st, force=True, global_corr=True)
x = torch.randn(B, 3, 32, 32, device=dev); y = torch.randint(0, 10, (B,), device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "graph":
    st.capture(x, y, warmup=3)
    x, y = st.static_inputs()
for _ in range(3):
    out = st(x, y)
torch.cuda.synchronize()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
for _ in range(n):
    out = st(x, y)
torch.cuda.synchronize()
print("B", B, sys.argv[3:] , "global-corr step %.2f ms" % ((time.perf_counter() - t0) / n * 1e3), "ce", float(out[1]))
dist.destroy_process_group()
