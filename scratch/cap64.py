import sys, faulthandler, torch
faulthandler.enable()
sys.path.insert(0, '.')
from alignq_amd import config
from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
from alignq_amd.train_step import TrainStep
B = int(sys.argv[1])
dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = 8
config.args.train_batch_size = B
torch.manual_seed(5)
x = torch.randn(B, 3, 32, 32, device=dev); y = torch.randint(0, 10, (B,), device=dev)
net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 8, 8, "second", 10).to(dev).train()
step = TrainStep(net, lr=0.01, channels_last=True)
step.capture(x, y, warmup=1)
out = step(*step.static_inputs())
print("B", B, "captured ok", float(out[1]))
