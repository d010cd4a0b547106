"""One Office bottleneck site (batch 28, 256 channels, 56 x 56, channels-last) forward + backward, either as the folded chain
(fused.bn_act_relu: alignq_bnq_fwd / _bwd) or as the modules the reference composes (nn.BatchNorm2d -> activation_quantize_fn ->
relu): the program tools/bnq_pmc.sh runs under rocprofv3 --pmc to count HBM bytes.  argv[1]: fused | plain"""
import sys, torch
sys.path.insert(0, '.')
from alignq_amd import config, fused
import alignq_amd.office as Off
mode = sys.argv[1]
dev = torch.device('cuda:0')
config.args.abitW = 8
B, C, H = 28, 256, 56
torch.manual_seed(0)
bn = torch.nn.BatchNorm2d(C).to(dev).train()
act = Off.activation_quantize_fn(8, "aligned").to(dev)
z = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
for it in range(3):
    z.grad = None
    if mode == "fused":
        y = fused.bn_act_relu(bn, act, z, 0, relu=True)
    else:
        y = torch.nn.functional.relu(act(bn(z)))
    y.backward(g)
torch.cuda.synchronize()
print(mode, "ok", float(y.sum()))
