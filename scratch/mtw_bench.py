import sys, torch, numpy as np
sys.path.insert(0, '.')
from alignq_amd import config
from alignq_amd.resnet import resnet20_quant
from alignq_amd.fused import prequantize_weights

dev = torch.device('cuda:0')
config.args.bitW = config.args.abitW = 8
net = resnet20_quant(8, 8).to(dev).train()
convs = [m for m in net.modules() if hasattr(m, 'quantize_fn') and isinstance(m, torch.nn.Conv2d)]
print(len(convs), sorted(set(c.weight.numel() for c in convs)))
def t(fn, n=30):
    for i in range(5): fn()
    ev = [torch.cuda.Event(True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]) * 1e3)
print("prequantize_weights: %.1f us" % t(lambda: prequantize_weights(convs)))
