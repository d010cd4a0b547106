import sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import config
from alignq_amd.resnet import PreActBlock_conv_Q, PreActResNet
from oracle import torch_ref as R
from tests.conftest import load_golden
from tests.test_gpu_parity import _load_ref_state, _ref_to_oracle_name
dev = torch.device('cuda:0')
g = load_golden("g8_tiny_resnet_admm")
config.args.bitW = config.args.abitW = 4; config.args.train_batch_size = 8
net = PreActResNet(PreActBlock_conv_Q, [1, 1, 1], 4, 4, "second", 10); _load_ref_state(net, g, "init/"); net = net.to(dev).train()
cfg = R.Config(tree="admm", bitW=4, abitW=4, train_batch_size=8)
onet = R.PreActResNet(cfg, [1, 1, 1], 4, 4)
onet.load_state_dict({_ref_to_oracle_name(k[5:]): torch.from_numpy(v) for k, v in g.items() if k.startswith("init/") and _ref_to_oracle_name(k[5:]) is not None}, strict=True)
onet = onet.to(dev).train()
rec, orec = {}, {}
def hook(store, name):
    def f(m, inp, out):
        store[name] = (inp[0].detach().clone(), (out[0] if isinstance(out, tuple) else out).detach().clone())
    return f
for n, m in net.named_modules():
    if n.endswith(('act_q0', 'act_q1', 'act_skip_q')) or 'conv' in n.split('.')[-1]:
        m.register_forward_hook(hook(rec, n))
for n, m in onet.named_modules():
    if n.endswith(('site0', 'site1', 'site_skip')) or 'conv' in n.split('.')[-1]:
        m.register_forward_hook(hook(orec, n.replace('site_skip', 'act_skip_q').replace('site0', 'act_q0').replace('site1', 'act_q1')))
x = torch.from_numpy(g["xs"][0]).to(dev)
lo, tl = net(x); olo, otl = onet(x)
for n in rec:
    i, o = rec[n]; oi, oo = orec[n]
    print(f"{n:28s} in maxdiff {float((i-oi).abs().max()):.3e}  out ndiff {int((o!=oo).sum())}/{o.numel()} maxdiff {float((o-oo).abs().max()):.3e}")
print('logits diff', float((lo-olo).abs().max()), 'tl', float(tl), float(otl))
