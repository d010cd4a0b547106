import sys, numpy as np, torch
sys.path.insert(0, '.')
from alignq_amd import ops
from tests import oracle_c as O
dev = torch.device('cuda:0')
for k, formula, n in [(8, 0, 1 << 18), (4, 0, 1025), (4, 0, 5)]:
    rng = np.random.default_rng(10 + k)
    x = (rng.standard_normal(n) * 1.7).astype(np.float32)
    xq, bins = ops.act_quant_bins(torch.from_numpy(x).to(dev), k, 2.0, formula)
    oq, ot, ob = O.act_quant_fwd(x, k, 2.0, formula)
    a = xq.cpu().numpy(); b = bins.cpu().numpy()
    bad = np.nonzero(a.view(np.uint32) != oq.view(np.uint32))[0]
    print(k, formula, n, 'bad', bad.size, 'bins bad', int((b != ob).sum()))
    for i in bad[:5]:
        print('  i', i, 'x', float(x[i]).hex(), 'ours', float(a[i]).hex(), 'oracle', float(oq[i]).hex(), 'bin', b[i], ob[i], 't', float(ot[i]).hex())
