"""How sensitive is the REFERENCE's own Office iteration (eager restatement, pinned to fixture G10 at 1e-6) to a 1e-6 / 1e-5
relative perturbation of its inputs?  4-bit bins flip, so two iterations amplify it to O(1) in the logits.  The numbers
printed here are the floor for any cross-implementation comparison of that iteration and set the bars of
tests/test_gpu_round2.py::test_office_tiny_dann_two_iterations_vs_reference.
Measured (torch 2.10 CPU): 1e-6 -> it0: logits 0, D <= 3.0e-3, conv1.weight 7.7e-4; it1: logits 0.79, D <= 1.3e-2, conv1.w 2.8e-3."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from det_init import det_init_, sample
from oracle import torch_ref as R
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g10_office_tiny_dann.npz')))
torch.set_num_threads(8)
def run(pert):
    cfg = R.Config(tree="office", bitW=4, abitW=4, train_batch_size=6)
    torch.manual_seed(0)
    net = R.OfficeDANN(cfg, 4, 4, "aligned", (1,1,1,1), width_per_group=8).train()
    det_init_(net)
    step = R.OfficeTrainStep(net, cfg, lr=0.004, alpha=0.5)
    outs=[]
    for it, epoch in enumerate((1,2)):
        step.new_epoch(epoch, 10, 0.004)
        xs = torch.from_numpy(g["xs"][it]); xt = torch.from_numpy(g["xt"][it])
        if pert: 
            gen = torch.Generator().manual_seed(5)
            xs = xs * (1 + pert*torch.randn(xs.shape, generator=gen)); xt = xt*(1+pert*torch.randn(xt.shape, generator=gen))
        o = step(xs, torch.from_numpy(g["ys"][it]), xt)
        outs.append((o["cls_s"].detach().numpy().copy(), [b.admm0.D.detach().numpy().copy() for b in net.feature.blocks()],
                     {n: p.detach().clone() for n,p in net.named_parameters()}))
    return outs
a = run(0); 
for pert in (1e-6, 1e-5):
    b = run(pert)
    for it in range(2):
        print(pert, it, "cls", np.abs(a[it][0]-b[it][0]).max(), "D", [float(np.abs(x-y).max()) for x,y in zip(a[it][1], b[it][1])],
              "conv1.w", float((a[it][2]['feature.conv1.weight']-b[it][2]['feature.conv1.weight']).abs().max()))
