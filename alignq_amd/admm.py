"""ADMM loss module — drop-in for the reference's utils/admm.py:12-33 (same in every tree).

Interface kept: `ADMM(dim)`; Parameters named `alterD`, `gamma` (matched by substring in the reference
drivers, main.py:86,90,256-260), plain attributes `mu`, `rho`, and `D` (the last D, read by
main.py:330-369 for ADMM_OPT.step).  forward(D) -> scalar loss, computed (with its gradients) by one
HIP launch (alignq_admm_loss)."""
import torch
from torch.nn.parameter import Parameter

from . import ops


class ADMM(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.mu = 0.2
        self.rho = 0.3
        self.alterD = Parameter(torch.rand(dim, dim))
        self.gamma = Parameter(torch.rand(dim, dim))
        self.D = None

    def forward(self, D):
        self.D = D
        return ops.AdmmLossFn.apply(D, self.alterD, self.gamma, self.mu, self.rho)
