"""Deterministic parameter initialisation shared by the golden generator (reference side) and the tests (this repo's
model): fixtures of whole-model iterations then need no copy of the initial state_dict.  Values come from torch's CPU
generator seeded per tensor (position in named_parameters order), which is stable across machines for one torch build; the
torch version used at capture time is stored in every fixture."""
import math

import torch


def det_init_(module, seed=1000):
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            g = torch.Generator().manual_seed(seed + i)
            if "alterD" in name or "gamma" in name:                   # utils/admm.py:19-20: torch.rand
                v = torch.rand(p.shape, generator=g)
            elif p.dim() == 4:                                        # conv weight: kaiming fan_out scale
                fan_out = p.shape[0] * p.shape[2] * p.shape[3]
                v = torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan_out)
            elif p.dim() == 2:                                        # linear weight
                v = torch.randn(p.shape, generator=g) * 0.02
            elif "bn" in name and name.endswith("weight") or "downsample.1.weight" in name:
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:                                                     # biases
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(v.to(p.device))
    return module


def sample(t, n=256):
    """A fixed strided sample (<= n elements) of a tensor, flattened; the whole tensor when it is small."""
    f = t.detach().reshape(-1)
    if f.numel() <= 2048:
        return f.clone()
    step = -(-f.numel() // n)
    return f[::step].clone()
