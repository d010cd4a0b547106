"""SGD and ADMM_OPT — drop-ins for the reference's utils/optimizer.py (ADMM_OPT :15-135, SGD :138-262),
same constructor and step() signatures, both torch.optim.Optimizer subclasses (state_dict keeps the
`momentum_buffer` key, main.py:142-143).  The arithmetic runs in HIP kernels (alignq_sgd_step,
alignq_sgd_grad_approx, alignq_admm_update); parameters are updated IN PLACE at fixed addresses so the
whole step is HIP-graph capturable (the reference rebinds `p.data` to fresh tensors).

Reference quirks kept:
  * SGD: for tensors in `idx` the update still uses the un-approximated direction; only the value left
    in p.grad is `dir * sigmoid_d(transform(w_cdf)) * w_pdf`;
  * ADMM_OPT: parameters whose grad is None are skipped; the dual update uses the padded D and the NEW
    alterD of the alterD parameter that precedes it in the parameter list.
"""
from __future__ import annotations

from typing import List

import torch
from torch.optim.optimizer import Optimizer

from . import _lib as L
from . import config


class SGD(Optimizer):
    def __init__(self, params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if momentum < 0.0:
            raise ValueError("Invalid momentum value: {}".format(momentum))
        if weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("nesterov", False)

    @torch.no_grad()
    def step(self, idx, w_cdf, w_pdf, lam, lam2, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        st = L.stream_ptr()
        bitW = int(config.args.bitW)
        for group in self.param_groups:
            wd, mom, damp, nest, lr = (group["weight_decay"], group["momentum"], group["dampening"],
                                       group["nesterov"], group["lr"])
            for i, p in enumerate(group["params"]):
                if p.grad is None:
                    continue
                L.dev_f32(p, "parameter")
                g = p.grad
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                first = 0
                buf = None
                if mom != 0:
                    state = self.state[p]
                    if "momentum_buffer" not in state:
                        state["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                        first = 1
                    buf = state["momentum_buffer"]
                L.check(lib.alignq_sgd_step(p.data_ptr(), g.data_ptr(), L.ptr(buf), p.numel(), float(lr), float(mom),
                                            float(damp), float(wd), int(bool(nest)), first, st), "alignq_sgd_step")
                if bitW < 32 and i in idx:
                    j = idx.index(i)
                    c, pdf = L.dev_f32(w_cdf[j].detach(), "w_cdf"), L.dev_f32(w_pdf[j].detach(), "w_pdf")
                    L.check(lib.alignq_sgd_grad_approx(g.data_ptr(), c.data_ptr(), pdf.data_ptr(), g.data_ptr(),
                                                       p.numel(), bitW, float(lam), float(lam2), st),
                            "alignq_sgd_grad_approx")
        return loss


class ADMM_OPT(Optimizer):
    def __init__(self, params):
        super().__init__(params, dict())
        self._tab_host = None      # pinned [3, S] int64
        self._tab_dev = None
        self._tab_key = None

    def _tables(self, Dp: List[int], Ap: List[int], Gp: List[int], device):
        key = (tuple(Dp), tuple(Ap), tuple(Gp))
        S = len(Dp)
        if self._tab_dev is None or self._tab_dev.shape[1] != S or self._tab_dev.device != device:
            self._tab_host = torch.empty(3, S, dtype=torch.int64).pin_memory()
            self._tab_dev = torch.empty(3, S, dtype=torch.int64, device=device)
            self._tab_key = None
        if key != self._tab_key:
            self._tab_host.copy_(torch.tensor([Dp, Ap, Gp], dtype=torch.int64))
            # pinned source: capturable as a memcpy node; outside capture make it blocking so the pinned
            # buffer is never rewritten under an in-flight copy
            self._tab_dev.copy_(self._tab_host, non_blocking=torch.cuda.is_current_stream_capturing())
            self._tab_key = key
        return self._tab_dev

    @torch.no_grad()
    def step(self, alterD_idx, gamma_idx, Ds, alterDs, gammas, mus, rhos, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if int(config.args.bitW) >= 32:
            raise KeyError("lr")   # the reference's non-quantised branch reads group['lr'], which ADMM_OPT never defines
        lib = L.load()
        # walk the parameter list like utils/optimizer.py:76-124 and collect (site j, alterD param, gamma param)
        groups = {}
        for group in self.param_groups:
            params = group["params"]
            pending = None            # (j, alterD param) waiting for its dual
            for i, p in enumerate(params):
                if p.grad is None:
                    continue
                if i in alterD_idx:
                    if pending is not None:
                        raise NotImplementedError("alterD parameter without its gamma right after it")
                    pending = (alterD_idx.index(i), p)
                elif i in gamma_idx:
                    if pending is None:
                        raise NotImplementedError("gamma parameter whose alterD was skipped (grad None)")
                    j, pa = pending
                    if gammas[j] is not p:
                        raise NotImplementedError("gamma parameter order does not pair with its alterD")
                    pending = None
                    D = L.dev_f32(Ds[j].detach(), "D")
                    L.dev_f32(pa, "alterD")
                    L.dev_f32(p, "gamma")
                    b, dim = D.shape[0], pa.shape[0]
                    groups.setdefault((float(mus[j]), float(rhos[j]), b, dim), []).append((D, pa, p))
                else:
                    raise KeyError("lr")
            if pending is not None:
                # alterD updated, dual skipped: do the primal only by pairing with a scratch dual
                raise NotImplementedError("alterD parameter without a following gamma parameter")
        st = L.stream_ptr()
        for (mu, rho, b, dim), sites in groups.items():
            keep = [s[0] for s in sites]   # keep D tensors alive until the launch is enqueued
            tab = self._tables([s[0].data_ptr() for s in sites], [s[1].data_ptr() for s in sites],
                               [s[2].data_ptr() for s in sites], sites[0][1].device)
            L.check(lib.alignq_admm_update(tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr(), len(sites), b, dim,
                                           mu, rho, st), "alignq_admm_update")
            del keep
        return loss
