"""Drop-in for the reference's `model/quantization_uniform_admm.py` (the paper's use_cdf=False ablation: uniform
quantisers + ADMM loss; cdf_alignment_admm/resnet-20-cifar-10/model/resnet_ours_uniform_admm.py:22 imports it with `*`):

    from alignq_amd.uniform_admm import *        # instead of: from .quantization_uniform_admm import *
"""
from .admm import ADMM
from .optimizer import ADMM_OPT, SGD
from .quantization import make_uniform_admm_namespace as _mk

_ns = _mk()
globals().update(vars(_ns))
__all__ = sorted(vars(_ns)) + ["ADMM", "ADMM_OPT", "SGD"]
