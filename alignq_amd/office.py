"""Drop-in for the reference's `model/quantization.py` + `utils/admm.py` + `utils/optimizer.py` of the
"office" tree (see alignq_amd/quantization.py for the file:line map).  Usage in a reference-style model file:

    from alignq_amd.office import *        # instead of: from .quantization import *
"""
from .admm import ADMM
from .optimizer import ADMM_OPT, SGD
from .quantization import make_namespace as _mk

_ns = _mk("office")
globals().update(vars(_ns))
__all__ = sorted(vars(_ns)) + ["ADMM", "ADMM_OPT", "SGD"]
