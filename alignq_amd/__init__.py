"""alignq_amd — MI355X-native implementation of AlignQ's hot path (CDF-alignment quantise/dequantise,
sample-correlation Gram pair, ADMM loss and primal/dual update, SGD step) behind the reference's Python
module API.  See DESIGN.md.  Importing the package does not need a GPU; calling an op does."""
from . import config  # noqa: F401

__version__ = "0.1.0"
