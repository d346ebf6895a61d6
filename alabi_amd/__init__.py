"""alabi_amd: MI355X-native GP-surrogate + ensemble-MCMC hot path of jbirky/alabi.

The numerical work lives in ``csrc/libalabi_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/alabi_hip.h``); this package is the Python host side mirroring the reference's
``SurrogateModel`` interface.  Importing the package does not need a GPU; using it does.
"""
from . import benchmarks, gp_utils, mcmc_utils, utility  # noqa: F401
from .core import CachedSurrogateLikelihood, SurrogateModel  # noqa: F401
from .gp import HipGP  # noqa: F401
from .sampler import EnsembleSampler  # noqa: F401
from .utility import *  # noqa: F401,F403

__version__ = "0.1.0"
