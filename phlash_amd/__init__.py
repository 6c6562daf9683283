"""phlash_amd: an MI355X-native (gfx950) engine for the one hot path of jthlab/phlash -- the
PSMC / SMC' coalescent-HMM log-likelihood and its gradient -- behind phlash's own kernel-plugin
surface (``get_kernel(M, data, double_precision)`` -> ``.loglik`` / ``.__call__``).

Importing the package does not touch the GPU; the HIP library is loaded on first use
(reference: src/phlash/kernel.py:9-12 defers the CUDA import the same way).
"""

__version__ = "0.1.0"


def __getattr__(name):
    # lazy re-exports (reference: src/phlash/__init__.py:18-24) -- nothing touches the GPU on import
    if name in ("fit",):
        from .mcmc import fit
        return fit
    if name == "psmc":
        from .psmc import psmc
        return psmc
    if name == "get_kernel":
        from .kernel import get_kernel
        return get_kernel
    if name in ("DemographicModel", "SizeHistory"):
        from . import size_history
        return getattr(size_history, name)
    if name in ("PSMCParams", "MCMCParams"):
        from . import params
        return getattr(params, name)
    if name == "RawContig":
        from .data import RawContig
        return RawContig
    raise AttributeError(name)
