"""phlash_amd: an MI355X-native (gfx950) engine for the one hot path of jthlab/phlash -- the
PSMC / SMC' coalescent-HMM log-likelihood and its gradient -- behind phlash's own kernel-plugin
surface (``get_kernel(M, data, double_precision)`` -> ``.loglik`` / ``.__call__``).

Importing the package does not touch the GPU; the HIP library is loaded on first use
(reference: src/phlash/kernel.py:9-12 defers the CUDA import the same way).
"""

__version__ = "0.1.0"
