"""CPU oracle for the phlash coalescent-HMM hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import or execute it, and only as the checker / the reported CPU baseline --
never as the thing shipped or measured as "the GPU path".  ``phlash_amd`` must
never import this package (``tests/test_layout.py`` enforces that).

What it is: an independent float64 restatement (numpy loops + plain C) of the
reference's algorithm for this path, each function citing the reference
file:line it follows (paths relative to the upstream repo jthlab/phlash
@ v1.0.6).

PARITY PINNING -- read this before trusting a number
----------------------------------------------------
The reference's own implementation can be neither imported nor compiled in the
build container or on the GPU box: ``import phlash`` needs jax / jaxlib /
jax_dataclasses / jaxtyping / loguru / blackjax / optax (all absent, no
network) and its CUDA kernel is an NVRTC string that needs an NVIDIA driver.
The reference's tests hold **no numeric golden log-likelihoods or gradients**
for this path -- every hot-path check there is a self-consistency identity.
So this oracle is pinned by exactly those identities and known answers
(``tests/test_oracle_pins.py``), each against the reference test it restates:

* ``v @ transition_matrix(dm) == matvec_smc(v, from_dm(dm))``  (tests/test_hmm.py:10-19)
* ``_expQ(r,c,n) == scipy.linalg.expm(Q)``                      (tests/test_transition.py:21-28)
* transition rows >= 0 and sum to 1 for n in 2,5,10,50          (tests/test_transition.py:31-35)
* ``surv == [0.9,0.8,0.7,0]`` and ``pi == 0.25`` closed forms   (tests/test_size_history.py:30-40)
* the psmcfa fixture decodes to 100 sites / 82 hets             (tests/test_data.py:31-38)
* chunking 10,000 sites / chunk 4,567 / overlap 123 -> 3 rows   (tests/test_data.py:18-28)
* grad-kernel ll == no-grad ll; O(K) scan == dense forward == brute-force
  path enumeration; reverse-mode gradient == autograd == finite differences
  (the roles of tests/test_gpu.py:27-64, tests/test_model.py:8-19)

Beyond those identities the numeric values of ll / gradients are
**parity unpinned** (restatement-derived, not reference-captured); the golden
vectors in ``tests/golden`` are produced by ``oracle/make_golden.py`` from this
oracle and are labelled as such.  The SVGD / AMSGrad arithmetic of the
reference lives in third-party blackjax==1.2.5 / optax==0.2.6 (uv.lock) whose
sources are not in the reference tree: also parity unpinned.
"""
