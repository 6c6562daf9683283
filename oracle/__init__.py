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
**Kernel level: PINNED to the reference's own code.**  The
reference's native code for this path is one CUDA C++ translation unit held as a string
(``KERNEL_SRC``, src/phlash/gpu.py:474-693).  ``oracle/build_ref.py`` reads that text where it lies
under ``/root/reference`` and compiles it UNMODIFIED with hipcc for gfx950 into ``oracle/_ref/``
(binaries only, git-ignored; plain CUDA C++ is HIP C++, no stand-in headers are involved).
``oracle/make_ref_golden.py`` ran those kernels on an MI355X on the reference's test inputs and
stored their outputs in ``tests/golden/ref_cuda_golden.npz`` (reference-captured).  The float64
restatement here (numpy loops and C) reproduces the reference's float64 kernels to 1e-15 relative
on the log-likelihood and 1e-14 (row-scaled) on the gradient (``tests/test_ref_cuda.py``, CPU);
the HIP kernels are compared with the same vectors and with the reference kernels run live
(``-m gpu``).  ``tests/golden/ref_afs_golden.npz`` is captured from the reference's
``src/phlash/afs.py`` (numpy/scipy only; loaded by file path).

**Parameter map and SVGD (rows A11-A14): parity unpinned.**  Everything above the kernel in the
reference is JAX code: ``import phlash`` needs jax / jaxlib / jax_dataclasses / jaxtyping / loguru
/ blackjax / optax (all absent, no network), so ``from_dm`` / ``transition_matrix`` /
``SizeHistory`` cannot be executed here and the reference's tests hold no numeric vectors for
them -- every check there is an identity or a closed form.  The restatement of those rows is
pinned by exactly those identities and known answers (``tests/test_oracle_pins.py``), each
against the reference test it restates:

* ``v @ transition_matrix(dm) == matvec_smc(v, from_dm(dm))``  (tests/test_hmm.py:10-19)
* ``_expQ(r,c,n) == scipy.linalg.expm(Q)``                      (tests/test_transition.py:21-28)
* transition rows >= 0 and sum to 1 for n in 2,5,10,50          (tests/test_transition.py:31-35)
* ``surv == [0.9,0.8,0.7,0]`` and ``pi == 0.25`` closed forms   (tests/test_size_history.py:30-40)
* the psmcfa fixture decodes to 100 sites / 82 hets             (tests/test_data.py:31-38)
* chunking 10,000 sites / chunk 4,567 / overlap 123 -> 3 rows   (tests/test_data.py:18-28)
* grad-kernel ll == no-grad ll; O(K) scan == dense forward == brute-force
  path enumeration; reverse-mode gradient == autograd == finite differences
  (the roles of tests/test_gpu.py:27-64, tests/test_model.py:8-19)

Round 3 added oracles for those rows that share no code (and, where possible, no formula) with the product,
so that the GPU tests of the parameter map, the prior, the SVGD step and the AFS term compare the HIP kernels
with an independent statement instead of with the product's own torch code: ``psmc_numpy.from_dm`` /
``particle_to_dm`` / ``log_prior`` (loops), ``psmc_torch`` (autograd Jacobians), ``svgd_numpy`` (loop-form
blackjax 1.2.5 ``svgd`` + optax 0.2.6 ``amsgrad``), ``afs_numpy`` (``etjj`` by scipy quadrature, ``etbl`` from
the lineage-count Markov chain and Fu's subtending probabilities -- no W matrix).  Still unpinned: agreement
with an independent restatement is not agreement with the reference.

``tests/golden/psmc_golden.npz`` (``oracle/make_golden.py``) is restatement-derived and labelled
as such; it overlaps with the reference-captured file on the conftest inputs, where the two agree
to 1e-13.  The SVGD / AMSGrad arithmetic of the reference lives in third-party blackjax==1.2.5 /
optax==0.2.6 (uv.lock) whose sources are not in the reference tree: also parity unpinned.
"""
