"""The gradient metric of the parity tests, in one place (used by tests/test_full_size.py; tests/test_hip_parity.py
documents its derivation next to the random-shape test and profiles/r02_f32_gradient_bars.txt holds the measured
distribution it was set from).

Per gradient row (particle b, chunk s, parameter row r) the largest absolute error over the K states is compared with
    bound = a * own + c * full,
    own  = max |that row of the oracle's gradient|          (pi row: pi_i * d ll / d pi_i, the form gpu.py:303-313
           returns, and at least 1 -- its W = 0 scale, sum_i pi_i d ll / d pi_i = 1),
    full = max |the same row of the oracle's W = 0 gradient| (0 when the test has no warm-up prefix).
With a warm-up prefix every row is the difference of two sweeps and ``full`` is the size of the terms of that
difference: no evaluation in a given precision can promise more than eps x (steps) x |terms|.
"""

import numpy as np


def grad_error_ratios(g, g_ref, g_full, P, a, c):
    """g, g_ref, g_full: [B, S, 7, K] (g_full None for W = 0); P: [B, 1|S, 7, K] parameters.
    Returns (worst err / bound, worst err / own, worst err / full)."""
    g, g_ref = np.array(g, dtype=np.float64), np.array(g_ref, dtype=np.float64)
    pi = np.asarray(P, dtype=np.float64)[..., 6, :]
    g[..., 6, :] *= pi
    g_ref[..., 6, :] *= pi
    own = np.abs(g_ref).max(-1, keepdims=True)
    own[..., 6, :] = np.maximum(own[..., 6, :], 1.0)
    full = np.zeros_like(own)
    if g_full is not None:
        gf = np.array(g_full, dtype=np.float64)
        gf[..., 6, :] *= pi
        full = np.abs(gf).max(-1, keepdims=True)
    err = np.abs(g - g_ref).max(-1, keepdims=True)
    r_bound = float((err / (a * own + c * full + 1e-300)).max())
    r_own = float((err / np.maximum(own, 1e-300)).max())
    r_full = float((err / np.maximum(full, 1e-300)).max()) if g_full is not None else 0.0
    return r_bound, r_own, r_full
