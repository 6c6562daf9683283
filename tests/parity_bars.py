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
    # (emis0 row = total posterior mass - het - missing in the kernels: at least the het row's scale, see _check in
    # tests/test_hip_parity.py)
    own[..., 4, :] = np.maximum(own[..., 4, :], own[..., 5, :])
    full = np.zeros_like(own)
    if g_full is not None:
        gf = np.array(g_full, dtype=np.float64)
        gf[..., 6, :] *= pi
        full = np.abs(gf).max(-1, keepdims=True)
        full[..., 4, :] = np.maximum(full[..., 4, :], full[..., 5, :])
    err = np.abs(g - g_ref).max(-1, keepdims=True)
    r_bound = float((err / (a * own + c * full + 1e-300)).max())
    r_own = float((err / np.maximum(own, 1e-300)).max())
    r_full = float((err / np.maximum(full, 1e-300)).max()) if g_full is not None else 0.0
    return r_bound, r_own, r_full


def rowscaled(got, ref, floor=1e-300):
    """max over rows of |got - ref| / max|that row of ref| (rows = last axis); ``floor`` keeps all-zero rows judged
    against an absolute figure only where the caller says so (W > 0 rows: their natural scale is the W = 0 row's)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / np.maximum(np.abs(ref).max(-1, keepdims=True), floor)).max())


# Bars of the float32 gradient comparisons that are NOT the random-shape / full-size metric above: every one is <= 5x
# the worst value measured on the MI355X for that comparison (profiles/r04_full_size_parity.txt lists the
# measurements, printed by ``check`` as "PARITY <name>: measured ... bar ..." under ``pytest -s``).
BARS = {  # name: bar                                  measured worst (round 4: profiles/r04_full_size_parity.txt, written by scripts/parity_maxima.py)
    "ref_cuda.captured_vectors.f32": 6e-5,         # 1.26e-5
    "ref_cuda.captured_vectors.f64": 6.5e-14,        # 1.31e-14
    "ref_cuda.dlog_blocks.f32": 3.4e-05,            # 6.86e-06
    "ref_cuda.dlog_blocks.f64": 6.8e-14,             # 1.37e-14
    "ref_cuda.cfg1.f32": 8e-6,                     # 1.67e-6
    "ref_cuda.cfg1.f64": 1.2e-13,                  # 2.52e-14
    "ref_cuda.live_short.f32": 5.6e-05,               # 1.13e-05
    "ref_cuda.live_short.f64": 8e-14,              # 1.62e-14
    "ref_cuda.live_cfg2_rows.f32": 0.00039,         # 7.92e-05
    "ref_cuda.live_cfg2_rows.f64": 3e-13,          # 6.11e-14
    "ref_cuda.live_dense_het_runs.f32": 2.1e-5,    # 4.25e-6 (20,000-site rows at 5 % / 10 % hets, one-state-per-lane kernels)
    "ref_cuda.live_dense_het_runs.f64": 2.5e-13,   # 5.13e-14
    "golden.row0.f32": 2.5e-5,                     # 5.03e-6
    "golden.row0.f64": 4e-14,                      # 8.0e-15
    "golden.row1_W100.f32": 1.5e-5,                # 3.05e-6 of (own row + W = 0 row)
    "golden.row1_W100.f64": 3.4e-14,               # 6.88e-15
    "c_abi_client.f32": 3.4e-05,                    # 6.86e-06
    "c_abi_client.f64": 1.1e-08,                    # 2.32e-09
    "integration_stub.f32": 2.7e-5,                # 5.41e-6
    "integration_stub.f64": 8e-14,                 # 1.62e-14
    "smoke.f32": 8.1e-6,                           # 1.62e-6 of (own row + W = 0 row)
    "full_size.identity_pi.f32": 2e-3,             # 8.0e-4
    "full_size.identity_gamma.f32": 1.5e-3,        # 3.5e-4
    "full_size.ll_grad_vs_nograd.f32": 1e-2,       # round 6: 1.9e-3 .. 3.3e-3 absolute on rows of 60,500 / 100,500 sites, |ll| 2e3 .. 3e4 (1e-7
                                                   # relative): the gradient call's ll carries the first-order correction for the float32 model's
                                                   # rounding (phk_ll_first_order), the no-gradient call's cannot; round 5, neither did: 3.1e-4
    "full_size.ll_variants.f32": 2e-3,             # 7.7e-4 absolute
    "full_size.ll_plans.f32": 1e-3,                # 0 (the plans compared share their forward kernel); a variant change is ~3e-4
}


def check(name, value, bar=None):
    """Print the measured figure (so that a run under ``pytest -s`` documents it) and hold it against its bar."""
    bar = BARS[name] if bar is None else bar
    print(f"PARITY {name}: measured {value:.3e} bar {bar:.1e}")
    assert value < bar, f"{name}: {value:.3e} >= {bar:.1e}"
