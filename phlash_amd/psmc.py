"""Convenience entry point for PSMC-formatted input (reference: src/phlash/psmc.py:8-29)."""

from __future__ import annotations

from .data import RawContig
from .mcmc import fit


def psmc(psmcfa_files: list[str], window_size: int = 100, hold_out: bool = True, **options):
    """Read ``.psmcfa`` files and run ``fit``; with ``hold_out`` and more than one contig the
    first contig is kept aside for the expected log-predictive density."""
    contigs = [c for f in psmcfa_files for c in RawContig.from_psmcfa_iter(f, window_size)]
    test_data = None
    if hold_out and len(contigs) > 1:
        test_data = contigs.pop(0)
    return fit(contigs, test_data=test_data, **options)
