"""Small helpers on the path: the PSMC pattern grammar and the softplus inverse.
Behaviour follows the reference's ``Pattern`` / ``softplus_inv`` (src/phlash/util.py:8-37, 49-51)."""

from __future__ import annotations

import functools
import re

import torch

_TERM = re.compile(r"^\s*(?:(\d+)\s*\*\s*)?(\d+)\s*$")


class Pattern:
    """PSMC-style pattern: ``+``-separated terms ``k*w`` (k epochs, each spanning w hidden states)
    or ``w`` (one epoch of width w).  ``"14*1+1*2"`` = 15 free epochs over M = 16 states.
    Raises ValueError for anything else, for an empty pattern and for zero widths (util.py:19-26)."""

    def __init__(self, pattern: str):
        widths: list[int] = []
        for term in str(pattern).split("+"):
            m = _TERM.match(term)
            if m is None:
                raise ValueError("could not parse pattern")
            reps = int(m.group(1)) if m.group(1) is not None else 1
            widths.extend([int(m.group(2))] * reps)
        if not widths:
            raise ValueError("pattern must contain at least one epoch")
        if min(widths) <= 0:
            raise ValueError("epochs must be positive")
        self.pattern = pattern
        self._widths = tuple(widths)
        self._index_cache: dict = {}

    @property
    def widths(self) -> tuple[int, ...]:
        return self._widths

    @property
    def M(self) -> int:
        """number of hidden states"""
        return sum(self._widths)

    def __len__(self) -> int:
        """number of free epochs"""
        return len(self._widths)

    def expand(self, x):
        """One value per epoch -> one per hidden state.  A tensor [..., P] becomes [..., M]; any
        other sequence becomes a list of length M."""
        if isinstance(x, torch.Tensor):
            if x.shape[-1] != len(self):
                raise AssertionError("one value per epoch expected")
            # gather with a cached per-device index (repeat_interleave with tensor repeats would
            # synchronise the stream to learn its output size)
            idx = self._index_cache.get(x.device)
            if idx is None:
                host = [e for e, w in enumerate(self._widths) for _ in range(w)]
                idx = self._index_cache[x.device] = torch.tensor(host, dtype=torch.int64, device=x.device)
            return x.index_select(-1, idx)
        if len(x) != len(self):
            raise AssertionError("one value per epoch expected")
        out = []
        for w, v in zip(self._widths, x):
            out.extend([v] * w)
        return out


@functools.lru_cache(maxsize=64)
def get_pattern(pattern: str) -> Pattern:
    """Shared, cached Pattern (keeps its per-device expansion index across calls)."""
    return Pattern(pattern)


def softplus_inv(y):
    """log(exp(y) - 1) for y > 0, written as y + log1p(-exp(-y)) to stay finite for large y."""
    y = torch.as_tensor(y, dtype=torch.float64)
    return y + torch.log1p(-torch.exp(-y))
