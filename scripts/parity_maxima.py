#!/usr/bin/env python3
"""Collect what the GPU suite measured against its bars: every ``PARITY`` line of ``pytest -m gpu -s`` (written by
tests/parity_bars.check and the full-size tests), the largest measurement per bar, the W = 0 identities and the
rows above the scan.

    python3 -m pytest tests -q -m gpu -s > gpurun_out/r4z/gpu_suite.log      (on the GPU box)
    python3 scripts/parity_maxima.py gpurun_out/r4z/gpu_suite.log > profiles/r04_full_size_parity.txt
"""
import collections
import re
import sys


def main():
    log = sys.argv[1]
    bars = collections.OrderedDict()  # key -> [max measured, bar, count]
    other = []
    summary = ""
    for raw in open(log, errors="replace"):
        for line in re.split(r"(?=PARITY )", raw.rstrip("\n")):
            line = line.lstrip(".").strip()
            m = re.match(r"PARITY (\S+): measured ([\d.e+-]+) bar ([\d.e+-]+)", line)
            if m:
                k, v, b = m.group(1), float(m.group(2)), float(m.group(3))
                e = bars.setdefault(k, [0.0, b, 0])
                e[0] = max(e[0], v)
                e[2] += 1
            elif line.startswith("PARITY "):
                other.append(line)
            elif "W=0 identities" in line or "param map vs numpy oracle" in line or "row-scaled max error" in line:
                other.append(line.strip())
        if re.search(r"\d+ passed", raw):
            summary = raw.strip()
    print(f"GPU suite of the round's final build ({log}): {summary}")
    print()
    print("Largest measurement per bar (tests/parity_bars.py: every bar is <= 5 x the largest value seen on this hardware):")
    print(f"  {'bar':44s} {'max measured':>13s} {'bar':>10s} {'bar / max':>10s} {'checks':>7s}")
    for k, (v, b, n) in bars.items():
        ratio = f"{b / v:10.1f}" if v > 0 else "       (0)"
        print(f"  {k:44s} {v:13.3e} {b:10.1e} {ratio} {n:7d}")
    print()
    print("Full-size and sampled comparisons as the tests printed them:")
    for line in other:
        print("  " + line)


if __name__ == "__main__":
    main()
