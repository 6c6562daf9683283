#!/usr/bin/env python3
"""Loops of one kernel in a gfx950 object file: for every backward branch, the instruction mix of the address range it
closes (VALU / SALU / LDS / global / scratch / branches).  A hot loop that touches scratch shows up here.

    python scripts/loop_report.py <object or .so> <substring of the mangled kernel name> [min instructions]

tests/test_layout.py imports ``kernel_loops`` to hold the sweeps' block loops to "no scratch access".
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _device_code(obj):
    """Path of the gfx950 code object inside a host object / shared library (or ``obj`` itself if it is one)."""
    d = tempfile.mkdtemp()
    shutil.copy(obj, os.path.join(d, "x.o"))
    subprocess.run([OBJDUMP, "--offloading", "x.o"], cwd=d, capture_output=True, text=True)
    cos = [f for f in os.listdir(d) if "amdgcn" in f]
    return (os.path.join(d, cos[0]), d) if cos else (obj, d)


def kernel_loops(obj, pat):
    """{mangled kernel name: {"n": instructions, "scratch": scratch ops, "loops": [dict per backward branch]}} for every
    kernel of ``obj`` whose mangled name contains ``pat``."""
    co, tmp = _device_code(obj)
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for f in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", txt):
        m = re.match(r"([0-9a-f]+) <([^>]+)>:", f)
        if not m or pat not in m.group(2):
            continue
        insts = []  # (address, mnemonic, rest of the line)
        for line in f.splitlines()[1:]:
            im = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", line)
            if im:
                insts.append((int(im.group(3), 16), im.group(1), im.group(2) + " " + im.group(4)))
        if not insts:
            continue
        index = {a: i for i, (a, _, _) in enumerate(insts)}
        base = insts[0][0]
        loops = set()
        for i, (a, mn, ops) in enumerate(insts):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                t = re.search(r"\+0x([0-9a-f]+)>", ops)
                tgt = base + int(t.group(1), 16) if t else (base if re.search(r"<[^>+]+>", ops) else None)
                if tgt is not None and tgt <= a and tgt in index:
                    loops.add((index[tgt], i))
        rows = []
        for lo, hi in sorted(loops, key=lambda x: x[1] - x[0]):
            body = insts[lo:hi + 1]

            def cnt(pred):
                return sum(1 for _, mn, _ in body if pred(mn))

            rows.append({
                "lo": lo, "hi": hi, "n": len(body),
                "valu": cnt(lambda m: m.startswith("v_")),
                "pk": cnt(lambda m: m.startswith("v_pk_")),
                "dpp": sum(1 for _, mn, ops in body if mn.startswith("v_") and ("dpp" in mn or "row_" in ops or "quad_perm" in ops)),
                "salu": cnt(lambda m: m.startswith("s_") and not m.startswith("s_cbranch") and m not in ("s_branch", "s_waitcnt", "s_nop")),
                "branch": cnt(lambda m: m.startswith("s_cbranch") or m == "s_branch"),
                "waitcnt": cnt(lambda m: m == "s_waitcnt"),
                "lds": cnt(lambda m: m.startswith("ds_")),
                "global": cnt(lambda m: m.startswith(("global_", "buffer_", "flat_"))),
                "scratch": cnt(lambda m: m.startswith("scratch_")),
            })
        out[m.group(2)] = {"n": len(insts), "scratch": sum(1 for _, mn, _ in insts if mn.startswith("scratch_")), "loops": rows}
    return out


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    for name, k in kernel_loops(obj, pat).items():
        print(f"== {name}: {k['n']} instructions, {k['scratch']} scratch ops")
        for r in k["loops"]:
            if r["n"] >= min_n:
                print(f"  loop [{r['lo']:5d},{r['hi']:5d}] {r['n']:5d} insts: VALU {r['valu']} (pk {r['pk']}, dpp {r['dpp']})  SALU {r['salu']}  "
                      f"branch {r['branch']}  waitcnt {r['waitcnt']}  LDS {r['lds']}  global {r['global']}  scratch {r['scratch']}")


if __name__ == "__main__":
    main()
