#!/usr/bin/env python3
"""Loops of one kernel in a gfx950 object file: for every backward branch, the instruction mix of the address range it
closes (VALU / SALU / LDS / global / scratch / branches).  A hot loop that touches scratch shows up here.

    python scripts/loop_report.py <object or .so> <substring of the mangled kernel name> [min instructions]
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    if obj.endswith(".o") or obj.endswith(".so"):
        # host object with an embedded offload bundle: unbundle the gfx950 code object first
        import os, shutil, tempfile
        d = tempfile.mkdtemp()
        shutil.copy(obj, os.path.join(d, "x.o"))
        subprocess.run([OBJDUMP, "--offloading", "x.o"], cwd=d, capture_output=True, text=True)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        if cos:
            obj = os.path.join(d, cos[0])
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    # split into functions
    funcs = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", txt)
    for f in funcs:
        m = re.match(r"([0-9a-f]+) <([^>]+)>:", f)
        if not m or pat not in m.group(2):
            continue
        name = m.group(2)
        insts = []  # (addr, mnemonic, operands)
        labels = {}
        for line in f.splitlines()[1:]:
            lm = re.match(r"\s*([0-9a-f]+) <([^>]+)>:", line)
            if lm:
                continue
            im = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if im:
                insts.append((int(im.group(3), 16), im.group(1), im.group(2) + " " + line.split("//", 1)[1]))
        if not insts:
            continue
        addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
        print(f"== {name}: {len(insts)} instructions, {sum(1 for _, mn, _ in insts if mn.startswith('scratch_'))} scratch ops")
        loops = []
        for i, (a, mn, ops) in enumerate(insts):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                tm = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", ops) or re.search(r"<[^>+]+>", ops)
                # objdump prints the target as symbol+offset; compute from operand value when present
                om = re.match(r"(-?\d+|0x[0-9a-f]+)?", ops)
                tgt = None
                t2 = re.search(r"\+0x([0-9a-f]+)>", ops)
                if t2:
                    base = insts[0][0]
                    tgt = base + int(t2.group(1), 16)
                elif re.search(r"<[^>]+>", ops) and not t2:
                    tgt = insts[0][0]
                if tgt is not None and tgt <= a and tgt in addr_index:
                    loops.append((addr_index[tgt], i))
        for lo, hi in sorted(set(loops), key=lambda x: x[1] - x[0]):
            body = insts[lo:hi + 1]
            if len(body) < min_n:
                continue
            def cnt(pred):
                return sum(1 for _, mn, _ in body if pred(mn))
            valu = cnt(lambda m: m.startswith("v_"))
            pk = cnt(lambda m: m.startswith("v_pk_"))
            dpp = sum(1 for _, mn, ops in body if mn.startswith("v_") and ("dpp" in mn or "row_" in ops or "quad_perm" in ops))
            salu = cnt(lambda m: m.startswith("s_") and not m.startswith("s_cbranch") and m not in ("s_branch", "s_waitcnt", "s_nop"))
            br = cnt(lambda m: m.startswith("s_cbranch") or m == "s_branch")
            wait = cnt(lambda m: m == "s_waitcnt")
            ds = cnt(lambda m: m.startswith("ds_"))
            gl = cnt(lambda m: m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_"))
            scr = cnt(lambda m: m.startswith("scratch_"))
            print(f"  loop [{lo:5d},{hi:5d}] {len(body):5d} insts: VALU {valu} (pk {pk}, dpp {dpp})  SALU {salu}  branch {br}  "
                  f"waitcnt {wait}  LDS {ds}  global {gl}  scratch {scr}")


if __name__ == "__main__":
    main()
