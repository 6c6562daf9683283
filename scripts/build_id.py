#!/usr/bin/env python3
"""Identity of the build a measurement was taken on: sha256 of the library the process loads and the git head of the tree
(when the tree is a checkout; the GPU box gets a snapshot without .git, so ``git_head`` is read from GIT_HEAD if the caller
exported it).  Printed as one JSON object; scripts/profile.sh and scripts/scaling_expectation.py store it next to their
results, and bench.py compares ``lib_sha256`` with the library it has loaded (roofline.traffic_build_matches)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_path():
    return os.environ.get("PHK_LIB") or os.path.join(ROOT, "phlash_amd", "libphlash_hip.so")


def lib_sha256(path=None):
    h = hashlib.sha256()
    with open(path or lib_path(), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def git_head():
    if os.environ.get("GIT_HEAD"):
        return os.environ["GIT_HEAD"]
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None


def build_id():
    return {"lib_sha256": lib_sha256(), "git_head": git_head()}


if __name__ == "__main__":
    json.dump(build_id(), sys.stdout)
    print()
