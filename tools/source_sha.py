#!/usr/bin/env python3
"""Identity of the pair kernels' sources: one sha256 over the files a pair kernel is compiled from.
A profile records it (tools/summarize_prof.py) and bench.py quotes a profile's counters only while
it still matches, so that counters of an older kernel never ride in a freshly timed line."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("nbody_amd/csrc/nbpp_kernels.hip.h", "nbody_amd/csrc/nbpp_sym.hip.h",
                  "nbody_amd/csrc/nbpp_api.hip", "nbody_amd/csrc/nb_common.hip.h")


def kernel_sources_sha():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    print(kernel_sources_sha(), git_head())
