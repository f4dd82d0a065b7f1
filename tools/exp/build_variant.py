#!/usr/bin/env python
"""Builds a variant of the library for same-box A/B runs (tools/exp/ab.sh):
   tools/exp/build_variant.py NAME [-DFLAG ...] [--csrc DIR]   ->  stego_amd/lib/NAME.so
--csrc: take the sources from another tree (e.g. `git archive HEAD stego_amd/csrc include | tar -x -C /tmp/base`)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import _build

def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-D")]
    csrc = _build.CSRC
    if "--csrc" in sys.argv:
        csrc = sys.argv[sys.argv.index("--csrc") + 1]
    srcs = [os.path.join(csrc, s) for s in _build.SOURCES if os.path.exists(os.path.join(csrc, s))]
    out = os.path.join(_build.LIB_DIR, name + ".so")
    cmd = [_build._hipcc()] + _build.CFLAGS + flags + ["-shared"] + srcs + ["-o", out]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr[-4000:])
    print(out)

main()
