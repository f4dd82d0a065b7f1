#!/usr/bin/env python
"""Round 5: a variant of the library that differs only in the fused forward's three translation units (compiled in parallel with the
given -D flags), linked with the cached objects of everything else:  tools/exp/build_fused_variant.py NAME [-DFLAG ...]  ->  stego_amd/lib/NAME.so
Prints the register / spill / scratch numbers of the headline instantiation (f16x3, C = 384, K <= 96, even)."""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import _build

FUSED = ["corr_fused.hip", "corr_fused_odd.hip", "corr_fused_c192.hip"]
HEAD = "_ZN5stego17corr_fused_kernelILi1ELi3ELi3ELb0EEEvNS_11FusedParamsE"


def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-D")]
    _build.build()                                   # the cached objects of the other sources
    odir = os.path.join(_build.OBJ_DIR, "var_" + name)
    os.makedirs(odir, exist_ok=True)

    def one(src):
        o = os.path.join(odir, src.replace(".hip", ".o"))
        cmd = [_build._hipcc()] + _build.CFLAGS + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(_build.CSRC, src), "-o", o]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(res.stderr[-6000:])
        return o, res.stderr

    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(one, FUSED))
    objs = [_build._obj(s) for s in _build.sources() if os.path.basename(s) not in FUSED] + [o for o, _ in outs]
    out = os.path.join(_build.LIB_DIR, name + ".so")
    res = subprocess.run([_build._hipcc()] + _build.LDFLAGS + objs + ["-o", out], capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr[-4000:])
    cur, stats, worst = None, {}, (0, "")
    for _, err in outs:
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1); stats[cur] = {}; continue
            m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill): (\d+)", line)
            if m and cur:
                stats[cur][m.group(1)] = int(m.group(2))
    for k, v in stats.items():
        if "corr_fused_kernel" in k and v.get("VGPRs Spill", 0) > worst[0]:
            worst = (v["VGPRs Spill"], k)
    print(out)
    print("headline:", stats.get(HEAD))
    print("worst spill:", worst)


main()
