#!/usr/bin/env python
"""ISA of the headline fused-forward instantiation with the given -D flags: where the scratch (spill) accesses and the full vmcnt(0) drains sit
relative to the s_barriers (a spill reload inside the gather stream drains it).  usage: tools/exp/isa_scratch.py [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = '_ZN5stego17corr_fused_kernelILi1ELi3ELi3ELb0EEEvNS_11FusedParamsE'
flags = [a for a in sys.argv[1:] if a.startswith('-D')]
out = os.path.join(tempfile.gettempdir(), 'fused_%s.s' % abs(hash(tuple(flags))))
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S'] + flags +
               ['-I', os.path.join(ROOT, 'stego_amd/csrc'), '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'stego_amd/csrc/corr_fused.hip'), '-o', out], check=True)
s = open(out).read()
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].splitlines()
bars = [n for n, l in enumerate(body) if 's_barrier' in l]
print('lines', len(body), 'barriers', len(bars))
ev = []
for n, l in enumerate(body):
    if 'scratch_' in l: ev.append((n, 'S ' + l.strip()[:60]))
    elif 's_barrier' in l: ev.append((n, 'B'))
    elif re.search(r's_waitcnt vmcnt\(0\)', l): ev.append((n, 'W0'))
    elif 'v_mfma' in l: ev.append((n, 'M'))
# compress: print runs
line = []
last = None
for n, e in ev:
    k = e[0]
    if k == 'M':
        if last == 'M': continue
    if e.startswith('S'):
        print(n, e)
    elif e == 'B': print(n, 'BARRIER')
    elif e == 'W0': print(n, '   vmcnt(0)')
    elif k == 'M': print(n, '   mfma...')
    last = k
print(out)
