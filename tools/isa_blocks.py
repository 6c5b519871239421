#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in the ISA dump `make -C pluto-gps-sim_amd/csrc asm` leaves in /tmp.
   python tools/isa_blocks.py k_synth_ev [min_instructions]"""
import re
import sys

s = open('/tmp/gpsbb-hip-amdgcn-amd-amdhsa-gfx950.s').read()
pat = sys.argv[1]
m = re.search(r'^(_ZN10gpsbb_impl\d+%s[^\n]*):(.*?)\.Lfunc_end\d+:' % pat, s, re.S | re.M)
body = m.group(2)
open('/tmp/%s.s' % pat, 'w').write(body)
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
names = ['entry'] + blocks[1::2]
texts = [blocks[0]] + blocks[2::2]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for n, t in zip(names, texts):
    lines = [l.strip() for l in t.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    v = sum(1 for l in lines if l.startswith('v_'))
    sp = sum(1 for l in lines if l.startswith('v_readlane') or l.startswith('v_writelane'))
    sa = sum(1 for l in lines if l.startswith('s_'))
    ds = sum(1 for l in lines if l.startswith('ds_'))
    g = sum(1 for l in lines if l.startswith('global_') or l.startswith('scratch_'))
    w = sum(1 for l in lines if l.startswith('s_waitcnt'))
    if v + ds + g >= lim:
        print('%-10s valu %3d (lane moves %2d)  salu %3d  ds %2d  vmem %2d  waitcnt %2d' % (n, v, sp, sa, ds, g, w))
