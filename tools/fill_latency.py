#!/usr/bin/env python3
"""Latency of the drop-in call gpsbb_fill_block (one block, IQ into pageable host memory) with the NCO pre-pass on host
threads (the default for one block) and on the device — by the row walks of rounds 1-4 (one whole-block walk per chain) and
lap-parallel (round 5).   python tools/fill_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
CASES = (("reference block: 12 ch, 2.6 MS/s, 300000 samples", 12, 2.6e6, 300000),
         ("16 ch, 25 MS/s, 2500000 samples", 16, 25e6, 2500000))
with pkg.Synth(0) as s:
    for name, nch, fs, nsamp in CASES:
        ch = pkg.synth_descriptors(8, nch=nch, seed=0xBEEF)
        for where, label in ((2, "host threads"), (1, "device, row walks"), (3, "device, lap-parallel"), (0, "automatic")):
            s.set_option(pkg.OPT_SEED_WHERE, where)
            for k in range(3):
                s.fill_block(ch[k % 8], 1.0 / fs, nsamp)
            ts = []
            for k in range(24):
                t0 = time.perf_counter()
                s.fill_block(ch[k % 8], 1.0 / fs, nsamp)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print("%-52s pre-pass on %-21s median %.3f ms  min %.3f ms" % (name, label, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
