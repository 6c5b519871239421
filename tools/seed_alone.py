#!/usr/bin/env python3
"""k_walk + k_tiles (or k_seed) timed alone: every run is waited for before the next starts, so no synthesis
kernel shares the chip with the pre-pass.   python tools/seed_alone.py [blocks]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from __graft_entry__ import load_package
pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ch = pkg.synth_descriptors(nb, nch=16, seed=0x5EED)
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 1)
    b = s.batch(ch, 1.0 / 25e6, 2500000)
    for k in range(6):
        b.run()
        s.sync()
        print("run %d: seed %.3f ms  synth %.3f ms  total %.3f ms" % ((k,) + tuple(b.timing()[f] for f in ("ms_seed", "ms_synth", "ms_total"))))
    b.close()
