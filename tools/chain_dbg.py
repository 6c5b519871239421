#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from __graft_entry__ import load_package
pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 2500000
ch = pkg.synth_descriptors(nb, nch=16, seed=0x5EED)
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 1)
    b = s.batch(ch, 1.0 / 25e6, nsamp, flags=pkg.CHAIN_CARRIER)
    for k in range(3):
        f0 = s.info(pkg.INFO_CHAIN_FALLBACKS)
        b.run()
        s.sync()
        t = b.timing()
        print("run %d: seed %.3f ms synth %.3f ms; chained on device %d; fallbacks %d of %d; ties so far %d" %
              (k, t["ms_seed"], t["ms_synth"], s.info(pkg.INFO_CHAIN_ON_DEVICE), s.info(pkg.INFO_CHAIN_FALLBACKS) - f0, nb * 16, s.info(pkg.INFO_CHAIN_TIES)))
    b.close()
