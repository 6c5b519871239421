#!/usr/bin/env python3
"""One channel of a saved descriptor array as a chained batch through a -DGPSBB_CHAIN_DEBUG build (fix_block prints what it does):
   make -C pluto-gps-sim_amd/csrc broken EXTRA=-DGPSBB_CHAIN_DEBUG ... see the command in tools/chain_dbg1.sh"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa
from __graft_entry__ import load_package
import oracle_binding as ob
pkg = load_package()
ch = np.load(sys.argv[1]); fs = float(sys.argv[2]); nsamp = int(sys.argv[3]); i = int(sys.argv[4]); cw = int(sys.argv[5]) if len(sys.argv) > 5 else 0
one = ch[:, i:i + 1].copy()
want_iq, want_st, _ = ob.Oracle().fill_blocks(one, 1 / fs, nsamp, chain=True)
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 1); s.set_option(pkg.OPT_CHAIN_WHERE, cw)
    b = s.batch(one, 1 / fs, nsamp, flags=pkg.CHAIN_CARRIER); b.run(); s.sync(); iq, st = b.read(); b.close()
print("got ", [repr(float(x)) for x in st["carr_phase"][:, 0]])
print("want", [repr(float(x)) for x in want_st["carr_phase"][:, 0]], "iq equal", bool((iq == want_iq).all()))
