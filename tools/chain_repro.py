#!/usr/bin/env python3
"""re-run a failing chained stream saved by fuzz_parity --stream: python tools/chain_repro.py <ch.npy> fs nsamp bps [channel]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa
from __graft_entry__ import load_package
import oracle_binding as ob
pkg = load_package(); oracle = ob.Oracle()
ch = np.load(sys.argv[1]); fs = float(sys.argv[2]); nsamp = int(sys.argv[3]); bps = int(sys.argv[4])
if len(sys.argv) > 5:
    i = int(sys.argv[5]); ch = np.ascontiguousarray(ch[:, i:i + 1])
nb, nch = ch.shape
want_iq, want_st, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True, fixed=False)
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 1)
    # one batch
    b = s.batch(ch, 1 / fs, nsamp, flags=pkg.CHAIN_CARRIER); b.run(); s.sync(); iq, st = b.read(); b.close()
    bad = np.argwhere((st["carr_phase"] != want_st["carr_phase"]) & (ch["prn"] > 0))
    print("batch: on device %d, fallbacks %d, ties %d; bad block-channels %d, first %r" %
          (s.info(pkg.INFO_CHAIN_ON_DEVICE), s.info(pkg.INFO_CHAIN_FALLBACKS), s.info(pkg.INFO_CHAIN_TIES), len(bad), bad[:3].tolist()))
    # the stream
    for depth in (2, 4):
        st_ = s.stream(nch, 1 / fs, nsamp, bps, depth=depth, flags=pkg.CHAIN_CARRIER)
        got = []
        for k in range(nb // bps):
            if st_.pending == depth:
                got.append(st_.pop()[1])
            st_.push(ch[k * bps:(k + 1) * bps])
        while st_.pending:
            got.append(st_.pop()[1])
        st_.close()
        g = np.concatenate(got)
        bad = np.argwhere((g["carr_phase"] != want_st["carr_phase"][:len(g)]) & (ch["prn"][:len(g)] > 0))
        print("stream depth %d: bad %d first %r" % (depth, len(bad), bad[:3].tolist()))
        if len(bad):
            bb, ii = bad[0]
            print("   got %r want %r start (want prev end) %r f_carr %r step %r" % (g["carr_phase"][bb, ii], want_st["carr_phase"][bb, ii],
                  want_st["carr_phase"][bb - 1, ii] if bb else None, ch["f_carr"][bb, ii], ch["f_carr"][bb, ii] / fs))
