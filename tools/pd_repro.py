#!/usr/bin/env python3
"""Replay the descriptors a failing stream soak left in gpurun_out/fuzz_stream_fail_ch.npy as one chained batch and list where
the IQ differs from the oracle's.   python tools/pd_repro.py <fs> <nsamp> [file]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa
from __graft_entry__ import load_package
import oracle_binding as ob
pkg = load_package(); oracle = ob.Oracle()
fs, nsamp = float(sys.argv[1]), int(sys.argv[2])
ch = np.load(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "fuzz_stream_fail_ch.npy"))
want_iq, want_st, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=True, fixed=False)
with pkg.Synth(0) as synth:
    synth.set_option(pkg.OPT_SEED_WHERE, 1)
    b = synth.batch(ch, 1.0 / fs, nsamp, flags=pkg.CHAIN_CARRIER); b.run(); synth.sync(); iq, st = b.read(); b.close()
    iq = np.asarray(iq).reshape(want_iq.shape)
    bad = np.argwhere((iq != want_iq).any(axis=-1)) if iq.ndim == 3 else np.argwhere(iq != want_iq)
    print("kernel", synth.info(pkg.INFO_LAST_KERNEL), "mismatching samples", len(bad), "of", iq.shape)
    tiles = sorted(set((int(b_), int(n_) // 1024) for b_, n_ in bad))
    print("tiles touched", len(tiles), tiles[:12])
    for blk, t in tiles[:6]:
        seg = (iq[blk, t * 1024:(t + 1) * 1024] != want_iq[blk, t * 1024:(t + 1) * 1024]).any(axis=-1)
        print("  block %d tile %d: %d bad samples; distinct got %s" % (blk, t, int(seg.sum()), np.unique(iq[blk, t * 1024:(t + 1) * 1024][seg], axis=0)[:4].tolist()))
    for blk, n in bad[:3]:
        t, r = divmod(int(n), 1024)
        print("block %d sample %d tile %d j %d lane %d  got %s want %s  f_carr %s prn %s" % (blk, n, t, r // 64, r % 64, iq[blk, n].tolist(), want_iq[blk, n].tolist(),
              ch["f_carr"][blk].tolist(), ch["prn"][blk].tolist()))
