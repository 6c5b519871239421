#!/usr/bin/env python3
"""Replay ONE case of tools/fuzz_parity.py (batch mode) and show where it differs, for every way the chain can be resolved.
   python tools/fuzz_repro.py --seed 401 --case 138"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=401); ap.add_argument("--case", type=int, default=138)
a = ap.parse_args()
import torch  # noqa
from __graft_entry__ import load_package
import oracle_binding as ob
pkg = load_package(); oracle = ob.Oracle()
rng = np.random.default_rng(a.seed)
for case in range(a.case + 1):
    fs = float(rng.choice([1e6, 2.6e6, 3e6, 4.092e6, 10e6, 16e6, 25e6, 30e6, 2.0 ** 25, 50e6]))
    nsamp = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 120000), 1024 * int(rng.integers(1, 60))]))
    nch = int(rng.integers(1, 17)); nblocks = int(rng.integers(1, 6))
    fixed = bool(rng.integers(0, 4) == 0); chain = bool(rng.integers(0, 2))
    mode = int(rng.integers(1, 3)); kern = int(rng.integers(0, 2))
    ch = pkg.synth_descriptors(nblocks, nch=nch, seed=int(rng.integers(1, 2 ** 31)))
    scale = 10.0 ** rng.uniform(-3, np.log10(0.124 * fs), size=(nblocks, nch))
    ch["f_carr"] = np.where(rng.random((nblocks, nch)) < 0.5, -1.0, 1.0) * scale
    if rng.random() < 0.3:
        ch["f_carr"] = np.sign(ch["f_carr"]) * fs * 2.0 ** rng.integers(-20, -4, size=(nblocks, nch))
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    if rng.random() < 0.2:
        ch["code_phase"] = np.floor(ch["code_phase"])
    if rng.random() < 0.2:
        ch["carr_phase"] = np.floor(ch["carr_phase"] * 512.0) / 512.0
    ch["prn"][rng.random((nblocks, nch)) < 0.15] = 0
    if fixed:
        ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
print(dict(case=case, fs=fs, nsamp=nsamp, nch=nch, nblocks=nblocks, fixed=fixed, chain=chain, mode=mode, kern=kern))
flags = (pkg.FIXED_CARRIER if fixed else 0) | (pkg.CHAIN_CARRIER if chain else 0)
want_iq, want_st, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=chain, fixed=fixed)
np.save(os.path.join(ROOT, "gpurun_out", "fuzz_repro_ch.npy"), ch)
with pkg.Synth(0) as synth:
    for cw in (0, 3, 2, 1):
        synth.set_option(pkg.OPT_SEED_WHERE, mode); synth.set_option(pkg.OPT_SYNTH_KERNEL, kern); synth.set_option(pkg.OPT_CHAIN_WHERE, cw)
        b = synth.batch(ch, 1.0 / fs, nsamp, flags=flags); b.run(); synth.sync(); iq, st = b.read(); b.close()
        act = ch["prn"] > 0
        bad = np.argwhere((st["carr_phase"] != want_st["carr_phase"]) & act)
        print("chain_where", cw, "iq equal", bool((iq == want_iq).all()), "bad end phases", bad.tolist())
        for blk, i in bad[:4]:
            print("   block %d ch %d prn %s f_carr %.17g step %.17g start %.17g got %.17g want %.17g diff %.3e" % (
                blk, i, ch["prn"][:, i].tolist(), ch["f_carr"][blk, i], ch["f_carr"][blk, i] / fs, ch["carr_phase"][blk, i], st["carr_phase"][blk, i], want_st["carr_phase"][blk, i], st["carr_phase"][blk, i] - want_st["carr_phase"][blk, i]))
